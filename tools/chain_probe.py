import os, sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from vector_db_id_compression_amd import _lib
from vector_db_id_compression_amd.codecs import RocLists
ctx = _lib.default_context(0)
rng = np.random.default_rng(5)
def run(n, nbits, tag):
    ids = np.sort(rng.choice(1 << nbits, size=n, replace=False)).astype(np.uint64)
    off = np.array([0, n], dtype=np.uint64)
    d = torch.from_numpy(ids.view(np.int64)).cuda()
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    best = [1e9, 1e9]
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = RocLists.encode(off, d, ctx=ctx)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        r.decode_all(out)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        best = [min(best[0], t1 - t0), min(best[1], t2 - t1)]
    print(tag, "n", n, "bits", nbits, "encode us/step %.3f decode us/step %.3f" % (1e6 * best[0] / n, 1e6 * best[1] / n), flush=True)
for n, nb in ((52114, 20), (52114, 24), (52114, 30), (165129, 24), (65536, 30)):
    run(n, nb, os.environ.get("TAG", "default"))
