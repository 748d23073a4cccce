"""Cheap stress of concurrent decode classes: G8K-class general lists + lane<64> + lane<256> lists in one object, decoded
repeatedly and compared with the first decode (torch.equal: one pass).  usage: repro_mix.py <iters> [cfg ...]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("VIDC_PKG_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vector_db_id_compression_amd import _lib, synth
from vector_db_id_compression_amd.codecs import RocLists
ctx = _lib.default_context(0)
t0 = time.time()
rng = np.random.default_rng(3)
scale = int(os.environ.get("SCALE", "1"))
sizes = np.concatenate([rng.integers(4100, 8192, 16000 * scale), rng.integers(520, 1024, 200000 * scale), rng.integers(1030, 4096, 40000 * scale),
                        rng.integers(8200, 30000, 3000 * scale), rng.integers(33000, 65536, 300 * scale),
                        rng.integers(100, 257, int(os.environ.get("N256", "0"))), rng.integers(257, 513, int(os.environ.get("N512", "0")))]).astype(np.int64)
rng.shuffle(sizes)
N = int(sizes.sum())
off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
# ids: a random permutation cut into lists, sorted inside each list (device)
perm = torch.randperm(N, device="cuda")
bounds = torch.from_numpy(off[1:].astype(np.int64)).cuda()
seg = torch.searchsorted(bounds, torch.arange(N, device="cuda"), right=True)
ids = torch.sort((seg << 40) + perm).values & ((1 << 40) - 1)
del perm, seg
print("ids", N, "lists", sizes.size, "setup %.1f s" % (time.time() - t0), flush=True)
out = torch.empty(N, dtype=torch.int64, device="cuda")
ref = torch.empty_like(out)
r = RocLists.encode(off, ids, ctx=ctx, want_perm=True)
r.decode_all(ref)
srt = torch.sort((torch.searchsorted(bounds, torch.arange(N, device="cuda"), right=True) << 40) + ref).values & ((1 << 40) - 1)
print("reference decode == ids per list:", bool(torch.equal(srt, ids)), "nonclean", r.last_decode_nonclean, flush=True)
del srt
iters = int(sys.argv[1])
for cfg in sys.argv[2:] or ["X=1"]:
    for kv in cfg.split(","):
        k, v = kv.split("=")
        os.environ[k] = v
    bad = 0
    t1 = time.time()
    for it in range(iters):
        out.fill_(-1)
        r.decode_all(out)
        if not torch.equal(out, ref):
            bad += 1
            pos = torch.nonzero(out != ref).flatten()
            lists = torch.unique(torch.searchsorted(bounds, pos, right=True)).cpu().numpy()
            print(cfg, "it", it, "differs in", lists.size, "lists", [(int(l), int(sizes[l])) for l in lists[:5]], "wrong elements", pos.numel(), "nonclean", r.last_decode_nonclean, flush=True)
    print(cfg, "->", bad, "bad of", iters, "in %.1f s" % (time.time() - t1), flush=True)
