cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from vector_db_id_compression_amd import _lib
from vector_db_id_compression_amd.codecs import RocLists
ctx = _lib.default_context(0)
rng = np.random.default_rng(5)
for n, nb in ((52114, 18), (52114, 17), (52114, 20)):
    for perm in (False, True):
        ids = np.sort(rng.choice(1 << nb, size=n, replace=False)).astype(np.uint64)
        off = np.array([0, n], dtype=np.uint64)
        d = torch.from_numpy(ids.view(np.int64)).cuda()
        out = torch.empty(n, dtype=torch.int64, device="cuda")
        e = dd = 1e9
        for it in range(5):
            r = RocLists.encode(off, d, ctx=ctx, want_perm=perm); e = min(e, ctx.phase_ms(0))
            r.decode_all(out); dd = min(dd, ctx.phase_ms(2))
        ok = bool(torch.equal(torch.sort(out).values, d))
        print("n", n, "bits", nb, "perm", perm, "encode us/step %.4f decode us/step %.4f" % (1e3 * e / n, 1e3 * dd / n), "ok", ok, flush=True)
PY
timeout 900 python -m pytest tests/test_gpu_roc.py tests/test_gpu_full_configs.py -m gpu -q 2>&1 | tail -3
timeout 300 python tools/fuzz_chain.py 21 90 2>&1 | tail -1
