import os, sys, time, numpy as np, torch
T0 = time.time()
sys.path.insert(0, os.environ.get("VIDC_PKG_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from vector_db_id_compression_amd import _lib, synth
from vector_db_id_compression_amd.codecs import RocLists
ctx = _lib.default_context(0)
if os.environ.get("ZIPF"):  # "N:L" -> Zipf(0.75) lists capped at 65536 (e.g. 500000000:2097152: half of S2's list sizes)
    N_, L_ = (int(v) for v in os.environ["ZIPF"].split(":"))
    off, ids = synth.make_lists_torch(N_, L_, 0.75, seed=1043, cap=65536)
else:
    w = synth.workload("s2", seed=1043)
    off, ids = w["offsets"], w["ids"]
sizes = (off[1:] - off[:-1]).astype(np.int64)
n = int(off[-1])
out = torch.empty(n, dtype=torch.int64, device="cuda")
r = RocLists.encode(off, ids, ctx=ctx, want_perm=True)
os.environ["VIDC_DEC_NQ"] = "8"
ref = torch.empty_like(out)
r.decode_all(ref)
print("reference decode nonclean", r.last_decode_nonclean, "t=%.1f" % (time.time() - T0), flush=True)
starts = torch.from_numpy(off[:-1].astype(np.int64)).cuda()
bounds = torch.from_numpy(off[1:].astype(np.int64)).cuda()
for cfg in sys.argv[1:]:
    for kv in cfg.split(","):
        if kv:
            k, v = kv.split("=")
            os.environ[k] = v
    r._plan = None
    nqs = [int(v) for v in os.environ.get("NQS", "5,6,8,7").split(",")]
    for it in range(int(os.environ.get("ITERS", "10"))):
        os.environ["VIDC_DEC_NQ"] = str(nqs[it % len(nqs)])
        out.fill_(-1)
        r.decode_all(out)
        ne = out != ref
        if bool(ne.any()):
            pos = torch.nonzero(ne).flatten()
            lists = torch.unique(torch.searchsorted(bounds, pos, right=True)).cpu().numpy()
            l0 = int(lists[0]); a0, b0 = int(off[l0]), int(off[l0 + 1])
            g = out[a0:b0].cpu().numpy(); rf = ref[a0:b0].cpu().numpy()
            nbad = int((g != rf).sum())
            print(cfg, "it", it, "nq", os.environ["VIDC_DEC_NQ"], "differs in", lists.size, "lists:", [(int(l), int(sizes[l])) for l in lists[:6]], "list", l0, "wrong elements", nbad, "of", b0 - a0,
                  "minus-ones", int((g == -1).sum()), "first wrong idx", int(np.flatnonzero(g != rf)[0]), "last wrong idx", int(np.flatnonzero(g != rf)[-1]), "nonclean", r.last_decode_nonclean, flush=True)
        else:
            print(cfg, "it", it, "ok t=%.1f" % (time.time() - T0), flush=True)
