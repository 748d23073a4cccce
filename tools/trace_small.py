import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_db_id_compression_amd import _lib, synth
from vector_db_id_compression_amd.codecs import RocLists
ctx = _lib.default_context(0)
w = synth.workload(sys.argv[1] if len(sys.argv) > 1 else "s1_uniform", seed=1)
ids = w["ids"]; off = w["offsets"]
if isinstance(ids, np.ndarray): ids = torch.from_numpy(ids.view(np.int64)).cuda()
out = torch.empty(w["ntotal"], dtype=torch.int64, device="cuda")
for it in range(6):
    if it == 5: os.environ["VIDC_TRACE"] = "1"
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = RocLists.encode(off, ids, ctx=ctx, want_perm=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    r.decode_all(out)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("encode %.3f ms (kernels %.3f)  decode %.3f ms (kernels %.3f)" % (1e3*(t1-t0), ctx.phase_ms(0)+ctx.phase_ms(1), 1e3*(t2-t1), ctx.phase_ms(2)))
