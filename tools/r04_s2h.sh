cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | grep -v "amdgpu.ids" | tail -40
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from vector_db_id_compression_amd import _lib, synth
from vector_db_id_compression_amd.codecs import RocLists
ctx = _lib.default_context(0)
w = synth.workload("s2", seed=1043)
ids = w["ids"]; off = w["offsets"]
out = torch.empty(w["ntotal"], dtype=torch.int64, device="cuda")
for it in range(4):
    if it == 3: os.environ["VIDC_TRACE"] = "1"
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = RocLists.encode(off, ids, ctx=ctx, want_perm=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    r.decode_all(out)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("encode %.3f ms (kernels %.3f)  decode %.3f ms (kernels %.3f)" % (1e3*(t1-t0), ctx.phase_ms(0)+ctx.phase_ms(1), 1e3*(t2-t1), ctx.phase_ms(2)))
PY
