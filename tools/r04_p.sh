cd $GRAFT_REPO_ROOT
ROUNDS=8 timeout 1500 python tools/s2_ab.py - VIDC_R2_TAKE=1600 VIDC_R2_TAKE=1650 VIDC_R2_TAKE=1700 VIDC_R2_TAKE=1450 2>&1 | grep -v amdgpu.ids | tail -6
VIDC_TRACE=1 timeout 300 python - <<'PY' 2>&1 | grep "offsets\|classify\|sort work" | tail -3
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from vector_db_id_compression_amd import _lib, synth
from vector_db_id_compression_amd.codecs import RocLists
ctx = _lib.default_context(0)
w = synth.workload("s2", seed=1043)
for it in range(3):
    r = RocLists.encode(w["offsets"], w["ids"], ctx=ctx, want_perm=True); del r
PY
