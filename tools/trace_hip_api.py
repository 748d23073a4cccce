#!/usr/bin/env python3
"""Dev tool: N encode + decode_all pairs of one codec on one workload, nothing else -- run under
`rocprofv3 --hip-trace --stats` to see which HIP calls the host side of a call is made of, or alone for the wall / kernel split.
usage: trace_hip_api.py <workload> <ef|packed|roc> [reps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_db_id_compression_amd import _lib, synth
from vector_db_id_compression_amd.codecs import EfLists, PackedLists, RocLists
ctx = _lib.default_context(0)
w = synth.workload(sys.argv[1] if len(sys.argv) > 1 else "uniform_16m", seed=1)
cls = {"ef": EfLists, "packed": PackedLists, "roc": RocLists}[sys.argv[2] if len(sys.argv) > 2 else "ef"]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ids, off = w["ids"], w["offsets"]
if isinstance(ids, np.ndarray): ids = torch.from_numpy(ids.view(np.int64)).cuda()
out = torch.empty(w["ntotal"], dtype=torch.int64, device="cuda")
te = td = ke = kd = 0.0
for it in range(reps + 3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = cls.encode(off, ids, ctx=ctx)
    t1 = time.perf_counter(); k1 = ctx.last_kernel_ms()
    r.decode_all(out)
    t2 = time.perf_counter(); k2 = ctx.last_kernel_ms()
    if it >= 3:
        te += t1 - t0; td += t2 - t1; ke += k1; kd += k2
print("encode %.4f ms wall / %.4f kernels, decode %.4f / %.4f; host %.4f ms per pair" % (1e3*te/reps, ke/reps, 1e3*td/reps, kd/reps, 1e3*(te+td)/reps - (ke+kd)/reps))
