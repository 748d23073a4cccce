#!/usr/bin/env python3
"""Host-side phase trace (VIDC_TRACE=1) and python-level wall times of one encode + decode of a workload."""
import os
import sys
import time

import numpy as np
import torch

os.environ["VIDC_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_db_id_compression_amd import _lib, synth
from vector_db_id_compression_amd.codecs import RocLists

ctx = _lib.default_context(0)
wl = synth.workload(sys.argv[1] if len(sys.argv) > 1 else "uniform_16m", seed=7)
ids = torch.from_numpy(wl["ids"].view(np.int64)).cuda() if isinstance(wl["ids"], np.ndarray) else wl["ids"]
out = torch.empty(wl["ntotal"], dtype=torch.int64, device="cuda")
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = RocLists.encode(wl["offsets"], ids, ctx=ctx)
    t1 = time.perf_counter()
    r.decode_all(out)
    t2 = time.perf_counter()
    del r
    t3 = time.perf_counter()
    print(f"--- iteration {it}: encode {1e3*(t1-t0):.3f} ms, decode {1e3*(t2-t1):.3f} ms, destroy {1e3*(t3-t2):.3f} ms", file=sys.stderr, flush=True)
