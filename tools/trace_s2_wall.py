#!/usr/bin/env python3
"""Wall time of the Python-level pieces of one S2 step (encode call, decode call, object destruction)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_db_id_compression_amd import _lib, synth
from vector_db_id_compression_amd.codecs import RocLists

ctx = _lib.default_context(0)
wl = synth.workload(sys.argv[1] if len(sys.argv) > 1 else "s2", seed=1043)
ids = wl["ids"] if not isinstance(wl["ids"], np.ndarray) else torch.from_numpy(wl["ids"].view(np.int64)).cuda()
out = torch.empty(wl["ntotal"], dtype=torch.int64, device="cuda")
r = None
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r2 = RocLists.encode(wl["offsets"], ids, want_perm=True, ctx=ctx)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    r = None  # (destroy the previous object)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    r = r2; r2 = None
    r.decode_all(out)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"iter {it}: encode call {1e3*(t1-t0):.1f} ms (kernels {ctx.phase_ms(0):.1f}), destroy previous {1e3*(t2-t1):.1f} ms, decode call {1e3*(t3-t2):.1f} ms (kernels {ctx.phase_ms(2):.1f})", flush=True)
