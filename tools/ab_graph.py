#!/usr/bin/env python3
"""A/B of the lane-per-row ROC graph kernels vs the wave-per-row kernels (VIDC_NO_LANE=1): identical streams + timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_db_id_compression_amd import synth, _lib
from vector_db_id_compression_amd.codecs import RocLists

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rows = torch.from_numpy(synth.make_graph_rows(N, K, seed=44, dmin=K // 2)).cuda()
ctx = _lib.default_context()
nodes = np.arange(N, dtype=np.uint64)
res = {}
for mode in ("1", "0"):
    os.environ["VIDC_NO_LANE"] = mode; os.environ["VIDC_FORCE_LANE"] = "0" if mode == "1" else "1"
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g = RocLists.encode_rows(rows)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        k_enc = ctx.last_kernel_ms()
        dec, cnt = g.decode_rows(nodes, K)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        k_dec = ctx.last_kernel_ms()
    info = g.info()
    res[mode] = (info["heads"], info["nwords"], info["precision"], info["mt_draws"], g.all_words(), dec.cpu().numpy(), np.asarray(cnt))
    print(f"N={N} K={K} NO_LANE={mode}: encode kernels {k_enc:.3f} ms (wall {1e3*(t1-t0):.2f}), decode kernels {k_dec:.3f} ms (wall {1e3*(t2-t1):.2f})", flush=True)
print("identical:", all(np.array_equal(a, b) for a, b in zip(res["0"], res["1"])))
