#!/usr/bin/env python3
"""Dev tool: S3 graph shape (N nodes x K=64 int32 rows) through the three graph containers (BASELINE configs[3])."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_db_id_compression_amd import synth, _lib
from vector_db_id_compression_amd.codecs import RocLists, EfLists, CompactRows

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rows = torch.from_numpy(synth.make_graph_rows(N, 64, seed=44)).cuda()
ctx = _lib.default_context()
nodes = np.arange(N, dtype=np.uint64)
edges = int((rows >= 0).sum().item())
out = {}
for name, cls in [("roc", RocLists), ("elias-fano", EfLists), ("compact", CompactRows)]:
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g = cls.encode_rows(rows)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        k_enc = ctx.last_kernel_ms()
        dec, cnt = g.decode_rows(None, 64)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        k_dec = ctx.last_kernel_ms()
    size = g.compressed_bytes if name != "compact" else g.size_in_bytes
    out[name] = dict(encode_s=t1 - t0, decode_s=t2 - t1, kernel_ms_enc=k_enc, kernel_ms_dec=k_dec,
                     bits_per_edge=8.0 * size / edges, edges_per_s=edges / (t1 - t0 + t2 - t1))
print(json.dumps(dict(N=N, K=64, edges=edges, **out)))
