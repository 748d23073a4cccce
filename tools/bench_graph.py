#!/usr/bin/env python3
"""Dev tool: S3 graph shape (N nodes x K=64 int32 rows) through the three graph containers (BASELINE configs[3]).
Per codec: kernel ms (hipEvents inside the C-ABI), wall ms, bits/edge, fraction of the 8 TB/s HBM peak on SURVEY 8(d)'s graph
variant of the algorithmic bytes (8 + 2c per edge), and a real round trip (sorted neighbour sets of every row)."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_db_id_compression_amd import synth, _lib
from vector_db_id_compression_amd.codecs import RocLists, EfLists, CompactRows

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
only = sys.argv[4].split(",") if len(sys.argv) > 4 else None
rows = torch.from_numpy(synth.make_graph_rows(N, K, seed=44)).cuda()
ctx = _lib.default_context()
edges = int((rows >= 0).sum().item())
big = torch.iinfo(torch.int32).max
want = torch.sort(torch.where(rows >= 0, rows, torch.full_like(rows, big)), dim=1).values
out = {}
for name, cls in [("roc", RocLists), ("elias-fano", EfLists), ("compact", CompactRows)]:
    if only and name not in only:
        continue
    te = td = ke = kd = 0.0
    for rep in range(REPS + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g = cls.encode_rows(rows)
        t1 = time.perf_counter()
        k_enc = ctx.last_kernel_ms()
        dec, _ = g.decode_rows(None, K, want_counts=False)
        t2 = time.perf_counter()
        k_dec = ctx.last_kernel_ms()
        if rep:
            te += t1 - t0; td += t2 - t1; ke += k_enc; kd += k_dec
    te /= REPS; td /= REPS; ke /= REPS; kd /= REPS
    got = torch.sort(torch.where(dec >= 0, dec, torch.full_like(dec, big)), dim=1).values
    ok = bool(torch.equal(got, want))
    size = g.compressed_bytes if name != "compact" else g.size_in_bytes
    alg = (8.0 + 2.0 * size / edges) * edges
    out[name] = dict(encode_ms=1e3 * te, decode_ms=1e3 * td, kernel_ms_enc=ke, kernel_ms_dec=kd, bits_per_edge=8.0 * size / edges,
                     edges_per_s=edges / (te + td), algorithmic_MB=alg / 1e6, frac=alg / ((ke + kd) * 1e-3) / 8e12,
                     frac_wall=alg / (te + td) / 8e12, host_ms=1e3 * (te + td) - ke - kd, roundtrip_ok=ok)
print(json.dumps(dict(N=N, K=K, edges=edges, **out)))
