import os, sys, time
os.environ["VIDC_TRACE"]="1"
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from vector_db_id_compression_amd import synth, _lib
from vector_db_id_compression_amd.codecs import RocLists
N=1000000; K=64
rows = torch.from_numpy(synth.make_graph_rows(N, K, seed=44)).cuda()
ctx = _lib.default_context()
nodes = np.arange(N, dtype=np.uint64)
for rep in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    g = RocLists.encode_rows(rows)
    torch.cuda.synchronize(); t1=time.perf_counter()
    dec, cnt = g.decode_rows(nodes, K, want_counts=(rep == 0))
    torch.cuda.synchronize(); t2=time.perf_counter()
    print(f"--- rep {rep}: encode {1e3*(t1-t0):.2f} ms decode {1e3*(t2-t1):.2f} ms", file=sys.stderr)
