"""Dev tool: device memory must not grow across repeated encode / decode calls (block cache reuse)."""
import sys, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_db_id_compression_amd import _lib, synth
from vector_db_id_compression_amd.codecs import RocLists, EfLists, PackedLists
ctx = _lib.default_context(0)
wl = synth.workload("uniform_16m", seed=3)
ids = wl["ids"] if not isinstance(wl["ids"], np.ndarray) else torch.from_numpy(wl["ids"].view(np.int64)).cuda()
out = torch.empty(wl["ntotal"], dtype=torch.int64, device="cuda")
rows = torch.from_numpy(synth.make_graph_rows(100000, 64, seed=1)).cuda()
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**20
f0 = None
for it in range(60):
    for cls in (RocLists, EfLists, PackedLists):
        o = cls.encode(wl["offsets"], ids, ctx=ctx); o.decode_all(out)
    g = RocLists.encode_rows(rows, ctx=ctx); g.decode_rows(None, 64)
    g = EfLists.encode_rows(rows, ctx=ctx); g.decode_rows(None, 64)
    if it in (4, 20, 59):
        print(f"iteration {it}: free {free():.0f} MiB")
