"""Repeated S2 encode / decode with a per-list check that names the lists (and their sizes) that come back wrong."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from vector_db_id_compression_amd import _lib, synth
from vector_db_id_compression_amd.codecs import RocLists
ctx = _lib.default_context(0)
wname = sys.argv[1] if len(sys.argv) > 1 else "s2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
w = synth.workload(wname, seed=1043)
off, ids = w["offsets"], w["ids"]
if isinstance(ids, np.ndarray):
    ids = torch.from_numpy(ids.view(np.int64)).cuda()
sizes = (off[1:] - off[:-1]).astype(np.int64)
n = int(off[-1])
out = torch.empty(n, dtype=torch.int64, device="cuda")
bounds = torch.from_numpy(off[1:].astype(np.int64)).cuda()
def bad_lists(got):
    bad = []
    chunk = 1 << 27
    start = 0
    ends = off[1:].astype(np.int64)
    while start < n:
        j = int(np.searchsorted(ends, min(n, start + chunk), side="right"))
        end = int(ends[j - 1]) if j > 0 and ends[j - 1] > start else int(ends[min(j, ends.size - 1)])
        seg = torch.searchsorted(bounds, torch.arange(start, end, device="cuda"), right=True)
        a = torch.sort((seg << 40) + got[start:end]).values
        b = torch.sort((seg << 40) + ids[start:end]).values
        ne = a != b
        if bool(ne.any()):
            bad += torch.unique(a[ne] >> 40).cpu().tolist()
        start = end
    return bad
for rep in range(reps):
    r = RocLists.encode(off, ids, ctx=ctx, want_perm=True)
    for nq in (8, 5, 6, 4):
        os.environ["VIDC_DEC_NQ"] = str(nq)
        out.fill_(-1)
        try:
            r.decode_all(out)
        except Exception as e:
            print("rep", rep, "nq", nq, "EXCEPTION", str(e)[:300], flush=True)
            continue
        bad = bad_lists(out)
        print("rep", rep, "nq", nq, "bad lists", len(bad), [(l, int(sizes[l])) for l in bad[:12]], "nonclean", r.last_decode_nonclean, flush=True)
    del r
