"""Row-per-list kernels (roc_grp.h) against the wave-per-list kernels on equal-sized lists: kernel ms and us per chain step.
usage: python tools/probe_grp.py [nlist:n ...]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_db_id_compression_amd import _lib, synth
from vector_db_id_compression_amd.codecs import RocLists
ctx = _lib.default_context(0)
cases = [c.split(":") for c in sys.argv[1:]] or [("1", "32768"), ("4", "32768"), ("4096", "8192"), ("2048", "32768"), ("16384", "6000")]
for nl, n in cases:
    nl, n = int(nl), int(n)
    off, ids = synth.make_lists_torch(nl * n, nl, 0.0, seed=3)
    out = torch.empty(nl * n, dtype=torch.int64, device="cuda")
    for mode in ("grp", "wave"):
        os.environ["VIDC_FORCE_GRP"] = "1" if mode == "grp" else "0"
        os.environ["VIDC_NO_GRP"] = "0" if mode == "grp" else "1"
        e = d = 1e9
        for it in range(3):
            r = RocLists.encode(off, ids, ctx=ctx, want_perm=True)
            e = min(e, ctx.phase_ms(0))
            r.decode_all(out)
            d = min(d, ctx.phase_ms(2))
        print(f"{nl:6d} lists x {n:6d} ids  {mode:5s} encode {e:8.3f} ms ({1e3 * e / n:.3f} us/step, {nl * n / e / 1e6:7.2f} G steps/s)   "
              f"decode {d:8.3f} ms ({1e3 * d / n:.3f} us/step, {nl * n / d / 1e6:7.2f} G steps/s)", flush=True)
