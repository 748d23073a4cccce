#!/usr/bin/env python3
"""A/B of the lane-per-list kernels against the wave-per-list kernels (VIDC_NO_LANE=1): identical streams, timing."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_db_id_compression_amd import _lib, synth
from vector_db_id_compression_amd.codecs import RocLists

ctx = _lib.default_context(0)
for name in sys.argv[1:] or ["uniform_16m", "c5", "s1"]:
    wl = synth.workload(name, seed=7)
    ids = torch.from_numpy(wl["ids"].view(np.int64)).cuda() if isinstance(wl["ids"], np.ndarray) else wl["ids"]
    out = torch.empty(wl["ntotal"], dtype=torch.int64, device="cuda")
    res = {}
    for mode in ("1", "0"):
        os.environ["VIDC_NO_LANE"] = mode; os.environ["VIDC_FORCE_LANE"] = "0" if mode == "1" else "1"
        for it in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = RocLists.encode(wl["offsets"], ids, want_perm=True, ctx=ctx)
            e_ms = ctx.phase_ms(0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            r.decode_all(out)
            d_ms = ctx.phase_ms(2)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        info = r.info()
        res[mode] = dict(heads=info["heads"], nwords=info["nwords"], draws=info["mt_draws"], words=r.all_words(),
                         perm=r.perm(), out=out.cpu().numpy().copy())
        print(f"{name} NO_LANE={mode}: enc kernels {e_ms:.3f} ms (wall {1e3*(t1-t0):.2f}), dec kernels {d_ms:.3f} ms "
              f"(wall {1e3*(t2-t1):.2f})", flush=True)
    same = all(np.array_equal(res["0"][k], res["1"][k]) for k in res["0"])
    print(f"{name}: streams/perm/decoded identical = {same}", flush=True)
    if not same:
        for k in res["0"]:
            if not np.array_equal(res["0"][k], res["1"][k]):
                bad = np.nonzero(res["0"][k] != res["1"][k])[0] if res["0"][k].shape == res["1"][k].shape else []
                print("  differs:", k, res["0"][k].shape, res["1"][k].shape, bad[:5])
