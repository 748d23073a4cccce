#!/usr/bin/env python3
"""Re-run a batch saved by fuzz_families.py / fuzz_chain.py (gpurun_out/fuzz_fail.npz) through the three kernel families
and print which list and which field differs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_db_id_compression_amd.codecs import RocLists  # noqa: E402

MODES = {"lane": {"VIDC_FORCE_LANE": "1", "VIDC_NO_LANE": "0", "VIDC_FORCE_GENERAL": "0"},
         "wave": {"VIDC_FORCE_LANE": "0", "VIDC_NO_LANE": "1", "VIDC_FORCE_GENERAL": "0"},
         "general": {"VIDC_FORCE_LANE": "0", "VIDC_NO_LANE": "1", "VIDC_FORCE_GENERAL": "1"}}
d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fuzz_fail.npz")
off, ids, mode = d["off"], d["ids"], int(d["mode"])
for want_perm in (False, True):
    got = {}
    for name, env in MODES.items():
        os.environ.update(env)
        r = RocLists.encode(off, ids, precision_mode=mode, want_perm=want_perm)
        info = r.info()
        dec = r.decode_all().cpu().numpy().copy()
        got[name] = dict(heads=info["heads"], nwords=info["nwords"], prec=info["precision"], draws=info["mt_draws"],
                         words=r.all_words(), dec=dec, perm=r.perm() if want_perm else np.zeros(0), nonclean=np.array([r.last_decode_nonclean]))
    for name in ("wave", "general"):
        for k in got["lane"]:
            a, b = got["lane"][k], got[name][k]
            if a.shape != b.shape or not np.array_equal(a, b):
                bad = np.nonzero(a != b)[0] if a.shape == b.shape else []
                where = ""
                if k in ("dec", "perm") and len(bad):
                    where = f"lists {sorted(set(np.searchsorted(off, bad, side='right') - 1))[:10]}"
                print(f"want_perm={want_perm}: lane vs {name}: {k} differs ({len(bad)} entries) first {bad[:8]} {where}", flush=True)
print("done")
