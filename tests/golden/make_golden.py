#!/usr/bin/env python3
"""Generate tests/golden/roc_golden.json from the REFERENCE codec (oracle/_ref).

Run in the build container (needs /root/reference to build oracle/_ref/libvidc_ref.so):
    python tests/golden/make_golden.py
Every expected value in the fixture comes from the reference's own functions
(compress / decompress / pop_with_finer_precision / codec_push, codec.h:47-52, driven by
oracle/ref_driver.cpp); none comes from the clean-room oracle or the HIP kernels.
Inputs are either stored verbatim (small cases) or described by a deterministic recipe
(`gen`) that tests/golden_cases.py re-creates; large outputs are stored as FNV-1a-64 digests.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_cases import CASES, fnv_stream, fnv_u64, make_ids  # noqa: E402
from oracle.pyoracle import Oracle, Ref  # noqa: E402

FULL_LIMIT = 96  # store complete vectors up to this n


def main():
    ref = Ref()
    orc = Oracle()  # only for the precision rule helper (double log2/ceil like the reference)
    out = []
    for case in CASES:
        ids = make_ids(case)
        n = int(ids.size)
        prec = case.get("precision")
        if prec is None:
            # container rule custom_invlists_impl.cpp:163-164
            prec = orc.list_precision(ids) if n else 0
        rt = ref.roundtrip(ids, prec) if n else dict(head=1 << 31, words=np.zeros(0, np.uint32), decoded=ids)
        enc = ref.container_encode(ids, prec, shuffle_seed=3) if n else dict(order=ids, perm=np.zeros(0, np.uint32))
        assert n == 0 or (enc["head"] == rt["head"] and np.array_equal(enc["words"], rt["words"]))
        if case.get("also_compress"):
            c2 = ref.compress(ids, prec)  # data-order insertion must give the same stream (Q1)
            assert c2["head"] == rt["head"] and np.array_equal(c2["words"], rt["words"])
        rec = dict(name=case["name"], n=n, precision=int(prec), head=int(rt["head"]), nwords=int(rt["words"].size),
                   stream_fnv=fnv_stream(rt["head"], rt["words"]),
                   order_fnv=fnv_u64(enc["order"]), perm_fnv=fnv_u64(enc["perm"].astype(np.uint64)),
                   decoded_fnv=fnv_u64(rt["decoded"]),
                   roundtrip_is_order=bool(np.array_equal(rt["decoded"], enc["order"])),
                   roundtrip_set_ok=bool(np.array_equal(np.sort(rt["decoded"]), np.sort(ids))))
        if n <= FULL_LIMIT:
            rec.update(ids=[int(x) for x in ids], words=[int(x) for x in rt["words"]],
                       order=[int(x) for x in enc["order"]], perm=[int(x) for x in enc["perm"]],
                       decoded=[int(x) for x in rt["decoded"]])
        out.append(rec)
        print(f"{case['name']:40s} n={n:6d} P={prec:2d} head={rec['head']} words={rec['nwords']} "
              f"set_ok={rec['roundtrip_set_ok']}")
    path = os.path.join(ROOT, "tests", "golden", "roc_golden.json")
    with open(path, "w") as f:
        json.dump(dict(source="reference codec.cpp via oracle/_ref (oracle/ref_driver.cpp)", cases=out), f, indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
