#!/usr/bin/env python3
"""Differential fuzz of the wavelet-tree build (dev tool, run through gpurun): random list counts (1 .. 2^17), id counts on both sides of
the 2^18 from which list_nos[id] is built by the partitioned scatter, list-length distributions (equal, geometric, one huge list, thousands
of empty lists), both level codings.  The tree built through the partitioned scatter and the one built through the direct scatter
(VIDC_WT_SCATTER=1) must decode to the same ids, answer the same selects and have the same size; decode_all must give back the input;
a sample of selects is checked against the definition (the id at that offset of that list).
usage: fuzz_wt.py seed seconds"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_db_id_compression_amd.codecs import WaveletTreeLists  # noqa: E402


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    rng = np.random.default_rng(seed)
    t0 = time.time()
    nb = nids = 0
    while time.time() - t0 < budget:
        ntotal = int(rng.choice([int(rng.integers(1, 5000)), int(rng.integers(200_000, 300_000)), int(rng.integers(1 << 18, 1 << 21))]))
        nlist = int(rng.choice([1, 2, int(rng.integers(1, 300)), int(rng.integers(300, 5000)), int(rng.integers(5000, 1 << 17))]))
        shape = rng.choice(["uniform", "geometric", "huge", "empties"])
        if shape == "uniform":
            assign = rng.integers(0, nlist, ntotal)
        elif shape == "geometric":
            assign = np.minimum(rng.geometric(min(0.5, 8.0 / nlist), ntotal) - 1, nlist - 1)
        elif shape == "huge":
            assign = np.where(rng.random(ntotal) < 0.7, int(rng.integers(0, nlist)), rng.integers(0, nlist, ntotal))
        else:  # most lists empty
            used = rng.choice(nlist, size=max(1, nlist // 50), replace=False)
            assign = used[rng.integers(0, used.size, ntotal)]
        order = np.argsort(assign, kind="stable")
        counts = np.bincount(assign, minlength=nlist)
        off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        ids = order.astype(np.uint64)
        d_ids = torch.from_numpy(ids.view(np.int64)).cuda()
        wt_type = int(rng.integers(0, 2))
        ql = rng.integers(0, nlist, 300)
        ql = ql[counts[ql] > 0]
        qo = (rng.random(ql.size) * counts[ql]).astype(np.int64)
        want = ids[off[ql].astype(np.int64) + qo].astype(np.int64)
        res = []
        for direct in (False, True):
            if direct:
                os.environ["VIDC_WT_SCATTER"] = "1"
            else:
                os.environ.pop("VIDC_WT_SCATTER", None)
            wt = WaveletTreeLists.build(off, d_ids, wt_type=wt_type)
            dec = wt.decode_all().cpu().numpy().view(np.uint64)
            assert np.array_equal(dec, ids), (seed, nb, nlist, ntotal, shape, wt_type, direct, "decode_all")
            got = wt.select(ql, qo)
            assert np.array_equal(got, want), (seed, nb, nlist, ntotal, shape, wt_type, direct, "select")
            res.append(wt.size_in_bytes)
            del wt
        os.environ.pop("VIDC_WT_SCATTER", None)
        assert res[0] == res[1], (seed, nb, "size")
        nb += 1
        nids += ntotal
    print(f"fuzz ok: seed {seed}, {nb} trees, {nids} ids: partitioned scatter == direct scatter (decode_all, selects, sizes), both == the input", flush=True)


if __name__ == "__main__":
    main()
