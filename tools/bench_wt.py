#!/usr/bin/env python3
"""Dev tool: wavelet-tree container (custom_invlists_impl.cpp:346-397) at the S1 shape and larger: build, decode_all,
random select (get_single_id)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vector_db_id_compression_amd import _lib, synth
from vector_db_id_compression_amd.codecs import WaveletTreeLists

ctx = _lib.default_context()
for name in sys.argv[1:] or ["s1", "uniform_16m"]:
    wl = synth.workload(name, seed=5)
    ids = wl["ids"]
    if isinstance(ids, np.ndarray):
        ids = torch.from_numpy(ids.view(np.int64)).cuda()
    # the wavelet tree stores the list number of every id: ids must be a permutation of 0..ntotal-1
    perm_ok = bool(torch.equal(torch.sort(ids).values, torch.arange(wl["ntotal"], device="cuda")))
    if not perm_ok:
        ids = torch.randperm(wl["ntotal"], device="cuda")
    for wt_type in (0, 1):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            wt = WaveletTreeLists.build(wl["offsets"], ids, wt_type=wt_type)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            out = wt.decode_all()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            rng = np.random.default_rng(1)
            q = 100000
            ln = rng.integers(0, wl["nlist"], q).astype(np.uint64)
            sizes = (wl["offsets"][1:] - wl["offsets"][:-1])[ln.astype(np.int64)]
            keep = sizes > 0
            ln = ln[keep]
            of = (rng.random(ln.size) * sizes[keep]).astype(np.uint64)
            t3 = time.perf_counter()
            got = wt.select(ln, of)
            t4 = time.perf_counter()
        print(f"{name} wt_type={wt_type}: build {1e3*(t1-t0):.2f} ms, decode_all {1e3*(t2-t1):.2f} ms ({wl['ntotal']/(t2-t1)/1e6:.0f} M ids/s), "
              f"{ln.size} selects {1e3*(t4-t3):.2f} ms ({ln.size/(t4-t3)/1e6:.1f} M/s), levels {wt.levels}, "
              f"{8*wt.size_in_bytes/wl['ntotal']:.2f} bit/id", flush=True)
