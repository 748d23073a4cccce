#!/usr/bin/env python3
"""Latency of the calls a search makes (dev tool, run through gpurun): decode of the probed lists and single-id
translation on the S1 index (1 M ids, 1024 Zipf lists), per container.

  decode_lists(m lists)   m = 16 (one query, nprobe 16) and 1600 (100 queries): what IndexIVF::search_preassigned
                          does through get_ids when ids are not deferred
  get(m pairs)            m = 2000: (list, offset) -> id for the k results of a deferred search (get_single_id)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_db_id_compression_amd import synth
from vector_db_id_compression_amd.codecs import EfLists, PackedLists, RocLists, WaveletTreeLists

w = synth.workload("s1")
off, ids = w["offsets"], w["ids"]
sizes = np.diff(off.astype(np.int64))
rng = np.random.default_rng(3)
objs = {"roc": RocLists.encode(off, ids), "elias-fano": EfLists.encode(off, ids), "packed-bits": PackedLists.encode(off, ids),
        "wavelet-tree": WaveletTreeLists.build(off, ids)}


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(t))


for m in (16, 1600):
    probe = rng.choice(off.size - 1, size=m, replace=True, p=sizes / sizes.sum()).astype(np.uint64)  # probes follow list mass
    nids = int(sizes[probe.astype(np.int64)].sum())
    for name in ("roc", "elias-fano"):
        ms = timeit(lambda: objs[name].decode_lists(probe))
        print(f"decode_lists {name:12s} m={m:5d} ({nids} ids) {ms:8.3f} ms  {nids / ms / 1e3:8.1f} M ids/s", flush=True)
ln = rng.integers(0, off.size - 1, 2000).astype(np.uint64)
ln = ln[sizes[ln.astype(np.int64)] > 0]
of = (rng.random(ln.size) * sizes[ln.astype(np.int64)]).astype(np.uint64)
for name, fn in (("elias-fano", lambda: objs["elias-fano"].get(ln, of)), ("packed-bits", lambda: objs["packed-bits"].get(ln, of)),
                 ("wavelet-tree", lambda: objs["wavelet-tree"].select(ln, of))):
    print(f"single ids   {name:12s} m={ln.size:5d} {timeit(fn):8.3f} ms", flush=True)
