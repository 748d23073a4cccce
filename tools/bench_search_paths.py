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

# The decode section of a deferred search (custom_invlists_impl.cpp:508-525) on the C3 shape (IVF1024, 1 M vectors): nq = 10^4
# queries x nprobe 16, k = 20 results each.  "lists to host" = what the adapter did until round 4 (decode the touched lists, copy
# EVERY touched list over PCIe, index on the host); "device gather" = vidc_*_decode_gather (the scatter labels[r] = ids[offset] on
# the device, 8 bytes per result over PCIe).
nq, nprobe, k = 10_000, 16, 20
probes = rng.choice(off.size - 1, size=(nq, nprobe), replace=True, p=sizes / sizes.sum())
pick = probes[np.arange(nq)[:, None], rng.integers(0, nprobe, size=(nq, k))]  # the list each result came from
res_off = (rng.random((nq, k)) * sizes[pick]).astype(np.uint64)
touched, inv = np.unique(pick.ravel(), return_inverse=True)
touched = touched.astype(np.uint64)
slot = inv.astype(np.uint64)
flat_off = res_off.ravel()
n_touched_ids = int(sizes[touched.astype(np.int64)].sum())
print(f"deferred search, decode section: {nq} queries x k {k} = {slot.size} results from {touched.size} touched lists ({n_touched_ids} ids)")
for name in ("roc", "elias-fano", "packed-bits", "wavelet-tree"):
    o = objs[name]

    def lists_to_host():
        ids_d, lo = o.decode_lists(touched)
        h = ids_d.cpu().numpy()
        return h[lo[inv].astype(np.int64) + flat_off.astype(np.int64)]

    def device_gather():
        return o.decode_gather(touched, slot, flat_off)

    a, b = lists_to_host(), device_gather()
    assert np.array_equal(a, b)
    d0 = o.ctx.d2h_bytes()
    device_gather()
    moved = o.ctx.d2h_bytes() - d0
    t_old, t_new = timeit(lists_to_host, 10), timeit(device_gather, 10)
    print(f"  {name:12s} lists to host {t_old:8.3f} ms ({8 * n_touched_ids} B over PCIe)   device gather {t_new:8.3f} ms "
          f"({moved} B over PCIe)", flush=True)
