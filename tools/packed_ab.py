"""Interleaved A/B of packed-bits switches (environment, `K=V,K=V`; "-" = defaults) on one workload: encode + decode of the same lists under
every configuration in every round; medians of the kernel times.  usage: WORKLOAD=s2 ROUNDS=10 python tools/packed_ab.py - VIDC_PACKED_TILE16=1"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("VIDC_PKG_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vector_db_id_compression_amd import _lib, synth
from vector_db_id_compression_amd.codecs import PackedLists
ctx = _lib.default_context(0)
w = synth.workload(os.environ.get("WORKLOAD", "s2"), seed=1043)
off, ids = w["offsets"], w["ids"]
if isinstance(ids, np.ndarray): ids = torch.from_numpy(ids.view(np.int64)).cuda()
out = torch.empty(int(off[-1]), dtype=torch.int64, device="cuda")
cfgs = sys.argv[1:] or ["-"]
touched = set(kv.partition("=")[0] for c in cfgs if c != "-" for kv in c.split(","))
res = {c: ([], []) for c in cfgs}
for rnd in range(int(os.environ.get("ROUNDS", "10")) + 2):
    for c in cfgs:
        for k in touched: os.environ.pop(k, None)
        if c != "-":
            for kv in c.split(","):
                k, _, v = kv.partition("="); os.environ[k] = v
        r = PackedLists.encode(off, ids, ctx=ctx); ke = ctx.last_kernel_ms()
        r.decode_all(out); kd = ctx.last_kernel_ms()
        assert torch.equal(out, ids)
        if rnd >= 2: res[c][0].append(ke); res[c][1].append(kd)
for c in cfgs:
    e, d = (np.array(x) for x in res[c])
    print("%-40s encode med %.4f mean %.4f | decode med %.4f mean %.4f" % (c, np.median(e), e.mean(), np.median(d), d.mean()), flush=True)
