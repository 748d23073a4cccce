"""Interleaved A/B/C... comparison of library switches on S2 (round 4): every round encodes + decodes the same 10^9-id object once
under each configuration (environment switches, `K=V,K=V`; "-" = defaults), so that drift over a run (clocks, the block cache)
hits every configuration alike; medians / means over ROUNDS rounds.  The decode time of ONE configuration varies by +-5 ms from
call to call (which class's wavefronts become resident first), so three repetitions of A followed by three of B prove nothing.
usage: ROUNDS=10 python tools/s2_ab.py - VIDC_B2_TOP_OLD=1 ..."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("VIDC_PKG_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from vector_db_id_compression_amd import _lib, synth
from vector_db_id_compression_amd.codecs import RocLists
ctx = _lib.default_context(0)
w = synth.workload(os.environ.get("WORKLOAD", "s2"), seed=1043)
off, ids = w["offsets"], w["ids"]
if isinstance(ids, np.ndarray):
    ids = torch.from_numpy(ids.view(np.int64)).cuda()
out = torch.empty(int(off[-1]), dtype=torch.int64, device="cuda")
cfgs = sys.argv[1:] or ["-"]
touched = set()
for c in cfgs:
    if c != "-":
        touched.update(kv.partition("=")[0] for kv in c.split(","))
res = {c: ([], [], []) for c in cfgs}
ref = None
for rnd in range(int(os.environ.get("ROUNDS", "10")) + 2):
    for c in cfgs:
        for k in touched: os.environ.pop(k, None)
        if c != "-":
            for kv in c.split(","):
                k, _, v = kv.partition("=")
                os.environ[k] = v
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = RocLists.encode(off, ids, ctx=ctx, want_perm=True)
        ke = ctx.phase_ms(0) + ctx.phase_ms(1)
        r.decode_all(out)
        kd = ctx.phase_ms(2)
        torch.cuda.synchronize(); wall = 1e3 * (time.perf_counter() - t0)
        if ref is None: ref = out.clone()
        elif not torch.equal(out, ref): print("DECODE DIFFERS under", c, flush=True)
        del r
        if rnd >= 2:
            res[c][0].append(ke); res[c][1].append(kd); res[c][2].append(wall)
for c in cfgs:
    e, d, wl = (np.array(x) for x in res[c])
    print("%-60s encode med %.2f mean %.2f | decode med %.2f mean %.2f (min %.1f max %.1f) | step wall med %.2f mean %.2f" %
          (c, np.median(e), e.mean(), np.median(d), d.mean(), d.min(), d.max(), np.median(wl), wl.mean()), flush=True)
