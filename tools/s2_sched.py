"""Schedule experiments on S2 (round 4): the same object encoded / decoded under explicit kernel-class schedules
(VIDC_ENC_SCHED / VIDC_DEC_SCHED, csrc/roc.hip), kernel time per schedule from the library's hipEvents.
usage: python tools/s2_sched.py [E:<enc sched> | D:<dec sched> | X:<VAR=val,VAR=val> (environment for what follows)] ...
"E:" / "D:" alone = the library's default schedule.  Every decode is compared with the first one."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("VIDC_PKG_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from vector_db_id_compression_amd import _lib, synth
from vector_db_id_compression_amd.codecs import RocLists
ctx = _lib.default_context(0)
REPS = int(os.environ.get("REPS", "3"))
W = os.environ.get("WORKLOAD", "s2")
w = synth.workload(W, seed=1043)
off, ids = w["offsets"], w["ids"]
n = int(off[-1])
out = torch.empty(n, dtype=torch.int64, device="cuda")
ref = None
r = None
def enc():
    global r
    r = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = RocLists.encode(off, ids, ctx=ctx, want_perm=True)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0), ctx.phase_ms(0) + ctx.phase_ms(1)
def dec():
    global ref
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r.decode_all(out)
    torch.cuda.synchronize()
    wall = 1e3 * (time.perf_counter() - t0)
    k = ctx.phase_ms(2)
    if ref is None: ref = out.clone()
    ok = bool(torch.equal(out, ref))
    return wall, k, ok
for _ in range(2): enc(); dec()   # warm the block cache
for cfg in sys.argv[1:]:
    kind, _, val = cfg.partition(":")
    if kind == "X":
        for kv in val.split(","):
            if kv:
                k, _, v = kv.partition("=")
                if v == "": os.environ.pop(k, None)
                else: os.environ[k] = v
        print("env", val, flush=True)
        enc()  # (plan-time switches: the decode plan of an object is built while it is encoded)
        continue
    var = "VIDC_ENC_SCHED" if kind == "E" else "VIDC_DEC_SCHED"
    if val: os.environ[var] = val
    else: os.environ.pop(var, None)
    res = []
    for _ in range(REPS):
        if kind == "E":
            res.append(enc())
        else:
            r._plan = None
            res.append(dec())
    ks = sorted(x[1] for x in res)
    print("%s %-72s kernels min %.2f med %.2f ms (wall med %.2f)%s%s" % (kind, val or "(default)", ks[0], ks[len(ks) // 2], sorted(x[0] for x in res)[len(res) // 2],
          "" if kind == "E" or all(x[2] for x in res) else "  DECODE DIFFERS",
          ("  all: " + " ".join("%.1f" % x[1] for x in res)) if os.environ.get("SHOW_ALL") else ""), flush=True)
    os.environ.pop(var, None)
    if kind == "E": dec()
