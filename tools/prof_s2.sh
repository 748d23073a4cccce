cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/s2p
W=${1:-s2}
timeout 900 rocprofv3 --kernel-trace -d gpurun_out/s2p/prof -o s2 -- python bench.py --workload $W --no-cpu-baseline --no-extra --no-verify --steps 2 --warmup 1 > /dev/null 2> gpurun_out/s2p/err.txt
python - <<'PY'
import sqlite3, glob, re
db = sqlite3.connect(glob.glob("gpurun_out/s2p/prof/*results.db")[0])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t][0] if any("kernel_dispatch" in t for t in tabs) else None
print([t for t in tabs if "kernel" in t][:10])
rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(start), max(end) from kernels group by name order by sum(duration) desc")) if "kernels" in tabs else []
for r in rows[:14]:
    m = re.search(r"(k_\w+(<[^>]*>)?)", r[0]); print((m.group(1) if m else r[0][:40]).ljust(34), "calls", r[1], "total_ms", round(r[2]/1e6,2), "avg_ms", round(r[3]/1e6,3))
# timeline of the last step: every dispatch with start/end relative to the first dispatch of the last encode
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]; print(cols)
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "0"); wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else "1")
rows = list(cur.execute(f"select name, start, end, {gx}, {wx} from kernels order by start")) if "kernels" in tabs else []
if rows:
    big = [i for i, r in enumerate(rows) if "k_roc_prepass" in r[0]]  # (k_roc_prepass or k_roc_prepass_last)
    i0 = big[-1] if big else 0
    t0 = rows[i0][1]
    for r in rows[i0:]:
        if r[2] - r[1] < int(__import__("os").environ.get("MIN_NS", "200000")): continue
        m = re.search(r"(k_\w+(<[^>]*>)?)", r[0]); print((m.group(1) if m else r[0][:40]).ljust(34), "start", round((r[1]-t0)/1e6,2), "end", round((r[2]-t0)/1e6,2), "wg", r[3]//max(r[4],1))
PY
rm -rf gpurun_out/s2p/prof
