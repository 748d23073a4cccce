# HBM-side traffic of the default bench (S1) from the PMC counters, separate passes (FETCH_SIZE and WRITE_SIZE do not fit
# one pass); per-kernel sums -> profiles/<round>_bench_s1_pmc_{fetch,write}.csv + a per-step total (KiB)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
R=${1:-r01g}; mkdir -p gpurun_out/$R
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d gpurun_out/$R/pmc_$c -o p -- $CMD > /dev/null 2> gpurun_out/$R/pmc_$c.err
done
python - <<PY
import sys, glob, json
sys.path.insert(0, "profiles")
from extract_rocprof import pmc_summary
import sqlite3
out = {}
for c, name in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    db = glob.glob(f"gpurun_out/$R/pmc_{c}/*results.db")[0]
    pmc_summary(db, f"gpurun_out/$R/bench_s1_pmc_{name}.csv")
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select kernel_name, sum(value), count(*) from counters_collection where kernel_name like '%vidc::%' or kernel_name like '%(anonymous namespace)::k_%' group by kernel_name"))
    # 3 bench steps (1 warm-up + 2 timed) ran: per-step KiB = sum / 3
    out[name + "_KiB_per_step"] = sum(r[1] for r in rows) / 3.0
    out[name + "_by_kernel_KiB_per_step"] = {(__import__("re").search(r"k_\w+(<[^>]*>)?", r[0]) or [r[0][:40]])[0]: r[1] / 3.0 for r in rows}
print(json.dumps(out, indent=1))
json.dump(out, open(f"gpurun_out/$R/pmc_traffic.json", "w"), indent=1)
PY
rm -rf gpurun_out/$R/pmc_FETCH_SIZE gpurun_out/$R/pmc_WRITE_SIZE
