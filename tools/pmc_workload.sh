# HBM-side traffic of `bench.py --workload W` from the PMC counters (separate FETCH_SIZE / WRITE_SIZE passes), per kernel
# and per step -> gpurun_out/<tag>/pmc_traffic_<W>.json      usage: bash tools/pmc_workload.sh <tag> <workload> [codec]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
R=${1:-r02b}; W=${2:-s2}; C=${3:-roc}; mkdir -p gpurun_out/$R
CMD="python bench.py --workload $W --codec $C --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-verify"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c -d gpurun_out/$R/pmcw_$c -o p -- $CMD > gpurun_out/$R/pmcw_$c.out 2> gpurun_out/$R/pmcw_$c.err
done
python - <<PY
import glob, json, sqlite3
out = {"workload": "$W", "codec": "$C", "steps_profiled": 2}
for c, name in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    db = glob.glob(f"gpurun_out/$R/pmcw_{c}/*results.db")[0]
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select kernel_name, sum(value), count(*) from counters_collection where kernel_name like '%vidc::%' or kernel_name like '%(anonymous namespace)::k_%' group by kernel_name"))
    out[name + "_KiB_per_step"] = sum(r[1] for r in rows) / 2.0   # 1 warm-up + 1 timed step ran
    out[name + "_by_kernel_KiB_per_step"] = {(__import__("re").search(r"k_\w+(<[^>]*>)?", r[0]) or [r[0][:44]])[0]: r[1] / 2.0 for r in sorted(rows, key=lambda r: -r[1])}
try:
    line = [l for l in open("gpurun_out/$R/pmcw_FETCH_SIZE.out") if l.startswith("{")][-1]
    d = json.loads(line)
    # (16 + 2c B/id of SURVEY 8d + the 4 B/id of the sampling permutation the container path writes: what the kernels move by design)
    out["algorithmic_bytes_per_step"] = d["roofline"]["algorithmic_bytes_per_id_with_perm"] * d["config"]["ids_per_gpu"]
    out["traffic_over_algorithmic"] = 1024.0 * (out["fetch_KiB_per_step"] + out["write_KiB_per_step"]) / out["algorithmic_bytes_per_step"]
except Exception as e:
    out["note"] = str(e)
print(json.dumps(out, indent=1))
json.dump(out, open(f"gpurun_out/$R/pmc_traffic_${W}_$C.json".replace(":", "_"), "w"), indent=1)
PY
rm -rf gpurun_out/$R/pmcw_FETCH_SIZE gpurun_out/$R/pmcw_WRITE_SIZE
