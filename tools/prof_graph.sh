# BASELINE configs[3] (10^6 nodes x K = 64 graph rows) evidence: the tool's own line, rocprofv3 kernel stats of the same command and
# the HBM-side traffic of every kernel from separate FETCH_SIZE / WRITE_SIZE passes -> gpurun_out/<tag>/
#   usage: bash tools/prof_graph.sh <tag>        (then: cp gpurun_out/<tag>/{bench_graph.json,bench_graph_kernel_stats.csv,pmc_traffic_graph_*.json} profiles/)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
R=${1:-r06}; mkdir -p gpurun_out/$R
python tools/bench_graph.py 1000000 64 5 2>/dev/null | tail -1 > gpurun_out/$R/bench_graph.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/kt -o g -- python tools/bench_graph.py 1000000 64 5 > /dev/null 2> gpurun_out/$R/kt.err
cp gpurun_out/$R/kt/g_kernel_stats.csv gpurun_out/$R/bench_graph_kernel_stats.csv 2>/dev/null || find gpurun_out/$R/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/$R/bench_graph_kernel_stats.csv \;
rm -rf gpurun_out/$R/kt
for codec in elias-fano compact roc; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c -d gpurun_out/$R/pg_$c -o p -- python tools/bench_graph.py 1000000 64 1 $codec > gpurun_out/$R/pg_$c.out 2> gpurun_out/$R/pg_$c.err
  done
  python - <<PY
import glob, json, re, sqlite3
out = {"workload": "10^6 graph nodes x K=64 int32 rows (tools/bench_graph.py 1000000 64 1 $codec)", "codec": "$codec", "calls_profiled": 2,
       "note": "gfx950: FETCH_SIZE counts 16-byte-per-lane reads at half their bytes (MI355X_MICROARCH.md); ratios between kernels are unaffected"}
for c, name in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    db = glob.glob(f"gpurun_out/$R/pg_{c}/*results.db")[0]
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select kernel_name, sum(value), count(*) from counters_collection where kernel_name like '%vidc::%' or kernel_name like '%(anonymous namespace)::k_%' group by kernel_name"))
    out[name + "_KiB_per_call_by_kernel"] = {(re.search(r"k_\w+(<[^>]*>)?", r[0]) or [r[0][:44]])[0]: r[1] / r[2] for r in sorted(rows, key=lambda r: -r[1])}
    out[name + "_KiB_per_encode_plus_decode"] = sum(r[1] / r[2] for r in rows)
try:
    d = json.loads(open("gpurun_out/$R/pg_FETCH_SIZE.out").read().strip().split("\n")[-1])
    out["algorithmic_bytes"] = 1e6 * d["$codec"]["algorithmic_MB"]
    out["traffic_over_algorithmic"] = 1024.0 * (out["fetch_KiB_per_encode_plus_decode"] + out["write_KiB_per_encode_plus_decode"]) / out["algorithmic_bytes"]
except Exception as e:
    out["parse_note"] = str(e)
json.dump(out, open("gpurun_out/$R/pmc_traffic_graph_" + {"elias-fano": "ef"}.get("$codec", "$codec") + ".json", "w"), indent=1)
print(json.dumps(out)[:600])
PY
  rm -rf gpurun_out/$R/pg_FETCH_SIZE gpurun_out/$R/pg_WRITE_SIZE
done
