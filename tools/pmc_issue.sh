# Where the issue slots of a workload go, per kernel class (rocprofv3 --pmc, two passes of 8 SQ counters + GRBM_GUI_ACTIVE):
# instructions by type, wave cycles, cycles with an instruction in flight, cycles waiting -> gpurun_out/<tag>/pmc_issue_<W>.json
#   usage: bash tools/pmc_issue.sh <tag> <workload> [codec]
# (counter passes serialise the dispatches of a call: the numbers are per class with the machine to itself)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
R=${1:-r05}; W=${2:-s2}; C=${3:-roc}; mkdir -p gpurun_out/$R
CMD="python bench.py --workload $W --codec $C --steps 1 --warmup 1 --no-cpu-baseline --no-extra --no-verify"
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU"
i=0
for p in "$P1" "$P2"; do
  i=$((i+1))
  timeout 1200 rocprofv3 --pmc $p -d gpurun_out/$R/pmci_$i -o p -- $CMD > gpurun_out/$R/pmci_$i.out 2> gpurun_out/$R/pmci_$i.err
done
python - <<PY
import glob, json, re, sqlite3
out = {"workload": "$W", "codec": "$C", "steps_profiled": 2, "per_kernel_per_step": {}}
for i in (1, 2):
    dbs = glob.glob(f"gpurun_out/$R/pmci_{i}/*results.db")
    if not dbs:
        out[f"pass{i}"] = "no results: " + open(f"gpurun_out/$R/pmci_{i}.err").read()[-400:]
        continue
    cur = sqlite3.connect(dbs[0]).cursor()
    q = ("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like '%vidc::%' or "
         "kernel_name like '%(anonymous namespace)::k_%' group by kernel_name, counter_name")
    for kn, cn, v, n in cur.execute(q):
        k = (re.search(r"k_\w+(<[^>]*>)?", kn) or [kn[:44]])[0]
        out["per_kernel_per_step"].setdefault(k, {})[cn] = v / 2.0   # 1 warm-up + 1 timed step ran
json.dump(out, open(f"gpurun_out/$R/pmc_issue_${W}_$C.json".replace(":", "_"), "w"), indent=1)
tot = {}
for k, d in out["per_kernel_per_step"].items():
    for c, v in d.items():
        tot[c] = tot.get(c, 0) + v
print(json.dumps({"total_per_step": tot}, indent=1))
for k, d in sorted(out["per_kernel_per_step"].items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    print(k, {c: f"{v:.4g}" for c, v in d.items()})
PY
rm -rf gpurun_out/$R/pmci_1 gpurun_out/$R/pmci_2
