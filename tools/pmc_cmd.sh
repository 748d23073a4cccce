# Per-kernel counters of ANY command (rocprofv3 --pmc, separate passes: issue slots, waits, memory-side bytes) ->
# gpurun_out/<tag>/pmc_<name>.json; values are per dispatch (sum / dispatch count) with the dispatch count beside them.
#   usage: bash tools/pmc_cmd.sh <tag> <name> <command ...>
# (counter passes serialise the dispatches: every kernel has the machine to itself)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
R=$1; NAME=$2; shift 2
mkdir -p gpurun_out/$R
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_BRANCH TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"
i=0
for p in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $p -d gpurun_out/$R/pmcc_$i -o p -- "$@" > gpurun_out/$R/pmcc_$i.out 2> gpurun_out/$R/pmcc_$i.err
done
python - <<PY
import glob, json, re, sqlite3
out = {"command": """$*""", "per_kernel_per_dispatch": {}}
for i in range(1, 6):
    dbs = glob.glob(f"gpurun_out/$R/pmcc_{i}/*results.db")
    if not dbs:
        out[f"pass{i}"] = "no results: " + open(f"gpurun_out/$R/pmcc_{i}.err").read()[-300:]
        continue
    cur = sqlite3.connect(dbs[0]).cursor()
    q = ("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like '%vidc::%' or "
         "kernel_name like '%(anonymous namespace)::k_%' group by kernel_name, counter_name")
    for kn, cn, v, n in cur.execute(q):
        k = (re.search(r"k_\w+(<[^>]*>)?", kn) or [kn[:44]])[0]
        d = out["per_kernel_per_dispatch"].setdefault(k, {})
        d[cn] = v / n
        d["dispatches"] = n
json.dump(out, open("gpurun_out/$R/pmc_$NAME.json", "w"), indent=1)
for k, d in sorted(out["per_kernel_per_dispatch"].items()):
    print(k, {c: (f"{v:.4g}" if isinstance(v, float) else v) for c, v in d.items()})
PY
rm -rf gpurun_out/$R/pmcc_?
