cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/lane
CMD="python bench.py --workload uniform_16m --no-cpu-baseline --no-extra --no-verify --steps 2 --warmup 1"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d gpurun_out/lane/pmc$i -o p -- $CMD > /dev/null 2> gpurun_out/lane/pmc$i.err
  python - <<PY
import sqlite3,glob
for db in glob.glob("gpurun_out/lane/pmc$i/*results.db"):
    c=sqlite3.connect(db).cursor()
    for name,cn,v,n in c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where kernel_name like '%lane%' group by kernel_name, counter_name"):
        print(name.split('(')[0][-28:], cn, f"{v/n:.4g} per dispatch", n)
PY
done
