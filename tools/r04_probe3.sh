cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04c
for b in 0 1; do
  VIDC_LANE_BATCH=$b timeout 600 python bench.py --workload uniform_64m_1k --no-cpu-baseline --no-extra --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch=$b uniform_64m_1k', d['ms_per_step'], d['kernel_ms'], d['verified_roundtrip'])"
done > gpurun_out/r04c/u64.txt 2>&1
timeout 900 python tools/s2_sched.py "D:" "X:VIDC_LANE_BATCH=1" "D:" "X:VIDC_LANE_BATCH=0" "D:" "X:VIDC_LANE_BATCH=1" "D:" "X:VIDC_SERIAL=1" "D:" > gpurun_out/r04c/s2.txt 2>&1
cat gpurun_out/r04c/u64.txt gpurun_out/r04c/s2.txt
