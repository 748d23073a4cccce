cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/lane
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/lane/prof -o u16 -- python bench.py --workload uniform_16m --no-cpu-baseline --no-extra --steps 5 --warmup 2 > gpurun_out/lane/bench_u16.json 2> gpurun_out/lane/err.txt
python profiles/extract_rocprof.py gpurun_out/lane/prof/u16_results.db gpurun_out/lane/u16_stats.csv
head -8 gpurun_out/lane/u16_stats.csv | cut -c1-200
cat gpurun_out/lane/bench_u16.json | cut -c1-400
VIDC_TRACE=1 python bench.py --workload uniform_16m --no-cpu-baseline --no-extra --steps 1 --warmup 1 2>&1 | grep vidc | tail -16
