# rocprofv3 kernel stats of the 16 M ids / 65 536 lists workload (lane-per-list encoder + register decoder)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/lane
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/lane/prof -o u16 -- python bench.py --workload ${1:-uniform_16m} --no-cpu-baseline --no-extra --steps 8 --warmup 2 > gpurun_out/lane/bench_u16.json 2> gpurun_out/lane/err.txt
python profiles/extract_rocprof.py gpurun_out/lane/prof/u16_results.db gpurun_out/lane/u16_stats.csv
grep "k_roc" gpurun_out/lane/u16_stats.csv | cut -c1-60,120-300 | sed 's/,"[^"]*$//' | head -8
python -c "
import json
d=json.loads(open('gpurun_out/lane/bench_u16.json').read().strip().splitlines()[-1])
print('ms/step', round(d['ms_per_step'],3), d['kernel_ms'], d['verified_roundtrip'])"
rm -rf gpurun_out/lane/prof
