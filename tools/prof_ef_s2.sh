# rocprofv3 kernel stats of the Elias-Fano and packed-bits benches on S2 / 16 M x 256 / S1 -> gpurun_out/<tag>/{ef,packed}_<workload>_kernel_stats.csv
# usage: bash tools/prof_ef_s2.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=${1:-efp}; mkdir -p gpurun_out/$R
for C in ef packed; do for W in s2 uniform_16m s1; do
  timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/$R/prof -o p -- python bench.py --workload $W --codec $C --no-cpu-baseline --no-extra --no-verify --steps 5 --warmup 2 > gpurun_out/$R/bench_${C}_${W}.json 2> gpurun_out/$R/err.txt
  python profiles/extract_rocprof.py gpurun_out/$R/prof/p_results.db gpurun_out/$R/${C}_${W}_kernel_stats.csv
  rm -rf gpurun_out/$R/prof
  echo "== $C $W"; grep "k_ef\|k_packed\|k_scan\|k_fill\|k_count" gpurun_out/$R/${C}_${W}_kernel_stats.csv | sed 's/^"\?[^,]*\(k_[a-z0-9_]*\)[^,]*"\?,/\1,/' | cut -c1-90 | head -12
  python -c "import json; d=json.loads(open('gpurun_out/$R/bench_${C}_${W}.json').read().strip().split('\n')[-1]); print('   ms/step', round(d['ms_per_step'],4), 'kernel', d['kernel_ms'], 'frac', round(d['roofline']['frac'],4))"
done; done
