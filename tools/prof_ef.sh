# rocprofv3 kernel stats of the Elias-Fano and packed-bits benches -> gpurun_out/<tag>/{ef,packed}_<workload>_kernel_stats.csv
# usage: bash tools/prof_ef.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=${1:-efp}; mkdir -p gpurun_out/$R
for C in ef packed; do for W in s1 uniform_16m uniform_64m_1k; do
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/$R/prof -o p -- python bench.py --workload $W --codec $C --no-cpu-baseline --no-extra --steps 5 --warmup 2 > /dev/null 2> gpurun_out/$R/err.txt
  python profiles/extract_rocprof.py gpurun_out/$R/prof/p_results.db gpurun_out/$R/${C}_${W}_kernel_stats.csv
  rm -rf gpurun_out/$R/prof
  echo "== $C $W"; grep "k_ef\|k_packed\|k_scan\|k_fill\|k_count\|rocclr" gpurun_out/$R/${C}_${W}_kernel_stats.csv | cut -c1-70,150-400 | sed 's/,"[^"]*$//' | head -12
done; done
