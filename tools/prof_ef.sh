cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/efp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/efp/prof -o ef -- python bench.py --workload uniform_64m_1k --codec ef --no-cpu-baseline --no-extra --steps 5 --warmup 2 > /dev/null 2> gpurun_out/efp/err.txt
python profiles/extract_rocprof.py gpurun_out/efp/prof/ef_results.db gpurun_out/efp/ef_stats.csv
rm -rf gpurun_out/efp/prof
grep -i "vidc\|rocclr" gpurun_out/efp/ef_stats.csv | cut -c1-150
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/efp/prof -o pk -- python bench.py --workload uniform_64m_1k --codec packed --no-cpu-baseline --no-extra --steps 5 --warmup 2 > /dev/null 2> gpurun_out/efp/err.txt
python profiles/extract_rocprof.py gpurun_out/efp/prof/pk_results.db gpurun_out/efp/pk_stats.csv
rm -rf gpurun_out/efp/prof
grep -i "vidc\|rocclr" gpurun_out/efp/pk_stats.csv | cut -c1-150
