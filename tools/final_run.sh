set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01d
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r01d/pytest_gpu.txt
VIDC_FORCE_GENERAL=1 timeout 900 python -m pytest tests/test_gpu_roc.py -m gpu -q 2>&1 | tail -3 > gpurun_out/r01d/pytest_force_general.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r01d/smoke.txt 2>&1
timeout 600 python bench.py > gpurun_out/r01d/bench.json 2> gpurun_out/r01d/bench.err
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r01d/prof -o s1 -- python bench.py --no-cpu-baseline --no-extra > gpurun_out/r01d/bench_prof.json 2> gpurun_out/r01d/prof.err
ls -R gpurun_out/r01d/prof | head -20
cat gpurun_out/r01d/pytest_gpu.txt gpurun_out/r01d/pytest_force_general.txt gpurun_out/r01d/smoke.txt
cat gpurun_out/r01d/bench.json
