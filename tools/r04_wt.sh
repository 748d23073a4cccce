cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_containers.py tests/test_gpu_full_configs.py tests/test_boundary.py -m gpu -q 2>&1 | tail -4
timeout 600 python tools/bench_wt.py 2>&1 | tail -4
