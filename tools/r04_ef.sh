cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_packed_ef.py -m gpu -q 2>&1 | tail -3
bash tools/prof_ef_s2.sh r04h 2>&1 | grep "ms/step\|^==\|k_ef_build"
