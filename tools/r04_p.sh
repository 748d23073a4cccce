cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_roc.py -m gpu -q 2>&1 | tail -3
timeout 300 python tools/fuzz_families.py 51 60 2>&1 | tail -1
for w in uniform_16m uniform_64m_1k; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --no-extra --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step'],3), d['kernel_ms'], round(d['value']/1e9,2), 'G ids/s', d['verified_roundtrip'])"
done
timeout 900 python tools/s2_sched.py "E:" "D:" "X:VIDC_NO_LANE128=1" "D:" "X:VIDC_NO_LANE128=0" "D:" "E:" 2>&1 | tail -8
