cd $GRAFT_REPO_ROOT
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/trace_u16.py 2>&1 | grep "offsets\|classify\|sync\|compaction\|^encode" | tail -9
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
for w in uniform_16m c5 s1_uniform; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --no-extra --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', round(d['ms_per_step'],3), d['kernel_ms'], 'host', round(d['host_ms_per_step'],3), round(d['value']/1e9,2), 'G ids/s', d['verified_roundtrip'])"
done
