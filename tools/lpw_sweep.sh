#!/bin/bash
# lists per wavefront of the lane-per-list kernels (VIDC_LPW) on workloads with few / many lists per class
mkdir -p gpurun_out
for w in ${WORKLOADS:-uniform_16m uniform:16384:256 uniform:16384:1024 uniform_64m_1k uniform:8192:4000 c5}; do
  for lpw in 64 32 16 8 auto; do
    if [ $lpw = auto ]; then unset VIDC_LPW; else export VIDC_LPW=$lpw; fi
    python bench.py --workload $w --no-cpu-baseline --no-extra --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w lpw=$lpw', round(d['value']/1e9,3),'G ids/s', 'ms/step', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms'].items()})"
  done
done
