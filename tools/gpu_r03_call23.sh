#!/usr/bin/env bash
mkdir -p gpurun_out/r03
for i in 1 2; do timeout 300 python bench.py --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('shipped', 'ms_per_step', round(d['ms_per_step'],3), 'frac', round(r['frac'],4), 'achieved', r['achieved'], d.get('boundary_rays'))"; done
SSDNERF_HIP_LIB=.variants/base/libssdnerf_hip.so timeout 300 python bench.py --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('base', 'ms_per_step', round(d['ms_per_step'],3), 'frac', round(r['frac'],4))"
timeout 1500 python -m pytest tests/test_render_gpu.py tests/test_rows_gpu.py tests/test_golden.py tests/test_hip_ops_gpu.py tests/test_decode_gpu.py -q -m gpu 2>&1 | tail -8
