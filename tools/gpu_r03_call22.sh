#!/usr/bin/env bash
# call 22: shade kernel VALU diet A/B (prescale in all): base | +gather pairs | +asm packed heads | +packed split | +k-step-1 packing | in-tree (all but pairs)
mkdir -p gpurun_out/r03
for v in base v1 v2 v3 v4 intree base v4; do
  if [ $v = intree ]; then unset SSDNERF_HIP_LIB; else export SSDNERF_HIP_LIB=.variants/$v/libssdnerf_hip.so; fi
  timeout 300 python bench.py --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', 'ms_per_step', round(d['ms_per_step'],3), 'shade_ms', r.get('kernel_ms', r.get('ms')), 'frac', round(r['frac'],4))"
done 2>&1 | tee gpurun_out/r03/h_shade_valu_diet.txt
unset SSDNERF_HIP_LIB
timeout 1500 python -m pytest tests/test_render_gpu.py tests/test_rows_gpu.py tests/test_golden.py tests/test_hip_ops_gpu.py -x -q -m gpu 2>&1 | tail -6
