#!/usr/bin/env bash
mkdir -p gpurun_out/r03
pr() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', 'ms_per_step', round(d['ms_per_step'],3), 'frac', round(r['frac'],4), {k:v for k,v in d.items() if 'stage' in k or 'first' in k})"; }
for i in 1 2; do
SSDNERF_MARCH_NO_LDS=1 timeout 300 python bench.py --no-extras 2>/dev/null | tail -1 | pr global
timeout 300 python bench.py --no-extras 2>/dev/null | tail -1 | pr lds
done
timeout 1500 python -m pytest tests/test_render_gpu.py tests/test_rows_gpu.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -3
