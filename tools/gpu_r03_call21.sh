#!/usr/bin/env bash
# call 21: SiLU scaling folded into the weights: render parity tests + bench
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_render_gpu.py tests/test_rows_gpu.py tests/test_golden.py tests/test_hip_ops_gpu.py -x -q -m gpu 2>&1 | tail -6
timeout 600 python bench.py --no-extras 2>&1 | tail -1 | cut -c1-900
