#!/usr/bin/env bash
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_unet_fast_gpu.py tests/test_unet_golden.py tests/test_render_gpu.py tests/test_rows_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/bench_unet.py --modes fast --iters 30 2>&1 | tail -1
bash tools/prof_step.sh i_headskip3 | head -6
