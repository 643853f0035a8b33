#!/usr/bin/env bash
# call 17: partial channel tiles (multiples of 8): conv tests, tiled UNet golden with library_fallbacks == 0, tiled base-80 timing
mkdir -p gpurun_out/r03
python -m pytest tests/test_host_cpu.py -x -q 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_unet_fast_gpu.py tests/test_unet_golden.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python tools/bench_unet.py --layout tiled --iters 10 2>&1 | tee gpurun_out/r03/f_unet_tiled_base80.txt | tail -6
timeout 600 python tools/bench_unet.py --modes fast --iters 20 2>&1 | tail -2
