#!/usr/bin/env bash
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_unet_fast_gpu.py tests/test_unet_golden.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python tools/bench_unet.py --modes fast --iters 30 2>&1 | tail -1
timeout 600 python tools/bench_unet.py --modes fast --iters 10 --layout tiled 2>&1 | tail -1
