#!/usr/bin/env bash
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_unet_fast_gpu.py -x -q -m gpu -k "f32x2" 2>&1 | tail -8
timeout 600 python tools/bench_conv.py --no-lib --dtype fp32 --hints 0,1,5,6 2>&1 | head -12 | cut -c1-220
timeout 600 python tools/bench_unet.py --modes fast --iters 30 2>&1 | tail -1
