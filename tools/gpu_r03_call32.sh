#!/usr/bin/env bash
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_unet_fast_gpu.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python tools/bench_conv.py --no-lib --dtype bf16 --hints 0,1,5,6 2>&1 | head -7 | cut -c1-200
timeout 600 python tools/bench_unet.py --modes fast --iters 30 2>&1 | tail -1
