#!/usr/bin/env bash
# call 18: A/B of the cars UNet with the generalised convolution kernels against the previous commit's, same box
mkdir -p gpurun_out/r03
for i in 1 2; do
SSDNERF_HIP_LIB=.variants/prev/libssdnerf_hip.so timeout 600 python tools/bench_unet.py --modes fast --iters 30 2>&1 | tail -1 | sed 's/^/prev /'
timeout 600 python tools/bench_unet.py --modes fast --iters 30 2>&1 | tail -1 | sed 's/^/new  /'
done
timeout 900 python -m pytest tests/test_unet_fast_gpu.py tests/test_unet_golden.py -x -q -m gpu -k "partial or golden or reference_unet" 2>&1 | tail -4
