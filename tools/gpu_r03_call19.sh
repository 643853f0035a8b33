#!/usr/bin/env bash
# call 19: per-layer A/B, generalised kernels vs previous commit
mkdir -p gpurun_out/r03
for dt in bf16 fp32; do
SSDNERF_HIP_LIB=.variants/prev/libssdnerf_hip.so timeout 600 python tools/bench_conv.py --no-lib --dtype $dt --extra "128,64,128,3;128,128,64,3" > gpurun_out/r03/ab_prev_$dt.jsonl 2>&1
timeout 600 python tools/bench_conv.py --no-lib --dtype $dt --extra "128,64,128,3;128,128,64,3;128,24,128,3;128,128,24,3" > gpurun_out/r03/ab_new_$dt.jsonl 2>&1
tail -1 gpurun_out/r03/ab_prev_$dt.jsonl; tail -1 gpurun_out/r03/ab_new_$dt.jsonl
done
