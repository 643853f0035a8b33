#!/usr/bin/env bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
for i in 1 2; do timeout 300 python tools/bench_unet.py --modes fast --dtypes fp32,bf16 --iters 30 2>&1 | grep -v amdgpu.ids | tail -1; done | tee $O/unet_c.json
SSDNERF_CONV_NO_TWO_GROUP=1 timeout 300 python tools/bench_unet.py --modes fast --dtypes bf16 --iters 30 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/unet_c.json
timeout 300 python tools/bench_unet.py --modes fast --dtypes bf16 --iters 5 --no-graph --profile fast:bf16 2>&1 | grep -v amdgpu.ids > $O/unet_profile_bf16_b.txt; grep "anonymous\|Self CUDA time" $O/unet_profile_bf16_b.txt | cut -c1-90,150-200 | head -16
timeout 300 python tools/bench_unet.py --modes fast --dtypes fp32 --iters 5 --no-graph --profile fast:fp32 2>&1 | grep -v amdgpu.ids > $O/unet_profile_fp32_b.txt; grep "anonymous\|Self CUDA time" $O/unet_profile_fp32_b.txt | cut -c1-90,150-200 | head -16
