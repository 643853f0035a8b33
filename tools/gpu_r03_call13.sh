#!/usr/bin/env bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_unet_fast_gpu.py tests/test_unet_golden.py -x -q -m gpu > $O/test_unet2.log 2>&1; echo "tests rc=$?"; tail -3 $O/test_unet2.log
timeout 300 python tools/bench_unet.py --modes fast --dtypes fp32,bf16 --iters 20 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/unet_a.json
timeout 300 python tools/bench_unet.py --modes fast --dtypes bf16 --iters 5 --no-graph --profile fast:bf16 2>&1 | grep -v amdgpu.ids > $O/unet_profile_bf16.txt; head -45 $O/unet_profile_bf16.txt | cut -c1-200
