#!/usr/bin/env bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_unet_fast_gpu.py -x -q -m gpu -k "conv_igemm or concat" > $O/test_pp.log 2>&1; echo "tests rc=$?"; tail -3 $O/test_pp.log
timeout 200 python tools/bench_conv_few.py 0 5 6 2>&1 | grep -v amdgpu.ids | tee $O/pp_few.txt
