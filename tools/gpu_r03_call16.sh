#!/usr/bin/env bash
# call 16: (f)4 train-step parity, Option A on hardware, conv tests for the buffer-load port
mkdir -p gpurun_out/r03
python -m pytest tests/test_host_cpu.py -x -q 2>&1 | tail -2
timeout 600 python -m pytest tests/test_diffusion_gpu.py -x -q -m gpu -k "train_step" 2>&1 | tail -15
timeout 300 python tools/option_a_on_gpu.py 2>&1 | tee gpurun_out/r03/option_a.txt | tail -12
timeout 900 python -m pytest tests/test_unet_fast_gpu.py -x -q -m gpu 2>&1 | tail -4
