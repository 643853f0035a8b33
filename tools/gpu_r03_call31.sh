#!/usr/bin/env bash
for i in 1 2; do
echo two_group_off $(SSDNERF_CONV_NO_TWO_GROUP=1 timeout 300 python tools/bench_unet.py --modes fast --dtypes fp32 --iters 40 2>&1 | tail -1 | cut -c1-120)
echo two_group_on  $(timeout 300 python tools/bench_unet.py --modes fast --dtypes fp32 --iters 40 2>&1 | tail -1 | cut -c1-120)
done
timeout 300 python tools/bench_finetune.py 2>&1 | tail -1
