#!/usr/bin/env bash
for i in 1 2; do
for v in gn_ur1 gn_ur2 intree; do
  if [ $v = intree ]; then unset SSDNERF_HIP_LIB; else export SSDNERF_HIP_LIB=.variants/$v/libssdnerf_hip.so; fi
  echo $v $(timeout 300 python tools/bench_unet.py --modes fast --iters 40 2>&1 | tail -1 | cut -c1-200)
done; done
