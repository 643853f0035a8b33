#!/usr/bin/env bash
# round 3, GPU call 1: wait-state micro-benchmark, asm-patched side builds of the shading kernel under the repeat test, the new bench-shape UNet
# parity tests, and the full bench line (baseline of the round).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
( timeout 120 .variants/swap_mfma_hazard 3000 ) > $O/hazard_ubench.txt 2>&1
echo "ubench rc=$?"
{
for v in in-tree hz_none hz_raw4 hz_raw5 hz_raw6 hz_raw8 hz_swap6 hz_valu6 hz_swapraw4; do
  if [ $v = in-tree ]; then L=""; else L="SSDNERF_HIP_LIB=$R/.variants/$v/libssdnerf_hip.so"; fi
  echo "== $v"; env $L RR_ONLY1=1 timeout 200 python tools/render_repeat.py ${RR_N:-40} 2>&1 | tail -2
done
} > $O/hz_variants.txt 2>&1
cat $O/hz_variants.txt
timeout 900 python -m pytest tests/test_unet_fast_gpu.py tests/test_unet_golden.py -x -q -m gpu > $O/test_unet.log 2>&1; echo "unet tests rc=$?"; tail -3 $O/test_unet.log
timeout 600 python -m pytest tests/test_diffusion_gpu.py tests/test_rows_gpu.py -x -q -m gpu > $O/test_diff.log 2>&1; echo "diffusion/rows tests rc=$?"; tail -3 $O/test_diff.log
timeout 900 python bench.py > $O/bench_a.json 2> $O/bench_a.err; echo "bench rc=$?"; tail -5 $O/bench_a.err
head -c 1500 $O/hazard_ubench.txt
