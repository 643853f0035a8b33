#!/usr/bin/env bash
# round 3, GPU call 6: the library built with the transcendental post-pass and WITHOUT the r02 scheduling barriers: reproducibility, speed, parity
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
{
for v in in-tree legacy nopost ws3; do
  if [ $v = in-tree ]; then L="SSDNERF_DUMMY=1"; else L="SSDNERF_HIP_LIB=$R/.variants/$v/libssdnerf_hip.so"; fi
  echo "== $v"; env $L timeout 300 python tools/render_repeat.py ${RR_N:-120} 2>&1 | tail -2
done
} > $O/hz_final_repeat.txt 2>&1
cat $O/hz_final_repeat.txt
bash tools/ab_shade.sh legacy nopost ws3 > $O/hz_final_ab.txt 2>&1; cat $O/hz_final_ab.txt
timeout 300 python tools/kernel_repeat.py 40 > $O/kernel_repeat.txt 2>&1; tail -12 $O/kernel_repeat.txt
timeout 900 python -m pytest tests/test_render_gpu.py tests/test_rows_gpu.py tests/test_hip_ops_gpu.py tests/test_unet_fast_gpu.py -x -q -m gpu > $O/test_b.log 2>&1; echo "tests rc=$?"; tail -3 $O/test_b.log
