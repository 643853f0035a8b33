#!/usr/bin/env bash
# round 3, GPU call 7: stage-ordered SiLU heads + post-pass at 2 / 3 / 4 wait states: reproducibility, speed, parity
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
bash tools/ab_shade.sh ws3 ws2 nopost hg2 hg8 > $O/silu_ab.txt 2>&1; cat $O/silu_ab.txt
{
for v in in-tree ws3 ws2 nopost; do
  if [ $v = in-tree ]; then L="SSDNERF_DUMMY=1"; else L="SSDNERF_HIP_LIB=$R/.variants/$v/libssdnerf_hip.so"; fi
  echo "== $v"; env $L timeout 300 python tools/render_repeat.py ${RR_N:-200} 2>&1 | tail -2
done
} > $O/silu_repeat.txt 2>&1
cat $O/silu_repeat.txt
timeout 900 python -m pytest tests/test_render_gpu.py tests/test_rows_gpu.py tests/test_golden.py -x -q -m gpu > $O/test_c.log 2>&1; echo "tests rc=$?"; tail -3 $O/test_c.log
