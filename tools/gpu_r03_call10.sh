#!/usr/bin/env bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
for v in in-tree skip_epi skip_k; do
  if [ $v = in-tree ]; then L="SSDNERF_DUMMY=1"; else L="SSDNERF_HIP_LIB=$R/.variants/$v/libssdnerf_hip.so"; fi
  echo "== $v"; env $L timeout 200 python tools/bench_conv_few.py 5 6 2>&1 | grep -v amdgpu.ids
done > $O/pp_split.txt 2>&1
cat $O/pp_split.txt
