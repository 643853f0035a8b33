#!/usr/bin/env bash
# round 3, GPU call 4: bisect of the position from which a 4-byte shift no longer restores reproducibility (argument: list of variant names)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
{
for v in "$@"; do
  echo -n "$v: "; env SSDNERF_HIP_LIB=$R/.variants/$v/libssdnerf_hip.so RR_ONLY1=1 timeout 200 python tools/render_repeat.py ${RR_N:-24} 2>&1 | tail -1
done
} >> $O/hz_bisect.txt 2>&1
tail -n $# $O/hz_bisect.txt
