#!/usr/bin/env bash
# round 3, GPU call 3: does ANY 4-byte shift of the unguarded shading kernel's instruction stream restore reproducibility, and from which position on?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
{
for v in hz_none sh_0_1 sh_0_2 sh_0_16 sh_1000 sh_1900 sh_2100 sh_2300 sh_2500 sh_2700 sh_2900 sh_3200 sh_3600 sh_4000; do
  echo "== $v"; env SSDNERF_HIP_LIB=$R/.variants/$v/libssdnerf_hip.so RR_ONLY1=1 timeout 200 python tools/render_repeat.py 40 2>&1 | tail -1
done
} > $O/hz_shift.txt 2>&1
cat $O/hz_shift.txt
