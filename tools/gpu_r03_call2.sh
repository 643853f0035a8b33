#!/usr/bin/env bash
# round 3, GPU call 2: wait-state micro-benchmark with a stressing partner wave; per-site bisect of the four VALU -> MFMA pairs at 3 slots
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
( timeout 300 .variants/swap_mfma_hazard_stress 2000 ) > $O/hazard_ubench_stress.txt 2>&1; echo "ubench rc=$?"
{
for v in hz_none hs_only0 hs_only1 hs_only2 hs_only3 hs_but0 hs_but1 hs_but2 hs_but3; do
  echo "== $v"; env SSDNERF_HIP_LIB=$R/.variants/$v/libssdnerf_hip.so RR_ONLY1=1 timeout 200 python tools/render_repeat.py 40 2>&1 | tail -1
done
echo "== hz_raw4 x 300 (1 scene) and 8 scenes"; env SSDNERF_HIP_LIB=$R/.variants/hz_raw4/libssdnerf_hip.so timeout 300 python tools/render_repeat.py 300 2>&1 | tail -2
} > $O/hz_sites.txt 2>&1
cat $O/hz_sites.txt; grep -c . $O/hazard_ubench_stress.txt
