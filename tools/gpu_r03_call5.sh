#!/usr/bin/env bash
# round 3, GPU call 5: transcendental -> use wait states: micro-benchmark, and the unguarded shading kernel with that pad lengthened IN PLACE
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
( timeout 300 .variants/trans_use_hazard 2000 ) > $O/trans_use_ubench.txt 2>&1; echo "ubench rc=$?"
{
for v in hz_none ti2 ti2_s16 ti2r ti2ra ti2rb ti2rc ti2rd t2 t2_s16 t2_s4; do
  echo -n "$v: "; env SSDNERF_HIP_LIB=$R/.variants/$v/libssdnerf_hip.so RR_ONLY1=1 timeout 200 python tools/render_repeat.py 40 2>&1 | tail -1
done
} > $O/hz_trans.txt 2>&1
cat $O/hz_trans.txt; grep -v "     0     0     0     0     0     0     0     0     0     0     0     0     0     0     0     0$" $O/trans_use_ubench.txt | head -80
