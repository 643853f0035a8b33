#!/usr/bin/env bash
# tools/prof_decode_bwd.sh -- per-kernel times of the decode backward (GPU box): in-tree library and any .variants/db_* side builds
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for mode in "" "--ray-like" "--ray-like --cone"; do
for lib in in-tree $(ls -d .variants/db_* 2>/dev/null); do
  [ "$lib" = in-tree ] && unset SSDNERF_HIP_LIB || export SSDNERF_HIP_LIB=$R/$lib/libssdnerf_hip.so
  rm -rf /tmp/rp_db
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_db -o db -- python tools/bench_decode_bwd.py $mode 2>/dev/null | grep samples_total
  f=$(find /tmp/rp_db -name "*kernel_stats.csv" | head -1)
  grep "k_decode_bwd\|k_point_decode" $f | awk -F'",' '{print substr($1,1,40), $2}' | sed 's/^/    /'
done; done
