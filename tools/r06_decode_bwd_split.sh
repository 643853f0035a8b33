# r06: the decode backward's binned reduction: what the LDS float atomics cost (side builds -DDB_EXP_NO_ATOMICS / -DDB_EXP_SCAN_ONLY of the r05 kernel, run with
# SSDNERF_DECODE_BWD_ATOMICS=1) and the ownership kernel that replaces them; tools/bench_decode_bwd.py: 8 scenes x 875 000 samples, uniform / ray-ordered / one cone of rays
for arm in "owned:" "atomics:SSDNERF_DECODE_BWD_ATOMICS=1" "atomics_plain_adds_instead(racy,measurement):SSDNERF_DECODE_BWD_ATOMICS=1 SSDNERF_HIP_LIB=.variants/db_noatom/libssdnerf_hip.so" "atomics_scan_only(measurement):SSDNERF_DECODE_BWD_ATOMICS=1 SSDNERF_HIP_LIB=.variants/db_scanonly/libssdnerf_hip.so"; do
  name=${arm%%:*}; envs=${arm#*:}
  for mode in "" "--ray-like" "--ray-like --cone"; do
    [ -n "$envs" ] && [ ! -e "$(echo $envs | sed -n 's/.*SSDNERF_HIP_LIB=\([^ ]*\).*/\1/p')" ] && [ -n "$(echo $envs | grep HIP_LIB)" ] && continue
    echo -n "$name $mode: "; env $envs python tools/bench_decode_bwd.py --samples 875000 $mode 2>&1 | grep kernels_ms | cut -c1-250
  done
done
