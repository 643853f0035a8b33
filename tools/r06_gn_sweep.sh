# r06: GroupNorm backward, sweep of (rows per trip, least blocks) side builds over the cars UNet's norm shapes (tools/bench_gn_bwd.py), then the GPU tests of the kernel
# variants: python -m ssdnerf_amd.build --variant gn_u2b1024 --source groupnorm.hip -- -DGN_BWD_UNROLL=2   (u: rows per trip, b: GN_BWD_STATS_BLOCKS = GN_BWD_APPLY_BLOCKS, t: GN_BWD_MIN_TRIPS)
mkdir -p gpurun_out/r06y
out=gpurun_out/r06y/gn_sweep.txt
: > $out
for v in shipped gn_u4b2048 gn_u2b1024 gn_u4b512 gn_u4b1024t2; do
  echo "== $v" >> $out
  if [ $v = shipped ]; then unset SSDNERF_HIP_LIB; else export SSDNERF_HIP_LIB=.variants/$v/libssdnerf_hip.so; fi
  timeout 300 python tools/bench_gn_bwd.py 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r06y/gn_shapes_$v.txt
  grep "all norms" gpurun_out/r06y/gn_shapes_$v.txt >> $out
done
unset SSDNERF_HIP_LIB
cat $out
cat gpurun_out/r06y/gn_shapes_shipped.txt
timeout 600 python -m pytest tests/test_unet_fast_gpu.py -x -q -m gpu -k "group_norm or norms" 2>&1 | tail -3
