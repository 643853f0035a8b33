# r06: GroupNorm backward kernels, A/B of side builds (tools/bf16_grad_probe.py --profile --kernels k_gn): per-call totals on the cars UNet, 8 scenes, both arithmetic classes
mkdir -p gpurun_out/r06y
out=gpurun_out/r06y/gn_ab.txt
: > $out
for v in shipped gn_u2 gn_r64 gn_u2r64 gn_b4096; do
  echo "== $v" >> $out
  if [ $v = shipped ]; then unset SSDNERF_HIP_LIB; else export SSDNERF_HIP_LIB=.variants/$v/libssdnerf_hip.so; fi
  timeout 300 python tools/bf16_grad_probe.py --profile --kernels k_gn_bwd,k_gn_apply --iters 10 2>&1 | grep -v "Warning\|warn\|amdgpu.ids" | grep "k_gn\|input-gradient call\|second eager" >> $out
done
cat $out
