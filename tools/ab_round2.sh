#!/usr/bin/env bash
# tools/ab_round2.sh [section ...] -- run on the GPU box (gpurun): the A/B measurements DESIGN.md section 6 lists as the first calls of the
# next round.  Every step has its own timeout and reads no stdin; results go to gpurun_out/ab/.  Sections: parity bench finetune sample ubench (default: all).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
OUT=$R/gpurun_out/ab
mkdir -p "$OUT"
SECTIONS="${@:-parity bench finetune sample ubench}"
GP=$R/.variants/gather_pairs/libssdnerf_hip.so
run() { name=$1; shift; echo "== $name" | tee -a "$OUT/log.txt"; ( timeout "${T:-120}" "$@" ) > "$OUT/$name.out" 2> "$OUT/$name.err" < /dev/null; echo "rc=$? $(tail -n 1 "$OUT/$name.out" | cut -c1-300)" | tee -a "$OUT/log.txt"; }
for s in $SECTIONS; do
  case $s in
    parity)
      T=200 run parity_rescheduled env SSDNERF_TEST_EXPERIMENTAL=1 python -m pytest tests/test_render_gpu.py -q -m gpu -k rescheduled -p no:cacheprovider
      [ -f "$GP" ] && T=300 run parity_gather_pairs env SSDNERF_HIP_LIB=$GP python -m pytest tests/test_render_gpu.py tests/test_golden.py tests/test_hip_ops_gpu.py -q -m gpu -x -p no:cacheprovider
      T=120 run parity_grad_att env SSDNERF_UNET_GRAD_GN=1 SSDNERF_UNET_GRAD_ATT=1 python -m pytest tests/test_unet_fast_gpu.py tests/test_diffusion_gpu.py -q -m gpu -k "input_gradient or val_optim or guidance_loss" -p no:cacheprovider
      ;;
    bench)
      B="python bench.py --no-cpu-baseline --steps 10 --warmup 3"
      T=120 run bench_default $B
      T=120 run bench_variant6 env SSDNERF_SHADE_VARIANT=6 $B
      T=120 run bench_compact env SSDNERF_FIRST_HIT_COMPACT=1 $B
      T=120 run bench_variant6_compact env SSDNERF_SHADE_VARIANT=6 SSDNERF_FIRST_HIT_COMPACT=1 $B
      if [ -f "$GP" ]; then
        T=120 run bench_pairs env SSDNERF_HIP_LIB=$GP $B
        T=120 run bench_pairs_variant6_compact env SSDNERF_HIP_LIB=$GP SSDNERF_SHADE_VARIANT=6 SSDNERF_FIRST_HIT_COMPACT=1 $B
      fi
      ;;
    finetune)
      T=150 run finetune_default python tools/bench_finetune.py
      T=150 run finetune_grad_gn env SSDNERF_UNET_GRAD_GN=1 python tools/bench_finetune.py
      T=150 run finetune_grad_gn_att env SSDNERF_UNET_GRAD_GN=1 SSDNERF_UNET_GRAD_ATT=1 python tools/bench_finetune.py
      T=150 run finetune_bf16_default python tools/bench_finetune.py --dtype bf16
      T=150 run finetune_bf16_grad_conv env SSDNERF_UNET_GRAD_CONV_BF16=1 python tools/bench_finetune.py --dtype bf16
      ;;
    sample)
      T=150 run sample_bf16 python tools/bench_sample.py --dtype bf16
      T=150 run sample_fp32 python tools/bench_sample.py --dtype fp32
      ;;
    ubench)
      [ -x .variants/trans_rate ] && T=60 run ubench_trans_rate .variants/trans_rate
      ;;
  esac
done
cat "$OUT/log.txt"
