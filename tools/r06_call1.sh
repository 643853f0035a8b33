#!/usr/bin/env bash
# tools/r06_call1.sh -- round 6, GPU call 1: transcendental-rate microbenchmark, A/B of the side builds, the loud arrangements with and without the swap rule,
# and the soaks (positive control without the swap rule, the fp32-swap form, the r05 shipped form).  Everything lands in gpurun_out/r06a/.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06a; mkdir -p $O
export TMPDIR=/tmp
V=$R/.variants
rocm-smi --showclocks > $O/clocks.txt 2>&1
$V/trans_rate > $O/a_trans_rate.txt 2>&1
timeout 600 python -m pytest tests/test_render_gpu.py -x -q > $O/test_render_gpu.txt 2>&1; tail -3 $O/test_render_gpu.txt
AB_REPEAT=2 AB_SKIP_1WAVE=1 bash tools/ab_shade.sh U S1R G8 A1 A1S1 > $O/b_ab.txt 2>&1; cat $O/b_ab.txt
for v in Q4 Q4S Q4S1 WE WES WES1; do
  echo "== $v"; SSDNERF_HIP_LIB=$V/$v/libssdnerf_hip.so timeout 300 python tools/repro_check.py 12 2>&1 | tail -4
done > $O/c_loud.txt 2>&1
cat $O/c_loud.txt | cut -c1-300
soak() { # label, renders, env...
  l=$1; n=$2; shift 2
  echo "== $l ($n renders) $*"; /usr/bin/time -f "wall %e s" env "$@" timeout 1500 python tools/repro_check.py $n 2>&1 | tail -40
}
{
soak U_noswaprule 40000 SSDNERF_HIP_LIB=$V/U/libssdnerf_hip.so
soak S1_fp32swap_norule 80000 SSDNERF_HIP_LIB=$V/S1/libssdnerf_hip.so
soak H_r05_shipped_swaprule8 40000 SSDNERF_DUMMY=0
soak U_noswaprule_1wave_per_simd 15000 SSDNERF_HIP_LIB=$V/U/libssdnerf_hip.so SSDNERF_SHADE_BLOCKS_PER_CU=1
} > $O/d_soak.txt 2>&1
cat $O/d_soak.txt | cut -c1-400
