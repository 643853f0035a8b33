#!/usr/bin/env bash
# tools/r06_call2.sh -- round 6: the loud arrangements with the crossed packed instructions split, A/B timing, and the soaks (positive control = the compiler's packed instructions kept)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06c; mkdir -p $O
V=$R/.variants
TAILN=3 bash tools/r06_loud.sh 12 WEx Q4S1x Q4S1d4x WEx1 Q4S1d4x1 Q4x1 2>&1 | grep -v "^render" > $O/a_loud_fixed.txt; cat $O/a_loud_fixed.txt
AB_REPEAT=2 AB_SKIP_1WAVE=1 bash tools/ab_shade.sh U NS NSS1 NT0 > $O/b_ab.txt 2>&1; cat $O/b_ab.txt
soak() { l=$1; n=$2; shift 2; echo "== $l ($n renders) $*"; S=$SECONDS; env "$@" timeout 1700 python tools/repro_check.py $n 2>&1 | grep -v amdgpu.ids | tail -30; echo "wall $((SECONDS-S)) s"; }
{
soak U_compilers_packed_instructions_kept 30000 SSDNERF_HIP_LIB=$V/U/libssdnerf_hip.so
soak NT0_split_only_no_padding_rules 60000 SSDNERF_HIP_LIB=$V/NT0/libssdnerf_hip.so
soak NSS1_split_trans4_fp32swap 40000 SSDNERF_HIP_LIB=$V/NSS1/libssdnerf_hip.so
} > $O/c_soak.txt 2>&1
cat $O/c_soak.txt | cut -c1-400
