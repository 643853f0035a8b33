#!/usr/bin/env bash
# tools/r06_final.sh -- the closing session of round 6 on the shipped build: soak (with the positive control beside it), profiling session, full bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06f; mkdir -p $O
soak() { l=$1; n=$2; shift 2; echo "== $l ($n renders) $*"; S=$SECONDS; env "$@" timeout 1700 python tools/repro_check.py $n 2>&1 | grep -v amdgpu.ids | tail -30; echo "wall $((SECONDS-S)) s"; }
{
soak shipped_build_1fe0acb 100000 SSDNERF_DUMMY=0
soak positive_control_same_source_compilers_packed_instructions_kept 30000 SSDNERF_HIP_LIB=$R/.variants/U/libssdnerf_hip.so
} > $O/soak.txt 2>&1
cat $O/soak.txt | cut -c1-300
SSDNERF_PROFILED_COMMIT=1fe0acb SSDNERF_PROFILE_ROUND=r06 bash tools/prof_render.sh z > $O/prof.log 2>&1; tail -5 $O/prof.log | cut -c1-300
python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -c 600 $O/bench_full.json
