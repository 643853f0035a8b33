#!/usr/bin/env bash
# tools/r06_final2.sh -- second closing session of round 6 (the gradient-path work came after tools/r06_final.sh): a soak of the shipped build, the profiling session
# (rocprofv3 kernel trace + PMC passes, tools/prof_render.sh) with traffic_latest.json bound to THIS commit, the full bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06g; mkdir -p $O
C=${1:-HEAD}
{ echo "== shipped build $C (30 000 renders)"; S=$SECONDS; timeout 1200 python tools/repro_check.py 30000 2>&1 | grep -v amdgpu.ids | tail -6; echo "wall $((SECONDS-S)) s"; } > $O/soak.txt 2>&1
cat $O/soak.txt | cut -c1-300
SSDNERF_PROFILED_COMMIT=$C SSDNERF_PROFILE_ROUND=r06 bash tools/prof_render.sh zz > $O/prof.log 2>&1; tail -5 $O/prof.log | cut -c1-300
python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -c 400 $O/bench_full.json
