#!/usr/bin/env bash
# tools/r06_trace.sh [N] VARIANT[:ENV=VAL,...] ... -- tools/trace_check.py on side builds (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06b; mkdir -p $O
N=7; case "$1" in [0-9]*) N=$1; shift;; esac
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  echo "== $spec"
  env SSDNERF_HIP_LIB=$R/.variants/$v/libssdnerf_hip.so $envs timeout 900 python tools/trace_check.py $N 2>&1 | grep -v amdgpu.ids | tail -40 | cut -c1-300
done 2>&1 | tee -a $O/trace_$(date +%H%M%S).txt
