#!/usr/bin/env bash
# tools/r06_dual.sh [N] VARIANT[:ENV=VAL,...] ... -- tools/dual_check.py on side builds (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06b; mkdir -p $O
N=12; case "$1" in [0-9]*) N=$1; shift;; esac
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  echo "== $spec"
  env SSDNERF_HIP_LIB=$R/.variants/$v/libssdnerf_hip.so $envs timeout 900 python tools/dual_check.py $N 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-300
done 2>&1 | tee -a $O/dual_$(date +%H%M%S).txt
