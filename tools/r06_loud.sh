#!/usr/bin/env bash
# tools/r06_loud.sh [N=12] VARIANT[:ENV=VAL,...] ... -- repro_check of side builds on the GPU box (loud arrangements: seconds each)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06b; mkdir -p $O
N=12; case "$1" in [0-9]*) N=$1; shift;; esac
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  lib=$R/.variants/$v/libssdnerf_hip.so; [ "$v" = base ] && lib=$R/ssdnerf_amd/lib/libssdnerf_hip.so
  echo "== $spec"
  S=$SECONDS
  env SSDNERF_HIP_LIB=$lib $envs timeout 1800 python tools/repro_check.py $N 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-4} | cut -c1-420
  echo "wall $((SECONDS-S)) s"
done 2>&1 | tee -a $O/loud_$(date +%H%M%S).txt
