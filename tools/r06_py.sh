#!/usr/bin/env bash
# tools/r06_py.sh SCRIPT [N] VARIANT[:ENV=VAL,...] ... -- run tools/SCRIPT.py N on side builds (GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06b; mkdir -p $O
SC=$1; shift
N=12; case "$1" in [0-9]*) N=$1; shift;; esac
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  lib=$R/.variants/$v/libssdnerf_hip.so; [ "$v" = base ] && lib=$R/ssdnerf_amd/lib/libssdnerf_hip.so
  echo "== $spec"
  env SSDNERF_HIP_LIB=$lib $envs timeout 1800 python tools/$SC.py $N 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-40} | cut -c1-400
done 2>&1 | tee -a $O/${SC}_$(date +%H%M%S).txt
