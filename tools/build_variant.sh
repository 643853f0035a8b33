#!/usr/bin/env bash
# tools/build_variant.sh <name> <file.hip> [-D flags...] -- side build for A/B runs (build container, no GPU needed): recompiles ONE source of
# ssdnerf_amd/csrc with extra flags THROUGH THE BUILD'S POST-PASS (ssdnerf_amd.build.build_variant) and links it with the in-tree objects into
# .variants/<name>/libssdnerf_hip.so (git-ignored, shipped by gpurun).  Use on the GPU box with SSDNERF_HIP_LIB=.variants/<name>/libssdnerf_hip.so.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
cd $R && python -m ssdnerf_amd.build --variant "$NAME" --source "$SRC" -- "$@"
