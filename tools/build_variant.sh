#!/usr/bin/env bash
# tools/build_variant.sh <name> <file.hip> [-D flags...] -- side build for A/B runs (build container, no GPU needed): recompiles ONE source of
# ssdnerf_amd/csrc with extra flags and links it with the in-tree objects into .variants/<name>/libssdnerf_hip.so (git-ignored, shipped by
# gpurun).  Use on the GPU box with SSDNERF_HIP_LIB=.variants/<name>/libssdnerf_hip.so.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
[ -n "$SSDNERF_SKIP_BUILD" ] || python -m ssdnerf_amd.build > /dev/null 2>&1
mkdir -p $R/.variants/$NAME
OBJ=$R/.variants/$NAME/${SRC%.hip}.o
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -Wno-shift-op-parentheses -c $R/ssdnerf_amd/csrc/$SRC -o $OBJ
OBJS=""
for o in $R/ssdnerf_amd/lib/*.o; do
  [ "$(basename $o)" = "${SRC%.hip}.o" ] && OBJS="$OBJS $OBJ" || OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $R/.variants/$NAME/libssdnerf_hip.so $OBJS
rm -f $OBJ
echo "built .variants/$NAME/libssdnerf_hip.so ($*)"
