#!/usr/bin/env bash
# tools/asm_patch_build.sh <name> <file.hip> <mode> <arg> [-D flags...] -- side build (no GPU needed) whose DEVICE code is the compiler's own listing
# of ssdnerf_amd/csrc/<file.hip> edited by tools/asm_patch.py <mode> <arg> (idle issue slots at named places, schedule otherwise untouched):
#   hipcc -S --cuda-device-only -> asm_patch.py -> clang (assembler) -> lld -> clang-offload-bundler -> host object with that fat binary ->
#   .variants/<name>/libssdnerf_hip.so (linked with the in-tree objects).  Use on the GPU box with SSDNERF_HIP_LIB=.variants/<name>/libssdnerf_hip.so.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; MODE=$3; ARG=$4; shift 4
# modes with a second argument (site list): pass it as "N:SITES"
ARG2=""; case "$ARG" in *:*) ARG2="${ARG#*:}"; ARG="${ARG%%:*}";; esac
LL=/opt/rocm/lib/llvm/bin
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -I$R/ssdnerf_amd/csrc -I$R/include"
[ -n "$SSDNERF_SKIP_BUILD" ] || python -m ssdnerf_amd.build > /dev/null 2>&1
D=$R/.variants/$NAME; mkdir -p $D; T=$(mktemp -d)
/opt/rocm/bin/hipcc "$@" $FLAGS -S --cuda-device-only $R/ssdnerf_amd/csrc/$SRC -o $T/dev.s 2> /dev/null
python $R/tools/asm_patch.py $T/dev.s $T/dev_p.s $MODE $ARG $ARG2 | grep -v "__half" | tail -8
if [ -n "$PATCH2" ]; then mv $T/dev_p.s $T/dev_q.s; python $R/tools/asm_patch.py $T/dev_q.s $T/dev_p.s $PATCH2 | tail -1; fi   # PATCH2="mode arg [arg2]": a second pass
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/dev_p.s -o $T/dev.o
$LL/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -plugin-opt=-amdgpu-internalize-symbols -plugin-opt=mcpu=gfx950 -o $T/dev.out $T/dev.o
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev.out -output=$T/dev.hipfb
OBJ=$D/${SRC%.hip}.o
/opt/rocm/bin/hipcc "$@" $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -c $R/ssdnerf_amd/csrc/$SRC -o $OBJ 2> /dev/null
OBJS=""
for o in $R/ssdnerf_amd/lib/*.o; do
  [ "$(basename $o)" = "${SRC%.hip}.o" ] && OBJS="$OBJS $OBJ" || OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $D/libssdnerf_hip.so $OBJS
cp $T/dev_p.s $D/${SRC%.hip}.s
rm -rf $T $OBJ
echo "built .variants/$NAME/libssdnerf_hip.so ($SRC, asm_patch $MODE $ARG, $*)"
