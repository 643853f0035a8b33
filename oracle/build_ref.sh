#!/usr/bin/env bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE.  Builds oracle/_ref/libssdnerf_ref_{fma,nofma}.so:
# the reference's OWN kernels (raymarching.cu, shencoder.cu), compiled for the host CPU from the
# sources where they lie under $REF (default /root/reference).  Nothing from the reference is
# copied into the repo: the only rewrite (CUDA's `k<<<g,b>>>(a)` launch syntax, which no C++
# compiler parses, -> CPU_LAUNCH(g,b,k,a)) is applied by sed on a pipe straight into g++.
# Outputs go only into oracle/_ref/ (git-ignored; travels to the GPU box with the snapshot).
#   _fma   : -mfma -ffp-contract=fast  (mirrors nvcc's default -fmad=true contraction policy)
#   _nofma : -ffp-contract=off         (to measure how much of the result depends on contraction)
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${REF:-/root/reference}"
RM="$REF/lib/ops/raymarching/src"; SH="$REF/lib/ops/shencoder/src"
[ -f "$RM/raymarching.cu" ] || { echo "reference not present at $REF - keeping prebuilt oracle/_ref" >&2; exit 0; }
mkdir -p "$HERE/_ref"
LAUNCH_SED='s/\([A-Za-z_0-9]*\(<scalar_t>\)\{0,1\}\)<<<\(.*\)>>>(/CPU_LAUNCH(\3, \1, /'
build() {  # $1 = tag, rest = flags
  tag="$1"; shift
  tmp="$(mktemp -d)"
  for src in "$RM/raymarching.cu" "$SH/shencoder.cu"; do
    sed "$LAUNCH_SED" "$src" | g++ -x c++ -std=c++17 -O2 -fPIC -w "$@" -I"$HERE/ref_shim" -I"$RM" -I"$SH" -c - -o "$tmp/$(basename "$src").o"
  done
  g++ -std=c++17 -O2 -fPIC -w "$@" -I"$HERE" -I"$HERE/ref_shim" -I"$RM" -I"$SH" -c "$HERE/ref_glue.cpp" -o "$tmp/glue.o"
  g++ -shared -o "$HERE/_ref/libssdnerf_ref_$tag.so" "$tmp"/*.o -lm
  rm -rf "$tmp"
}
build fma   -mfma -ffp-contract=fast
build nofma -ffp-contract=off
echo "built: $(ls "$HERE/_ref")"
