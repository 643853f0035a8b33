#!/usr/bin/env bash
# tools/build_variants.sh "name -Dflag ..." ["name2 -Dflag ..." ...] -- side builds of shade_mfma.hip for A/B runs, in parallel, after ONE in-tree build
# (build_variant() rebuilds the in-tree library when a source changed; concurrent rebuilds would trample each other).
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
python -m ssdnerf_amd.build > /dev/null || exit 1
for v in "$@"; do set -- $v; n=$1; shift; (tools/build_variant.sh $n ${AB_SOURCE:-shade_mfma.hip} "$@" 2>&1 | tail -1 | cut -c1-150) & done; wait
