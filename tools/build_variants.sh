#!/usr/bin/env bash
# tools/build_variants.sh -- run in the BUILD container (no GPU needed): side builds for A/B runs on the GPU box.  Everything lands under
# .variants/ (git-ignored, but shipped by gpurun), so that no GPU-box time is spent compiling.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
python -m ssdnerf_amd.build                                                      # the default library, in-tree
SSDNERF_LIB_DIR=$R/.variants/gather_pairs SSDNERF_EXTRA_FLAGS=-DSSD_GATHER_PAIRS=1 python -m ssdnerf_amd.build --force
rm -f .variants/gather_pairs/*.o
for u in trans_rate mfma_valu_overlap; do
  hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/ubench/$u.hip -o .variants/$u
done
ls -la .variants .variants/gather_pairs
