#!/usr/bin/env bash
# tools/stage_reference.sh -- stage the reference's Python (lib/ only, no data, no configs) into the git-ignored oracle/_ref/reference_py so that
# tools/option_a_on_gpu.py can execute it on the GPU box, where /root/reference does not exist.  Nothing staged here is ever committed.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=${1:-/root/reference}
[ -d "$SRC/lib/ops" ] || { echo "no reference checkout at $SRC"; exit 0; }
D=$R/oracle/_ref/reference_py; rm -rf $D; mkdir -p $D
(cd $SRC && find lib -name "*.py" -size -200k -print0 | xargs -0 cp --parents -t $D)
echo "staged $(find $D -name '*.py' | wc -l) files under oracle/_ref/reference_py (git-ignored)"
