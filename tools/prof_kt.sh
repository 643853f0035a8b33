#!/usr/bin/env bash
# tools/prof_kt.sh <name> [bench args...] -- rocprofv3 kernel trace + stats of a short bench run (GPU box); summary -> gpurun_out/<name>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=${1:-kt}; shift
mkdir -p $R/gpurun_out
cd $R
rm -rf /tmp/rp_$NAME
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$NAME -o $NAME -- python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline "$@" > $R/gpurun_out/${NAME}_bench.json 2> $R/gpurun_out/${NAME}_err.log
f=$(find /tmp/rp_$NAME -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-260 "$f" | head -14 > $R/gpurun_out/${NAME}_kernel_stats.csv
cat $R/gpurun_out/${NAME}_kernel_stats.csv | cut -c1-200
