#!/usr/bin/env bash
# tools/prof_step.sh <tag> [bench args] -- rocprofv3 --kernel-trace --stats of `python bench.py --no-extras` on the GPU box; per-kernel average
# durations (us) into gpurun_out/r03/<tag>_kernel_stats.txt (copy what is judged into profiles/)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
mkdir -p $R/gpurun_out/r03; rm -rf /tmp/rc_step
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rc_step -o st -- python bench.py --no-extras "$@" > $R/gpurun_out/r03/${TAG}_bench.json 2> /tmp/rc_step_err.log
python - <<'PY' > $R/gpurun_out/r03/${TAG}_kernel_stats.txt
import csv, glob, collections
for f in glob.glob("/tmp/rc_step/**/*kernel_trace.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"{'kernel':90s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'total_ms':>10s}")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[:90]:90s} {len(v):6d} {sum(v) / len(v):10.1f} {min(v):10.1f} {sum(v) / 1e3:10.2f}")
PY
head -14 $R/gpurun_out/r03/${TAG}_kernel_stats.txt
