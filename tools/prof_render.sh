#!/usr/bin/env bash
# tools/prof_render.sh <tag> [bench args...] -- ONE profiling session of the render bench on the GPU box; everything the bench line's `roofline`
# object cites comes out of it (r04: the r03 verdict's "make roofline reproducible from profiles/"):
#   1. the un-profiled bench line of the same build and box                                  -> <tag>_bench_unprofiled.json
#   2. rocprofv3 --kernel-trace --stats of `python bench.py --warmup 10 --steps 20`            -> <tag>_kernel_stats.csv (every launch of the run),
#      <tag>_bench_under_rocprof.json (the bench line of THAT run: its HIP-event launch_ms is over the same 20 launches) and
#      <tag>_launch_avg.txt: per kernel, the rocprofv3 average over the LAST 20 launches (= the timed steps, warm) next to the all-launch one
#   3. the PMC passes, each in its own run with --kernel-trace only (never combined with sys/hip tracing)   -> <tag>_pmc.txt
#   4. profiles/traffic_latest.json regenerated from passes 3 + 2 of THIS session (FETCH_SIZE doubled per MI355X_MICROARCH.md), with the commit
#      it profiled; bench.py copies it into `roofline.traffic` / `roofline.traffic_source`.
# Outputs land in gpurun_out/prof_<tag>/; copy them to profiles/rNN/ and commit traffic_latest.json.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-a}; shift
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $R
ARGS="${@:---no-extras --no-cpu-baseline}"      # the default (8 scenes x 251 views) workload: traffic is per launch of THAT
python bench.py --warmup 10 --steps 20 $ARGS > $OUT/${TAG}_bench_unprofiled.json 2> $OUT/unprofiled_err.log
rm -rf /tmp/rp_kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_kt -o kt -- python bench.py --warmup 10 --steps 20 $ARGS > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/kt_err.log
f=$(find /tmp/rp_kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-300 "$f" | head -24 > $OUT/${TAG}_kernel_stats.csv
python - "$OUT/${TAG}_launch_avg.txt" <<'PY'
import csv, glob, collections, sys
d = collections.defaultdict(list)
for f in glob.glob("/tmp/rp_kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
with open(sys.argv[1], "w") as out:
    out.write(f"{'kernel':70s} {'calls':>6s} {'avg_us_all':>11s} {'avg_us_last20':>14s} {'min_us':>9s} {'max_us_last20':>14s}\n")
    for k, v in sorted(d.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
        v.sort()
        durs = [x[1] for x in v]
        last = durs[-20:]
        out.write(f"{k[:70]:70s} {len(durs):6d} {sum(durs) / len(durs):11.1f} {sum(last) / len(last):14.1f} {min(durs):9.1f} {max(last):14.1f}\n")
print(open(sys.argv[1]).read()[:2400])
# (r06) the timeline of the last three timed steps: which kernels overlap (stage A of step i+1 runs on a second stream beside the shading kernel of step i), and the gaps
rows = sorted((s, s + int(dur * 1e3), k) for k, v in d.items() for s, dur in v)
shade = [r for r in rows if "k_shade_mfma" in r[2]]
if len(shade) >= 4:
    t0 = shade[-4][1]
    with open(sys.argv[1].replace("_launch_avg.txt", "_timeline.txt"), "w") as out:
        out.write("start_us    end_us      dur_us   kernel   (last three timed steps; 0 = the end of the shading kernel of the step before them)\n")
        for s, e, k in rows:
            if s >= shade[-4][0]:
                out.write(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f}   {k[:60]}\n")
    print(open(sys.argv[1].replace("_launch_avg.txt", "_timeline.txt")).read()[:3000])
PY
: > $OUT/${TAG}_pmc.txt
pmc() { # name, counters...
  n=$1; shift
  rm -rf /tmp/rp_$n
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/rp_$n -o $n -- python bench.py --steps 1 --warmup 0 $ARGS > /dev/null 2> $OUT/${n}_err.log
  python - "$n" "$OUT/${TAG}_pmc.txt" <<'PY'
import csv, glob, sys, collections
n, path = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f"/tmp/rp_{n}/**/*counter_collection*.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")[:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
with open(path, "a") as out:
    for k, d in acc.items():
        if not any(w in k for w in ("k_shade", "k_ray_cull", "k_survivor", "k_density", "k_view_masks", "k_queue_close", "k_bitfield")):
            continue
        for c, v in sorted(d.items()):
            out.write(f"{k:60s} {c:28s} total={v:.6g} dispatches={cnt[(k, c)]} per_dispatch={v / cnt[(k, c)]:.6g}\n")
PY
}
pmc pmc1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pmc pmc2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pmc pmc3 FETCH_SIZE
pmc pmc6 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES
pmc pmc4 WRITE_SIZE GRBM_GUI_ACTIVE
pmc pmc5 TCC_HIT_sum TCC_MISS_sum
# traffic_latest.json from THIS session
python tools/make_traffic_json.py "$OUT" "$TAG"
cat $OUT/${TAG}_bench_unprofiled.json | tail -1 | cut -c1-300
du -sh $OUT
