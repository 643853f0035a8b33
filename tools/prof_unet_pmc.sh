#!/usr/bin/env bash
# tools/prof_unet_pmc.sh [dtype=bf16] -- matrix-pipe evidence for the UNet executor (north_star: "MFMA utilisation on the UNet against chip peak"): one rocprofv3 PMC
# session of executor replays WITHOUT graph capture (a captured graph is one dispatch to the profiler), counters per kernel:
#   SQ_VALU_MFMA_BUSY_CYCLES (cycles a SIMD's matrix pipe is busy, summed over SIMDs), SQ_BUSY_CU_CYCLES, SQ_INSTS_MFMA / SQ_INSTS_VALU, GRBM_GUI_ACTIVE
# in separate passes (each with --kernel-trace only), then a kernel-trace pass for durations.  Output: gpurun_out/unet_pmc/unet_pmc_<dtype>.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; DT=${1:-bf16}; OUT=$R/gpurun_out/unet_pmc; mkdir -p $OUT; cd $R
CMD="python tools/bench_unet.py --modes fast --dtypes $DT --no-graph --iters 3"
pass() { n=$1; shift; rm -rf /tmp/up_$n; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/up_$n -o $n -- $CMD > /dev/null 2> $OUT/${n}_err.log; }
pass p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
pass p2 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES
pass p3 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
rm -rf /tmp/up_kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/up_kt -o kt -- $CMD > $OUT/bench_${DT}.txt 2> $OUT/kt_err.log
python - "$DT" "$OUT/unet_pmc_${DT}.txt" <<'PY'
import csv, glob, sys, collections
dt, path = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for n in ("p1", "p2", "p3"):
    for f in glob.glob(f"/tmp/up_{n}/**/*counter_collection*.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "?")
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/up_kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot_us = sum(sum(v) for k, v in dur.items() if k.startswith(("k_", "void k_")) or "k_conv" in k or "k_attn" in k or "k_gn" in k)
rows = []
for k, d in acc.items():
    if not any(w in k for w in ("k_conv", "k_attn", "k_gn", "k_bias", "k_group", "k_split", "k_ddim", "k_nhwc", "k_time", "k_silu", "k_")):
        continue
    n = cnt[(k, "SQ_VALU_MFMA_BUSY_CYCLES")] or 1
    mfma_busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / n            # per launch, summed over the chip's SIMDs
    gui = d.get("GRBM_GUI_ACTIVE", 0.0) / (cnt[(k, "GRBM_GUI_ACTIVE")] or 1)
    insts_mfma = d.get("SQ_INSTS_MFMA", 0.0) / (cnt[(k, "SQ_INSTS_MFMA")] or 1)
    us = sum(dur.get(k, [0.0])) / max(len(dur.get(k, [])), 1)
    # matrix-pipe utilisation of the launch = busy cycles / (1024 SIMDs x the launch's cycles).  rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (check: the
    # 177 us launch of k_conv_pp_bf16<false,..> reads 3.11e6 = 8 x 388.6 k cycles = 177 us at 2.19 GHz), so the launch's cycles are GUI / 8
    util = mfma_busy / (1024.0 * gui / 8.0) if gui else float("nan")
    rows.append((sum(dur.get(k, [0.0])), k, len(dur.get(k, [])), us, insts_mfma, mfma_busy, gui, util))
rows.sort(reverse=True)
with open(path, "w") as out:
    out.write(f"# UNet executor ({dt}), 8 scenes, no graph capture, 4 forwards per process (1 warm-up + 3): per kernel, rocprofv3 PMC per launch; util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); a 32x32x16 bf16 MFMA holds the pipe 32 cycles\n")
    out.write(f"{'kernel':64s} {'launches':>8s} {'avg_us':>8s} {'total_us':>9s} {'MFMA insts':>11s} {'MFMA busy cyc':>14s} {'GUI cyc':>10s} {'matrix-pipe util':>16s}\n")
    tb = tg = 0.0
    for tot, k, n, us, im, mb, gui, util in rows:
        out.write(f"{k[:64]:64s} {n:8d} {us:8.1f} {tot:9.0f} {im:11.0f} {mb:14.0f} {gui:10.0f} {util:16.3f}\n")
        if gui == gui and n:
            tb += mb * n; tg += gui * n
    out.write(f"\nall kernels of the table, launch-weighted: matrix pipe busy {tb / (1024.0 * tg / 8.0) if tg else float('nan'):.3f} of the SIMD-cycles inside kernels\n")
print(open(path).read()[:3500])
PY
