#!/usr/bin/env bash
# tools/prof_conv.sh [hint] -- rocprofv3 kernel trace + PMC counters of the implicit-GEMM convolution kernel (run on the GPU box).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_conv_h${1:-0}
rm -rf $OUT && mkdir -p $OUT
cd $R
HINT=${1:-0}
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rc_kt -o kt -- python tools/prof_conv.py --iters 5 --hint $HINT > /dev/null 2> $OUT/kt_err.log
find /tmp/rc_kt -name "*kernel_trace.csv" -exec cp {} $OUT/ \;
python - <<'PY' > $OUT/kernel_durations.txt
import csv, glob, collections
for f in glob.glob("/tmp/rc_kt/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_conv" in r["Kernel_Name"]]
    for i, r in enumerate(rows):
        print(i, r["Kernel_Name"][:70], "grid", r.get("Grid_Size_X", r.get("Grid_Size", "?")), "us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
pmc() { n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/rc_$n -o $n -- python tools/prof_conv.py --iters 1 --hint $HINT > /dev/null 2> $OUT/${n}_err.log
  python - "$n" <<'PY' >> $OUT/pmc_summary.txt
import csv, glob, sys, collections
n = sys.argv[1]
rows = collections.OrderedDict()
for f in glob.glob(f"/tmp/rc_{n}/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_conv" not in r["Kernel_Name"]: continue
        rows.setdefault((r["Dispatch_Id"], r["Grid_Size"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (d, g), c in rows.items():
    print(n, "dispatch", d, "grid", g, " ".join(f"{k}={v:.6g}" for k, v in sorted(c.items())))
PY
}
pmc p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
pmc p2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU
pmc p3 TCC_HIT_sum TCC_MISS_sum
pmc p4 FETCH_SIZE
cat $OUT/kernel_durations.txt | tail -8; cat $OUT/pmc_summary.txt
