#!/usr/bin/env bash
# tools/prof_render.sh [bench args...] -- rocprofv3 kernel trace + PMC counters of the fused render kernel (run on the GPU box).
# Kernel-trace/stats and each --pmc set are separate runs (never combined with sys/hip traces).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
cd $R
ARGS="${@:---no-extras --no-cpu-baseline}"      # the default (8 scenes x 251 views) workload: traffic is per launch of THAT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_kt -o kt -- python bench.py --steps 3 --warmup 1 $ARGS > $OUT/kt_bench.json 2> $OUT/kt_err.log
find /tmp/rp_kt -name "*stats*.csv" -exec cp {} $OUT/ \;
pmc() { # name, counters...
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/rp_$n -o $n -- python bench.py --steps 1 --warmup 0 $ARGS > /dev/null 2> $OUT/${n}_err.log
  python - "$n" <<'PY'
import csv, glob, sys, collections
n = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f"/tmp/rp_{n}/**/*counter_collection*.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")[:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
        cnt[(k, row["Counter_Name"])] += 1
import os
out = open(os.environ.get("OUT", ".") + f"/{n}_summary.txt", "w")
for k, d in acc.items():
    if not any(w in k for w in ("k_shade", "k_ray_cull", "k_survivor", "k_density", "k_quantize", "k_bitfield")): continue
    for c, v in sorted(d.items()):
        line = f"{k:60s} {c:28s} total={v:.6g} dispatches={cnt[(k,c)]} per_dispatch={v/cnt[(k,c)]:.6g}"
        print(line); out.write(line + "\n")
PY
}
export OUT
pmc pmc1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pmc pmc2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pmc pmc3 FETCH_SIZE
pmc pmc6 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES
pmc pmc4 WRITE_SIZE GRBM_GUI_ACTIVE
pmc pmc5 TCC_HIT_sum TCC_MISS_sum
head -30 $OUT/kt_kernel_stats.csv 2>/dev/null || ls $OUT
cat $OUT/kt_bench.json | tail -1 | cut -c1-400
du -sh $OUT
