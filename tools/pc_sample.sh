#!/usr/bin/env bash
# tools/pc_sample.sh [method] [interval] -- on the GPU box: PC-sample the render bench with rocprofv3 (beta feature), aggregate the samples per
# (code object, offset[, stall reason]) on the box and leave only that histogram under gpurun_out/pcs/ (raw sample files stay in /tmp).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
M=${1:-host_trap}; I=${2:-1}; U=${3:-time}
OUT=/tmp/pcs_$M; rm -rf $OUT; mkdir -p $OUT gpurun_out/pcs
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $U --pc-sampling-method $M --pc-sampling-interval $I --kernel-trace --output-format csv -d $OUT -o pcs -- \
    python $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $R/gpurun_out/pcs/${M}_run.log 2>&1
echo "rc=$?" >> $R/gpurun_out/pcs/${M}_run.log
ls -la $OUT $OUT/* >> $R/gpurun_out/pcs/${M}_run.log 2>&1
cd $R
python - "$OUT" "$M" <<'PY'
import sys, glob, os, collections, csv
out, m = sys.argv[1], sys.argv[2]
files = [f for f in glob.glob(out + "/**/*.csv", recursive=True)]
with open(f"gpurun_out/pcs/{m}_files.txt", "w") as fo:
    for f in files:
        fo.write(f"{f} {os.path.getsize(f)}\n")
        with open(f) as fi:
            for k, line in enumerate(fi):
                if k < 4: fo.write("    " + line)
                else: break
for f in files:
    if "pc_sampling" not in os.path.basename(f): continue
    hist = collections.Counter()
    with open(f) as fi:
        rd = csv.DictReader(fi)
        cols = rd.fieldnames
        keyc = [c for c in cols if c.lower() in ("code_object_id", "code_object_offset", "instruction_index", "stall_reason", "wave_issued", "instruction_type", "snapshot_reason_not_issued", "inst_type", "reason_not_issued", "arb_state_issue", "arb_state_stall")]
        for row in rd:
            hist[tuple(row[c] for c in keyc)] += 1
    with open(f"gpurun_out/pcs/{m}_{os.path.basename(f)}.hist.csv", "w") as fo:
        fo.write(",".join(keyc) + ",count\n")
        for k, n in hist.most_common():
            fo.write(",".join(k) + f",{n}\n")
PY
ls -la gpurun_out/pcs
