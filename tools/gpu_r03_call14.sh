#!/usr/bin/env bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_unet_fast_gpu.py tests/test_unet_golden.py -x -q -m gpu > $O/test_unet3.log 2>&1; echo "tests rc=$?"; tail -3 $O/test_unet3.log
timeout 600 python tools/bench_conv.py --hints 0 --no-lib --iters 30 > $O/bench_conv_b.jsonl 2> /dev/null; python - <<'PY'
import json
for l in open("gpurun_out/r03/bench_conv_b.jsonl"):
    d = json.loads(l)
    if "H" in d: print(d["H"], d["Cin"], d["Cout"], d["k"], d["stride"], d["up"], "x%d" % d["n"], d["own_us"], d["own_tf"])
    else: print(d)
PY
timeout 300 python tools/bench_unet.py --modes fast --dtypes fp32,bf16 --iters 20 2>&1 | grep -v amdgpu.ids | tail -1 | tee $O/unet_b.json
