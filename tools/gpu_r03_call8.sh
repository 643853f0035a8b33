#!/usr/bin/env bash
# round 3, GPU call 8: the two-group convolution kernel: parity tests, then per-layer timing against the r02 kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r03; mkdir -p $O
timeout 600 python -m pytest tests/test_unet_fast_gpu.py -x -q -m gpu -k "conv_igemm or concat" > $O/test_pp.log 2>&1; echo "tests rc=$?"; tail -5 $O/test_pp.log
timeout 600 python tools/bench_conv.py --hints 0,5,6 --no-lib --iters 30 > $O/bench_conv_pp.jsonl 2> $O/bench_conv_pp.err; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r03/bench_conv_pp.jsonl"):
    d = json.loads(l)
    if "H" in d: print(d["H"], d["Cin"], d["Cout"], d["k"], d["stride"], d["up"], "x%d" % d["n"], d["own_us"])
    else: print(d)
PY
