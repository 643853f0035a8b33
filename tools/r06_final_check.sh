set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06z
S=$SECONDS
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06z/pytest_gpu.txt 2>&1; echo "pytest rc $? in $((SECONDS-S)) s" >> gpurun_out/r06z/pytest_gpu.txt
tail -5 gpurun_out/r06z/pytest_gpu.txt
S=$SECONDS
timeout 900 python bench.py > gpurun_out/r06z/bench.json 2> gpurun_out/r06z/bench.err; echo "bench rc $? in $((SECONDS-S)) s"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r06z/bench.json').read().strip().splitlines()[-1])
print({k:j[k] for k in ('metric','value','ms_per_step')}, j['roofline'])
PY
