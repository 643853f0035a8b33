#!/usr/bin/env bash
# tools/ab_shade.sh <variant names...> -- on the GPU box: bench each .variants/<name> side build (tools/build_variant.sh) and the in-tree library
# ("base"); prints ms/step, shade ms, first-hit ms, samples (a changed sample total flags a broken variant).  AB_REPEAT=n repeats the whole list
# (boxes drift by a few % over a session: interleaved repeats separate a variant from the drift).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/ab
one() { # label, env...
  l=$1; shift
  env "$@" timeout 120 python bench.py --steps 20 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/ab/$l.json 2> gpurun_out/ab/$l.err < /dev/null
  python - "$l" <<'PY'
import json, sys
l = sys.argv[1]
try:
    d = json.loads([x for x in open(f"gpurun_out/ab/{l}.json") if x.startswith("{")][-1])
    r = d["roofline"]
    print(f"{l:28s} step {d['ms_per_step']:.3f}  shade {r['launch_ms']:.3f}  frac {r['frac']:.3f}  first_hit {list(r['other_kernels'].values())[0]['launch_ms']:.3f}  samples {d['boundary_rays']['samples_per_step_per_gpu']}")
except Exception as e:
    print(l, "FAILED", e)
PY
}
for rep in $(seq 1 ${AB_REPEAT:-1}); do
  one base SSDNERF_DUMMY=0
  for v in "$@"; do
    [ -f .variants/$v/libssdnerf_hip.so ] || { echo "$v: not built"; continue; }
    one $v SSDNERF_HIP_LIB=$R/.variants/$v/libssdnerf_hip.so
  done
done
[ -n "$AB_SKIP_1WAVE" ] || one base_1wave SSDNERF_SHADE_BLOCKS_PER_CU=1
