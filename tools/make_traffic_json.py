#!/usr/bin/env python
"""tools/make_traffic_json.py <dir> <tag> -- profiles/traffic_latest.json from ONE tools/prof_render.sh session: the PMC passes (<tag>_pmc.txt), the
rocprofv3 launch averages (<tag>_launch_avg.txt) and the bench line printed under the profiler (<tag>_bench_under_rocprof.json) of the same session;
written to <dir>/traffic_latest.json AND to <repo>/profiles/traffic_latest.json.  bench.py copies it into `roofline.traffic` / `roofline.traffic_source`.
SSDNERF_PROFILED_COMMIT: the commit that was profiled (.git does not travel to the GPU box: the caller passes `git rev-parse --short HEAD`)."""
import json
import os
import re
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

out, tag = sys.argv[1], sys.argv[2]


def per_dispatch(counter):
    for line in open(f"{out}/{tag}_pmc.txt"):
        if "k_shade_mfma" in line[:60] and f" {counter} " in line:
            return line[:60].strip().replace("void ", ""), float(re.search(r"per_dispatch=([0-9.e+]+)", line).group(1))
    return None, None


kern, fetch = per_dispatch("FETCH_SIZE")
_, write = per_dispatch("WRITE_SIZE")
_, hit = per_dispatch("TCC_HIT_sum")
_, miss = per_dispatch("TCC_MISS_sum")
bench = json.loads([l for l in open(f"{out}/{tag}_bench_under_rocprof.json") if l.startswith("{")][-1])
avg_last = n_last = None
for line in open(f"{out}/{tag}_launch_avg.txt"):
    if "k_shade_mfma" in line[:70]:
        cols = line[70:].split()
        avg_last, n_last = float(cols[2]) / 1e3, min(20, int(cols[0]))
cfg = bench["config"]
m = re.search(r"direction term (\d) of 6", cfg.get("mlp_arithmetic", ""))
tj = {"workload": {"scenes": cfg["scenes_per_gpu"], "views": cfg["views_per_scene"], "size": int(cfg["image"].split("x")[0]), "variant": cfg["scene_variant"],
                   "plane_dtype": cfg["plane_dtype"], "ray_source": "cameras" if cfg["ray_source"].startswith("cameras") else "arrays"},
      "kernel": kern, "dir_products": int(m.group(1)) if m else 3,
      "fetch_size_kib_raw": fetch, "write_size_kib_raw": write, "tcc_hit_requests": hit, "tcc_miss_requests": miss,
      "hbm_bytes_per_launch": None if fetch is None or write is None else (2 * fetch + write) * 1024,
      "rocprof_launch_ms_avg_timed_steps": avg_last, "rocprof_launches_averaged": n_last,
      "hip_event_launch_ms_same_run": bench["roofline"]["launch_ms"],
      "method": "tools/prof_render.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE GRBM_GUI_ACTIVE in separate passes (stat pass + 1 step: 2 launches each, "
                "averaged); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); launch average = rocprofv3 --kernel-trace over the "
                "20 timed launches of `bench.py --warmup 10 --steps 20`, HIP events of the same run beside it",
      "profiled_commit": os.environ.get("SSDNERF_PROFILED_COMMIT"),
      "render_build_id": __import__("ssdnerf_amd.build", fromlist=["render_build_id"]).render_build_id(),      # (r06) bench.py quotes the traffic only for a library built from these sources + settings
      "source": f"profiles/{os.environ.get('SSDNERF_PROFILE_ROUND', 'r05')}/{tag}_pmc.txt, {tag}_launch_avg.txt, {tag}_bench_under_rocprof.json"}
json.dump(tj, open(f"{out}/traffic_latest.json", "w"), indent=1)
# ... and installed where bench.py reads it (on the GPU box that copy is scratch: run this tool again on the merged gpurun_out/prof_<tag>/ and commit)
import os
_prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
if os.path.isdir(_prof):
    json.dump(tj, open(os.path.join(_prof, "traffic_latest.json"), "w"), indent=1)
print(json.dumps(tj, indent=1))
