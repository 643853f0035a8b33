#!/usr/bin/env python
"""tools/bench_decode_bwd.py -- ssdnerf_point_decode_backward alone on synthetic samples (fog-like: uniform in the box), the guidance
workload's shape: --scenes x --samples points, every point with a gradient.  Prints ms per call (HIP events); run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split (k_decode_bwd_feat / _bin / _sum)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssdnerf_amd  # noqa
from ssdnerf_amd.registry import MODULES
from ssdnerf_amd import synthetic as S

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=8); ap.add_argument("--samples", type=int, default=625000); ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--cone", action="store_true", help="with --ray-like: all rays within a narrow cone around -z (one view looking down an axis: the xy plane sees whole rays on one texel)")
ap.add_argument("--ray-like", action="store_true", help="consecutive samples 0.0135 apart along random rays (the march's order) instead of i.i.d. points")
a = ap.parse_args()
dec = MODULES.build(dict(type="TriPlaneDecoder", interp_mode="bilinear", base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3],
                         use_dir_enc=True, dir_layers=[16, 64], activation="silu", sigma_activation="trunc_exp", sigmoid_saturation=0.001, max_steps=256))
dec.load_state_dict(S.make_decoder_params(), strict=False)
dec = dec.cuda().train(True).requires_grad_(False)
g = torch.Generator().manual_seed(1)
code = torch.stack([S.make_triplane(3 + i) for i in range(a.scenes)]).cuda().requires_grad_(True)
if a.ray_like:
    n_rays = a.samples // 128
    o = torch.rand(a.scenes, n_rays, 1, 3, generator=g) * 2 - 1
    d = torch.randn(a.scenes, n_rays, 1, 3, generator=g)
    if a.cone:
        d = d * 0.15 + torch.tensor([0.0, 0.0, -1.0])
    d = torch.nn.functional.normalize(d, dim=-1)
    t = (torch.arange(128).float() - 64)[None, None, :, None] * 0.0135
    pts = (o + d * t).clamp(-1, 1).reshape(a.scenes, -1, 3)
    xyzs = [p.cuda() for p in pts]
    dirs = [d[s].expand(-1, 128, -1).reshape(-1, 3).contiguous().cuda() for s in range(a.scenes)]
else:
    xyzs = [(torch.rand(a.samples, 3, generator=g) * 2 - 1).cuda() for _ in range(a.scenes)]
    dirs = [torch.nn.functional.normalize(torch.randn(a.samples, 3, generator=g), dim=-1).cuda() for _ in range(a.scenes)]
n = sum(x.size(0) for x in xyzs)
gs, gc = torch.randn(n, generator=g).cuda() * 0.1, torch.randn(n, 3, generator=g).cuda()
sig, rgb, _ = dec.point_decode(xyzs, dirs, code)
loss = (sig * gs).sum() + (rgb * gc).sum()
torch.autograd.grad(loss, code, retain_graph=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    (gcode,) = torch.autograd.grad(loss, code, retain_graph=True)
e1.record(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        torch.autograd.grad(loss, code, retain_graph=True)
    torch.cuda.synchronize()
kern = {e.key.split("(")[0].replace("void ", "")[:40]: round(e.self_device_time_total / e.count * 1e-3, 3) for e in prof.key_averages() if "k_decode_bwd" in e.key}
print(json.dumps({"kernels_ms": kern, "scenes": a.scenes, "samples_total": n, "ray_like": a.ray_like, "cone": a.cone, "ms_per_backward": e0.elapsed_time(e1) / a.iters,
                  "grad_abs_sum": float(gcode.abs().sum()), "lib": os.environ.get("SSDNERF_HIP_LIB", "in-tree")}))
