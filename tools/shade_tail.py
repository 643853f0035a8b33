#!/usr/bin/env python
"""tools/shade_tail.py -- how evenly the persistent shading kernel ends (GPU box).  Needs the instrumented side build
`tools/build_variant.sh dbgiters shade_mfma.hip -DSM_DEBUG_ITERS` and SSDNERF_HIP_LIB=.variants/dbgiters/libssdnerf_hip.so: every wave then leaves
its iteration count, live-lane count and start / end time (100 MHz wall clock) in spare words of scene 0's boundary-counter line.  Prints the
kernel's span, the MEAN wave end time (span - mean = idle tail) and the lane utilisation of the bench workload."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssdnerf_amd import synthetic as S, nerf
from ssdnerf_amd.decoders import TriPlaneDecoder, pack_triplanes
from ssdnerf_amd.density import get_density
dev = torch.device("cuda")
dec = TriPlaneDecoder(base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], dir_layers=[16, 64], max_steps=256)
dec.load_state_dict(S.make_decoder_params(2021), strict=False); dec = dec.to(dev).eval()
g = torch.Generator().manual_seed(7); jit = [torch.rand(64 ** 3, 3, generator=g).to(dev) for _ in range(8)]
ns, nv, hw = 8, 251, 128
poses = S.spiral_poses(nv).to(dev)[None].expand(ns, -1, -1, -1).contiguous(); intr = S.cars_intrinsics(hw, hw).to(dev)[None, None].expand(ns, nv, -1).contiguous()
code = torch.stack([S.make_triplane(2021 + s, "object") for s in range(ns)]).to(dev)
_, bits = get_density(dec, code, 64, density_thresh=0.1, density_step=8, jitters=jit)
planes = pack_triplanes(code, dec.plane_dtype)
for it in range(2):
    dec.render_packed(planes, None, None, bits, 64, [0.0] * ns, 1e-4, bg_color=1.0, check_overflow=False, cams=(poses, intr, hw, hw))
torch.cuda.synchronize()
ws = list(dec._ws_cache.values())[0]
c = ws[:4 * ns * 128].view(torch.int32).view(4, ns, 32)[3, 0, :4].tolist()
c64 = ws[:4 * ns * 128].view(torch.int64).view(4, ns, 16)[3, 0, :6].tolist()
t0, tmax, tsum = (~c64[2]) & (2 ** 64 - 1), c64[3], c64[4]
print("start", t0, "max_end-start (ms)", (tmax - t0) / 1e5, "mean_end-start (ms)", ((tsum / 2048) - (t0 & 0xffffffff)) / 1e5)
print("boundary", c[0], "sum_iters", c[1], "max_iters", c[2], "mean_iters", c[1] / 2048, "max/mean", c[2] / (c[1] / 2048), "live_lane_frac", c[3] / c[1])

nk = 5
lines = ws[:nk * ns * 128].view(torch.int32).view(nk, ns, 32)[3]
hist = lines[1, 1:32].tolist() + lines[2, 1:32].tolist()
its = lines[3, 1:32].tolist() + lines[4, 1:32].tolist()
print("waves per 0.1 ms bin of their own duration (mean iterations per wave in the bin):")
print("  " + "  ".join(f"{0.1 * b:.1f}ms:{n}({its[b] // max(n, 1)})" for b, n in enumerate(hist) if n))
