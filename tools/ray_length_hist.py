#!/usr/bin/env python
"""tools/ray_length_hist.py -- distribution of samples per hitting ray on the bench workload (GPU box): share of rays and of samples in rays of
at most k samples.  Input for the queue-order / drain-tail discussion in DESIGN.md section 5.2."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssdnerf_amd import synthetic as S
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
dec.render_packed(planes, None, None, bits, 64, [0.0] * ns, 1e-4, bg_color=1.0, want_counts=True, cams=(poses, intr, hw, hw))
cn = dec.last_render_stats["sample_counts"].flatten()
cn = cn[cn > 0].long()
h = torch.bincount(cn, minlength=257).double()
rays, samples = h.sum().item(), (h * torch.arange(257, device=dev)).sum().item()
print(f"hitting rays {int(rays)}  samples {int(samples)}  mean {samples / rays:.2f}  max {int(cn.max())}")
cr, cs = torch.cumsum(h, 0) / rays, torch.cumsum(h * torch.arange(257, device=dev), 0) / samples
for k in (1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256):
    print(f"  <= {k:3d} samples: {100 * cr[k].item():5.1f} % of rays  {100 * cs[k].item():5.1f} % of samples")
