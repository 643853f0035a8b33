#!/usr/bin/env python
"""tools/fresh_process_check.py -- ONE render of one bench scene (251 views) in a fresh process, first use of a fresh workspace; prints a checksum of
image / depth / sample counts.  Run it several times: differing lines point at a read of memory the call did not write (which repeated renders in
one process, on a recycled workspace, cannot show)."""
import hashlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssdnerf_amd import synthetic as S
from ssdnerf_amd.decoders import TriPlaneDecoder, pack_triplanes
from ssdnerf_amd.density import get_density
dev = torch.device("cuda")
# dirty the allocator's pool first so that "fresh" memory is not zero pages
junk = torch.full((int(sys.argv[1]) if len(sys.argv) > 1 else 300_000_000,), float("nan"), device=dev); del junk
dec = TriPlaneDecoder(base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], dir_layers=[16, 64], max_steps=256)
dec.load_state_dict(S.make_decoder_params(), strict=False); dec = dec.to(dev).eval()
g = torch.Generator().manual_seed(7); jit = [torch.rand(64 ** 3, 3, generator=g).to(dev) for _ in range(8)]
nv = 251
code = S.make_triplane(2021, "object").to(dev)[None]
_, bits = get_density(dec, code, 64, density_thresh=0.1, density_step=8, jitters=jit)
poses = S.spiral_poses(251)[:nv].to(dev)[None]; intr = S.cars_intrinsics(128, 128)[None].expand(nv, -1).to(dev)[None].contiguous()
out = dec.render_packed(pack_triplanes(code), None, None, bits, 64, [0.0], 1e-4, bg_color=1.0, want_counts=True, check_overflow=False, cams=(poses, intr, 128, 128))
cn = dec.last_render_stats["sample_counts"]
parts = []
for t in (out["image"], out["depth"], out["weights_sum"], cn, bits):
    parts.append(hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:8])
print("image %s depth %s weights %s counts %s bitfield %s" % tuple(parts), "samples", int(cn.sum()), "nan", int(torch.isnan(out["image"]).sum()))
