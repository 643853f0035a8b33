#!/usr/bin/env python
"""tools/render_repeat.py [N] -- stress test of run-to-run reproducibility (GPU box): renders the 251-view bench scene N times (1 scene, then 8
scenes) and counts the renders whose sample counts / image / depth differ from the first by even one bit.  r02 final build: 0 of 400 and 0 of 50
(before the MFMA operand guard of csrc/shade_mfma.hip: every repeat differed on ~30 rays)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssdnerf_amd import synthetic as S
from ssdnerf_amd.decoders import TriPlaneDecoder, pack_triplanes
from ssdnerf_amd.density import get_density
dev = torch.device("cuda")
dec = TriPlaneDecoder(base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], dir_layers=[16, 64], max_steps=256)
dec.load_state_dict(S.make_decoder_params(), strict=False); dec = dec.to(dev).eval()
g = torch.Generator().manual_seed(7); jit = [torch.rand(64 ** 3, 3, generator=g).to(dev) for _ in range(8)]
nv, hw = 251, 128
cases = ((1, [2022]), (8, list(range(2021, 2029))))
if os.environ.get("RR_ONLY1"):                      # side-build sweeps: the one-scene case only
    cases = cases[:1]
for ns, seeds in cases:
    poses = S.spiral_poses(251).to(dev)[None].expand(ns, -1, -1, -1).contiguous(); intr = S.cars_intrinsics(hw, hw).to(dev)[None, None].expand(ns, nv, -1).contiguous()
    code = torch.stack([S.make_triplane(sd, "object") for sd in seeds]).to(dev)
    _, bits = get_density(dec, code, 64, density_thresh=0.1, density_step=8, jitters=jit)
    planes = pack_triplanes(code)
    def render():
        out = dec.render_packed(planes, None, None, bits, 64, [0.0] * ns, 1e-4, bg_color=1.0, want_counts=True, check_overflow=False, cams=(poses, intr, hw, hw))
        return dec.last_render_stats["sample_counts"], out["image"], out["depth"]
    ref = [t.clone() for t in render()]
    bad = 0
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    for k in range(n // ns if ns > 1 else n):
        r = render()
        if not (torch.equal(r[0], ref[0]) and torch.equal(r[1], ref[1]) and torch.equal(r[2], ref[2])):
            bad += 1
    print(f"scenes {ns}: {bad} of {n // ns if ns > 1 else n} repeated renders differ from the first (sample total of the first render {int(ref[0].sum().item())})")
