#!/usr/bin/env python
"""tools/blend_check.py [renders] -- GPU box, side build with -DSM_DEBUG_BLEND2: counters of the in-kernel re-evaluation of the bilinear blend."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssdnerf_amd import synthetic as S
from ssdnerf_amd import _cabi as C
from ssdnerf_amd.decoders import TriPlaneDecoder, pack_triplanes
from ssdnerf_amd.density import get_density
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda")
dec = TriPlaneDecoder(base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], dir_layers=[16, 64], max_steps=256)
dec.load_state_dict(S.make_decoder_params(2021), strict=False); dec = dec.to(dev).eval()
g = torch.Generator().manual_seed(7); jit = [torch.rand(64 ** 3, 3, generator=g).to(dev) for _ in range(8)]
ns, nv, hw = 8, 251, 128
poses = S.spiral_poses(nv).to(dev)[None].expand(ns, -1, -1, -1).contiguous(); intr = S.cars_intrinsics(hw, hw).to(dev)[None, None].expand(ns, nv, -1).contiguous()
code = torch.stack([S.make_triplane(2021 + s, "object") for s in range(ns)]).to(dev)
_, bits = get_density(dec, code, 64, density_thresh=0.1, density_step=8, jitters=jit)
planes = pack_triplanes(code, dec.plane_dtype)
need = C.lib().ssdnerf_render_queue_workspace(ns, nv * hw * hw, 64)
tot = torch.zeros(32, dtype=torch.int64); ref = None; ndiff = 0
for it in range(n):
    out = dec.render_packed(planes, None, None, bits, 64, [0.0] * ns, 1e-4, bg_color=1.0, want_counts=True, cams=(poses, intr, hw, hw))
    wsp = dec._workspace(need, dev, tag="render0")
    tot += wsp[:4 * ns * 128].view(torch.int32).view(4, ns, 32)[3, 0].clone().cpu().to(torch.int64)
    cur = (out["image"].clone(), out["depth"].clone())
    if ref is None: ref = cur
    else: ndiff += int(((cur[0] != ref[0]).any(-1) | (cur[1] != ref[1])).any())
t = tot.tolist()
print(f"{n} renders, {ndiff} differ from render 0; samples {int(dec.last_render_stats['sample_counts'].sum())}")
print(f"lanes whose second packed blend disagrees with the first: {t[8]}; whose scalar blend disagrees: {t[9]}; wave-iterations checked: {t[10]}; lane mask {t[30] & 0xffffffff:08x} {t[31] & 0xffffffff:08x}")
print("per feature (c*3 + plane):", t[12:30]); print("line", t); print("boundary_tests", dec.last_render_stats["boundary_tests"].tolist(), wsp.numel(), need, wsp.data_ptr())
print(wsp[:4 * ns * 128].view(torch.int32).view(4, ns, 32)[:, :, 0].tolist())
