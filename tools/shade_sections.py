#!/usr/bin/env python
"""tools/shade_sections.py -- where a wave of the persistent shading kernel spends its cycles (GPU box).  Needs the instrumented side build
`tools/build_variant.sh dbgsec shade_mfma.hip -DSM_DEBUG_SECTIONS [...]` and SSDNERF_HIP_LIB=.variants/dbgsec/libssdnerf_hip.so: every wave sums the
shader cycles between four marks of its loop body (event | gather | split + MLP | composite + search) and adds them to spare words of scene 0's
boundary-counter line.  Prints cycles per loop iteration and the share of each section on the bench workload."""
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
reps = 20
for it in range(5):
    dec.render_packed(planes, None, None, bits, 64, [0.0] * ns, 1e-4, bg_color=1.0, check_overflow=False, cams=(poses, intr, hw, hw))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(reps):
    dec.render_packed(planes, None, None, bits, 64, [0.0] * ns, 1e-4, bg_color=1.0, check_overflow=False, cams=(poses, intr, hw, hw))
e1.record()
torch.cuda.synchronize()
ms_per_render = e0.elapsed_time(e1) / reps
ws = list(dec._ws_cache.values())[0]
nk = 5
line = ws[:nk * ns * 128].view(torch.int32).view(nk, ns, 32)[3, 0]
sec = [16 * v for v in line[10:30].contiguous().view(torch.int64).tolist()]
events, loops, waves, marches, stages = (line[i].item() for i in (4, 5, 6, 7, 8))
tot = sum(sec)
names = ["event: control", "gather", "split+MLP", "composite+search", "event: stores", "event: park", "event: pool refill", "event: stage refill", "event: SH operands", "event: march pass"]
print(f"waves {waves}  loop iterations {loops}  events {events} ({loops / max(events, 1):.2f} iterations per event)  march passes {marches}  stage fills {stages}  cycles per iteration {tot / max(loops, 1):.0f}")
print(f"render {ms_per_render:.3f} ms (both stages); wave cycles / render time = {tot / max(waves, 1) / (ms_per_render * 1e-3) / 1e9:.3f} GHz x (shade share of the render)")
tl = ws[:nk * ns * 128].view(torch.int32).view(nk, ns, 32)[1, 0]
mlp = [16 * v for v in tl[2:14].contiguous().view(torch.int64).tolist()]
if os.environ.get("SHADE_SECTIONS_MARCH") and sum(mlp):      # -DSM_DEBUG_MARCH: inside the march pass (per PASS)
    for n, v in zip(["(up to the pass)", "march: pool read + ray + bounds", "march: probe loop", "march: stores + pool writes"], mlp):
        print(f"  {n:36s} {v / max(marches, 1):8.0f} cycles per pass")
    tot += sum(mlp)
elif sum(mlp):                                  # -DSM_DEBUG_MLP_PHASES: the MLP section in phases (their sum replaces "split+MLP" above, whose mark then only holds the tail)
    for n, v in zip(["MLP: split", "MLP: A (layer 1, tile 0)", "MLP: B (layer 1 tile 1 | density 0)", "MLP: C (dir 0 | density 1)", "MLP: D (dir 1 | colour 0)", "MLP: E (colour 1)"], mlp):
        print(f"  {n:36s} {v / max(loops, 1):8.0f} cycles per iteration")
    tot += sum(mlp)
for n, v in zip(names, sec):
    print(f"  {n:18s} {v / max(loops, 1):8.0f} cycles per iteration  {100.0 * v / tot:5.1f} %")
