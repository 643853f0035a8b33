#!/usr/bin/env python
"""tools/trace_check.py [renders] -- GPU box, a side build with -DSM_DEBUG_TRACE (csrc/shade_mfma.hip): per ray, XOR hashes of 0 the march parameters of its samples,
1 the gathered features, 2 the MLP outputs + step, 3 the SH operands, 4 the ray constants / restored state at every load into a lane, 5 the composited state behind
every sample.  Renders the bench workload `renders` times; for every (render, ray) that deviates from the per-ray majority prints which hashes deviate."""
import os, sys, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssdnerf_amd import synthetic as S
from ssdnerf_amd.decoders import TriPlaneDecoder, pack_triplanes
from ssdnerf_amd.density import get_density
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device("cuda")
dec = TriPlaneDecoder(base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], dir_layers=[16, 64], max_steps=256)
dec.load_state_dict(S.make_decoder_params(2021), strict=False); dec = dec.to(dev).eval()
g = torch.Generator().manual_seed(7); jit = [torch.rand(64 ** 3, 3, generator=g).to(dev) for _ in range(8)]
ns, nv, hw = 8, 251, 128
poses = S.spiral_poses(nv).to(dev)[None].expand(ns, -1, -1, -1).contiguous(); intr = S.cars_intrinsics(hw, hw).to(dev)[None, None].expand(ns, nv, -1).contiguous()
code = torch.stack([S.make_triplane(2021 + s, "object") for s in range(ns)]).to(dev)
_, bits = get_density(dec, code, 64, density_thresh=0.1, density_step=8, jitters=jit)
planes = pack_triplanes(code, dec.plane_dtype)
N = nv * hw * hw
runs = []
for it in range(n):
    trace = torch.zeros(ns * N, 8, dtype=torch.int32, device=dev)
    os.environ["SSDNERF_DEBUG_TRACE_PTR"] = hex(trace.data_ptr())
    out = dec.render_packed(planes, None, None, bits, 64, [0.0] * ns, 1e-4, bg_color=1.0, want_counts=True, cams=(poses, intr, hw, hw))
    torch.cuda.synchronize()
    runs.append((out["image"].flatten(0, 1).clone(), out["depth"].flatten().clone(), dec.last_render_stats["sample_counts"].flatten().clone(), trace))
dis = torch.zeros(ns * N, dtype=torch.bool, device=dev)
for r in runs[1:]:
    dis |= (r[0] != runs[0][0]).any(-1) | (r[1] != runs[0][1]) | (r[2] != runs[0][2]) | (r[3] != runs[0][3]).any(-1)
idx = dis.nonzero().flatten()
print(f"{n} renders: {idx.numel()} rays on which they do not all agree (outputs or hashes); samples {int(runs[0][2].sum())}")
names = ["t", "features", "mlp_out", "sh", "reload", "composite", "texels", "weights"]
pat = collections.Counter(); first = collections.Counter(); shown = 0
cpu = [(r[0][idx].cpu(), r[1][idx].cpu(), r[2][idx].cpu(), r[3][idx].cpu()) for r in runs]
for j in range(idx.numel()):
    vals = [(tuple(c[3][j].tolist()), tuple(c[0][j].tolist()), float(c[1][j]), int(c[2][j])) for c in cpu]
    maj, votes = collections.Counter(vals).most_common(1)[0]
    for k, v in enumerate(vals):
        if v == maj:
            continue
        d = tuple(names[q] for q in range(8) if v[0][q] != maj[0][q])
        out_differs = v[1:] != maj[1:]
        pat[(d, out_differs)] += 1
        first[d[0] if d else ("(no hash)" if out_differs else "?")] += 1
        if shown < 12:
            shown += 1
            i = int(idx[j]); s_, rem = divmod(i, N); view, pix = divmod(rem, hw * hw); y, x = divmod(pix, hw)
            print(f" render {k} scene {s_} view {view} px ({x},{y}) votes {votes}/{n}: hashes differing {d}; count {v[3]} vs {maj[3]}; depth {v[2]:.6f} vs {maj[2]:.6f}")
print("deviating (render, ray) by set of differing hashes, outputs differ?:")
for (d, o), c_ in pat.most_common(20):
    print(f"   {c_:6d}  {d}  outputs differ: {o}")
import collections as _c
sc = _c.Counter(int(i) // N for i in idx.tolist())
print("deviating rays per scene:", sorted(sc.items()))
print("by first differing hash in the order t, features, mlp_out, sh, reload, composite:", dict(first))
