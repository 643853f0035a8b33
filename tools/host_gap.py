#!/usr/bin/env python
"""tools/host_gap.py [steps] -- where the host time of the timed step (``nerf.render``, 8 scenes x 251 views) goes: wall time per step against the
HIP-event time of its two launches, and perf_counter stamps at the entry / exit of the two C calls and of the flag read."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssdnerf_amd import synthetic as S, nerf, _cabi as C
from ssdnerf_amd.decoders import TriPlaneDecoder, pack_triplanes
from ssdnerf_amd.density import get_density
dev = torch.device("cuda")
dec = TriPlaneDecoder(base_layers=[18, 64], density_layers=[64, 1], color_layers=[64, 3], dir_layers=[16, 64], max_steps=256)
dec.load_state_dict(S.make_decoder_params(), strict=False); dec = dec.to(dev).eval()
g = torch.Generator().manual_seed(7); jit = [torch.rand(64 ** 3, 3, generator=g).to(dev) for _ in range(8)]
ns, nv, hw = 8, 251, 128
poses = S.spiral_poses(251).to(dev)[None].expand(ns, -1, -1, -1).contiguous(); intr = S.cars_intrinsics(hw, hw).to(dev)[None, None].expand(ns, nv, -1).contiguous()
code = torch.stack([S.make_triplane(sd, "object") for sd in range(2021, 2029)]).to(dev)
_, bits = get_density(dec, code, 64, density_thresh=0.1, density_step=8, jitters=jit)
planes = pack_triplanes(code)
stamps = []
real = C.lib()
class Proxy:
    def __getattr__(self, name):
        f = getattr(real, name)
        if name in ("ssdnerf_render_first_hit_cams", "ssdnerf_render_shade_queue_mfma_cams"):
            def w(*a):
                stamps.append((name[15:24] + ":in", time.perf_counter())); r = f(*a); stamps.append((name[15:24] + ":out", time.perf_counter())); return r
            return w
        return f
prox = Proxy()
C.lib = lambda: prox
import ssdnerf_amd.decoders as D
D.C.lib = C.lib
def step():
    return nerf.render(dec, code, bits, hw, hw, intr, poses, grid_size=64, bg_color=1.0, cfg={}, planes=planes, return_u8=True)
for _ in range(10): step()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rows = []
t_prev_end = None
for i in range(n):
    stamps.clear(); dec.stage_events = []
    t0 = time.perf_counter(); step(); t1 = time.perf_counter()
    ev = dec.stage_events; torch.cuda.synchronize()
    d = dict(stamps)
    rows.append(dict(wall=(t1 - t0) * 1e3, a_ms=ev[0].elapsed_time(ev[1]), b_ms=ev[1].elapsed_time(ev[2]), to_first_launch=(d["first_hit:in"] - t0) * 1e3,
                     first_call=(d["first_hit:out"] - d["first_hit:in"]) * 1e3, between=(d["shade_que:in"] - d["first_hit:out"]) * 1e3,
                     shade_call=(d["shade_que:out"] - d["shade_que:in"]) * 1e3, after_launch=(t1 - d["shade_que:out"]) * 1e3))
dec.stage_events = None
import statistics as st
for k in rows[0]:
    print(f"{k:16s} median {st.median(r[k] for r in rows):8.3f} ms   min {min(r[k] for r in rows):8.3f}   max {max(r[k] for r in rows):8.3f}")
print("wall - (A + B) median:", round(st.median(r['wall'] - r['a_ms'] - r['b_ms'] for r in rows), 3), "ms  (with stage events recorded: two extra event records per step)")
# the same loop un-instrumented
C.lib = lambda: real; D.C.lib = C.lib
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize(); print("un-instrumented ms/step:", round((time.perf_counter() - t0) / n * 1e3, 3))
# sustained rate: back-to-back steps, mean per block of 25 (clock behaviour over the first seconds), without and with the stage events
for label, with_ev in (("no events", False), ("stage events", True), ("no events", False)):
    torch.cuda.synchronize(); time.sleep(1.0)
    ts = [time.perf_counter()]
    for _ in range(250):
        dec.stage_events = [] if with_ev else None
        step(); ts.append(time.perf_counter())
    dec.stage_events = None
    print(f"sustained ({label}), ms/step per block of 25:", [round((ts[i + 25] - ts[i]) / 25 * 1e3, 3) for i in range(0, 250, 25)])
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(n): step()
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(18)
