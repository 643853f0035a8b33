#!/usr/bin/env python
"""tools/pipeline_probe2.py -- does stage A (ssdnerf_render_first_hit_cams) of render i+1 hide under the shading kernel of render i?  Back-to-back renders of the bench
workload, (a) both stages on one stream (the product's order), (b) stage A of the next render on a second stream with its own workspace and output tensors, the
shading launches in order on the main stream.  Prints ms per render and checks that (b) renders the same bits."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssdnerf_amd import synthetic as S
from ssdnerf_amd import _cabi as C
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
params = dec.packed_params()
pose = poses.reshape(ns, nv, 16).contiguous(); k = intr.contiguous()
n = nv * hw * hw
need = C.lib().ssdnerf_render_queue_workspace(ns, n, 64)
sets = []
for _ in range(2):
    sets.append(dict(ws=torch.empty(need, dtype=torch.uint8, device=dev), im=torch.empty(ns, n, 3, device=dev), dp=torch.empty(ns, n, device=dev), w=torch.empty(ns, n, device=dev),
                     im8=torch.empty(ns, n, 3, dtype=torch.uint8, device=dev), ov=torch.zeros(1, dtype=torch.int32, device=dev), fh=torch.cuda.Event(), sh=torch.cuda.Event()))
flags = dec._shade_flags()


def first_hit(b):
    C.check(C.lib().ssdnerf_render_first_hit_cams(C.ptr(bits), C.u32(64), C.ptr(pose), C.ptr(k), C.u32(ns), C.u32(nv), C.u32(hw), C.u32(hw), C.f32(dec.bound), C.f32(dec.min_near),
            C.f32(0.0), C.ptr(None), C.u32(dec.max_steps), C.f32(1.0), C.ptr(b["im"]), C.ptr(b["dp"]), C.ptr(b["w"]), C.ptr(None), C.ptr(b["im8"]), C.ptr(b["ws"]),
            C.ctypes.c_size_t(need), C.stream()), "first_hit")


def shade(b):
    C.check(C.lib().ssdnerf_render_shade_queue_mfma_cams(C.ptr(planes), C.dtype_code(planes) | flags, C.u32(128), C.u32(128), C.ptr(params), C.u32(64), C.ptr(pose), C.ptr(k), C.u32(ns),
            C.u32(nv), C.u32(hw), C.u32(hw), C.f32(dec.bound), C.f32(dec.min_near), C.f32(0.0), C.ptr(None), C.u32(dec.max_steps), C.f32(1e-4), C.f32(1.0),
            C.f32(dec.sigmoid_saturation), C.ptr(b["im"]), C.ptr(b["dp"]), C.ptr(b["w"]), C.ptr(None), C.ptr(b["ov"]), C.ptr(b["im8"]), C.ptr(b["ws"]), C.ctypes.c_size_t(need),
            C.stream()), "shade")


def serial(steps):
    for i in range(steps):
        b = sets[i & 1]
        first_hit(b); shade(b)


side = torch.cuda.Stream(priority=int(os.environ.get("PP_PRIO", "0")))


def pipelined(steps):
    main = torch.cuda.current_stream()
    for i in range(steps):
        b = sets[i & 1]
        with torch.cuda.stream(side):
            side.wait_event(b["sh"])                 # this set's previous shading pass (two renders ago) is done
            first_hit(b)
            b["fh"].record(side)
        main.wait_event(b["fh"])
        shade(b)
        b["sh"].record(main)


def timeit(fn, steps=40):
    fn(10); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(steps); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for b in sets:
    b["sh"].record(torch.cuda.current_stream())
serial(2); torch.cuda.synchronize()
ref = (sets[0]["im"].clone(), sets[0]["dp"].clone(), sets[0]["im8"].clone())
for rep in range(3):
    print(f"serial    {timeit(serial):.3f} ms per render")
    print(f"pipelined {timeit(pipelined):.3f} ms per render")
pipelined(6); torch.cuda.synchronize()
print("pipelined renders bit-identical to the serial one:", all(torch.equal(a, b) for a, b in zip(ref, (sets[1]["im"], sets[1]["dp"], sets[1]["im8"]))), int(sets[0]["ov"]), int(sets[1]["ov"]))
