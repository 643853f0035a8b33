import os, sys, torch
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
out = dec.render_packed(planes, None, None, bits, 64, [0.0] * ns, 1e-4, bg_color=1.0, want_counts=True, cams=(poses, intr, hw, hw))
for k, v in dec._ws_cache.items():
    line = v[:5 * ns * 128].view(torch.int32).view(5, ns, 32)
    print(k, v.numel(), v.data_ptr(), line[:, :, 0].tolist())
    print(line[3, 0].tolist())

import struct
def fl(u): return struct.unpack("f", struct.pack("I", u & 0xffffffff))[0]
for v in dec._ws_cache.values():
    line = v[:5 * ns * 128].view(torch.int32).view(5, ns, 32)
    for k in range(1, 8):
        e = line[3, k, 8:24].tolist()
        if e[0] == 0: continue
        print(f"case {k}: lane {e[0] & 255} feature {e[0] >> 8 & 255} f2!=f {e[0] >> 16 & 1} f3!=f {e[0] >> 17 & 1}: f {fl(e[1]):.9g} f2 {fl(e[2]):.9g} f3 {fl(e[3]):.9g} | t00 {fl(e[4]):.6g} t01 {fl(e[5]):.6g} t10 {fl(e[6]):.6g} t11 {fl(e[7]):.6g} | w {fl(e[8]):.6g} {fl(e[9]):.6g} {fl(e[10]):.6g} {fl(e[11]):.6g} | other-channel t00 {fl(e[12]):.6g} t01 {fl(e[13]):.6g}")
        t00, t01, t10, t11, w00, w01, w10, w11 = [fl(x) for x in e[4:12]]
        print("    partial sums: t00*w00", t00 * w00, " +t01*w01", t00 * w00 + t01 * w01, " +t10*w10", t00 * w00 + t01 * w01 + t10 * w10, " all", t00 * w00 + t01 * w01 + t10 * w10 + t11 * w11)
