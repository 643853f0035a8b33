#!/usr/bin/env python
"""tools/bench_conv.py -- every distinct convolution of the cars UNet (B scenes), hand-written implicit GEMM (csrc/conv_igemm.hip) vs the library
(MIOpen through torch, channels_last bf16, bias included): microseconds and TFLOP/s per layer, FLOP-weighted totals."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from ssdnerf_amd import unet_fast

ap = argparse.ArgumentParser(); ap.add_argument("--scenes", type=int, default=8); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--hints", default="0"); ap.add_argument("--no-lib", action="store_true")
ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"], help="bf16: k_conv_igemm_bf16; fp32: the fp32-class bf16x2 kernels vs the library's fp32 convolution")
ap.add_argument("--extra", default="", help="more layers, 'H,Cin,Cout,k;...' (stride 1): e.g. the stem / head after channel padding, 128,24,128,3;128,128,24,3")
a = ap.parse_args()
B = a.scenes
WS = torch.zeros(4 << 20, dtype=torch.float32, device="cuda")     # split-K scratch (all zero between calls)
# (H, Cin, Cout, k, stride, upsample, count) -- the layer census of DenoisingUnetMod(base 128, mult [1,2,2,4,4], 2 blocks/level, 128x128)
LAYERS = [
    (128, 128, 128, 3, 1, 0, 8), (128, 256, 128, 3, 1, 0, 2), (128, 256, 128, 1, 1, 0, 2), (128, 384, 128, 3, 1, 0, 1), (128, 384, 128, 1, 1, 0, 1),
    (128, 128, 128, 3, 2, 0, 1), (64, 256, 256, 3, 1, 1, 1),
    (64, 128, 256, 3, 1, 0, 1), (64, 128, 256, 1, 1, 0, 1), (64, 256, 256, 3, 1, 0, 6), (64, 512, 256, 3, 1, 0, 2), (64, 512, 256, 1, 1, 0, 2),
    (64, 384, 256, 3, 1, 0, 1), (64, 384, 256, 1, 1, 0, 1), (64, 256, 256, 3, 2, 0, 1), (32, 256, 256, 3, 1, 1, 1),
    (32, 256, 256, 3, 1, 0, 7), (32, 512, 256, 3, 1, 0, 2), (32, 512, 256, 1, 1, 0, 2), (32, 768, 256, 3, 1, 0, 1), (32, 768, 256, 1, 1, 0, 1),
    (32, 256, 256, 3, 2, 0, 1), (16, 512, 512, 3, 1, 1, 1),
    (16, 256, 512, 3, 1, 0, 1), (16, 256, 512, 1, 1, 0, 1), (16, 512, 512, 3, 1, 0, 6), (16, 1024, 512, 3, 1, 0, 2), (16, 1024, 512, 1, 1, 0, 2),
    (16, 768, 512, 3, 1, 0, 1), (16, 768, 512, 1, 1, 0, 1), (16, 512, 512, 3, 2, 0, 1), (8, 512, 512, 3, 1, 1, 1),
    (8, 512, 512, 3, 1, 0, 11), (8, 1024, 512, 3, 1, 0, 3), (8, 1024, 512, 1, 1, 0, 3),
]


for spec in [v for v in a.extra.split(";") if v]:
    h_, ci_, co_, k_ = (int(v) for v in spec.split(","))
    LAYERS.append((h_, ci_, co_, k_, 1, 0, 1))


def timeit(fn):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / a.iters * 1e3


tot = dict(lib=0.0, own=0.0, flop=0.0)
rows = []
for (H, Cin, Cout, k, stride, up, count) in LAYERS:
    DT = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    x = torch.randn(B, Cin, H, H, device="cuda").to(DT).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, device="cuda") * 0.02).to(DT).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(Cout, device="cuda")
    bias16 = bias.to(DT)
    if a.dtype == "fp32":
        w_hi, w_lo = [t.contiguous(memory_format=torch.channels_last) for t in unet_fast.split_bf16x2(w)]
    Hv = 2 * H if up else H
    Ho = (Hv + 2 * (k // 2) - k) // stride + 1
    flop = 2.0 * B * Ho * Ho * Cout * Cin * k * k
    if a.no_lib:
        t_lib = float("nan")
    elif up:
        t_lib = timeit(lambda: F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, bias16, 1, k // 2))
    else:
        t_lib = timeit(lambda: F.conv2d(x, w, bias16, stride, k // 2))
    best = None
    per_hint = {}
    for h in [int(v) for v in a.hints.split(",")]:
        if h in (1, 2, 4) and Cout % 128: continue
        if a.dtype == "fp32":
            if h in (2, 4): continue
            t = timeit(lambda: unet_fast.conv2d_nhwc_f32x2(x, w_hi, w_lo, bias, None, stride, bool(up), tile_hint=h))
        else:
            t = timeit(lambda: unet_fast.conv2d_nhwc_bf16(x, w, bias, None, stride, bool(up), tile_hint=h, splitk_ws=WS))
        per_hint[h] = round(t, 1)
        if h != 0 and (best is None or t < best[1]): best = (h, t)
    t_own = per_hint.get(0, best[1] if best else float("nan"))
    rows.append(dict(H=H, Cin=Cin, Cout=Cout, k=k, stride=stride, up=up, n=count, lib_us=round(t_lib, 1), own_us=per_hint, lib_tf=round(flop / t_lib / 1e6, 0),
                     own_tf=round(flop / t_own / 1e6, 0), best=best[0] if best else 0))
    print(json.dumps(rows[-1]), flush=True)
    tot["lib"] += t_lib * count; tot["own"] += t_own * count; tot["flop"] += flop * count
print(json.dumps(dict(total_lib_ms=tot["lib"] / 1e3, total_own_ms=tot["own"] / 1e3, gflop=tot["flop"] / 1e9, lib_tflops=tot["flop"] / tot["lib"] / 1e6,
                      own_tflops=tot["flop"] / tot["own"] / 1e6)))
