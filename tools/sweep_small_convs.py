#!/usr/bin/env python
"""tools/sweep_small_convs.py -- tile / split-K choice of the low-resolution UNet layers (<= 32 x 32): time every (tile hint, splits) pair per layer,
print the best against what the library's own plan picks."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssdnerf_amd import unet_fast
DT = sys.argv[1] if len(sys.argv) > 1 else "bf16"                  # bf16 | fp32 (the fp32-class bf16 x 2 kernels) | ps (r04: fp32-class on PRE-SPLIT activations, stride-1 layers)
B = 8
WS = torch.zeros(4 << 20, dtype=torch.float32, device="cuda")
LAYERS = [(64, 128, 256, 3, 1, 0), (64, 128, 256, 1, 1, 0), (64, 512, 256, 1, 1, 0), (64, 256, 256, 3, 2, 0), (128, 128, 128, 3, 2, 0), (128, 256, 128, 1, 1, 0),
          (32, 256, 256, 3, 1, 0), (32, 512, 256, 3, 1, 0), (32, 768, 256, 3, 1, 0), (32, 512, 256, 1, 1, 0), (32, 256, 256, 3, 2, 0),
          (16, 512, 512, 3, 1, 0), (16, 1024, 512, 3, 1, 0), (16, 256, 512, 3, 1, 0), (16, 1024, 512, 1, 1, 0), (16, 512, 512, 3, 2, 0), (16, 512, 512, 3, 1, 1),
          (8, 512, 512, 3, 1, 0), (8, 1024, 512, 3, 1, 0), (8, 1024, 512, 1, 1, 0), (8, 512, 512, 3, 1, 1)]


def timeit(fn, iters=30):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for (H, Cin, Cout, k, stride, up) in LAYERS:
    x = torch.randn(B, Cin, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, device="cuda") * 0.02).bfloat16().contiguous(memory_format=torch.channels_last)
    bias = torch.randn(Cout, device="cuda")
    res = {}
    if DT == "ps":
        if stride != 1 or up:
            continue
        x = x.float().contiguous(memory_format=torch.channels_last)                  # (timing only: any bytes do as the pre-split carrier)
        hi, lo = unet_fast.split_bf16x2_adjacent(w.float())
        fly_hi, fly_lo = hi, lo
        res["on_the_fly_auto"] = round(timeit(lambda: unet_fast.conv2d_nhwc_f32x2(x, fly_hi, fly_lo, bias, None, 1, False, splitk_ws=WS)), 1)
        run = lambda hint, sp: unet_fast.conv2d_nhwc_f32x2_presplit(x, hi, lo, bias, None, splitk_ws=WS, tile_hint=hint, splits_hint=sp)
    elif DT == "fp32":
        x = x.float().contiguous(memory_format=torch.channels_last)
        hi, lo = [t.contiguous(memory_format=torch.channels_last) for t in unet_fast.split_bf16x2(w.float())]
        run = lambda hint, sp: unet_fast.conv2d_nhwc_f32x2(x, hi, lo, bias, None, stride, bool(up), tile_hint=hint, splits_hint=sp)
    else:
        run = lambda hint, sp: unet_fast.conv2d_nhwc_bf16(x, w, bias, None, stride, bool(up), tile_hint=hint, splitk_ws=WS, splits_hint=sp)
    auto = timeit(lambda: run(0, 0))
    for hint in ((1, 3) if DT == "fp32" else (1, 2, 3)):
        for sp in (1, 2, 3, 4, 6, 8, 12, 16):
            try:
                res[f"{hint}/{sp}"] = round(timeit(lambda: run(hint, sp)), 1)
            except RuntimeError as e:
                res[f"{hint}/{sp}"] = None
    fly = res.pop("on_the_fly_auto", None)
    ok = {k2: v for k2, v in res.items() if v}
    best = min(ok, key=ok.get)
    print(json.dumps(dict(H=H, Cin=Cin, Cout=Cout, k=k, stride=stride, up=up, auto_us=round(auto, 1), on_the_fly_auto_us=fly, best=best, best_us=ok[best],
                          top=sorted(ok.items(), key=lambda kv: kv[1])[:5])), flush=True)
