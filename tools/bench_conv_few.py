#!/usr/bin/env python
"""tools/bench_conv_few.py hints... -- a handful of large UNet layers, microseconds per tile hint (quick A/B of kernel variants)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssdnerf_amd import unet_fast
hints = [int(v) for v in sys.argv[1:]] or [0, 5, 6]
LAYERS = [(128, 128, 128, 3, 1, 0), (128, 256, 128, 3, 1, 0), (128, 384, 128, 3, 1, 0), (64, 256, 256, 3, 1, 0), (64, 512, 256, 3, 1, 0), (64, 256, 256, 3, 1, 1),
          (128, 256, 128, 1, 1, 0), (128, 128, 128, 3, 2, 0)]
for (H, Cin, Cout, k, stride, up) in LAYERS:
    x = torch.randn(8, Cin, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, device="cuda") * 0.02).bfloat16().contiguous(memory_format=torch.channels_last)
    bias = torch.randn(Cout, device="cuda")
    Ho = ((2 * H if up else H) + 2 * (k // 2) - k) // stride + 1
    flop = 2.0 * 8 * Ho * Ho * Cout * Cin * k * k
    out = []
    for h in hints:
        fn = lambda: unet_fast.conv2d_nhwc_bf16(x, w, bias, None, stride, bool(up), tile_hint=h)
        for _ in range(3): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30): fn()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 30 * 1e3
        out.append(f"hint {h}: {us:7.1f} us {flop / us / 1e6:6.0f} TF")
    print(f"{H:4d} {Cin:4d}->{Cout:4d} k{k} s{stride} up{up}   " + "   ".join(out), flush=True)
