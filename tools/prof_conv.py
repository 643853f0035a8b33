#!/usr/bin/env python
"""tools/prof_conv.py -- run a few UNet convolution layers through the hand-written implicit GEMM, for rocprofv3 (--kernel-trace / --pmc)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssdnerf_amd import unet_fast
ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=3); ap.add_argument("--hint", type=int, default=0)
a = ap.parse_args()
LAYERS = [(8, 128, 128, 128, 3), (8, 64, 256, 256, 3), (8, 64, 512, 256, 3), (8, 32, 256, 256, 3)]
for (B, H, Cin, Cout, k) in LAYERS:
    x = torch.randn(B, Cin, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, device="cuda") * 0.02).bfloat16().contiguous(memory_format=torch.channels_last)
    bias = torch.randn(Cout, device="cuda")
    for _ in range(a.iters):
        unet_fast.conv2d_nhwc_bf16(x, w, bias, None, tile_hint=a.hint)
torch.cuda.synchronize()
