#!/usr/bin/env python
"""tools/pp_timing.py -- with a -DCV_PP_TIMING build of the library (SSDNERF_HIP_LIB): per-tile shader-clock stamps of the two-group convolution
kernel (tile start, prologue done, K loop done, stores acknowledged), summarised over the blocks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ssdnerf_amd import unet_fast
for (H, Cin, Cout, k, hint) in [(128, 128, 128, 3, 6), (128, 384, 128, 3, 6), (64, 256, 256, 3, 6), (128, 256, 128, 1, 5)]:
    x = torch.randn(8, Cin, H, H, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, device="cuda") * 0.02).bfloat16().contiguous(memory_format=torch.channels_last)
    bias = torch.randn(Cout, device="cuda")
    tiles = (8 * H * H // 256) * (Cout // 128)
    ws = torch.zeros(tiles * 8 + 64, dtype=torch.float32, device="cuda")
    for rep in range(3):
        ws.zero_()
        unet_fast.conv2d_nhwc_bf16(x, w, bias, None, tile_hint=hint, splitk_ws=ws)
    torch.cuda.synchronize()
    st = ws.cpu().numpy().view(np.int64)[:tiles * 4].reshape(tiles, 4).astype(np.float64)
    t0 = st[:, 0].min()
    st = (st - t0) / 100.0                                  # s_memtime ticks at 100 MHz on this part? printed raw as well
    grid = min(tiles, 256)
    first, second = st[:grid], st[grid:]
    def stat(a, name):
        print(f"   {name:28s} start {a[:,0].mean():8.2f}  prologue+{(a[:,1]-a[:,0]).mean():7.2f}  kloop+{(a[:,2]-a[:,1]).mean():7.2f}  epilogue(acked)+{(a[:,3]-a[:,2]).mean():7.2f}  end mean {a[:,3].mean():8.2f} max {a[:,3].max():8.2f}")
    print(f"{H} {Cin}->{Cout} k{k} hint {hint}: {tiles} tiles on {grid} blocks (units: s_memtime ticks / 100)")
    stat(first, "first tile of each block")
    if len(second): stat(second, "second tile")
