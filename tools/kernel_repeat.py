#!/usr/bin/env python
"""tools/kernel_repeat.py -- run-to-run reproducibility of the UNet's deterministic kernels (GPU box): attention (bf16 / fp32-class) and the unsplit
convolutions (bf16 / fp32-class, generic and row-reuse forms) called repeatedly on the same inputs must return the same bits (no atomics on
these paths).  A difference would be a hardware hazard of the kind found in the shading kernel (csrc/shade_mfma.hip, sm_operand_guard)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssdnerf_amd import unet_fast as UF
torch.manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
def repeat(name, fn):
    ref = fn().clone(); bad = 0
    for _ in range(N):
        if not torch.equal(fn(), ref): bad += 1
    print(f"{name:44s} {bad} of {N} repeats differ")
for (B, T, heads, ch) in ((8, 1024, 4, 64), (8, 256, 4, 128), (8, 64, 4, 128)):
    qkv = torch.randn(B, T, 3 * heads * ch, device="cuda")
    repeat(f"attention fp32-class T={T} ch={ch}", lambda: UF.attention_qkv_f32(qkv, heads))
    q16 = qkv.bfloat16()
    repeat(f"attention bf16 T={T} ch={ch}", lambda: UF.attention_qkv_bf16(q16, heads))
for (H, Cin, Cout, k) in ((128, 128, 128, 3), (64, 256, 256, 3), (64, 512, 256, 1), (32, 256, 256, 3)):
    x = torch.randn(8, Cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, device="cuda") * 0.02).contiguous(memory_format=torch.channels_last)
    bias = torch.randn(Cout, device="cuda")
    w_hi, w_lo = [t.contiguous(memory_format=torch.channels_last) for t in UF.split_bf16x2(w)]
    for hint, tag in ((0, "auto"), (1, "generic 128x128")):
        os.environ.pop("SSDNERF_CONV_NO_ROW_REUSE", None)
        repeat(f"conv fp32-class {Cin}->{Cout} k{k} @{H} [{tag}]", lambda: UF.conv2d_nhwc_f32x2(x, w_hi, w_lo, bias, None, 1, False, tile_hint=hint, splits_hint=1))
        x16, w16 = x.bfloat16().contiguous(memory_format=torch.channels_last), w.bfloat16().contiguous(memory_format=torch.channels_last)
        repeat(f"conv bf16 {Cin}->{Cout} k{k} @{H} [{tag}]", lambda: UF.conv2d_nhwc_bf16(x16, w16, bias, None, 1, False, tile_hint=hint, splits_hint=1))
