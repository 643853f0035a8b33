#!/usr/bin/env python
"""tools/bench_attention.py -- the UNet's attention sites (8 scenes x 4 heads; T = 1024 / 256 / 64, head width 64 / 128 / 128), fp32-class and bf16 forward, us per call
(SSDNERF_ATTN_WPB=4|2|1 forces the waves-per-block form of k_attn_fwd)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssdnerf_amd import unet_fast as UF
B, heads = int(os.environ.get("B", "8")), 4
for T, ch in ((1024, 64), (256, 128), (64, 128)):
    for dt in (torch.float32, torch.bfloat16):
        qkv = torch.randn(B, T, 3 * heads * ch, device="cuda").to(dt)
        for _ in range(5): UF.attention_qkv(qkv, heads)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): UF.attention_qkv(qkv, heads)
        e.record(); torch.cuda.synchronize()
        line = f"T={T:5d} ch={ch:4d} {str(dt)[6:]:9s} forward {s.elapsed_time(e) / 50 * 1e3:7.1f} us"
        if dt == torch.float32:                               # the gradient path's backward (k_attn_bwd_D + dq + dkv), fp32 only
            out, lse = UF.attention_qkv_f32_with_lse(qkv, heads)
            dout = torch.randn_like(out)
            for _ in range(3): UF.attention_qkv_f32_backward(qkv, out, dout, lse, heads)
            s.record()
            for _ in range(30): UF.attention_qkv_f32_backward(qkv, out, dout, lse, heads)
            e.record(); torch.cuda.synchronize()
            line += f"   backward {s.elapsed_time(e) / 30 * 1e3:7.1f} us"
        print(line)
