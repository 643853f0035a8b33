"""r06: the GroupNorm backward of the UNet's gradient path (ssdnerf_group_norm_nhwc_backward: k_gn_bwd_stats + k_gn_bwd_apply) per layer shape of the cars UNet,
fp32 and bf16: microseconds per call (both passes) and the bytes it has to move (2 reads of x and dy, 1 write of dx) as a rate.
  usage: python tools/bench_gn_bwd.py [--scenes 8] [--reps 50]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (channels, side, roughly how many norms of this shape one input-gradient call of the cars UNet runs: base 128, [1, 2, 2, 4, 4], 2 blocks per level, 71 norms)
SHAPES = [(128, 128, 8), (256, 128, 3), (128, 64, 1), (256, 64, 8), (512, 64, 2), (384, 64, 1), (256, 32, 12), (512, 32, 3), (768, 32, 1), (512, 16, 12), (256, 16, 1),
          (1024, 16, 2), (768, 16, 1), (512, 8, 14), (1024, 8, 3)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=8)
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    import ssdnerf_amd  # noqa: F401
    from ssdnerf_amd import unet_fast as UF
    B, G = args.scenes, 32
    g = torch.Generator().manual_seed(0)
    tot = {}
    for dtype in (torch.float32, torch.bfloat16):
        for C, side, count in SHAPES:
            x = torch.randn(B, C, side, side, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
            dy = torch.randn(B, C, side, side, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
            gamma, beta = torch.rand(C, generator=g).cuda() + 0.5, torch.randn(C, generator=g).cuda() * 0.1
            ss = (torch.randn(B, 2 * C, generator=g) * 0.1).cuda()
            sums = torch.zeros(B * G * 2, dtype=torch.float64, device="cuda")
            UF.group_norm_nhwc(x, G, gamma, beta, ss, 1e-5, True, sums, workspace_is_zero=True)
            ws = torch.zeros(args.reps + 5, UF.group_norm_backward_workspace_doubles(B, G), dtype=torch.float64, device="cuda")
            for i in range(5):
                UF.group_norm_nhwc_backward(x, dy, G, gamma, beta, ss, 1e-5, True, sums, workspace=ws[i])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(args.reps):
                UF.group_norm_nhwc_backward(x, dy, G, gamma, beta, ss, 1e-5, True, sums, workspace=ws[5 + i])
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / args.reps * 1e3
            from torch.profiler import profile, ProfilerActivity
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                for i in range(5):
                    UF.group_norm_nhwc_backward(x, dy, G, gamma, beta, ss, 1e-5, True, sums, workspace=ws[i + 5])
                torch.cuda.synchronize()
            per = {("stats" if "stats" in e.key else "apply"): e.self_device_time_total / e.count for e in prof.key_averages() if "k_gn_bwd" in e.key}
            nbytes = x.numel() * x.element_size() * 5
            tot[dtype] = tot.get(dtype, 0.0) + us * count
            print(f"{str(dtype):15s} C {C:5d} {side:3d}x{side:<3d} x{count:2d}: {us:7.1f} us per call (both passes)  {nbytes / us * 1e-6:6.2f} TB/s of 5 passes over the activation"
                  f"   [kernels: statistics {per.get('stats', 0):6.1f} us, apply {per.get('apply', 0):6.1f} us]")
        print(f"{str(dtype):15s} all norms of one input-gradient call: {tot[dtype] * 1e-3:.3f} ms")


if __name__ == "__main__":
    main()
