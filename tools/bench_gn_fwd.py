"""r06: the GroupNorm forward of the UNet (ssdnerf_group_norm_nhwc: statistics + normalisation passes; with run-level statistics only the normalisation) per layer shape,
fp32 and bf16: microseconds per call and the rate of the bytes it has to move.   usage: python tools/bench_gn_fwd.py [--scenes 8] [--reps 50]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_gn_bwd import SHAPES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=8)
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    import ssdnerf_amd  # noqa: F401
    from ssdnerf_amd import unet_fast as UF
    B, G = args.scenes, 32
    g = torch.Generator().manual_seed(0)
    for dtype in (torch.float32, torch.bfloat16):
        tot = [0.0, 0.0]
        for C, side, count in SHAPES:
            x = torch.randn(B, C, side, side, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
            gamma, beta = torch.rand(C, generator=g).cuda() + 0.5, torch.randn(C, generator=g).cuda() * 0.1
            ss = (torch.randn(B, 2 * C, generator=g) * 0.1).cuda()
            runs = torch.stack([x.double().sum((2, 3)).view(B, -1, 4).sum(2), x.double().square().sum((2, 3)).view(B, -1, 4).sum(2)], dim=-1).reshape(-1).contiguous()
            y = torch.empty_like(x)
            ws = torch.zeros(args.reps + 5, B * G * 2, dtype=torch.float64, device="cuda")
            us = []
            for mode in ("stats+apply", "runs"):
                def call(i):
                    if mode == "runs":
                        UF.group_norm_nhwc(x, G, gamma, beta, ss, 1e-5, True, None, out=y, runs=(runs, None))
                    else:
                        UF.group_norm_nhwc(x, G, gamma, beta, ss, 1e-5, True, ws[i], out=y, workspace_is_zero=True)
                for i in range(5):
                    call(i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(args.reps):
                    call(5 + i)
                e1.record()
                torch.cuda.synchronize()
                us.append(e0.elapsed_time(e1) / args.reps * 1e3)
            nb = x.numel() * x.element_size()
            tot[0] += us[0] * count; tot[1] += us[1] * count
            print(f"{str(dtype):15s} C {C:5d} {side:3d}x{side:<3d} x{count:2d}: statistics + normalisation {us[0]:7.1f} us ({3 * nb / us[0] * 1e-6:5.2f} TB/s of 3 passes)   "
                  f"normalisation from run-level statistics {us[1]:7.1f} us ({2 * nb / us[1] * 1e-6:5.2f} TB/s of 2 passes)")
        print(f"{str(dtype):15s} all norms of one forward: {tot[0] * 1e-3:.3f} ms with a statistics pass each, {tot[1] * 1e-3:.3f} ms from run-level statistics")


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
