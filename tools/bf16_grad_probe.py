"""r06: the native bf16 gradient path of the UNet (unet._ConvBf16Fn, DenoisingUnetMod.grad_path_bf16_native) against the fp32-class path and against the eager
modules under bf16 autocast (the reference's arithmetic for config 5), on the full cars UNet at the bench shape: output / input-gradient distances to the fp32-class
run, library calls, and the time of one input-gradient call (eager and through the captured graphs).
  usage: python tools/bf16_grad_probe.py [--scenes 8] [--iters 20] [--small]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--tiled", action="store_true", help="the tiled-triplane UNet (configs/new_cfgs/ssdnerf_cars_recons1v_tiled.py: base 80, 16 groups, 6 x 128 x 384 input)")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--aten", action="store_true", help="with --profile: the aten:: elementwise operators of one call by input shape (copies, casts, accumulations), both arithmetic classes")
    ap.add_argument("--kernels", default="", help="with --profile: print only the kernels whose name contains one of these comma-separated strings, for both arithmetic classes")
    args = ap.parse_args()
    import ssdnerf_amd  # noqa: F401
    from ssdnerf_amd import unet as U
    from test_unet_fast_gpu import _bench_unet, _unet
    net = _unet(seed=4) if args.small else _bench_unet()
    cin, hh, ww = 18, (32 if args.small else 128), (32 if args.small else 128)
    if args.tiled:
        from ssdnerf_amd.registry import MODULES
        net = MODULES.build(dict(type="DenoisingUnetMod", image_size=128, in_channels=6, base_channels=80, channels_cfg=[1, 1, 2, 2, 4, 4], resblocks_per_downsample=2, dropout=0.0,
                                 use_scale_shift_norm=True, downsample_conv=True, upsample_conv=True, num_heads=4, attention_res=[16, 8, 4],
                                 norm_cfg=dict(type="GN", num_groups=16))).cuda().eval()
        gw = torch.Generator().manual_seed(0)
        with torch.no_grad():
            for p_ in net.parameters():
                p_.copy_((torch.randn(p_.shape, generator=gw) * 0.03).cuda())
        cin, hh, ww = 6, 128, 384
    net.requires_grad_(False)
    B = args.scenes
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(B, cin, hh, ww, generator=g).cuda()
    t = torch.tensor([999, 979, 600, 339, 120, 59, 19, 0] * ((B + 7) // 8))[:B].cuda()
    probe = torch.randn(B, cin, hh, ww, generator=g).cuda()

    def call(autocast, native, eager, graph=False):
        net.grad_graph = graph
        net.grad_path_bf16_native = native
        net.grad_path_fp32_under_autocast = not eager
        x = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            y = net(x, t)
        (gx,) = torch.autograd.grad((y.float() * probe).sum(), x)
        return y.detach().float(), gx.float()

    arms = dict(fp32_class=(False, False, False), bf16_native=(True, True, False), eager_autocast=(True, False, True), fp32_class_under_autocast=(True, False, False))
    res = {}
    for name, a in arms.items():
        U._Conv2d.library_calls = 0
        res[name] = call(*a)
        print(f"{name:28s} library convolution calls {U._Conv2d.library_calls}; y finite {bool(torch.isfinite(res[name][0]).all())}, gx finite {bool(torch.isfinite(res[name][1]).all())}", flush=True)
    y0, g0 = res["fp32_class"]
    yb, gb = call(*arms["bf16_native"])                                         # the same arm again: what the split-K atomics' order alone moves
    print(f"bf16_native, second eager run vs the first: y rel L2 {float((yb - res['bf16_native'][0]).norm() / y0.norm()):.3e} | gx rel L2 {float((gb - res['bf16_native'][1]).norm() / g0.norm()):.3e}", flush=True)
    for name in ("bf16_native", "eager_autocast", "fp32_class_under_autocast"):
        y, gx = res[name]
        print(f"{name:28s} vs fp32_class: y rel L2 {float((y - y0).norm() / y0.norm()):.3e} max {float((y - y0).abs().max() / y0.abs().max()):.3e} | "
              f"gx rel L2 {float((gx - g0).norm() / g0.norm()):.3e} max {float((gx - g0).abs().max() / g0.abs().max()):.3e}", flush=True)

    def timeit(a, graph):
        for _ in range(5):
            call(*a, graph=graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            call(*a, graph=graph)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.iters * 1e3
    for name in ("fp32_class", "bf16_native", "eager_autocast"):
        print(f"{name:28s} input-gradient call: eager launches {timeit(arms[name], False):7.2f} ms, captured graphs {timeit(arms[name], True):7.2f} ms", flush=True)
    print("graphs:", net.grad_graph_info())
    if args.profile:
        from torch.profiler import profile, ProfilerActivity
        if args.aten:
            for name in ("fp32_class", "bf16_native"):
                with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as prof:
                    call(*arms[name], graph=False)
                    torch.cuda.synchronize()
                rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.self_device_time_total > 0]
                tot = sum(e.self_device_time_total for e in rows)
                print(f"== {name}: aten operators with device time, one call: {tot * 1e-3:.3f} ms")
                for e in sorted(rows, key=lambda e: -e.self_device_time_total)[:40]:
                    print(f"  {e.key:28s} x{e.count:3d} {e.self_device_time_total * 1e-3:7.3f} ms  {str(e.input_shapes)[:150]}")
            return
        want = [k for k in args.kernels.split(",") if k]
        for name in (("fp32_class", "bf16_native") if want else ("bf16_native",)):
            with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
                for _ in range(3):
                    call(*arms[name], graph=False)
                torch.cuda.synchronize()
            if not want:
                print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=70))
                continue
            for e in sorted(prof.key_averages(), key=lambda e: -e.self_device_time_total):
                if any(k in e.key for k in want) and e.self_device_time_total > 0:
                    print(f"  {name:12s} {e.key[:70]:70s} {e.count // 3:4d} launches per call  {e.self_device_time_total / 3e3:7.3f} ms per call  ({e.self_device_time_total / e.count:6.1f} us each)")


if __name__ == "__main__":
    main()
