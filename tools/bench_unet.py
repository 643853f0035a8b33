#!/usr/bin/env python
"""tools/bench_unet.py -- DDIM-step timing of the cars UNet (DenoisingUnetMod, 122.4 M params, 218 GFLOP/scene forward) on one MI355X:
eager module forward vs the inference executor (ssdnerf_amd/unet_fast.py), fp32 and bf16, optional per-kernel breakdown."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssdnerf_amd  # noqa
from ssdnerf_amd.registry import MODULES

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=8); ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--dtypes", default="fp32,bf16"); ap.add_argument("--modes", default="eager,fast")
ap.add_argument("--tune", action="store_true", help="torch.backends.cudnn.benchmark=True (MIOpen exhaustive find: minutes)")
ap.add_argument("--lib-fp32-conv", action="store_true", help="fp32 executor with the library convolution instead of the bf16 x 2 kernel")
ap.add_argument("--no-graph", action="store_true", help="launch the executor's kernels eagerly (needed for a complete profiler table)")
ap.add_argument("--profile", default="", help="mode:dtype to print a torch-profiler kernel table for")
ap.add_argument("--layout", default="cars", choices=["cars", "tiled"], help="tiled: configs/new_cfgs/ssdnerf_cars_recons1v_tiled.py:15-28 (base 80, 6 x 128 x 384 input, GroupNorm(16))")
a = ap.parse_args()
torch.backends.cudnn.benchmark = a.tune
from ssdnerf_amd import unet_fast
unet_fast.FastUnet.capture_by_default = not a.no_graph
if a.lib_fp32_conv: unet_fast._Conv.F32X2 = False
if a.layout == "tiled":
    net = MODULES.build(dict(type="DenoisingUnetMod", image_size=128, in_channels=6, base_channels=80, channels_cfg=[1, 1, 2, 2, 4, 4], resblocks_per_downsample=2,
                             dropout=0.0, use_scale_shift_norm=True, downsample_conv=True, upsample_conv=True, num_heads=4, attention_res=[16, 8, 4],
                             norm_cfg=dict(type="GN", num_groups=16))).cuda().eval()
else:
    net = MODULES.build(dict(type="DenoisingUnetMod", image_size=128, in_channels=18, base_channels=128, channels_cfg=[1, 2, 2, 4, 4], resblocks_per_downsample=2,
                             dropout=0.0, use_scale_shift_norm=True, downsample_conv=True, upsample_conv=True, num_heads=4, attention_res=[32, 16, 8])).cuda().eval()
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for p in net.parameters():
        p.copy_(torch.randn(p.shape, generator=g).cuda() * 0.02)
x = torch.randn(a.scenes, 6, 128, 384, device="cuda") if a.layout == "tiled" else torch.randn(a.scenes, 18, 128, 128, device="cuda")
t = torch.full((a.scenes,), 500, device="cuda")
FLOP = (1.49e11 if a.layout == "tiled" else 2.18e11) * a.scenes
DT = dict(fp32=None, bf16=torch.bfloat16, fp16=torch.float16)
res, ref = {}, None


def run(mode, dt):
    net.fast_inference = mode == "fast"
    with torch.no_grad(), torch.autocast("cuda", enabled=dt is not None, dtype=dt):
        return net(x, t)


for mode in a.modes.split(","):
    for name in a.dtypes.split(","):
        t0 = time.perf_counter()
        for _ in range(2): y = run(mode, DT[name])
        torch.cuda.synchronize(); setup = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(a.iters): y = run(mode, DT[name])
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / a.iters * 1e3
        if ref is None: ref = y.float()
        res[f"{mode}:{name}"] = dict(ms_per_forward=round(ms, 3), tflops=round(FLOP / ms / 1e9, 1), setup_s=round(setup, 1),
                                     rel_err_vs_first=float((y.float() - ref).norm() / ref.norm()))
        print(json.dumps({f"{mode}:{name}": res[f"{mode}:{name}"]}), flush=True)
if net.__dict__.get("_fast_cache"):
    res["library_fallback_ops"] = {str(k): ex.library_fallbacks for k, ex in net.__dict__["_fast_cache"].items()}
print(json.dumps(res))
if a.profile:
    from torch.profiler import profile, ProfilerActivity
    mode, name = a.profile.split(":")
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        run(mode, DT[name]); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=90))
