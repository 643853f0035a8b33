#!/usr/bin/env python
"""tools/bench_unet.py -- DDIM-step timing of the cars UNet (DenoisingUnetMod, 122.4 M params, 218 GFLOP/scene forward) on one MI355X:
fp32 vs bf16 autocast, per-op-class breakdown from the torch profiler.  Used to decide which UNet blocks to hand-write first."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssdnerf_amd  # noqa
from ssdnerf_amd.registry import MODULES

ap = argparse.ArgumentParser(); ap.add_argument("--scenes", type=int, default=8); ap.add_argument("--iters", type=int, default=5); ap.add_argument("--profile", action="store_true")
a = ap.parse_args()
torch.backends.cudnn.benchmark = True
net = MODULES.build(dict(type="DenoisingUnetMod", image_size=128, in_channels=18, base_channels=128, channels_cfg=[1, 2, 2, 4, 4], resblocks_per_downsample=2,
                         dropout=0.0, use_scale_shift_norm=True, downsample_conv=True, upsample_conv=True, num_heads=4, attention_res=[32, 16, 8])).cuda().eval()
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for p in net.parameters():
        p.copy_(torch.randn(p.shape, generator=g).cuda() * 0.02)
x = torch.randn(a.scenes, 18, 128, 128, device="cuda"); t = torch.full((a.scenes,), 500, device="cuda")
FLOP = 2.18e11 * a.scenes
res = {}
for name, dt in (("fp32", None), ("bf16", torch.bfloat16), ("fp16", torch.float16)):
    with torch.no_grad(), torch.autocast("cuda", enabled=dt is not None, dtype=dt):
        for _ in range(2): y = net(x, t)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.iters): y = net(x, t)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / a.iters * 1e3
    res[name] = dict(ms_per_forward=ms, tflops=FLOP / ms / 1e9)
print(json.dumps(res))
if a.profile:
    from torch.profiler import profile, ProfilerActivity
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16), profile(activities=[ProfilerActivity.CUDA]) as prof:
        net(x, t); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
