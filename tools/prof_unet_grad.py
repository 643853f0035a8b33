#!/usr/bin/env python
"""tools/prof_unet_grad.py -- where the NON-kernel time of the UNet's input-gradient path goes (guided sampling / fine-tuning): torch.profiler with
shapes and stacks over one forward + backward of the cars UNet (fp32, 8 scenes), grouped by (op, input shapes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssdnerf_amd  # noqa
from ssdnerf_amd.registry import MODULES
net = MODULES.build(dict(type="DenoisingUnetMod", image_size=128, in_channels=18, base_channels=128, channels_cfg=[1, 2, 2, 4, 4], resblocks_per_downsample=2,
                         dropout=0.0, use_scale_shift_norm=True, downsample_conv=True, upsample_conv=True, num_heads=4, attention_res=[32, 16, 8])).cuda().eval()
net.requires_grad_(False)
x = torch.randn(8, 18, 128, 128, device="cuda").requires_grad_(True)
t = torch.full((8,), 500, device="cuda", dtype=torch.long)
def step():
    y = net(x, t)
    (g,) = torch.autograd.grad(y.square().mean(), x)
    return g
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=48, max_shapes_column_width=70))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=40, max_src_column_width=110))
