"""tests/golden/unet_fill.py -- deterministic weights for the UNet fixtures (test infrastructure).

The UNet fixtures (unet_cars.npz / unet_tiled.npz, written by make_golden_unet.py from the reference's own ``DenoisingUnetMod``) do not store
the ~30 M parameters of the network: both the generator and the tests fill a state-dict with this function, key by key in sorted order
from one seeded CPU generator, and the fixture records the key list, the shapes and a checksum of the result.  Nothing is left at its
zero initialisation (mmgen zeroes conv_2 / proj / out, which would hide wiring errors)."""
import math

import torch

UNET_CONFIGS = dict(
    # configs/paper_cfgs/ssdnerf_cars_uncond.py:15-27 at half width / half resolution (attention at the same three scales 4, 8, 16;
    # channel counts stay multiples of 64 so that the hand-written kernels are the ones that run on the GPU)
    cars=dict(kwargs=dict(image_size=64, in_channels=18, base_channels=64, channels_cfg=[1, 2, 2, 4, 4], resblocks_per_downsample=2, dropout=0.0,
                          use_scale_shift_norm=True, downsample_conv=True, upsample_conv=True, num_heads=4, attention_res=[16, 8, 4]),
              x_shape=(2, 18, 64, 64), timesteps=(999, 339), seed=2021),
    # configs/new_cfgs/ssdnerf_cars_recons1v_tiled.py:16-28 reduced: non-square (6, H, 3H) input, GroupNorm(16), widths that are multiples
    # of 16 but not of 64 (the real config's base is 80), six levels
    tiled=dict(kwargs=dict(image_size=32, in_channels=6, base_channels=48, channels_cfg=[1, 1, 2, 2, 4, 4], resblocks_per_downsample=2, dropout=0.0,
                           use_scale_shift_norm=True, downsample_conv=True, upsample_conv=True, num_heads=4, attention_res=[4, 2, 1],
                           norm_cfg=dict(type="GN", num_groups=16)),
               x_shape=(2, 6, 32, 96), timesteps=(19, 640), seed=2022),
)


def fill_state_dict(sd, seed):
    """Overwrite every floating tensor of ``sd`` in place (sorted key order); returns the float64 checksum sum(|p|)."""
    g = torch.Generator().manual_seed(seed)
    total = 0.0
    with torch.no_grad():
        for key in sorted(sd.keys()):
            p = sd[key]
            if not torch.is_floating_point(p):
                continue
            r = torch.randn(p.shape, generator=g, dtype=torch.float32)
            if p.dim() >= 2:
                fan_in = p[0].numel()
                r = r * (1.0 / math.sqrt(fan_in))
            elif key.endswith("weight"):                       # GroupNorm scale
                r = 1.0 + 0.1 * r
            else:
                r = 0.1 * r
            p.copy_(r.to(p.dtype))
            total += float(r.double().abs().sum())
    return total


def make_input(x_shape, seed):
    g = torch.Generator().manual_seed(seed + 1000)
    return torch.randn(x_shape, generator=g)
