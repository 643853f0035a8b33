"""oracle/guidance.py -- TEST INFRASTRUCTURE.  The rendering-loss guidance closure on the CPU with autograd:
``BaseNeRF.loss`` through the TRAIN branch of the renderer (reference: lib/models/autodecoders/base_nerf.py:276-296,
lib/models/decoders/base_volume_renderer.py:59-77, lib/models/autodecoders/diffusion_nerf.py:282-294).

march_rays_train (C oracle, no gradient) -> point_decode (PyTorch-CPU, autograd) -> composite_rays_train (C oracle forward
and the reference's analytic backward wrapped in a torch.autograd.Function) -> MSE * 3 * scale + RegLoss."""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import ops as _ops
from .decoder import gather_point_code, sh_encode


class _CompositeTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh):
        o = _ops()
        ws, depth, image = o.composite_rays_train_forward(sigmas.detach().numpy(), rgbs.detach().numpy(), deltas, rays, T_thresh)
        ctx.save_for_backward(sigmas.detach(), rgbs.detach(), torch.from_numpy(ws), torch.from_numpy(image))
        ctx.aux = (deltas, rays, T_thresh)
        return torch.from_numpy(ws), torch.from_numpy(depth), torch.from_numpy(image)

    @staticmethod
    def backward(ctx, g_ws, g_depth, g_image):
        sigmas, rgbs, ws, image = ctx.saved_tensors
        deltas, rays, T_thresh = ctx.aux
        gs, gc = _ops().composite_rays_train_backward(g_ws.contiguous().numpy(), g_image.contiguous().numpy(), sigmas.numpy(), rgbs.numpy(),
                                                      deltas, rays, ws.numpy(), image.numpy(), T_thresh)
        return torch.from_numpy(gs), torch.from_numpy(gc), None, None, None


def decode_autograd(params: Dict[str, torch.Tensor], code: torch.Tensor, xyzs: torch.Tensor, dirs: torch.Tensor, sat: float = 0.001):
    f = gather_point_code(code, xyzs)
    base_x = F.linear(f, params["base_net.0.weight"], params["base_net.0.bias"])
    sigma = torch.exp(F.linear(F.silu(base_x), params["density_net.0.weight"], params["density_net.0.bias"]).squeeze(-1))
    color_in = F.silu(base_x + F.linear(sh_encode(dirs), params["dir_net.0.weight"], params["dir_net.0.bias"]))
    rgb = torch.sigmoid(F.linear(color_in, params["color_net.0.weight"], params["color_net.0.bias"])) * (1 + 2 * sat) - sat
    return sigma, rgb


def guidance_loss(params, code: torch.Tensor, bitfield: np.ndarray, rays_o: np.ndarray, rays_d: np.ndarray, target_rgbs: torch.Tensor,
                  noises: np.ndarray, dt_gamma: float, loss_weight: float = 20.0, loss_coef: float = 0.1 / (128 * 128), reg_weight: float = 3e-3,
                  reg_power: int = 2, bg_color: float = 1.0, grid_size: int = 64, max_steps: int = 256, T_thresh: float = 1e-4):
    """ONE scene; ``code`` (3,6,h,w) requires grad.  Returns loss (scalar tensor) and the integer march record."""
    o = _ops()
    aabb = np.array([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = o.near_far_from_aabb(rays_o, rays_d, aabb, 0.2)
    xyzs, dirs, deltas, rays, counter = o.march_rays_train(rays_o, rays_d, bitfield, 1.0, dt_gamma, max_steps, 1, grid_size, nears, fars, noises)
    m = int(counter[0])
    m_pad = m + 128 - m % 128
    xyzs, dirs, deltas = xyzs[:m_pad], dirs[:m_pad], deltas[:m_pad]
    sigma, rgb = decode_autograd(params, code, torch.from_numpy(xyzs), torch.from_numpy(dirs))
    ws, depth, image = _CompositeTrain.apply(sigma, rgb, deltas, rays, T_thresh)
    out_rgbs = image + bg_color * (1 - ws.unsqueeze(-1))
    n = rays_o.shape[0]
    scale = 1 - math.exp(-loss_coef * n)
    pixel = ((out_rgbs - target_rgbs) ** 2).mean() * loss_weight * (scale * 3)
    reg = (code.abs() ** reg_power).mean() * reg_weight
    return pixel + reg, dict(rays=rays, num_points=m, out_rgbs=out_rgbs.detach())
