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
from .diffusion import prior_loss_v


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
                  reg_power: int = 2, bg_color: float = 1.0, grid_size: int = 64, max_steps: int = 256, T_thresh: float = 1e-4,
                  scale_num_ray: int = None):
    """ONE scene; ``code`` (3,6,h,w) requires grad.  Returns loss (scalar tensor) and the integer march record.
    ``scale_num_ray``: the N of ``1 - exp(-loss_coef N)``; the number of rays by default (what guidance passes), all pixels of the
    conditioning views in ``inverse_code`` (base_nerf.py:446-449)."""
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
    n = rays_o.shape[0] if scale_num_ray is None else scale_num_ray
    scale = 1 - math.exp(-loss_coef * n)
    pixel = ((out_rgbs - target_rgbs) ** 2).mean() * loss_weight * (scale * 3)
    reg = (code.abs() ** reg_power).mean() * reg_weight
    return pixel + reg, dict(rays=rays, num_points=m, out_rgbs=out_rgbs.detach())


def finetune_code(params, denoise, code_: torch.Tensor, activation, tables, weight, density_grid: np.ndarray, rays_o: np.ndarray,
                  rays_d: np.ndarray, target_rgbs: torch.Tensor, dt_gamma: float, prior_timesteps, prior_noises, march_noises, density_jitters,
                  n_outer: int, n_inner: int, optimizer: dict, lr_gamma: float = None, weight_scale: float = 1.0, norm_factor: float = 1.0,
                  density_thresh: float = 0.1, update_extra_interval: int = 16, loss_kwargs: dict = None):
    """``DiffusionNeRF.val_optim`` for ONE scene with every random draw injected (diffusion_nerf.py:313-404 driving
    base_nerf.py:403-492): per outer step the diffusion-prior gradient of the pre-activation code seeds ``n_inner`` =
    ``extra_scene_step + 1`` rendering-loss iterations (grid refresh with decay 0.9 on the first of every ``update_extra_interval``),
    all sharing one optimizer and one exponential LR schedule.

    ``code_`` (3,6,h,w) pre-activation leaf, updated in place; ``activation(code_) -> code``; ``denoise(x_t (1,18,h,w), t) -> v``;
    ``density_grid`` (H^3,) float32 Morton grid, updated in place.  Returns (activated code, bitfield, list of rendering losses)."""
    from .render import update_extra_state
    code_.requires_grad_(True)
    opt_cfg = dict(optimizer)
    opt = getattr(torch.optim, opt_cfg.pop("type"))([code_], **opt_cfg)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=lr_gamma) if lr_gamma is not None else None
    march_noises, density_jitters = iter(march_noises), iter(density_jitters)
    bitfield, losses = None, []
    for k in range(n_outer):
        opt.zero_grad()
        x_0 = activation(code_).reshape(1, -1, *code_.shape[-2:])
        prior = prior_loss_v(denoise, x_0, torch.as_tensor(prior_timesteps[k]).long().reshape(-1), prior_noises[k], tables, weight, weight_scale,
                             norm_factor)
        prior.backward()
        prior_grad = code_.grad.detach().clone()
        for i in range(n_inner):
            code = activation(code_)
            if i % update_extra_interval == 0:
                bitfield, _ = update_extra_state(params, code.detach(), density_grid, next(density_jitters), density_thresh=density_thresh,
                                                 decay=0.9)
            loss, _ = guidance_loss(params, code, bitfield, rays_o, rays_d, target_rgbs, next(march_noises), dt_gamma,
                                    scale_num_ray=rays_o.shape[0], **(loss_kwargs or {}))
            code_.grad.copy_(prior_grad)
            loss.backward()
            opt.step()
            if sched is not None:
                sched.step()
            losses.append(float(loss.detach()))
    return activation(code_).detach(), bitfield, losses
