"""oracle/diffusion.py -- TEST INFRASTRUCTURE.  Independent restatement of the DDIM loop and of the UNet forward.

* ``schedule_tables`` / ``ddim_timesteps`` / ``ddim_sample`` follow the reference's
  lib/models/diffusions/gaussian_diffusion.py:64-154, :180-240, :264-331 (V-prediction, eta = 0, optional guidance),
  written as straight-line numpy/torch-CPU code with no module machinery.
* ``unet_forward`` is a FUNCTIONAL statement of ``DenoisingUnetMod.forward`` (lib/models/architecture/ddpm/denoising.py:
  191-216) over a plain state-dict, following SURVEY.md Appendix A for the mmgen block semantics.  **Parity unpinned**:
  mmgen 0.7.2 is neither vendored nor installable, so this restatement and ``ssdnerf_amd/unet.py`` are two independent
  readings of the same written spec, not a check against mmgen's code.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ schedule
def schedule_tables(num_timesteps: int = 1000, kind: str = "linear", beta_0: float = 1e-4, beta_T: float = 2e-2) -> Dict[str, np.ndarray]:
    if kind == "linear":
        scale = 1000 / num_timesteps
        betas = np.linspace(scale * beta_0, scale * beta_T, num_timesteps, dtype=np.float64)
    elif kind == "cosine":
        f = lambda t: np.cos((t / num_timesteps + 0.008) / 1.008 * np.pi / 2) ** 2
        betas = np.array([min(1 - f(t + 1) / f(t), 0.999) for t in range(num_timesteps)])
    else:
        raise ValueError(kind)
    ab = np.cumprod(1.0 - betas)
    ab_prev = np.append(1.0, ab[:-1])
    return dict(betas=betas, alphas_bar=ab, alphas_bar_prev=ab_prev, sqrt_alphas_bar=np.sqrt(ab), sqrt_one_minus_alphas_bar=np.sqrt(1 - ab),
                tilde_betas_t=betas * (1 - ab_prev) / (1 - ab))


def ddim_timesteps(num_timesteps: int, n: int) -> List[int]:
    return [int(v) for v in torch.arange(start=num_timesteps - 1, end=-1, step=-(num_timesteps / n)).long().tolist()]


def ddim_sample(denoise: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], noise: torch.Tensor, tables: Dict[str, np.ndarray],
                n_steps: int, clip_range=(-2.0, 2.0), grad_guide_fn: Optional[Callable] = None, guidance_gain: float = 1.0,
                snr_weight_power: float = 0.5, return_all: bool = False, langevin_steps: int = 0, langevin_delta: float = 0.1,
                langevin_t_range=(0, 1000), langevin_noises=None):
    """V-prediction DDIM, eta = 0.  ``denoise(x_t, t_batch) -> v``.  With ``grad_guide_fn`` the x0 prediction is corrected by
    grad_{x_t} L * sigma_t^(2-2w) * alpha_t^(2w-1) * gain exactly as gaussian_diffusion.py:193-227.  ``langevin_steps`` correction
    steps at the noise level just reached follow every DDIM step whose t_prev lies strictly inside ``langevin_t_range``
    (gaussian_diffusion.py:242-262, 318-323): x <- x - delta/2 * sigma * eps_pred(x, t_prev) + sqrt(delta) * sigma * z, with the same
    guided x0 prediction; ``langevin_noises`` is an iterable of the z draws."""
    T = len(tables["betas"])
    ts = ddim_timesteps(T, n_steps)
    x_t = noise.clone()
    trace = []
    zs = iter(langevin_noises) if langevin_noises is not None else None

    def predict_x0(x, t):
        a = torch.tensor(tables["sqrt_alphas_bar"], dtype=torch.float32)[t]
        b = torch.tensor(tables["sqrt_one_minus_alphas_bar"], dtype=torch.float32)[t]
        tb = torch.full((x.size(0),), t, dtype=torch.long)
        if grad_guide_fn is not None:
            x_in = x.detach().requires_grad_(True)
            with torch.enable_grad():
                v = denoise(x_in, tb)
                x0 = (a * x_in - b * v).clamp(*clip_range)
                loss = grad_guide_fn(x0)
                grad = torch.autograd.grad(loss, x_in)[0]
            x0 = x0.detach() - grad * ((b ** (2 - snr_weight_power * 2)) * (a ** (snr_weight_power * 2 - 1)) * guidance_gain)
        else:
            with torch.no_grad():
                v = denoise(x, tb)
            x0 = a * x - b * v
        return x0.clamp(*clip_range)

    for i, t in enumerate(ts):
        t_prev = ts[i + 1] if i + 1 < len(ts) else -1
        x0 = predict_x0(x_t, t)
        ab_prev = tables["alphas_bar"][t_prev] if t_prev >= 0 else tables["alphas_bar_prev"][0]
        eps = (x_t - tables["sqrt_alphas_bar"][t] * x0) / tables["sqrt_one_minus_alphas_bar"][t]
        x_t = (np.sqrt(ab_prev) * x0 + np.sqrt(1 - ab_prev) * eps).float()
        if langevin_steps > 0 and langevin_t_range[0] < t_prev < langevin_t_range[1]:
            sigma = tables["sqrt_one_minus_alphas_bar"][t_prev]
            for _ in range(langevin_steps):
                x0l = predict_x0(x_t, t_prev)
                eps_l = (x_t - tables["sqrt_alphas_bar"][t_prev] * x0l) / sigma
                x_t = (x_t - 0.5 * langevin_delta * sigma * eps_l + math.sqrt(langevin_delta) * sigma * next(zs)).float()
        trace.append((x0, x_t))
    return (x_t, trace) if return_all else x_t


# ------------------------------------------------------------------------------------------------ diffusion prior loss
def snr_timestep_weights(tables: Dict[str, np.ndarray], power: float = 0.5, mode: str = "V", bias: float = 0.0, prob_power: float = 0.0):
    """``SNRWeightedTimeStepSampler`` (lib/models/diffusions/sampler.py:14-44): per-timestep loss weight (float32) and sampling
    probabilities.  With the configs' power = 0.5, mode 'V', prob_power = 0 this is weight = sqrt(abar (1 - abar)), uniform sampling."""
    mean, std = tables["sqrt_alphas_bar"], tables["sqrt_one_minus_alphas_bar"]
    wx = (mean / std) ** (2 * power) + bias
    raw = {"EPS": wx * (std / mean) ** 2, "START_X": wx, "V": wx * std ** 2}[mode]
    prob = raw ** prob_power
    prob = prob / prob.sum()
    return (raw / (prob * len(mean))).astype(np.float32), prob


def prior_loss_v(denoise: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], x_0: torch.Tensor, t: torch.Tensor, noise: torch.Tensor,
                 tables: Dict[str, np.ndarray], weight: Optional[np.ndarray], weight_scale: float = 1.0, norm_factor: float = 1.0) -> torch.Tensor:
    """The V-prediction training loss ``val_optim`` back-propagates into the code (gaussian_diffusion.py:165-178, 389-433 with
    lib/models/losses/ddpm_loss.py:12-142): x_t = a x_0 + b noise, target v = a noise - b x_0,
    loss = mean_batch(0.5 mean_chw((v_pred - v)^2) w[t] c) / norm_factor.  ``t`` (B,) long."""
    a = torch.from_numpy(tables["sqrt_alphas_bar"])[t].float().reshape(-1, 1, 1, 1)
    b = torch.from_numpy(tables["sqrt_one_minus_alphas_bar"])[t].float().reshape(-1, 1, 1, 1)
    x_t = x_0 * a + noise * b
    v_pred = denoise(x_t, t)
    v = a * noise - b * x_0
    per = ((v_pred - v) ** 2).flatten(1).mean(dim=1) * 0.5
    if weight is not None:
        per = per * torch.from_numpy(weight)[t] * weight_scale
    return per.mean() / norm_factor


# ------------------------------------------------------------------------------------------------ UNet, functional
def _gn(x, sd, p, groups=32):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def _conv(x, sd, p, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def _time_embedding(sd, t, base):
    half = base // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    e = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    e = F.linear(e, sd["time_embedding.blocks.0.weight"], sd["time_embedding.blocks.0.bias"])
    return F.linear(F.silu(e), sd["time_embedding.blocks.2.weight"], sd["time_embedding.blocks.2.bias"])


def _resblock(x, emb, sd, p, groups):
    s = _conv(x, sd, p + ".shortcut", padding=0) if (p + ".shortcut.weight") in sd else x
    h = _conv(F.silu(_gn(x, sd, p + ".conv_1.0", groups)), sd, p + ".conv_1.2")
    e = F.linear(F.silu(emb), sd[p + ".norm_with_embedding.embedding_layer.1.weight"], sd[p + ".norm_with_embedding.embedding_layer.1.bias"])
    scale, shift = torch.chunk(e[:, :, None, None], 2, dim=1)
    h = _gn(h, sd, p + ".norm_with_embedding.norm", groups) * (1 + scale) + shift
    h = _conv(F.silu(h), sd, p + ".conv_2.1")
    return h + s


def _attention(x, sd, p, heads, groups):
    b, c, hh, ww = x.shape
    t = hh * ww
    xf = x.reshape(b, c, t)
    qkv = F.conv1d(_gn(xf, sd, p + ".norm", groups), sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])       # (b, 3c, t)
    qkv = qkv.reshape(b * heads, 3 * c // heads, t)                                                     # [head][q|k|v][c/heads]
    ch = c // heads
    q, k, v = qkv[:, :ch], qkv[:, ch:2 * ch], qkv[:, 2 * ch:]
    w = torch.softmax((q.transpose(1, 2) @ k) / math.sqrt(ch), dim=-1)                                  # == (q*s)(k*s), s = ch^-1/4
    h = (v @ w.transpose(1, 2)).reshape(b, c, t)
    h = F.conv1d(h, sd[p + ".proj.weight"], sd[p + ".proj.bias"])
    return (h + xf).reshape(b, c, hh, ww)


def unet_forward(sd: Dict[str, torch.Tensor], x_t: torch.Tensor, t: torch.Tensor, image_size: int = 128, base_channels: int = 128,
                 channels_cfg=(1, 2, 2, 4, 4), resblocks_per_downsample: int = 2, num_heads: int = 4, attention_res=(32, 16, 8),
                 num_timesteps: int = 1000, norm_groups: int = 32) -> torch.Tensor:
    """Walks the state-dict in the order the reference's constructor lays blocks out (denoising.py:106-187)."""
    emb = _time_embedding(sd, t.float() * (1000.0 / num_timesteps), base_channels)
    att_scales = [image_size // r for r in attention_res]
    h = _conv(x_t, sd, "in_blocks.0.0")
    hs = [h]
    scale, bi = 1, 1
    for level in range(len(channels_cfg)):
        for _ in range(resblocks_per_downsample):
            p = f"in_blocks.{bi}"
            h = _resblock(h, emb, sd, p + ".0", norm_groups)
            if scale in att_scales:
                h = _attention(h, sd, p + ".1", num_heads, norm_groups)
            hs.append(h); bi += 1
        if level != len(channels_cfg) - 1:
            h = _conv(h, sd, f"in_blocks.{bi}.0.downsample", stride=2)
            hs.append(h); bi += 1
            scale *= 2
    h = _resblock(h, emb, sd, "mid_blocks.0", norm_groups)
    h = _attention(h, sd, "mid_blocks.1", num_heads, norm_groups)
    h = _resblock(h, emb, sd, "mid_blocks.2", norm_groups)
    oi = 0
    for level in range(len(channels_cfg)):
        for idx in range(resblocks_per_downsample + 1):
            p = f"out_blocks.{oi}"
            h = _resblock(torch.cat([h, hs.pop()], dim=1), emb, sd, p + ".0", norm_groups)
            j = 1
            if scale in att_scales:
                h = _attention(h, sd, p + f".{j}", num_heads, norm_groups); j += 1
            if level != len(channels_cfg) - 1 and idx == resblocks_per_downsample:
                h = F.interpolate(h, scale_factor=2, mode="nearest")
                h = _conv(h, sd, p + f".{j}.conv")
                scale //= 2
            oi += 1
    return _conv(F.silu(_gn(h, sd, "out.gn", norm_groups)), sd, "out.conv")
