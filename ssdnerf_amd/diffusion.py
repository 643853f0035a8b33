"""``GaussianDiffusion``: noise schedules and the DDIM sampling loop over triplane latents, with the reference's
rendering-loss guidance hook (reference: lib/models/diffusions/gaussian_diffusion.py:14-464).

On the hot path: ``prepare_diffusion_vars`` (numpy float64 tables, :131-154),
``pred_x_0`` (:180-240), ``p_sample_ddim`` (:264-293), ``p_sample_langevin`` (:242-262), ``ddim_sample`` (:295-331).
Section 8(f) rank 1 (the fine-tuning half of ``cond_mode='guide_optim'``) adds the diffusion prior loss that
``val_optim`` back-propagates into the code: ``q_sample`` (:165-178), ``loss`` (:389-405), ``forward_train`` (:407-433)
with the timestep samplers (lib/models/diffusions/sampler.py) and ``DDPMMSELossMod`` (lib/models/losses/ddpm_loss.py).
The DDPM ancestral sampler (``p_sample_ddpm`` / ``ddpm_sample``, :333-385; ``sample_method='ddpm'``) is carried for API parity.

The per-step latent update (V-prediction -> x0, clamp, eps, x_prev) is one fused HIP kernel when no guidance closure is
active (``ssdnerf_ddim_step_v``); the guided path keeps the reference's exact PyTorch expression order because autograd
flows through it.
"""
from __future__ import annotations

import math
import sys
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn

from . import _cabi as C
from .registry import MODULES, build_module, get_module_device


def _noise_like(x):
    # mmgen's _get_noise_batch draws on the CPU and moves to the device
    return torch.randn(x.shape, dtype=torch.float32).to(x.device)


# ---------------------------------------------------------------------------------------------- timestep samplers
@MODULES.register_module()
class UniformTimeStepSampler:
    """mmgen ``UniformTimeStepSampler``: t ~ ``np.random.choice(T, p=prob)`` drawn on the HOST, so a seeded
    ``np.random`` gives the same timesteps on every device."""

    def __init__(self, num_timesteps, **kwargs):
        self.num_timesteps = num_timesteps
        self.prob = [1 / self.num_timesteps for _ in range(self.num_timesteps)]

    def sample(self, batch_size):
        return torch.from_numpy(np.random.choice(self.num_timesteps, size=(batch_size,), p=self.prob)).long()

    def __call__(self, batch_size):
        return self.sample(batch_size)


MODULES.register_module(name="UniformTimeStepSamplerMod", module=type("UniformTimeStepSamplerMod", (UniformTimeStepSampler,), {}))


@MODULES.register_module()
class SNRWeightedTimeStepSampler(UniformTimeStepSampler):
    """Per-timestep loss weight ``SNR^power`` expressed for the network's output parameterisation, and the sampling density
    ``weight^prob_power`` (lib/models/diffusions/sampler.py:14-44).  ``mean`` / ``std`` are the float64 sqrt(alpha_bar) tables."""

    def __init__(self, num_timesteps, mean, std, mode, power=1, min=-1, max=-1, bias=0, prob_power=0.0):
        self.num_timesteps = num_timesteps
        mean, std = np.asarray(mean, np.float64), np.asarray(std, np.float64)
        weight_x = (mean / std) ** (2 * power) + bias
        if min > 0:
            weight_x = weight_x.clip(min=min)
        if max > 0:
            weight_x = weight_x.clip(max=max)
        if mode == "EPS":
            weight_raw = weight_x * (std / mean) ** 2
        elif mode == "START_X":
            weight_raw = weight_x
        elif mode == "V":
            weight_raw = weight_x * (std ** 2)
        else:
            raise AttributeError(f"unknown denoising mean mode {mode!r}")
        prob = weight_raw ** prob_power
        prob /= prob.sum()
        self.weight = torch.from_numpy(weight_raw / (prob * self.num_timesteps)).to(torch.float)
        self.prob = prob.tolist()


# ---------------------------------------------------------------------------------------------- diffusion prior loss
@MODULES.register_module()
class DDPMMSELossMod(nn.Module):
    """``0.5 * mean_{chw}((pred - target)^2)`` per sample, times ``weight[t] * weight_scale`` (``rescale_mode='timestep_weight'``, the
    weight table coming from the timestep sampler), optionally divided by the running ``norm_factor`` = EMA of mean(x_0^2), then reduced
    over the batch (lib/models/losses/ddpm_loss.py:12-142 on top of mmgen's ``DDPMLoss``).

    ``log_vars`` holds the quartile means as 0-dim device tensors (the reference calls ``.item()`` on each, a sync per quartile)."""

    _default_data_info = dict(pred="eps_t_pred", target="noise")

    def __init__(self, rescale_mode=None, rescale_cfg=None, sampler=None, weight=None, weight_scale=1.0, log_cfgs=None, reduction="mean",
                 data_info=None, loss_name="loss_ddpm_mse", scale_norm=False, momentum=0.001):
        super().__init__()
        self.weight_scale = weight_scale
        self.reduction = reduction
        self.loss_name_ = loss_name
        self.data_info = dict(self._default_data_info if data_info is None else data_info)
        self.rescale_mode = rescale_mode
        self.timestep_weight = None
        if rescale_mode is not None:
            if rescale_mode != "timestep_weight":
                raise NotImplementedError(f"rescale_mode={rescale_mode!r}: the reference's configs only use 'timestep_weight'")
            if sampler is not None and hasattr(sampler, "weight"):
                weight = sampler.weight
            if weight is None:
                raise ValueError("rescale_mode='timestep_weight' needs a sampler with a weight table, or an explicit weight")
            self.timestep_weight = torch.as_tensor(weight, dtype=torch.float)
        self.log_cfgs = [log_cfgs] if isinstance(log_cfgs, dict) else list(log_cfgs or [])
        self.log_vars = dict()
        self.scale_norm = scale_norm
        self.freeze_norm = False
        if scale_norm:
            self.register_buffer("norm_factor", torch.ones(1, dtype=torch.float))
        self.momentum = momentum

    def _reduce(self, loss):
        if self.reduction == "mean":
            return loss.mean()
        if self.reduction == "sum":
            return loss.sum()
        if self.reduction == "none":
            return loss
        if self.reduction == "flatmean":
            return loss.flatten(1).mean(dim=1) if loss.dim() > 1 else loss
        raise ValueError(self.reduction)

    def _collect_log(self, loss, timesteps):
        self.log_vars = dict()
        for cfg in self.log_cfgs:
            if cfg.get("type") != "quartile":
                continue
            total, prefix = cfg.get("total_timesteps", 1000), cfg.get("prefix_name", "loss")
            quartile = (timesteps.float() / total * 4).long()
            ld = loss.detach()
            for q in range(4):
                m = (quartile == q).to(ld.dtype)
                self.log_vars[f"{prefix}_quartile_{q}"] = (ld * m).sum() / m.sum().clamp(min=1)

    def forward(self, output_dict):
        assert isinstance(output_dict, dict) and "timesteps" in output_dict, "DDPM losses take the dict of network outputs with 'timesteps'"
        timesteps = output_dict["timesteps"]
        pred, target = output_dict[self.data_info["pred"]], output_dict[self.data_info["target"]]
        loss = (pred - target).square().flatten(1).mean(dim=1) * 0.5
        if self.timestep_weight is not None:
            loss = loss * self.timestep_weight.to(timesteps.device)[timesteps] * self.weight_scale
        self._collect_log(loss, timesteps)
        loss = self._reduce(loss)
        if self.scale_norm:
            if self.training and not self.freeze_norm:
                from .parallel import reduce_mean
                norm_factor = reduce_mean(output_dict["x_0"].detach().square().mean())
                self.norm_factor[:] = (1 - self.momentum) * self.norm_factor + self.momentum * norm_factor
            loss = loss / self.norm_factor
        return loss


MODULES.register_module(name="DDPMMSELoss", module=type("DDPMMSELoss", (DDPMMSELossMod,), {}))


@MODULES.register_module()
class GaussianDiffusion(nn.Module):
    def __init__(self, denoising, ddpm_loss=dict(type="DDPMMSELoss", log_cfgs=dict(type="quartile", prefix_name="loss_mse", total_timesteps=1000)), betas_cfg=dict(type="cosine"), num_timesteps=1000, num_classes=0, sample_method="ddim",
                 timestep_sampler=dict(type="UniformTimeStepSampler"), denoising_var_mode="FIXED_LARGE", denoising_mean_mode="V", train_cfg=None, test_cfg=None):
        super().__init__()
        self.num_classes = num_classes
        self.num_timesteps = num_timesteps
        self.sample_method = sample_method
        self._denoising_cfg = deepcopy(denoising)
        self.denoising = build_module(denoising, default_args=dict(num_classes=num_classes, num_timesteps=num_timesteps))
        self.denoising_var_mode = denoising_var_mode
        self.denoising_mean_mode = denoising_mean_mode
        self.betas_cfg = deepcopy(betas_cfg)
        self.train_cfg = deepcopy(train_cfg) if train_cfg is not None else dict()
        self.test_cfg = deepcopy(test_cfg) if test_cfg is not None else dict()
        self.prepare_diffusion_vars()
        # timestep sampler + prior loss (:55-62): what ``forward_train`` / ``val_optim`` use
        self.sampler = build_module(timestep_sampler or dict(type="UniformTimeStepSampler"),
                                    default_args=dict(num_timesteps=num_timesteps, mean=self.sqrt_alphas_bar, std=self.sqrt_one_minus_alphas_bar,
                                                      mode=self.denoising_mean_mode))
        self.ddpm_loss = build_module(ddpm_loss or dict(type="DDPMMSELoss"), default_args=dict(sampler=self.sampler))
        self.use_fused_step = True

    # ------------------------------------------------------------------------------------------ schedules
    @staticmethod
    def linear_beta_schedule(diffusion_timesteps, beta_0=1e-4, beta_T=2e-2):
        scale = 1000 / diffusion_timesteps
        return np.linspace(scale * beta_0, scale * beta_T, diffusion_timesteps, dtype=np.float64)

    @staticmethod
    def cosine_beta_schedule(diffusion_timesteps, max_beta=0.999, s=0.008):
        def f(t, T, s):
            return np.cos((t / T + s) / (1 + s) * np.pi / 2) ** 2
        betas = []
        for t in range(diffusion_timesteps):
            betas.append(min(1 - f(t + 1, diffusion_timesteps, s) / f(t, diffusion_timesteps, s), max_beta))
        return np.array(betas)

    def get_betas(self):
        cfg = dict(self.betas_cfg)
        self.betas_schedule = cfg.pop("type")
        if self.betas_schedule == "linear":
            return self.linear_beta_schedule(self.num_timesteps, **cfg)
        if self.betas_schedule == "cosine":
            return self.cosine_beta_schedule(self.num_timesteps, **cfg)
        if self.betas_schedule == "scaled_linear":
            return np.linspace(cfg.get("beta_start", 0.0001) ** 0.5, cfg.get("beta_end", 0.02) ** 0.5, self.num_timesteps, dtype=np.float64) ** 2
        raise AttributeError(f"Unknown method name {self.betas_schedule} for beta schedule.")

    def prepare_diffusion_vars(self):
        self.betas = self.get_betas()
        self.alphas = 1.0 - self.betas
        self.alphas_bar = np.cumprod(self.alphas, axis=0)
        self.alphas_bar_prev = np.append(1.0, self.alphas_bar[:-1])
        self.alphas_bar_next = np.append(self.alphas_bar[1:], 0.0)
        self.sqrt_alphas_bar = np.sqrt(self.alphas_bar)
        self.sqrt_one_minus_alphas_bar = np.sqrt(1.0 - self.alphas_bar)
        self.log_one_minus_alphas_bar = np.log(1.0 - self.alphas_bar)
        self.sqrt_recip_alplas_bar = np.sqrt(1.0 / self.alphas_bar)
        self.sqrt_recipm1_alphas_bar = np.sqrt(1.0 / self.alphas_bar - 1)
        self.tilde_betas_t = self.betas * (1 - self.alphas_bar_prev) / (1 - self.alphas_bar)
        self.log_tilde_betas_t_clipped = np.log(np.append(self.tilde_betas_t[1], self.tilde_betas_t[1:]))
        self.tilde_mu_t_coef1 = np.sqrt(self.alphas_bar_prev) / (1 - self.alphas_bar) * self.betas
        self.tilde_mu_t_coef2 = np.sqrt(self.alphas) * (1 - self.alphas_bar_prev) / (1 - self.alphas_bar)

    # ------------------------------------------------------------------------------------------ forward process
    def q_sample(self, x_0, t, noise=None):
        """x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) noise; also returns the two broadcastable coefficients (:165-178)."""
        if noise is None:
            noise = _noise_like(x_0)
        t_host = torch.as_tensor(t).cpu()
        mean = x_0.new_tensor(self.sqrt_alphas_bar[t_host.numpy()], dtype=torch.float32).reshape(-1, 1, 1, 1)
        std = x_0.new_tensor(self.sqrt_one_minus_alphas_bar[t_host.numpy()], dtype=torch.float32).reshape(-1, 1, 1, 1)
        return x_0 * mean + noise * std, mean, std

    # ------------------------------------------------------------------------------------------ x0 prediction
    def pred_x_0(self, x_t, t, grad_guide_fn=None, concat_cond=None, cfg=dict(), update_denoising_output=False):
        clip_denoised = cfg.get("clip_denoised", True)
        clip_range = cfg.get("clip_range", [-1, 1])
        guidance_gain = cfg.get("guidance_gain", 1.0)
        grad_through_unet = cfg.get("grad_through_unet", True)
        snr_weight_power = cfg.get("snr_weight_power", 0.5)

        num_batches = x_t.size(0)
        t = torch.as_tensor(t).to(x_t.device)
        if t.dim() == 0 or len(t) != num_batches:
            t = t.expand(num_batches)
        sqrt_alpha_bar_t = x_t.new_tensor(self.sqrt_alphas_bar)[t].reshape(-1, 1, 1, 1)
        sqrt_one_minus_alpha_bar_t = x_t.new_tensor(self.sqrt_one_minus_alphas_bar)[t].reshape(-1, 1, 1, 1)

        grad_enabled_prev = torch.is_grad_enabled()
        if grad_guide_fn is not None and grad_through_unet:
            x_t = x_t.detach().requires_grad_(True)    # the reference flips the flag on the (leaf) latent in place (:193-196)
            torch.set_grad_enabled(True)

        denoising_output = self.denoising(x_t, t, concat_cond=concat_cond)
        mode = self.denoising_mean_mode.upper()
        if mode == "EPS":
            x_0_pred = (x_t - sqrt_one_minus_alpha_bar_t * denoising_output) / sqrt_alpha_bar_t
        elif mode == "START_X":
            x_0_pred = denoising_output
        elif mode == "V":
            x_0_pred = sqrt_alpha_bar_t * x_t - sqrt_one_minus_alpha_bar_t * denoising_output
        else:
            raise AttributeError(f"Unknown denoising mean output type [{self.denoising_mean_mode}].")

        if grad_guide_fn is not None:
            if clip_denoised:
                x_0_pred = x_0_pred.clamp(*clip_range)
            if grad_through_unet:
                loss = grad_guide_fn(x_0_pred)
                grad = torch.autograd.grad(loss, x_t)[0]
            else:
                x_0_pred.requires_grad = True
                torch.set_grad_enabled(True)
                loss = grad_guide_fn(x_0_pred)
                grad = torch.autograd.grad(loss, x_0_pred)[0]
            torch.set_grad_enabled(grad_enabled_prev)
            x_0_pred.detach_()
            x_0_pred -= grad * ((sqrt_one_minus_alpha_bar_t ** (2 - snr_weight_power * 2))
                                * (sqrt_alpha_bar_t ** (snr_weight_power * 2 - 1)) * guidance_gain)
        if clip_denoised:
            x_0_pred = x_0_pred.clamp(*clip_range)

        if update_denoising_output and grad_guide_fn is not None:
            if mode == "EPS":
                denoising_output = (x_t - x_0_pred * sqrt_alpha_bar_t) / sqrt_one_minus_alpha_bar_t
            elif mode == "START_X":
                denoising_output = x_0_pred
            elif mode == "V":
                denoising_output = (sqrt_alpha_bar_t * x_t - x_0_pred) / sqrt_one_minus_alpha_bar_t
        return x_0_pred, denoising_output

    # ------------------------------------------------------------------------------------------ samplers
    def p_sample_langevin(self, x_t, t, noise=None, cfg=dict(), grad_guide_fn=None, **kwargs):
        langevin_delta = cfg.get("langevin_delta", 0.1)
        sigma = self.sqrt_one_minus_alphas_bar[int(t)]
        x_0_pred, _ = self.pred_x_0(x_t, t, grad_guide_fn=grad_guide_fn, cfg=cfg, **kwargs)
        eps_t_pred = (x_t - self.sqrt_alphas_bar[int(t)] * x_0_pred) / sigma
        if noise is None:
            noise = _noise_like(x_t)
        return x_t - 0.5 * langevin_delta * sigma * eps_t_pred + math.sqrt(langevin_delta) * sigma * noise

    def _fused_v_step_ok(self, x_t, cfg, grad_guide_fn, eta):
        return (self.use_fused_step and grad_guide_fn is None and eta == 0 and self.denoising_mean_mode.upper() == "V"
                and x_t.is_cuda and x_t.dtype == torch.float32 and cfg.get("clip_denoised", True))

    def p_sample_ddim(self, x_t, t, t_prev, noise=None, cfg=dict(), grad_guide_fn=None, **kwargs):
        eta = cfg.get("eta", 0)
        t_i, tp_i = int(t), int(t_prev)
        alpha_bar_t_prev = self.alphas_bar[tp_i] if tp_i >= 0 else self.alphas_bar_prev[0]
        tilde_beta_t = self.tilde_betas_t[t_i]

        if self._fused_v_step_ok(x_t, cfg, grad_guide_fn, eta):
            # unguided V-prediction step: UNet forward, then ONE fused elementwise launch for
            # x0 = clamp(a*x_t - b*v), eps = (x_t - a*x0)/b, x_prev = c*x0 + d*eps      (:213, :235, :281-283)
            with torch.no_grad():
                tt = torch.full((x_t.size(0),), t_i, dtype=torch.long, device=x_t.device)
                v = self.denoising(x_t, tt, concat_cond=kwargs.get("concat_cond"))
            clip_range = cfg.get("clip_range", [-1, 1])
            x_t = x_t.contiguous()
            v = v.float().contiguous()
            x_prev, x_0_pred = torch.empty_like(x_t), torch.empty_like(x_t)
            C.check(C.lib().ssdnerf_ddim_step_v(C.ptr(x_t), C.ptr(v), C.ctypes.c_uint64(x_t.numel()), C.f32(self.sqrt_alphas_bar[t_i]),
                                                C.f32(self.sqrt_one_minus_alphas_bar[t_i]), C.f32(np.sqrt(alpha_bar_t_prev)),
                                                C.f32(np.sqrt(1 - alpha_bar_t_prev - tilde_beta_t * (eta ** 2))), C.f32(clip_range[0]),
                                                C.f32(clip_range[1]), C.ptr(x_0_pred), C.ptr(x_prev), C.stream()),
                    "ddim_step_v")
            return x_prev, x_0_pred

        x_0_pred, _ = self.pred_x_0(x_t, t, grad_guide_fn=grad_guide_fn, cfg=cfg, **kwargs)
        eps_t_pred = (x_t - self.sqrt_alphas_bar[t_i] * x_0_pred) / self.sqrt_one_minus_alphas_bar[t_i]
        pred_sample_direction = np.sqrt(1 - alpha_bar_t_prev - tilde_beta_t * (eta ** 2)) * eps_t_pred
        x_prev = np.sqrt(alpha_bar_t_prev) * x_0_pred + pred_sample_direction
        if eta > 0:
            if noise is None:
                noise = _noise_like(x_t)
            x_prev = x_prev + eta * np.sqrt(tilde_beta_t) * noise
        return x_prev, x_0_pred

    def ddim_timesteps(self, num_timesteps=None):
        """arange(T-1, -1, -T/n).long(): 50 -> 999, 979, ..., 19; 75 -> 999, 985, 972, ..., 12   (:300-302)."""
        n = self.test_cfg.get("num_timesteps", self.num_timesteps) if num_timesteps is None else num_timesteps
        return torch.arange(start=self.num_timesteps - 1, end=-1, step=-(self.num_timesteps / n)).long()

    def ddim_sample(self, noise, show_pbar=False, concat_cond=None, save_intermediates=False, **kwargs):
        device = noise.device
        x_t = noise
        langevin_steps = self.test_cfg.get("langevin_steps", 0)
        langevin_t_range = self.test_cfg.get("langevin_t_range", [0, 1000])
        # timesteps stay on the HOST: the reference moves them to the device and then indexes numpy tables with them,
        # which costs a device->host sync per table lookup (:275-283); the UNet gets a device copy inside pred_x_0.
        timesteps = self.ddim_timesteps()
        cond_step = 0
        x_0_x_t_list = [] if save_intermediates else None
        for step, t in enumerate(timesteps):
            t_prev = timesteps[step + 1] if step + 1 < len(timesteps) else torch.tensor(-1)
            tp_host = int(t_prev)
            x_t, x_0_pred = self.p_sample_ddim(
                x_t, t, t_prev,
                concat_cond=concat_cond[:, cond_step % concat_cond.size(1)] if concat_cond is not None else None,
                cfg=self.test_cfg, **kwargs)
            cond_step += 1
            if langevin_steps > 0 and langevin_t_range[0] < tp_host < langevin_t_range[1]:
                for _ in range(langevin_steps):
                    x_t = self.p_sample_langevin(
                        x_t, t_prev, concat_cond=concat_cond[:, cond_step % concat_cond.size(1)] if concat_cond is not None else None,
                        cfg=self.test_cfg, **kwargs)
                    cond_step += 1
            if x_0_x_t_list is not None:
                x_0_x_t_list.append(x_0_pred)
                x_0_x_t_list.append(x_t)
        return x_0_x_t_list if save_intermediates else x_t

    # ------------------------------------------------------------------------------------------ ancestral sampler
    def q_posterior_mean(self, x_0, x_t, t):
        """mean of q(x_{t-1} | x_t, x_0) (:154-163)"""
        t_host = torch.as_tensor(t).cpu().reshape(-1).numpy()
        c1 = x_0.new_tensor(self.tilde_mu_t_coef1[t_host], dtype=torch.float32).reshape(-1, 1, 1, 1)
        c2 = x_0.new_tensor(self.tilde_mu_t_coef2[t_host], dtype=torch.float32).reshape(-1, 1, 1, 1)
        return c1 * x_0 + c2 * x_t

    def p_sample_ddpm(self, x_t, t, noise=None, cfg=dict(), grad_guide_fn=None, **kwargs):
        """One ancestral step with the fixed-large / fixed-small variance (:333-365)."""
        t_host = torch.as_tensor(t).cpu().reshape(-1).numpy()
        mode = self.denoising_var_mode.upper()
        if mode == "FIXED_LARGE":
            table = np.append(self.tilde_betas_t[1], self.betas)
        elif mode == "FIXED_SMALL":
            table = self.tilde_betas_t
        else:
            raise AttributeError(f"Unknown denoising var output type [{self.denoising_var_mode}].")
        var_pred = x_t.new_tensor(table[t_host], dtype=torch.float32).reshape(-1, 1, 1, 1)
        x_0_pred, _ = self.pred_x_0(x_t, t, grad_guide_fn=grad_guide_fn, cfg=cfg, **kwargs)
        mean_pred = self.q_posterior_mean(x_0_pred, x_t, t)
        if noise is None:
            noise = _noise_like(x_t)
        nonzero = float(int(t_host[0]) != 0) if t_host.size == 1 else x_t.new_tensor((t_host != 0).astype(np.float32)).reshape(-1, 1, 1, 1)
        return mean_pred + nonzero * torch.sqrt(var_pred) * noise, x_0_pred

    def ddpm_sample(self, noise, show_pbar=False, concat_cond=None, **kwargs):
        x_t = noise
        cond_step = 0
        for t in self.ddim_timesteps():
            x_t, _ = self.p_sample_ddpm(x_t, t, concat_cond=concat_cond[:, cond_step % concat_cond.size(1)] if concat_cond is not None else None,
                                        cfg=self.test_cfg, **kwargs)
            cond_step += 1
        return x_t

    def sample_from_noise(self, noise, **kwargs):
        name = f"{self.sample_method.lower()}_sample"
        if not hasattr(self, name):
            raise AttributeError(f"Cannot find sample method [{name}] correspond to [{self.sample_method}].")
        return getattr(self, name)(noise=noise, **kwargs)

    # ------------------------------------------------------------------------------------------ prior loss
    def loss(self, denoising_output, x_0, noise, t, mean, std):
        mode = self.denoising_mean_mode.upper()
        if mode == "EPS":
            loss_kwargs = dict(eps_t_pred=denoising_output)
        elif mode == "START_X":
            loss_kwargs = dict(x_0_pred=denoising_output)
        elif mode == "V":
            loss_kwargs = dict(v_t_pred=denoising_output)
        else:
            raise AttributeError(f"Unknown denoising mean output type [{self.denoising_mean_mode}].")
        loss_kwargs.update(x_0=x_0, noise=noise, timesteps=t)
        if "v_t_pred" in loss_kwargs:
            loss_kwargs.update(v_t=mean * noise - std * x_0)
        return self.ddpm_loss(loss_kwargs)

    def forward_train(self, x_0, concat_cond=None, grad_guide_fn=None, cfg=dict(), x_t_detach=False, timesteps=None, noise=None, **kwargs):
        """Diffusion prior loss of ``x_0`` (:407-433).  ``timesteps`` / ``noise`` (extra): injected draws; by default both are drawn on
        the host exactly like the reference (``np.random.choice`` / CPU ``torch.randn``), so seeding reproduces them on any device.
        ``log_vars['loss_ddpm_mse']`` is a detached 0-dim tensor instead of a Python float (no device sync here)."""
        assert x_0.dim() == 4
        device = x_0.device
        t = (self.sampler(x_0.size(0)) if timesteps is None else torch.as_tensor(timesteps).long()).to(device)
        if noise is None:
            noise = _noise_like(x_0)
        x_t, mean, std = self.q_sample(x_0, t, noise.to(device))
        if x_t_detach:
            x_t = x_t.detach()
        _, denoising_output = self.pred_x_0(x_t, t, grad_guide_fn=grad_guide_fn, concat_cond=concat_cond, cfg=cfg, update_denoising_output=True)
        loss = self.loss(denoising_output, x_0, noise.to(device), t, mean, std)
        log_vars = self.ddpm_loss.log_vars
        log_vars.update(loss_ddpm_mse=loss.detach())
        return loss, log_vars

    def forward_test(self, data, **kwargs):
        assert data.dim() == 4
        return self.sample_from_noise(data, **kwargs)

    def forward(self, data, return_loss=False, **kwargs):
        if return_loss:
            return self.forward_train(data, **kwargs)
        return self.forward_test(data, **kwargs)
