"""``GaussianDiffusion`` for the triplane latents: noise schedule, DDIM / Langevin / ancestral sampling with the rendering-loss
guidance hook, and the diffusion prior loss (behaviour of lib/models/diffusions/gaussian_diffusion.py:14-464, sampler.py:7-44,
lib/models/losses/ddpm_loss.py:11-142).  Constructor keywords, registry names and the public
methods are the reference's; the implementation is organised for the MI355X instead of following the reference's per-step host code:

* ``NoiseSchedule`` holds the float64 tables once; everything a sampling trajectory needs from them is resolved ON THE HOST, ONCE, into a
  ``SamplingPlan`` -- a flat list of network evaluations, each with its timestep and its scalar coefficients.  The loops below never index
  a numpy table with a device tensor (the reference does, which costs a device->host sync per lookup, gaussian_diffusion.py:275-283) and
  never upload a table per call.
* An unguided DDIM step is: write ``t`` into the UNet session's static buffer, replay the captured forward (``unet_fast.UnetSession``), ONE
  fused elementwise launch (``ssdnerf_ddim_step_v``) that turns (x_t, v) into (x0, x_prev) and writes x_prev back IN PLACE into the
  session's input buffer.  Three launches per step, no copies, no allocation, no host sync in the loop.
* The guided step (rendering guidance, SSDNeRF's ``grad_guide_fn``) needs autograd through the UNet and the renderer; it keeps tensors in
  the graph exactly where the reference does, with the coefficients coming from the plan as Python floats.
"""
from __future__ import annotations

import math
from copy import deepcopy
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _cabi as C
from .registry import MODULES, build_module


class _PinnedRing:
    """Two pinned host buffers per (device, shape) and an event behind each upload: a host draw lands in pinned memory and goes to the device with an
    ASYNCHRONOUS copy, so the caller is not blocked until the stream reaches the copy (a pageable-memory upload is)."""
    rings: Dict = {}

    @classmethod
    def upload(cls, shape, device) -> torch.Tensor:
        key = (tuple(shape), str(device))
        ring = cls.rings.get(key)
        if ring is None:
            if len(cls.rings) >= 8:                             # (pinned memory is a scarce resource: keep the rings of the last few shapes only)
                old = cls.rings.pop(next(iter(cls.rings)))
                for ev in old["events"]:
                    if ev is not None:
                        ev.synchronize()
            ring = cls.rings[key] = {"bufs": [torch.empty(shape, dtype=torch.float32).pin_memory() for _ in range(2)], "events": [None, None], "i": 0}
        i = ring["i"]
        ring["i"] = i ^ 1
        if ring["events"][i] is not None:
            ring["events"][i].synchronize()                     # (the upload that last used this buffer: two draws ago)
        torch.randn(shape, dtype=torch.float32, out=ring["bufs"][i])
        out = ring["bufs"][i].to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        ring["events"][i] = ev
        return out

    @classmethod
    def release(cls) -> None:
        """give the pinned buffers back (r04 advisor: up to 8 shapes x 2 buffers stayed pinned for the life of the process); in-flight uploads are waited for"""
        for ring in cls.rings.values():
            for ev in ring["events"]:
                if ev is not None:
                    ev.synchronize()
        cls.rings.clear()


def release_pinned_buffers() -> None:
    """free the pinned host staging buffers of the seeded host draws (``_host_noise``); they are re-created on demand"""
    _PinnedRing.release()


def _host_noise(like: torch.Tensor) -> torch.Tensor:
    """Fresh N(0, 1) drawn with the CPU generator and moved to the latent's device: seeding the host generator reproduces a trajectory on
    any device (mmgen's ``_get_noise_batch`` does the same).  The draw costs ~2 ns per value on the host (4.8 ms for 8 cars
    latents): callers issue it where the GPU has work queued (``_run_plan``, ``DiffusionNeRF.val_optim``), and the upload does not block (r04)."""
    if like.is_cuda:
        return _PinnedRing.upload(like.shape, like.device)
    return torch.randn(like.shape, dtype=torch.float32).to(like.device)


# ============================================================================================== schedule
class NoiseSchedule:
    """float64 tables of a T-step variance schedule (gaussian_diffusion.py:64-154).  ``kind``: 'linear' (betas from 1e-4 to 2e-2, rescaled by
    1000/T), 'cosine' (Nichol & Dhariwal), 'scaled_linear' (linear in sqrt(beta))."""

    def __init__(self, cfg: Dict, T: int):
        cfg = dict(cfg)
        self.kind = cfg.pop("type")
        self.T = T
        if self.kind == "linear":
            k = 1000 / T
            betas = np.linspace(k * cfg.get("beta_0", 1e-4), k * cfg.get("beta_T", 2e-2), T, dtype=np.float64)
        elif self.kind == "cosine":
            s, cap = cfg.get("s", 0.008), cfg.get("max_beta", 0.999)
            g = [math.cos((i / T + s) / (1 + s) * math.pi / 2) ** 2 for i in range(T + 1)]
            betas = np.array([min(1 - g[i + 1] / g[i], cap) for i in range(T)])
        elif self.kind == "scaled_linear":
            betas = np.linspace(cfg.get("beta_start", 1e-4) ** 0.5, cfg.get("beta_end", 2e-2) ** 0.5, T, dtype=np.float64) ** 2
        else:
            raise AttributeError(f"Unknown method name {self.kind} for beta schedule.")
        ab = np.cumprod(1.0 - betas, axis=0)
        ab_prev = np.append(1.0, ab[:-1])
        post_var = betas * (1 - ab_prev) / (1 - ab)                       # variance of q(x_{t-1} | x_t, x_0)
        self.tables = dict(
            betas=betas, alphas=1.0 - betas, alphas_bar=ab, alphas_bar_prev=ab_prev, alphas_bar_next=np.append(ab[1:], 0.0),
            sqrt_alphas_bar=np.sqrt(ab), sqrt_one_minus_alphas_bar=np.sqrt(1.0 - ab), log_one_minus_alphas_bar=np.log(1.0 - ab),
            sqrt_recip_alplas_bar=np.sqrt(1.0 / ab), sqrt_recipm1_alphas_bar=np.sqrt(1.0 / ab - 1), tilde_betas_t=post_var,
            log_tilde_betas_t_clipped=np.log(np.append(post_var[1], post_var[1:])),
            tilde_mu_t_coef1=np.sqrt(ab_prev) / (1 - ab) * betas, tilde_mu_t_coef2=np.sqrt(1.0 - betas) * (1 - ab_prev) / (1 - ab))
        self._dev: Dict = {}

    def __getitem__(self, name: str) -> np.ndarray:
        return self.tables[name]

    def signal_noise(self, device) -> torch.Tensor:
        """(2, T) fp32 on ``device``: row 0 sqrt(alpha_bar), row 1 sqrt(1 - alpha_bar); uploaded once per device, gathered by ``t`` tensors."""
        key = str(device)
        if key not in self._dev:
            self._dev[key] = torch.from_numpy(np.stack([self["sqrt_alphas_bar"], self["sqrt_one_minus_alphas_bar"]])).float().to(device)
        return self._dev[key]


@dataclass(frozen=True)
class PlanStep:
    """One network evaluation of a sampling trajectory; every coefficient is a host float resolved from the float64 tables.
    After x0 has been predicted at timestep ``t`` (signal ``a`` = sqrt(abar_t), noise level ``b`` = sqrt(1 - abar_t)), with
    eps = (x_t - a x0) / b, the latent becomes
        kind 'ddim'     :  c x0 + d eps + n z            (c = sqrt(abar_prev), d = sqrt(1 - abar_prev - eta^2 var), n = eta sqrt(var))
        kind 'langevin' :  x_t - d eps + n z             (d = delta b / 2, n = sqrt(delta) b)
        kind 'ddpm'     :  c x0 + d x_t + n z            (posterior mean coefficients, n = sqrt(var), 0 at t = 0)"""
    kind: str
    t: int
    a: float
    b: float
    c: float
    d: float
    n: float
    emit: bool            # a trajectory point of the reference's ``save_intermediates`` list (the DDIM steps, not the Langevin corrections)


class SamplingPlan:
    """The whole trajectory of ``ddim_sample`` / ``ddpm_sample`` as a list of ``PlanStep``s (timesteps arange(T-1, -1, -T/n).long(): 50 ->
    999, 979, ..., 19; 75 -> 999, 985, 972, ..., 12; the last DDIM step lands on alpha_bar_prev[0] = 1, i.e. x_prev = x0 exactly)."""

    def __init__(self, sched: NoiseSchedule, method: str, n: int, eta: float = 0.0, langevin_steps: int = 0, langevin_delta: float = 0.1,
                 langevin_t_range: Sequence[float] = (0, 1000), var_mode: str = "FIXED_LARGE"):
        T = sched.T
        self.timesteps = torch.arange(start=T - 1, end=-1, step=-(T / n)).long()
        ts = self.timesteps.tolist()
        ab, sa, sb, var = sched["alphas_bar"], sched["sqrt_alphas_bar"], sched["sqrt_one_minus_alphas_bar"], sched["tilde_betas_t"]
        steps: List[PlanStep] = []
        if method == "ddim":
            for i, t in enumerate(ts):
                t_prev = ts[i + 1] if i + 1 < len(ts) else -1
                ab_prev = ab[t_prev] if t_prev >= 0 else sched["alphas_bar_prev"][0]
                steps.append(PlanStep("ddim", t, float(sa[t]), float(sb[t]), float(np.sqrt(ab_prev)),
                                      float(np.sqrt(1 - ab_prev - var[t] * eta ** 2)), float(eta * np.sqrt(var[t])), True))
                if langevin_steps > 0 and langevin_t_range[0] < t_prev < langevin_t_range[1]:
                    sigma = float(sb[t_prev])
                    steps += [PlanStep("langevin", t_prev, float(sa[t_prev]), sigma, 0.0, 0.5 * langevin_delta * sigma,
                                       math.sqrt(langevin_delta) * sigma, False)] * langevin_steps
        elif method == "ddpm":
            mode = var_mode.upper()
            if mode == "FIXED_LARGE":
                table = np.append(var[1], sched["betas"])
            elif mode == "FIXED_SMALL":
                table = var
            else:
                raise AttributeError(f"Unknown denoising var output type [{var_mode}].")
            for t in ts:
                steps.append(PlanStep("ddpm", t, float(sa[t]), float(sb[t]), float(sched["tilde_mu_t_coef1"][t]), float(sched["tilde_mu_t_coef2"][t]),
                                      float(np.sqrt(table[t])) if t != 0 else 0.0, True))
        else:
            raise AttributeError(f"Cannot find sample method [{method}_sample] correspond to [{method}].")
        self.steps = steps

    def __len__(self):
        return len(self.steps)

    def __iter__(self):
        return iter(self.steps)


# ============================================================================================== timestep samplers
@MODULES.register_module()
class UniformTimeStepSampler:
    """t ~ Categorical(prob) drawn with ``np.random`` on the HOST (mmgen's sampler): a seeded ``np.random`` yields the
    same timesteps whatever device the model lives on."""

    def __init__(self, num_timesteps, **kwargs):
        self.num_timesteps = num_timesteps
        self.prob = [1 / num_timesteps] * num_timesteps

    def sample(self, batch_size):
        return torch.from_numpy(np.random.choice(self.num_timesteps, size=(batch_size,), p=self.prob)).long()

    __call__ = sample


MODULES.register_module(name="UniformTimeStepSamplerMod", module=type("UniformTimeStepSamplerMod", (UniformTimeStepSampler,), {}))


@MODULES.register_module()
class SNRWeightedTimeStepSampler(UniformTimeStepSampler):
    """Loss weight per timestep = clip(SNR^power + bias) converted to the network's output parameterisation (x0-, eps- or v-prediction), and
    sampling density proportional to weight^prob_power (sampler.py:14-44).  ``mean`` / ``std``: float64 sqrt(abar), sqrt(1 - abar)."""

    _to_output_space = {"START_X": lambda w, m, s: w, "EPS": lambda w, m, s: w * (s / m) ** 2, "V": lambda w, m, s: w * (s ** 2)}

    def __init__(self, num_timesteps, mean, std, mode, power=1, min=-1, max=-1, bias=0, prob_power=0.0):
        self.num_timesteps = num_timesteps
        m, s = np.asarray(mean, np.float64), np.asarray(std, np.float64)
        w_x0 = (m / s) ** (2 * power) + bias
        lo, hi = (min if min > 0 else None), (max if max > 0 else None)
        if lo is not None or hi is not None:
            w_x0 = w_x0.clip(min=lo, max=hi)
        if mode not in self._to_output_space:
            raise AttributeError(f"unknown denoising mean mode {mode!r}")
        w_out = self._to_output_space[mode](w_x0, m, s)
        density = w_out ** prob_power
        density = density / density.sum()
        self.weight = torch.from_numpy(w_out / (density * num_timesteps)).to(torch.float)      # importance-corrected loss weight
        self.prob = density.tolist()


# ============================================================================================== diffusion prior loss
@MODULES.register_module()
class DDPMMSELossMod(nn.Module):
    """Per-sample ``0.5 * mean((pred - target)^2)`` x ``weight[t] * weight_scale`` (``rescale_mode='timestep_weight'``; the table comes from
    the timestep sampler), batch-reduced, optionally divided by ``norm_factor`` -- a running mean of mean(x_0^2) that only moves in training
    mode (ddpm_loss.py:11-142 over mmgen's ``DDPMLoss``).  ``log_vars``: per-quartile means as 0-dim DEVICE tensors (the reference calls
    ``.item()`` per quartile: four syncs per loss)."""

    _default_data_info = dict(pred="eps_t_pred", target="noise")
    _reductions = {"mean": lambda v: v.mean(), "sum": lambda v: v.sum(), "none": lambda v: v,
                   "flatmean": lambda v: v.flatten(1).mean(dim=1) if v.dim() > 1 else v}

    def __init__(self, rescale_mode=None, rescale_cfg=None, sampler=None, weight=None, weight_scale=1.0, log_cfgs=None, reduction="mean",
                 data_info=None, loss_name="loss_ddpm_mse", scale_norm=False, momentum=0.001):
        super().__init__()
        if reduction not in self._reductions:
            raise ValueError(reduction)
        self.weight_scale, self.reduction, self.loss_name_, self.momentum = weight_scale, reduction, loss_name, momentum
        self.data_info = dict(data_info or self._default_data_info)
        self.rescale_mode, self.timestep_weight = rescale_mode, None
        if rescale_mode == "timestep_weight":
            table = getattr(sampler, "weight", None) if sampler is not None else None
            table = weight if table is None else table
            if table is None:
                raise ValueError("rescale_mode='timestep_weight' needs a sampler with a weight table, or an explicit weight")
            self.timestep_weight = torch.as_tensor(table, dtype=torch.float)
        elif rescale_mode is not None:
            raise NotImplementedError(f"rescale_mode={rescale_mode!r}: the reference's configs only use 'timestep_weight'")
        self.log_cfgs = [log_cfgs] if isinstance(log_cfgs, dict) else list(log_cfgs or [])
        self.log_vars: Dict[str, torch.Tensor] = {}
        self.scale_norm, self.freeze_norm = scale_norm, False
        if scale_norm:
            self.register_buffer("norm_factor", torch.ones(1, dtype=torch.float))

    def _quartile_log(self, per_sample, timesteps):
        self.log_vars = {}
        for cfg in self.log_cfgs:
            if cfg.get("type") != "quartile":
                continue
            which = (timesteps.float() / cfg.get("total_timesteps", 1000) * 4).long()
            onehot = torch.nn.functional.one_hot(which.clamp(0, 3), 4).to(per_sample.dtype)             # (B, 4): no host round trip
            means = (per_sample.detach()[:, None] * onehot).sum(0) / onehot.sum(0).clamp(min=1)
            for q in range(4):
                self.log_vars[f"{cfg.get('prefix_name', 'loss')}_quartile_{q}"] = means[q]

    def forward(self, output_dict):
        assert isinstance(output_dict, dict) and "timesteps" in output_dict, "DDPM losses take the dict of network outputs with 'timesteps'"
        t = output_dict["timesteps"]
        diff = output_dict[self.data_info["pred"]] - output_dict[self.data_info["target"]]
        per_sample = diff.square().flatten(1).mean(dim=1) * 0.5
        if self.timestep_weight is not None:
            if self.timestep_weight.device != t.device:
                self.timestep_weight = self.timestep_weight.to(t.device)                                 # moved once, not per call
            per_sample = per_sample * self.timestep_weight[t] * self.weight_scale
        self._quartile_log(per_sample, t)
        loss = self._reductions[self.reduction](per_sample)
        if self.scale_norm:
            if self.training and not self.freeze_norm:
                from .parallel import reduce_mean
                self.norm_factor.mul_(1 - self.momentum).add_(self.momentum * reduce_mean(output_dict["x_0"].detach().square().mean()))
            loss = loss / self.norm_factor
        return loss


MODULES.register_module(name="DDPMMSELoss", module=type("DDPMMSELoss", (DDPMMSELossMod,), {}))


# ============================================================================================== the diffusion model
_FROM_X0 = {  # network output that corresponds to a given x0 (used when guidance has moved x0 and the loss wants the matching output)
    "EPS": lambda x_t, x0, a, b: (x_t - x0 * a) / b,
    "START_X": lambda x_t, x0, a, b: x0,
    "V": lambda x_t, x0, a, b: (a * x_t - x0) / b,
}
_TO_X0 = {
    "EPS": lambda x_t, out, a, b: (x_t - b * out) / a,
    "START_X": lambda x_t, out, a, b: out,
    "V": lambda x_t, out, a, b: a * x_t - b * out,
}


@MODULES.register_module()
class GaussianDiffusion(nn.Module):
    def __init__(self, denoising, ddpm_loss=dict(type="DDPMMSELoss", log_cfgs=dict(type="quartile", prefix_name="loss_mse", total_timesteps=1000)),
                 betas_cfg=dict(type="cosine"), num_timesteps=1000, num_classes=0, sample_method="ddim",
                 timestep_sampler=dict(type="UniformTimeStepSampler"), denoising_var_mode="FIXED_LARGE", denoising_mean_mode="V", train_cfg=None,
                 test_cfg=None):
        super().__init__()
        self.num_classes, self.num_timesteps, self.sample_method = num_classes, num_timesteps, sample_method
        self._denoising_cfg = deepcopy(denoising)
        self.denoising = build_module(denoising, default_args=dict(num_classes=num_classes, num_timesteps=num_timesteps))
        self.denoising_var_mode, self.denoising_mean_mode = denoising_var_mode, denoising_mean_mode
        if denoising_mean_mode.upper() not in _TO_X0:
            raise AttributeError(f"Unknown denoising mean output type [{denoising_mean_mode}].")
        self.betas_cfg = deepcopy(betas_cfg)
        self.train_cfg = deepcopy(train_cfg) if train_cfg is not None else dict()
        self.test_cfg = deepcopy(test_cfg) if test_cfg is not None else dict()
        self.schedule = NoiseSchedule(self.betas_cfg, num_timesteps)
        self.betas_schedule = self.schedule.kind
        self.sampler = build_module(timestep_sampler or dict(type="UniformTimeStepSampler"),
                                    default_args=dict(num_timesteps=num_timesteps, mean=self.schedule["sqrt_alphas_bar"],
                                                      std=self.schedule["sqrt_one_minus_alphas_bar"], mode=self.denoising_mean_mode))
        self.ddpm_loss = build_module(ddpm_loss or dict(type="DDPMMSELoss"), default_args=dict(sampler=self.sampler))
        self.use_fused_step = True          # False: every step through the generic tensor expressions (parity runs)
        self._plans: Dict = {}

    def __getattr__(self, name):
        # the schedule tables under the reference's attribute names (``betas``, ``alphas_bar``, ``sqrt_alphas_bar``, ``tilde_betas_t`` ...)
        if name != "schedule" and "schedule" in self.__dict__ and name in self.__dict__["schedule"].tables:
            return self.__dict__["schedule"].tables[name]
        return super().__getattr__(name)

    # ------------------------------------------------------------------------------------------ plans
    def ddim_timesteps(self, num_timesteps=None):
        n = self.test_cfg.get("num_timesteps", self.num_timesteps) if num_timesteps is None else num_timesteps
        return torch.arange(start=self.num_timesteps - 1, end=-1, step=-(self.num_timesteps / n)).long()

    def sampling_plan(self, method="ddim", cfg=None) -> SamplingPlan:
        cfg = self.test_cfg if cfg is None else cfg
        key = (method, cfg.get("num_timesteps", self.num_timesteps), cfg.get("eta", 0), cfg.get("langevin_steps", 0), cfg.get("langevin_delta", 0.1),
               tuple(cfg.get("langevin_t_range", [0, 1000])), self.denoising_var_mode)
        if key not in self._plans:
            self._plans[key] = SamplingPlan(self.schedule, *key)
        return self._plans[key]

    # ------------------------------------------------------------------------------------------ forward process
    def q_sample(self, x_0, t, noise=None):
        """x_t = a[t] x_0 + b[t] noise and the two broadcastable coefficient tensors (a = sqrt(abar), b = sqrt(1 - abar))."""
        noise = _host_noise(x_0) if noise is None else noise
        ab = self.schedule.signal_noise(x_0.device)[:, torch.as_tensor(t, device=x_0.device).reshape(-1)]
        a, b = ab[0].reshape(-1, 1, 1, 1), ab[1].reshape(-1, 1, 1, 1)
        return x_0 * a + noise * b, a, b

    # ------------------------------------------------------------------------------------------ x0 prediction (+ guidance)
    def pred_x_0(self, x_t, t, grad_guide_fn=None, concat_cond=None, cfg=dict(), update_denoising_output=False):
        """x0 predicted from (x_t, t) -- clipped to ``clip_range`` -- and the network output.  With a guidance closure ``grad_guide_fn(x0) ->
        scalar loss``: x0 <- x0 - grad * b^(2 - 2w) a^(2w - 1) * gain, where the gradient is taken w.r.t. x_t THROUGH the UNet
        (``grad_through_unet``, default) or w.r.t. x0 directly, and w = ``snr_weight_power`` (gaussian_diffusion.py:180-240)."""
        clip = cfg.get("clip_range", [-1, 1]) if cfg.get("clip_denoised", True) else None
        t = torch.as_tensor(t, device=x_t.device)
        if t.dim() == 0 or t.numel() != x_t.size(0):
            t = t.expand(x_t.size(0))
        ab = self.schedule.signal_noise(x_t.device)[:, t]
        a, b = ab[0].reshape(-1, 1, 1, 1), ab[1].reshape(-1, 1, 1, 1)
        mode = self.denoising_mean_mode.upper()
        if grad_guide_fn is None:
            out = self.denoising(x_t, t, concat_cond=concat_cond)
            x0 = _TO_X0[mode](x_t, out, a, b)
            return (x0.clamp(*clip) if clip else x0), out

        through_unet = cfg.get("grad_through_unet", True)
        w = cfg.get("snr_weight_power", 0.5)
        if through_unet:
            with torch.enable_grad():
                # sampling hands a detached latent in (under no_grad): make it a leaf.  A latent that already carries a graph (the prior loss
                # with guidance, x_t = q_sample(code)) stays IN that graph, so the UNet-path gradient still reaches the code.
                if not x_t.requires_grad:
                    x_t = x_t.detach().requires_grad_(True)
                out = self.denoising(x_t, t, concat_cond=concat_cond)
                x0 = _TO_X0[mode](x_t, out, a, b)
                if clip:
                    x0 = x0.clamp(*clip)
                (grad,) = torch.autograd.grad(grad_guide_fn(x0), x_t, retain_graph=x_t.grad_fn is not None)
        else:
            out = self.denoising(x_t, t, concat_cond=concat_cond)
            x0 = _TO_X0[mode](x_t, out, a, b)
            if clip:
                x0 = x0.clamp(*clip)
            with torch.enable_grad():
                x0 = x0.detach().requires_grad_(True)
                (grad,) = torch.autograd.grad(grad_guide_fn(x0), x0)
        x0 = x0.detach() - grad * (b ** (2 - 2 * w) * a ** (2 * w - 1) * cfg.get("guidance_gain", 1.0))
        if clip:
            x0 = x0.clamp(*clip)
        if update_denoising_output:
            out = _FROM_X0[mode](x_t, x0, a, b)
        return x0, out

    # ------------------------------------------------------------------------------------------ sampling
    def _advance(self, s: PlanStep, x_t, x0, noise=None):
        """the latent after plan step ``s`` (see PlanStep)"""
        if s.kind == "ddpm":
            x = s.c * x0 + s.d * x_t
        else:
            eps = (x_t - s.a * x0) / s.b
            x = s.c * x0 + s.d * eps if s.kind == "ddim" else x_t - s.d * eps
        if s.n != 0 or s.kind == "ddpm":                                     # (the ancestral sampler draws at t = 0 too and multiplies by 0)
            x = x + s.n * (_host_noise(x_t) if noise is None else noise)
        return x

    def _fused_ok(self, x_t, cfg, grad_guide_fn, concat_cond):
        return (self.use_fused_step and grad_guide_fn is None and concat_cond is None and self.denoising_mean_mode.upper() == "V" and x_t.is_cuda
                and x_t.dtype == torch.float32 and cfg.get("clip_denoised", True) and not torch.is_grad_enabled())

    def _run_plan(self, plan: SamplingPlan, noise, concat_cond=None, save_intermediates=False, grad_guide_fn=None, **kwargs):
        cfg = self.test_cfg
        x_t = noise
        B = x_t.size(0)
        kept: Optional[list] = [] if save_intermediates else None
        t_rows = torch.tensor([s.t for s in plan], dtype=torch.long).to(x_t.device)[:, None].expand(-1, B).contiguous()   # ONE upload for the loop
        clip = cfg.get("clip_range", [-1, 1])
        session = None
        if self._fused_ok(x_t, cfg, grad_guide_fn, concat_cond) and hasattr(self.denoising, "inference_session"):
            session = self.denoising.inference_session(x_t, t_rows[0])
            if session is not None:
                session.x.copy_(x_t)                                         # the latent lives in the UNet's static input buffer from here on
        pending_x0 = None                                                    # x0 of the last emitting (DDIM) step; its trajectory point is the latent AFTER
                                                                             # the Langevin corrections that follow it (gaussian_diffusion.py:313-326)
        for i, s in enumerate(plan):
            cond = concat_cond[:, i % concat_cond.size(1)] if concat_cond is not None else None
            if kept is not None and s.emit and pending_x0 is not None:
                kept += [pending_x0, session.x.clone() if session is not None else x_t]
                pending_x0 = None
            if session is not None and s.kind == "ddim" and s.n == 0:
                # ---- device-resident unguided DDIM step: t -> static buffer, graph replay, one fused update written back in place
                session.t.copy_(t_rows[i])
                v = session.run()
                x0 = torch.empty_like(session.x) if kept is not None else session.y       # (x0 is only materialised when the caller keeps it)
                C.check(C.lib().ssdnerf_ddim_step_v(C.ptr(session.x), C.ptr(v), C.ctypes.c_uint64(v.numel()), C.f32(s.a), C.f32(s.b), C.f32(s.c),
                                                    C.f32(s.d), C.f32(clip[0]), C.f32(clip[1]), C.ptr(x0), C.ptr(session.x), C.stream()), "ddim_step_v")
                if kept is not None:
                    pending_x0 = x0
                continue
            if session is not None:                                          # a step kind the fused form does not cover: leave the session
                x_t, session = session.x.clone(), None
            # a step that injects noise (Langevin, ancestral DDPM, eta > 0): the HOST draw first -- the device still has the previous step's work queued, the
            # draw (4.8 ms for 8 cars latents) hides under it; same generator, same order of draws as drawing inside _advance
            step_noise = _host_noise(x_t) if (s.n != 0 or s.kind == "ddpm") else None
            x0, _ = self.pred_x_0(x_t, t_rows[i], grad_guide_fn=grad_guide_fn, concat_cond=cond, cfg=cfg, **kwargs)
            x_t = self._advance(s, x_t.detach() if grad_guide_fn is not None else x_t, x0, noise=step_noise)
            if kept is not None and s.emit:
                pending_x0 = x0
        if session is not None:
            x_t = session.x.clone()
        if pending_x0 is not None:
            kept += [pending_x0, x_t]
        return kept if save_intermediates else x_t

    def ddim_sample(self, noise, show_pbar=False, concat_cond=None, save_intermediates=False, **kwargs):
        return self._run_plan(self.sampling_plan("ddim"), noise, concat_cond, save_intermediates, **kwargs)

    def ddpm_sample(self, noise, show_pbar=False, concat_cond=None, **kwargs):
        return self._run_plan(self.sampling_plan("ddpm"), noise, concat_cond, False, **kwargs)

    # single-step entry points of the reference API (each resolves its coefficients like one plan entry)
    def p_sample_ddim(self, x_t, t, t_prev, noise=None, cfg=dict(), grad_guide_fn=None, **kwargs):
        t, t_prev, eta = int(t), int(t_prev), cfg.get("eta", 0)
        sc = self.schedule
        ab_prev = sc["alphas_bar"][t_prev] if t_prev >= 0 else sc["alphas_bar_prev"][0]
        s = PlanStep("ddim", t, float(sc["sqrt_alphas_bar"][t]), float(sc["sqrt_one_minus_alphas_bar"][t]), float(np.sqrt(ab_prev)),
                     float(np.sqrt(1 - ab_prev - sc["tilde_betas_t"][t] * eta ** 2)), float(eta * np.sqrt(sc["tilde_betas_t"][t])), True)
        x0, _ = self.pred_x_0(x_t, t, grad_guide_fn=grad_guide_fn, cfg=cfg, **kwargs)
        return self._advance(s, x_t, x0, noise), x0

    def p_sample_langevin(self, x_t, t, noise=None, cfg=dict(), grad_guide_fn=None, **kwargs):
        t, delta = int(t), cfg.get("langevin_delta", 0.1)
        sigma = float(self.schedule["sqrt_one_minus_alphas_bar"][t])
        s = PlanStep("langevin", t, float(self.schedule["sqrt_alphas_bar"][t]), sigma, 0.0, 0.5 * delta * sigma, math.sqrt(delta) * sigma, False)
        x0, _ = self.pred_x_0(x_t, t, grad_guide_fn=grad_guide_fn, cfg=cfg, **kwargs)
        return self._advance(s, x_t, x0, noise)

    def q_posterior_mean(self, x_0, x_t, t):
        idx = torch.as_tensor(t).reshape(-1).cpu().numpy()
        c1 = x_0.new_tensor(self.schedule["tilde_mu_t_coef1"][idx], dtype=torch.float32).reshape(-1, 1, 1, 1)
        c2 = x_0.new_tensor(self.schedule["tilde_mu_t_coef2"][idx], dtype=torch.float32).reshape(-1, 1, 1, 1)
        return c1 * x_0 + c2 * x_t

    def p_sample_ddpm(self, x_t, t, noise=None, cfg=dict(), grad_guide_fn=None, **kwargs):
        key = ("ddpm-every-t", self.denoising_var_mode)
        if key not in self._plans:
            self._plans[key] = SamplingPlan(self.schedule, "ddpm", self.num_timesteps, var_mode=self.denoising_var_mode)   # every t once
        s = self._plans[key].steps[self.num_timesteps - 1 - int(t)]
        x0, _ = self.pred_x_0(x_t, int(t), grad_guide_fn=grad_guide_fn, cfg=cfg, **kwargs)
        return self._advance(s, x_t, x0, noise), x0

    def sample_from_noise(self, noise, **kwargs):
        method = self.sample_method.lower()
        if method not in ("ddim", "ddpm"):
            raise AttributeError(f"Cannot find sample method [{method}_sample] correspond to [{self.sample_method}].")
        return getattr(self, f"{method}_sample")(noise=noise, **kwargs)

    # ------------------------------------------------------------------------------------------ prior loss
    def loss(self, denoising_output, x_0, noise, t, mean, std):
        key = {"EPS": "eps_t_pred", "START_X": "x_0_pred", "V": "v_t_pred"}[self.denoising_mean_mode.upper()]
        fields = {key: denoising_output, "x_0": x_0, "noise": noise, "timesteps": t}
        if key == "v_t_pred":
            fields["v_t"] = mean * noise - std * x_0
        return self.ddpm_loss(fields)

    def forward_train(self, x_0, concat_cond=None, grad_guide_fn=None, cfg=dict(), x_t_detach=False, timesteps=None, noise=None, **kwargs):
        """Diffusion prior loss of ``x_0`` (gaussian_diffusion.py:407-433).  ``timesteps`` / ``noise`` (extra): injected draws; by default both
        are drawn on the host like the reference's, so seeding reproduces them on any device.  ``log_vars['loss_ddpm_mse']`` is a detached
        0-dim tensor, not a Python float (no device sync here)."""
        assert x_0.dim() == 4
        dev = x_0.device
        t = (self.sampler(x_0.size(0)) if timesteps is None else torch.as_tensor(timesteps).long()).to(dev)
        noise = (_host_noise(x_0) if noise is None else noise).to(dev)
        x_t, a, b = self.q_sample(x_0, t, noise)
        if x_t_detach:
            x_t = x_t.detach()
        _, out = self.pred_x_0(x_t, t, grad_guide_fn=grad_guide_fn, concat_cond=concat_cond, cfg=cfg, update_denoising_output=True)
        value = self.loss(out, x_0, noise, t, a, b)
        log_vars = self.ddpm_loss.log_vars
        log_vars.update(loss_ddpm_mse=value.detach())
        return value, log_vars

    def forward_test(self, data, **kwargs):
        assert data.dim() == 4
        return self.sample_from_noise(data, **kwargs)

    def forward(self, data, return_loss=False, **kwargs):
        return self.forward_train(data, **kwargs) if return_loss else self.forward_test(data, **kwargs)
