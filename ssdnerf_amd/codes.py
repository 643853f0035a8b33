"""Scene-code activations and the small losses of the guidance / fitting path, under the reference's registry names
(lib/models/autodecoders/base_nerf.py:25-76 ``TanhCode`` / ``IdentityCode`` / ``NormalizedTanhCode``; mmgen ``MSELoss``, SURVEY.md Appendix A;
lib/models/losses/reg_loss.py ``RegLoss``).  A scene is stored as a *pre-activation* code ``code_``; everything downstream (diffusion,
decoder) sees ``code = activation(code_)``, which is bounded, so the latents live in a box the diffusion can clip to."""
from __future__ import annotations

import torch
import torch.nn as nn

from .registry import MODULES


class _Bounded(nn.Module):
    """y = r * tanh(u / r') family: ``inverse`` clips to the open box first so that atanh stays finite."""
    eps = 1e-5

    def _atanh(self, v):
        return v.clamp(min=-1 + self.eps, max=1 - self.eps).atanh()


@MODULES.register_module()
class TanhCode(_Bounded):
    """code = scale * tanh(code_)"""

    def __init__(self, scale=1.0, eps=1e-5):
        super().__init__()
        self.scale, self.eps = scale, eps

    def forward(self, code_, update_stats=False):
        y = code_.tanh()
        return y if self.scale == 1 else y * self.scale

    def inverse(self, code):
        return self._atanh(code if self.scale == 1 else code / self.scale)


@MODULES.register_module()
class IdentityCode(nn.Module):
    @staticmethod
    def forward(code_, update_stats=False):
        return code_

    @staticmethod
    def inverse(code):
        return code


@MODULES.register_module()
class NormalizedTanhCode(_Bounded):
    """code = clip * tanh(((code_ - running_mean) * std / (running_std + eps) + mean) / clip): the pre-activation codes are whitened with running
    statistics (updated only in training mode, all-reduced over ranks) before the bounded squash (base_nerf.py:51-76)."""

    def __init__(self, mean=0.0, std=1.0, clip_range=1, eps=1e-5, momentum=0.001):
        super().__init__()
        self.mean, self.std, self.clip_range, self.momentum, self.eps = mean, std, clip_range, momentum, eps
        self.register_buffer("running_mean", torch.tensor([0.0]))
        self.register_buffer("running_var", torch.tensor([std ** 2]))

    def _gain(self, like, inverse=False):
        spread = self.running_var.sqrt() + self.eps
        return (spread / self.std if inverse else self.std / spread).to(like.device), self.running_mean.to(like.device)

    def forward(self, code_, update_stats=False):
        if update_stats and self.training:
            from .parallel import reduce_mean
            with torch.no_grad():
                var, mu = torch.var_mean(code_)
                self.running_mean.mul_(1 - self.momentum).add_(self.momentum * reduce_mean(mu))
                self.running_var.mul_(1 - self.momentum).add_(self.momentum * reduce_mean(var))
        gain, mu = self._gain(code_)
        return (code_ * gain + (self.mean - mu * gain)).div(self.clip_range).tanh().mul(self.clip_range)

    def inverse(self, code):
        inv, mu = self._gain(code, inverse=True)
        return self._atanh(code.div(self.clip_range)).mul(self.clip_range * inv) + (mu - self.mean * inv)


@MODULES.register_module()
class MSELoss(nn.Module):
    """``loss_weight * mean(weight * (pred - target)^2)`` -- the ``pixel_loss`` of the configs."""

    def __init__(self, loss_weight=1.0, reduction="mean", **kwargs):
        super().__init__()
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, **kwargs):
        sq = (pred - target).square()
        return (sq if weight is None else sq * weight).mean() * self.loss_weight


@MODULES.register_module()
class RegLoss(nn.Module):
    """``loss_weight * mean(|x|^power)`` on the scene codes."""

    def __init__(self, power=1, loss_weight=1.0):
        super().__init__()
        self.power, self.loss_weight = power, loss_weight

    def forward(self, tensor, weight=None, avg_factor=None, **kwargs):
        mag = tensor.abs()
        return (mag if self.power == 1 else mag ** self.power).mean() * self.loss_weight


class _ConfigOnly(nn.Module):
    """Config entries that only the (out-of-scope) training loop executes: constructible, so that the reference's configs build unchanged."""

    def __init__(self, **kwargs):
        super().__init__()
        self.cfg = kwargs

    def forward(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__} belongs to the training loop, which is outside the hot path")


for _name in ("TVLoss", "L1LossMod"):
    MODULES.register_module(name=_name, module=type(_name, (_ConfigOnly,), {}))


# ---------------------------------------------------------------------------------------------- small utilities
def attr_path_get(obj, path: str, *default):
    """``getattr`` along a dotted path (``'diffusion_ema.ddpm_loss.weight_scale'``)."""
    for part in path.split("."):
        obj = getattr(obj, part, *default)
    return obj


def attr_path_set(obj, path: str, value):
    head, _, leaf = path.rpartition(".")
    setattr(attr_path_get(obj, head) if head else obj, leaf, value)


class frozen:
    """``with frozen(module_a, module_b): ...`` -- requires_grad False on every parameter inside the block, previous flags restored after."""

    def __init__(self, *modules):
        self.params = [p for m in modules for p in m.parameters()]

    def __enter__(self):
        self.flags = [p.requires_grad for p in self.params]
        for p in self.params:
            p.requires_grad_(False)
        return self

    def __exit__(self, *exc):
        for p, f in zip(self.params, self.flags):
            p.requires_grad_(f)
