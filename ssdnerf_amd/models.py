"""Model-level glue of the hot path: ``BaseNeRF`` / ``MultiSceneNeRF`` / ``DiffusionNeRF`` with the reference's
constructor keywords (so ``configs/paper_cfgs/*.py`` build unchanged) and the test-time methods the north-star configs
exercise:

  render, get_density, update_extra_state, loss, ray_sample, load_scene / save_scene   (lib/models/autodecoders/base_nerf.py)
  val_uncond, val_guide + grad_guide_fn, code_diff_pr[_inv], val_step                 (lib/models/autodecoders/diffusion_nerf.py)

and, as the first row of SURVEY.md section 8(f), the fine-tuning half of ``cond_mode='guide_optim'``:

  get_init_code_, build_optimizer, build_scheduler, loss_decoder, inverse_code          (base_nerf.py:184-229, 298-316, 403-492)
  val_optim, the ``override_cfg`` switch in ``train()``                                  (diffusion_nerf.py:313-404, base_nerf.py:127-140)

and the remaining section 8(f) rows at the host level: the scene cache (``load_cache`` / ``save_cache``, ``scene_cache.py``) and the
training steps ``MultiSceneNeRF.train_step`` / ``DiffusionNeRF.train_step`` (multiscene_nerf.py:185-245, diffusion_nerf.py:66-189),
which compose the pieces above (the runner, hooks, EMA updates, datasets, evaluation and visualisation stay out of scope,
SURVEY.md section 2).
"""
from __future__ import annotations

import math
import os
from copy import deepcopy
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nerf
from .density import get_density as _get_density, update_density_grid
from .registry import MODELS, MODULES, build_module, get_module_device


# ---------------------------------------------------------------------------------------------- code activations
@MODULES.register_module()
class TanhCode(nn.Module):
    def __init__(self, scale=1.0, eps=1e-5):
        super().__init__()
        self.scale = scale
        self.eps = eps

    def forward(self, code_, update_stats=False):
        return code_.tanh() if self.scale == 1 else code_.tanh() * self.scale

    def inverse(self, code):
        c = code if self.scale == 1 else code / self.scale
        return c.clamp(min=-1 + self.eps, max=1 - self.eps).atanh()


@MODULES.register_module()
class IdentityCode(nn.Module):
    @staticmethod
    def forward(code_, update_stats=False):
        return code_

    @staticmethod
    def inverse(code):
        return code


@MODULES.register_module()
class NormalizedTanhCode(nn.Module):
    def __init__(self, mean=0.0, std=1.0, clip_range=1, eps=1e-5, momentum=0.001):
        super().__init__()
        self.mean, self.std, self.clip_range, self.momentum, self.eps = mean, std, clip_range, momentum, eps
        self.register_buffer("running_mean", torch.tensor([0.0]))
        self.register_buffer("running_var", torch.tensor([std ** 2]))

    def forward(self, code_, update_stats=False):
        if update_stats and self.training:          # running statistics of the pre-activation codes (base_nerf.py:64-68)
            from .parallel import reduce_mean
            with torch.no_grad():
                var, mean = torch.var_mean(code_)
                self.running_mean.mul_(1 - self.momentum).add_(self.momentum * reduce_mean(mean))
                self.running_var.mul_(1 - self.momentum).add_(self.momentum * reduce_mean(var))
        scale = (self.std / (self.running_var.sqrt() + self.eps)).to(code_.device)
        return (code_ * scale + (self.mean - self.running_mean.to(code_.device) * scale)).div(self.clip_range).tanh().mul(self.clip_range)

    def inverse(self, code):
        scale = ((self.running_var.sqrt() + self.eps) / self.std).to(code.device)
        return code.div(self.clip_range).clamp(min=-1 + self.eps, max=1 - self.eps).atanh().mul(self.clip_range * scale) + (
            self.running_mean.to(code.device) - self.mean * scale)


# ---------------------------------------------------------------------------------------------- losses on the guidance path
@MODULES.register_module()
class MSELoss(nn.Module):
    """mmgen ``MSELoss``: ``loss_weight * mean((pred - target)^2)``."""

    def __init__(self, loss_weight=1.0, reduction="mean", **kwargs):
        super().__init__()
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, **kwargs):
        d = (pred - target).square()
        if weight is not None:
            d = d * weight
        return d.mean() * self.loss_weight


@MODULES.register_module()
class RegLoss(nn.Module):
    """``loss_weight * mean(|code|^power)`` (lib/models/losses/reg_loss.py)."""

    def __init__(self, power=1, loss_weight=1.0):
        super().__init__()
        self.power, self.loss_weight = power, loss_weight

    def forward(self, tensor, weight=None, avg_factor=None, **kwargs):
        v = tensor.abs().mean() if self.power == 1 else (tensor.abs() ** self.power).mean()
        return v * self.loss_weight


def rgetattr(obj, attr, *default):
    """dotted-path getattr (lib/core/utils/misc.py:129-134)."""
    for name in attr.split("."):
        obj = getattr(obj, name, *default)
    return obj


def rsetattr(obj, attr, val):
    pre, _, post = attr.rpartition(".")
    return setattr(rgetattr(obj, pre) if pre else obj, post, val)


class _requires_grad:
    """``module_requires_grad`` (lib/core/utils/misc.py): set the flag on every parameter inside the block, restore after."""

    def __init__(self, module, flag):
        self.params, self.flag = list(module.parameters()), flag

    def __enter__(self):
        self.prev = [p.requires_grad for p in self.params]
        for p in self.params:
            p.requires_grad_(self.flag)

    def __exit__(self, *exc):
        for p, r in zip(self.params, self.prev):
            p.requires_grad_(r)


class _ConfigOnly(nn.Module):
    """Training-only config entries (losses / samplers / hooks): constructed so configs build, never executed."""

    def __init__(self, **kwargs):
        super().__init__()
        self.cfg = kwargs

    def forward(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__} belongs to the training loop, which is outside the hot path")


for _name in ("TVLoss", "L1LossMod"):
    MODULES.register_module(name=_name, module=type(_name, (_ConfigOnly,), {}))


# ---------------------------------------------------------------------------------------------- models
class BaseNeRF(nn.Module):
    def __init__(self, code_size=(3, 8, 64, 64), code_activation=dict(type="TanhCode", scale=1), grid_size=64,
                 decoder=dict(type="TriPlaneDecoder"), decoder_use_ema=False, bg_color=1, pixel_loss=dict(type="MSELoss"), reg_loss=None,
                 update_extra_interval=16, use_lpips_metric=True, init_from_mean=False, init_scale=1e-4, mean_ema_momentum=0.001,
                 mean_scale=1.0, train_cfg=dict(), test_cfg=dict(), pretrained=None):
        super().__init__()
        self.code_size = tuple(code_size)
        self.code_activation = build_module(code_activation)
        self.grid_size = grid_size
        self.decoder = build_module(decoder)
        self.decoder_use_ema = decoder_use_ema
        if self.decoder_use_ema:
            self.decoder_ema = deepcopy(self.decoder)
        self.bg_color = bg_color
        self.pixel_loss = build_module(pixel_loss)
        self.reg_loss = build_module(reg_loss) if reg_loss is not None else None
        self.train_cfg = dict(train_cfg or {})
        self.test_cfg = dict(test_cfg or {})
        self.update_extra_interval = update_extra_interval
        if init_from_mean:
            self.register_buffer("init_code", torch.zeros(self.code_size))
        else:
            self.init_code = None
        self.init_scale, self.mean_ema_momentum, self.mean_scale = init_scale, mean_ema_momentum, mean_scale
        if pretrained is not None and os.path.isfile(pretrained):
            sd = torch.load(pretrained, map_location="cpu")
            self.load_state_dict(sd.get("state_dict", sd), strict=False)
        self.train_cfg_backup = dict()
        self._backup_override_cfg()

    # ---- test-time attribute overrides (base_nerf.py:127-140): ``test_cfg['override_cfg']`` maps dotted attribute paths to the values
    # they take in eval mode, e.g. {'diffusion_ema.ddpm_loss.weight_scale': 1.0} in the recons configs -------------------------------
    def _backup_override_cfg(self):
        for key in self.test_cfg.get("override_cfg", dict()):
            self.train_cfg_backup[key] = rgetattr(self, key, None)

    def train(self, mode=True):
        if mode:
            for key, value in self.train_cfg_backup.items():
                rsetattr(self, key, value)
        else:
            for key, value in self.test_cfg.get("override_cfg", dict()).items():
                if self.training:
                    self.train_cfg_backup[key] = rgetattr(self, key)
                rsetattr(self, key, value)
        return super().train(mode)

    # ---- scene wire format (base_nerf.py:143-170) -----------------------------------------------------------------
    def load_scene(self, data, load_density=False):
        device = get_module_device(self)
        codes, grids, bits = [], [], []
        for st in data["code"]:
            p = st["param"]
            codes.append(p["code"] if "code" in p else self.code_activation(p["code_"]))
            if load_density:
                grids.append(p["density_grid"])
                bits.append(p["density_bitfield"])
        code = torch.stack(codes, dim=0).to(device)
        return (code, torch.stack(grids, dim=0).to(device) if load_density else None,
                torch.stack(bits, dim=0).to(device) if load_density else None)

    @staticmethod
    def save_scene(save_dir, code, density_grid, density_bitfield, scene_name):
        os.makedirs(save_dir, exist_ok=True)
        for i, name in enumerate(scene_name):
            torch.save(dict(scene_name=name, param=dict(code=code.data[i].cpu(), density_grid=density_grid.data[i].cpu(),
                                                        density_bitfield=density_bitfield.data[i].cpu())),
                       os.path.join(save_dir, name) + ".pth")

    def get_init_code_(self, num_scenes, device=None):
        """Pre-activation code leaf (base_nerf.py:184-192): U(-init_scale, init_scale), or the inverse-activated mean code."""
        code_ = torch.empty(self.code_size if num_scenes is None else (num_scenes, *self.code_size), device=device, requires_grad=True,
                            dtype=torch.float32)
        if self.init_code is None:
            code_.data.uniform_(-self.init_scale, self.init_scale)
        else:
            code_.data[:] = self.code_activation.inverse(self.init_code * self.mean_scale)
        return code_

    @staticmethod
    def build_optimizer(code_, cfg):
        """``cfg['optimizer'] = dict(type=<torch.optim class>, **kwargs)`` over the code leaf/leaves (base_nerf.py:204-214)."""
        optimizer_cfg = dict(cfg["optimizer"])
        optimizer_class = getattr(torch.optim, optimizer_cfg.pop("type"))
        if isinstance(code_, list):
            return [optimizer_class([c], **optimizer_cfg) for c in code_]
        return optimizer_class([code_], **optimizer_cfg)

    @staticmethod
    def build_scheduler(code_optimizer, cfg):
        if "lr_scheduler" not in cfg:
            return None
        scheduler_cfg = dict(cfg["lr_scheduler"])
        scheduler_class = getattr(torch.optim.lr_scheduler, scheduler_cfg.pop("type"))
        if isinstance(code_optimizer, list):
            return [scheduler_class(o, **scheduler_cfg) for o in code_optimizer]
        return scheduler_class(code_optimizer, **scheduler_cfg)

    def get_init_density_grid(self, num_scenes, device=None):
        """zero Morton grid, (H^3,) for one scene (``num_scenes=None``) or (S, H^3)   (base_nerf.py:194-197)"""
        return torch.zeros(self.grid_size ** 3 if num_scenes is None else (num_scenes, self.grid_size ** 3), device=device, dtype=torch.float16)

    def get_init_density_bitfield(self, num_scenes, device=None):
        return torch.zeros(self.grid_size ** 3 // 8 if num_scenes is None else (num_scenes, self.grid_size ** 3 // 8), device=device,
                           dtype=torch.uint8)

    # ---- density grid (base_nerf.py:318-401) ------------------------------------------------------------------------
    def update_extra_state(self, decoder, code, density_grid, density_bitfield, iter_density, density_thresh=0.01, decay=0.9, S=128,
                           jitter=None):
        if iter_density >= 16:
            raise NotImplementedError("the partial-update branch (base_nerf.py:353-376) is unreachable from the hot-path configs "
                                      "(SURVEY.md Appendix B.12)")
        with torch.no_grad():
            update_density_grid(decoder, code, density_grid, density_bitfield, density_thresh=density_thresh, decay=decay, jitter=jitter,
                                return_thresh=False)

    def get_density(self, decoder, code, cfg=dict(), jitters=None):
        return _get_density(decoder, code, self.grid_size, density_thresh=cfg.get("density_thresh", 0.01),
                            density_step=cfg.get("density_step", 8), jitters=jitters)

    # ---- guidance loss (base_nerf.py:231-261, 276-296) ---------------------------------------------------------------
    @staticmethod
    def ray_sample(cond_rays_o, cond_rays_d, cond_imgs, n_samples, sample_inds=None):
        device = cond_rays_o.device
        s, v, h, w, _ = cond_rays_o.size()
        npix = v * h * w
        rays_o, rays_d = cond_rays_o.reshape(s, npix, 3), cond_rays_d.reshape(s, npix, 3)
        target = cond_imgs.reshape(s, npix, 3)
        if npix > n_samples:
            if sample_inds is None:
                sample_inds = torch.stack([torch.randperm(npix, device=device)[:n_samples] for _ in range(s)], dim=0)
            ar = torch.arange(s, device=device)[:, None]
            rays_o, rays_d, target = rays_o[ar, sample_inds], rays_d[ar, sample_inds], target[ar, sample_inds]
        return rays_o, rays_d, target

    @staticmethod
    def get_raybatch_inds(cond_imgs, n_inverse_rays):
        device = cond_imgs.device
        s, v, h, w, _ = cond_imgs.size()
        npix = v * h * w
        if npix > n_inverse_rays:
            inds = torch.stack([torch.randperm(npix, device=device) for _ in range(s)], dim=0).split(n_inverse_rays, dim=1)
            return inds, len(inds)
        return None, None

    def loss(self, decoder, code, density_bitfield, target_rgbs, rays_o, rays_d, dt_gamma=0.0, return_decoder_loss=False,
             scale_num_ray=1.0, cfg=dict(), perturb=True, **kwargs):
        outputs = decoder(rays_o, rays_d, code, density_bitfield, self.grid_size, dt_gamma=dt_gamma, perturb=perturb,
                          return_loss=return_decoder_loss)
        out_weights = outputs["weights_sum"]
        out_rgbs = outputs["image"] + self.bg_color * (1 - out_weights.unsqueeze(-1))
        scale = 1 - math.exp(-cfg["loss_coef"] * scale_num_ray) if "loss_coef" in cfg else 1
        pixel_loss = self.pixel_loss(out_rgbs, target_rgbs, **kwargs) * (scale * 3)
        loss = pixel_loss
        loss_dict = dict(pixel_loss=pixel_loss)
        if self.reg_loss is not None:
            reg = self.reg_loss(code, **kwargs)
            loss = loss + reg
            loss_dict.update(reg_loss=reg)
        if return_decoder_loss and outputs.get("decoder_reg_loss") is not None:
            loss = loss + outputs["decoder_reg_loss"]
            loss_dict.update(decoder_reg_loss=outputs["decoder_reg_loss"])
        return out_rgbs, loss, loss_dict

    # ---- render (base_nerf.py:494-533) ---------------------------------------------------------------------------------
    def render(self, decoder, code, density_bitfield, h, w, intrinsics, poses, cfg=dict()):
        return nerf.render(decoder, code, density_bitfield, h, w, intrinsics, poses, grid_size=self.grid_size, bg_color=self.bg_color, cfg=cfg)

    def mean_ema_update(self, code):
        """EMA of the batch-mean code into ``init_code`` (``init_from_mean=True`` models; base_nerf.py:612-617)."""
        if self.init_code is None:
            return
        from .parallel import reduce_mean
        self.init_code.mul_(1 - self.mean_ema_momentum).add_(reduce_mean(code.detach().mean(dim=0)).data, alpha=self.mean_ema_momentum)

    def train_step(self, data, optimizer, running_status=None):
        raise NotImplementedError("BaseNeRF has no training step of its own (base_nerf.py:619-620); MultiSceneNeRF / DiffusionNeRF do")

    @staticmethod
    def _train_log(log_vars, out_rgbs, target_rgbs, code):
        """train_psnr / code_rms as 0-dim tensors (the reference converts every entry with float(), one sync each)."""
        log_vars.update(train_psnr=nerf.eval_psnr(out_rgbs.detach(), target_rgbs).mean(), code_rms=code.detach().square().flatten(1).mean().sqrt().mean())

    def loss_decoder(self, decoder, code, density_bitfield, cond_rays_o, cond_rays_d, cond_imgs, dt_gamma=0.0, cfg=dict(), **kwargs):
        """Rendering loss on ``n_decoder_rays`` freshly sampled rays (base_nerf.py:298-316); log values stay 0-dim tensors."""
        decoder_training_prev = decoder.training
        decoder.train(True)
        rays_o, rays_d, target_rgbs = self.ray_sample(cond_rays_o, cond_rays_d, cond_imgs, n_samples=cfg.get("n_decoder_rays", 4096))
        out_rgbs, loss, loss_dict = self.loss(decoder, code, density_bitfield, target_rgbs, rays_o, rays_d, dt_gamma, return_decoder_loss=True,
                                              scale_num_ray=cond_rays_o.shape[1:4].numel(), cfg=cfg, **kwargs)
        decoder.train(decoder_training_prev)
        return loss, {k: v.detach() for k, v in loss_dict.items()}, out_rgbs, target_rgbs

    def inverse_code(self, decoder, cond_imgs, cond_rays_o, cond_rays_d, dt_gamma=0, cfg=dict(), code_=None, density_grid=None,
                     density_bitfield=None, iter_density=None, code_optimizer=None, code_scheduler=None, prior_grad=None, show_pbar=False,
                     march_noises=None, density_jitters=None):
        """Optimisation-based inverse rendering of the scene codes (base_nerf.py:403-492): ``n_inverse_steps`` iterations of
        {activate code_, refresh the density grid every ``update_extra_interval`` steps, render a ray batch through the TRAIN branch,
        seed the gradient with ``prior_grad`` (the diffusion-prior gradient of ``val_optim``) or zero it, back-propagate, optimizer
        step, scheduler step}.  Works on the leaf ``code_`` in place.

        ``march_noises`` / ``density_jitters`` (extra): iterators (or lists) of injected per-step march jitter (S,R) and per-refresh grid
        jitter (H^3,3), replacing the reference's on-device ``torch.rand`` draws in parity runs."""
        device = get_module_device(self)
        decoder_training_prev = decoder.training
        decoder.train(True)
        march_noises = iter(march_noises) if march_noises is not None else None
        density_jitters = iter(density_jitters) if density_jitters is not None else None
        with _requires_grad(decoder, False):
            n_inverse_steps = cfg.get("n_inverse_steps", 1000)
            n_inverse_rays = cfg.get("n_inverse_rays", 4096)
            num_scenes, num_imgs, h, w, _ = cond_imgs.size()
            num_scene_pixels = num_imgs * h * w
            raybatch_inds, num_raybatch = self.get_raybatch_inds(cond_imgs, n_inverse_rays)
            if code_ is None:
                code_ = self.get_init_code_(num_scenes, device=device)
            if density_grid is None:
                density_grid = self.get_init_density_grid(num_scenes, device)
            if density_bitfield is None:
                density_bitfield = self.get_init_density_bitfield(num_scenes, device)
            if iter_density is None:
                iter_density = 0
            if code_optimizer is None:
                assert code_scheduler is None
                code_optimizer = self.build_optimizer(code_, cfg)
            if code_scheduler is None:
                code_scheduler = self.build_scheduler(code_optimizer, cfg)
            assert n_inverse_steps > 0
            optimizers = code_optimizer if isinstance(code_optimizer, list) else [code_optimizer]
            schedulers = [] if code_scheduler is None else (code_scheduler if isinstance(code_scheduler, list) else [code_scheduler])

            for inverse_step_id in range(n_inverse_steps):
                code = self.code_activation(torch.stack(code_, dim=0) if isinstance(code_, list) else code_)
                if inverse_step_id % self.update_extra_interval == 0:
                    self.update_extra_state(decoder, code.detach(), density_grid, density_bitfield, iter_density,
                                            density_thresh=cfg.get("density_thresh", 0.01),
                                            jitter=None if density_jitters is None else next(density_jitters))
                inds = raybatch_inds[inverse_step_id % num_raybatch] if raybatch_inds is not None else None
                rays_o, rays_d, target_rgbs = self.ray_sample(cond_rays_o, cond_rays_d, cond_imgs, n_inverse_rays, sample_inds=inds)
                if march_noises is not None:
                    decoder.injected_noises = next(march_noises)
                try:
                    out_rgbs, loss, loss_dict = self.loss(decoder, code, density_bitfield, target_rgbs, rays_o, rays_d, dt_gamma,
                                                          scale_num_ray=num_scene_pixels, cfg=cfg)
                finally:
                    decoder.injected_noises = None
                if prior_grad is not None:
                    if isinstance(code_, list):
                        for c, g in zip(code_, prior_grad):
                            c.grad.copy_(g)
                    else:
                        code_.grad.copy_(prior_grad)
                else:
                    for o in optimizers:
                        o.zero_grad()
                loss.backward()
                for o in optimizers:
                    o.step()
                for sch in schedulers:
                    sch.step()
        decoder.train(decoder_training_prev)
        return code.detach(), density_grid, density_bitfield, loss, loss_dict, out_rgbs, target_rgbs


@MODELS.register_module()
class MultiSceneNeRF(BaseNeRF):
    """Adds the per-scene cache of pre-activation codes + optimizer states (multiscene_nerf.py:31-183; wire format and 16-bit casting
    rules in ``ssdnerf_amd/scene_cache.py``).  The RAM cache is sharded over ranks with the same ``round(linspace)`` split as the
    scene sampler, so a rank only ever holds the scenes it is handed."""

    def __init__(self, *args, cache_size=0, cache_16bit=False, num_file_writers=0, **kwargs):
        super().__init__(*args, **kwargs)
        self.cache_size, self.cache_16bit, self.num_file_writers = cache_size, cache_16bit, num_file_writers
        if cache_size > 0:
            import torch.distributed as dist
            from .parallel import shard_scenes
            rank, ws = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
            self.cache = {ind: None for ind in shard_scenes(cache_size, rank, ws)}
        else:
            self.cache = None
        self.cache_loaded = False
        self.file_writers = None

    def load_cache(self, data):
        """-> (list of pre-activation code leaves, their optimizers, density_grid (S,H^3), density_bitfield (S,H^3/8)) for the scenes
        of ``data['scene_id']``: from the RAM cache (filled once from ``train_cfg['cache_load_from']`` when given), else from
        ``data['code']``, else freshly initialised (multiscene_nerf.py:74-129)."""
        from .scene_cache import optimizer_set_state
        device = get_module_device(self)
        num_scenes = len(data["scene_id"])
        if self.cache is not None:
            if not self.cache_loaded:
                cache_load_from = self.train_cfg.get("cache_load_from", None)
                if cache_load_from is not None:
                    cache_files = sorted(os.listdir(cache_load_from))
                    if len(cache_files) > 0:
                        assert len(cache_files) == self.cache_size
                        for ind in self.cache.keys():
                            self.cache[ind] = torch.load(os.path.join(cache_load_from, cache_files[ind]), map_location="cpu")
                self.cache_loaded = True
            cache_list = [self.cache[int(i)] for i in data["scene_id"]]
        elif "code" in data:
            cache_list = data["code"]
        else:
            cache_list = [None for _ in range(num_scenes)]
        code_list_, density_grid, density_bitfield = [], [], []
        for st in cache_list:
            if st is None:
                code_list_.append(self.get_init_code_(None, device))
                density_grid.append(self.get_init_density_grid(None, device))
                density_bitfield.append(self.get_init_density_bitfield(None, device))
            else:
                if "code_" in st["param"]:
                    code_ = st["param"]["code_"].to(dtype=torch.float32, device=device)
                else:       # a test-time scene file (activated code only): invert the activation, as the reference does with a warning
                    assert "code" in st["param"]
                    import warnings
                    warnings.warn("Pre-activation codes not found. Using on-the-fly inversion instead (which could be inconsistent).")
                    code_ = self.code_activation.inverse(st["param"]["code"].to(dtype=torch.float32, device=device))
                code_list_.append(code_.requires_grad_(True))
                density_grid.append(st["param"]["density_grid"].to(device))
                density_bitfield.append(st["param"]["density_bitfield"].to(device))
        density_grid = torch.stack(density_grid, dim=0)
        density_bitfield = torch.stack(density_bitfield, dim=0)
        code_optimizers = self.build_optimizer(code_list_, self.train_cfg)
        for ind, st in enumerate(cache_list):
            if st is not None and "optimizer" in st:
                optimizer_set_state(code_optimizers[ind], st["optimizer"])
        return code_list_, code_optimizers, density_grid, density_bitfield

    def save_cache(self, code_list_, code_optimizers, density_grid, density_bitfield, scene_id, scene_name):
        """Write the scenes back to the RAM cache (in place when the entry exists) and, with ``train_cfg['save_dir']``, to
        ``<save_dir>/<scene_name>.pth`` - fp16 code + bf16 optimizer moments when ``cache_16bit`` (multiscene_nerf.py:131-183)."""
        from .scene_cache import _FileWriters, load_tensor_to_dict, optimizer_state_copy, optimizer_state_to, out_dict_to
        if self.cache_16bit:
            code_dtype = torch.float16 if code_list_[0].dtype == torch.float32 else code_list_[0].dtype
            optimizer_dtype = torch.bfloat16
        else:
            code_dtype, optimizer_dtype = code_list_[0].dtype, torch.float32
        save_dir = self.train_cfg.get("save_dir", None)
        if save_dir is not None:
            os.makedirs(save_dir, exist_ok=True)
            if self.num_file_writers > 0 and self.file_writers is None:
                self.file_writers = _FileWriters(save_dir, self.num_file_writers)
        for ind, code_single_ in enumerate(code_list_):
            sid = int(scene_id[ind])
            out = dict(scene_id=scene_id[ind], scene_name=scene_name[ind],
                       param=dict(code_=code_single_.data, density_grid=density_grid[ind], density_bitfield=density_bitfield[ind]),
                       optimizer=code_optimizers[ind].state_dict())
            if self.cache is not None:
                if self.cache[sid] is None:
                    self.cache[sid] = out_dict_to(out, device="cpu", code_dtype=code_dtype, optimizer_dtype=optimizer_dtype)
                else:
                    entry = self.cache[sid]
                    entry.setdefault("scene_id", out["scene_id"])
                    entry.setdefault("scene_name", out["scene_name"])
                    entry["param"].pop("code", None)
                    for key, val in out["param"].items():
                        load_tensor_to_dict(entry["param"], key, val, device="cpu", dtype=code_dtype)
                    if "optimizer" in entry:
                        optimizer_state_copy(out["optimizer"], entry["optimizer"], device="cpu", dtype=optimizer_dtype)
                    else:
                        entry["optimizer"] = optimizer_state_to(out["optimizer"], device="cpu", dtype=optimizer_dtype)
            if save_dir is not None:
                obj = out_dict_to(out, device="cpu", code_dtype=code_dtype, optimizer_dtype=optimizer_dtype)
                if self.file_writers is not None:
                    self.file_writers.put(ind, obj)
                else:        # (the reference joins the LIST scene_name here, multiscene_nerf.py:182, which raises; the per-scene name is meant)
                    torch.save(obj, os.path.join(save_dir, scene_name[ind] + ".pth"))


    def _cond_rays(self, data):
        cond_imgs, cond_intrinsics, cond_poses = data["cond_imgs"], data["cond_intrinsics"], data["cond_poses"]
        _, _, h, w, _ = cond_imgs.size()
        cond_rays_o, cond_rays_d = nerf.get_cam_rays(cond_poses, cond_intrinsics, h, w)
        dt_gamma = self.train_cfg.get("dt_gamma_scale", 0.0) / cond_intrinsics[..., :2].mean(dim=(-2, -1))
        return cond_imgs, cond_rays_o, cond_rays_d, dt_gamma

    def train_step(self, data, optimizer, running_status=None):
        """Stage-1 auto-decoder step (multiscene_nerf.py:185-245): ``extra_scene_step`` code-only iterations, then one joint iteration of
        codes + decoder on ``n_decoder_rays`` rays, cache write-back.  ``optimizer`` = {'decoder': torch optimizer}; the per-scene code
        optimizers come from the cache.  Log values are 0-dim tensors."""
        code_list_, code_optimizers, density_grid, density_bitfield = self.load_cache(data)
        cond_imgs, cond_rays_o, cond_rays_d, dt_gamma = self._cond_rays(data)
        extra_scene_step = self.train_cfg.get("extra_scene_step", 0)
        if extra_scene_step > 0:
            cfg = dict(self.train_cfg)
            cfg["n_inverse_steps"] = extra_scene_step
            self.inverse_code(self.decoder, cond_imgs, cond_rays_o, cond_rays_d, dt_gamma=dt_gamma, cfg=cfg, code_=code_list_,
                              density_grid=density_grid, density_bitfield=density_bitfield, code_optimizer=code_optimizers)
        for o in code_optimizers:
            o.zero_grad()
        optimizer["decoder"].zero_grad()
        code = self.code_activation(torch.stack(code_list_, dim=0), update_stats=True)
        self.update_extra_state(self.decoder, code.detach(), density_grid, density_bitfield, 0, density_thresh=self.train_cfg.get("density_thresh", 0.01))
        loss, log_vars, out_rgbs, target_rgbs = self.loss_decoder(self.decoder, code, density_bitfield, cond_rays_o, cond_rays_d, cond_imgs, dt_gamma,
                                                                  cfg=self.train_cfg)
        loss.backward()
        log_vars.update(loss=loss.detach())
        optimizer["decoder"].step()
        for o in code_optimizers:
            o.step()
        self.save_cache(code_list_, code_optimizers, density_grid, density_bitfield, data["scene_id"], data["scene_name"])
        with torch.no_grad():
            self.mean_ema_update(code)
            self._train_log(log_vars, out_rgbs, target_rgbs, code)
        return dict(log_vars=log_vars, num_samples=len(data["scene_id"]))


@MODELS.register_module()
class DiffusionNeRF(MultiSceneNeRF):
    def __init__(self, *args, diffusion=dict(type="GaussianDiffusion"), diffusion_use_ema=True, freeze_decoder=True, image_cond=False,
                 code_permute=None, code_reshape=None, autocast_dtype=None, **kwargs):
        super().__init__(*args, **kwargs)
        diffusion = dict(diffusion)
        diffusion.update(train_cfg=self.train_cfg, test_cfg=self.test_cfg)
        self.diffusion = build_module(diffusion)
        self.diffusion_use_ema = diffusion_use_ema
        if self.diffusion_use_ema:
            self.diffusion_ema = deepcopy(self.diffusion)
        self.freeze_decoder = freeze_decoder
        if self.freeze_decoder:
            self.decoder.requires_grad_(False)
            if self.decoder_use_ema:
                self.decoder_ema.requires_grad_(False)
        self.image_cond = image_cond
        self.code_permute = code_permute
        self.code_reshape = code_reshape
        self.code_reshape_inv = [self.code_size[a] for a in self.code_permute] if code_permute is not None else self.code_size
        self.code_permute_inv = [self.code_permute.index(a) for a in range(len(self.code_permute))] if code_permute is not None else None
        self.autocast_dtype = autocast_dtype
        self._backup_override_cfg()     # the diffusion attributes exist only now (diffusion_nerf.py:47-48)

    # (3,6,128,128) <-> (18,128,128) [or the tiled (6,128,384) layout via code_permute]   (diffusion_nerf.py:50-64)
    def code_diff_pr(self, code):
        x = code
        if self.code_permute is not None:
            x = x.permute([0] + [a + 1 for a in self.code_permute])
        if self.code_reshape is not None:
            x = x.reshape(code.size(0), *self.code_reshape)
        return x

    def code_diff_pr_inv(self, code_diff):
        x = code_diff
        if self.code_reshape is not None:
            x = x.reshape(x.size(0), *self.code_reshape_inv)
        if self.code_permute_inv is not None:
            x = x.permute([0] + [a + 1 for a in self.code_permute_inv])
        return x

    def _autocast(self):
        return torch.autocast(device_type="cuda", enabled=self.autocast_dtype is not None,
                              dtype=getattr(torch, self.autocast_dtype) if self.autocast_dtype is not None else None)

    # ---- single-stage training step (diffusion_nerf.py:66-189) ------------------------------------------------------------
    def train_step(self, data, optimizer, running_status=None):
        """One SSDNeRF iteration: diffusion loss on the activated codes -> step the denoiser; its gradient on the codes seeds
        ``extra_scene_step`` rendering iterations (``inverse_code``) and the final joint iteration that also steps the decoder;
        cache write-back.  ``optimizer`` = {'diffusion': ..., ['decoder': ...]} (any key starting with 'diffusion' is stepped after the
        prior loss).  Without ``train_cfg['optimizer']`` the codes are fixed inputs (``data['code']``, stage-2 training)."""
        diffusion = self.diffusion
        decoder = self.decoder_ema if self.freeze_decoder and self.decoder_use_ema else self.decoder
        num_scenes = len(data["scene_id"])
        extra_scene_step = self.train_cfg.get("extra_scene_step", 0)
        if "optimizer" in self.train_cfg:
            code_list_, code_optimizers, density_grid, density_bitfield = self.load_cache(data)
            code = self.code_activation(torch.stack(code_list_, dim=0), update_stats=True)
        else:
            assert "code" in data
            code, density_grid, density_bitfield = self.load_scene(data, load_density="decoder" in optimizer)
            code_list_, code_optimizers = [], []
        for key in optimizer.keys():
            if key.startswith("diffusion"):
                optimizer[key].zero_grad()
        for o in code_optimizers:
            o.zero_grad()
        if "decoder" in optimizer:
            optimizer["decoder"].zero_grad()
        if self.image_cond:
            raise NotImplementedError("image-conditioned UNets (concat_cond) are not part of the north-star configs")
        if "cond_imgs" in data:
            cond_imgs, cond_rays_o, cond_rays_d, dt_gamma = self._cond_rays(data)
        with self._autocast():
            loss_diffusion, log_vars = diffusion(self.code_diff_pr(code), concat_cond=None, return_loss=True,
                                                 x_t_detach=self.train_cfg.get("x_t_detach", False), cfg=self.train_cfg)
        loss_diffusion.backward()
        for key in optimizer.keys():
            if key.startswith("diffusion"):
                optimizer[key].step()
        log_vars = dict(log_vars)
        prior_grad = None
        if extra_scene_step > 0:
            assert len(code_optimizers) > 0
            prior_grad = [c.grad.data.clone() for c in code_list_]
            cfg = dict(self.train_cfg)
            cfg["n_inverse_steps"] = extra_scene_step
            code, _, _, loss_decoder, loss_dict_decoder, out_rgbs, target_rgbs = self.inverse_code(
                decoder, cond_imgs, cond_rays_o, cond_rays_d, dt_gamma=dt_gamma, cfg=cfg, code_=code_list_, density_grid=density_grid,
                density_bitfield=density_bitfield, code_optimizer=code_optimizers, prior_grad=prior_grad)
            log_vars.update({k: v.detach() for k, v in loss_dict_decoder.items()})
        if "decoder" in optimizer or len(code_optimizers) > 0:
            if len(code_optimizers) > 0:
                code = self.code_activation(torch.stack(code_list_, dim=0))
            self.update_extra_state(decoder, code.detach(), density_grid, density_bitfield, 0,
                                    density_thresh=self.train_cfg.get("density_thresh", 0.01))
            loss_decoder, log_vars_decoder, out_rgbs, target_rgbs = self.loss_decoder(decoder, code, density_bitfield, cond_rays_o, cond_rays_d,
                                                                                      cond_imgs, dt_gamma, cfg=self.train_cfg)
            log_vars.update(log_vars_decoder)
            if prior_grad is not None:
                for c, g in zip(code_list_, prior_grad):
                    c.grad.copy_(g)
            loss_decoder.backward()
            if "decoder" in optimizer:
                optimizer["decoder"].step()
            for o in code_optimizers:
                o.step()
            self.save_cache(code_list_, code_optimizers, density_grid, density_bitfield, data["scene_id"], data["scene_name"])
            with torch.no_grad():
                if len(code_optimizers) > 0:
                    self.mean_ema_update(code)
                self._train_log(log_vars, out_rgbs, target_rgbs, code)
            log_vars.update(loss_decoder=loss_decoder.detach())
        return dict(log_vars=log_vars, num_samples=num_scenes)

    # ---- unconditional sampling (diffusion_nerf.py:191-239) -----------------------------------------------------------
    @torch.no_grad()
    def val_uncond(self, data, show_pbar=False, density_jitters=None, **kwargs):
        diffusion = self.diffusion_ema if self.diffusion_use_ema else self.diffusion
        decoder = self.decoder_ema if self.decoder_use_ema else self.decoder
        num_batches = len(data["scene_id"])
        noise = data.get("noise", None)
        if noise is None:
            noise = torch.randn((num_batches, *self.code_size), device=get_module_device(self))
        with self._autocast():
            code_out = diffusion(self.code_diff_pr(noise), return_loss=False, show_pbar=show_pbar, **kwargs)
        if self.test_cfg.get("n_inverse_steps", 0) > 0:
            raise NotImplementedError("post-sampling code optimisation (n_inverse_steps > 0) is not used by the hot-path configs")
        code = self.code_diff_pr_inv(code_out.float())
        density_grid, density_bitfield = self.get_density(decoder, code, cfg=self.test_cfg, jitters=density_jitters)
        return code, density_grid, density_bitfield

    # ---- rendering-guided sampling (diffusion_nerf.py:241-311) --------------------------------------------------------
    def val_guide(self, data, guide_noises=None, density_jitters=None, **kwargs):
        """``guide_noises`` / ``density_jitters`` (extra): per-step injected march jitter (S,R) and grid jitter (H^3,3) lists,
        replacing the reference's in-place ``torch.rand`` draws so that runs are reproducible across devices."""
        device = get_module_device(self)
        diffusion = self.diffusion_ema if self.diffusion_use_ema else self.diffusion
        decoder = self.decoder_ema if self.decoder_use_ema else self.decoder
        cond_imgs, cond_intrinsics, cond_poses = data["cond_imgs"], data["cond_intrinsics"], data["cond_poses"]
        num_scenes, num_imgs, h, w, _ = cond_imgs.size()
        cond_rays_o, cond_rays_d = nerf.get_cam_rays(cond_poses, cond_intrinsics, h, w)
        dt_gamma_scale = self.test_cfg.get("dt_gamma_scale", 0.0)
        dt_gamma = dt_gamma_scale / cond_intrinsics[..., :2].mean(dim=(-2, -1))
        if self.image_cond:
            raise NotImplementedError("image-conditioned UNets (concat_cond) are not part of the north-star configs")
        decoder_training_prev = decoder.training
        decoder.train(True)          # guidance uses the TRAIN branch of the renderer (diffusion_nerf.py:271-272)
        req = [p.requires_grad for p in list(diffusion.parameters()) + list(decoder.parameters())]
        for p in list(diffusion.parameters()) + list(decoder.parameters()):
            p.requires_grad_(False)
        try:
            n_inverse_rays = self.test_cfg.get("n_inverse_rays", 4096)
            raybatch_inds, num_raybatch = self.get_raybatch_inds(cond_imgs, n_inverse_rays)
            density_grid = torch.zeros((num_scenes, self.grid_size ** 3), device=device)
            density_bitfield = torch.zeros((num_scenes, self.grid_size ** 3 // 8), dtype=torch.uint8, device=device)
            step_id = [0]

            def grad_guide_fn(x_0_pred):
                code_pred = self.code_diff_pr_inv(x_0_pred)
                k = step_id[0]
                self.update_extra_state(decoder, code_pred.detach().float(), density_grid, density_bitfield, 0,
                                        density_thresh=self.test_cfg.get("density_thresh", 0.01),
                                        jitter=None if density_jitters is None else density_jitters[k])
                inds = raybatch_inds[k % num_raybatch] if raybatch_inds is not None else None
                rays_o, rays_d, target_rgbs = self.ray_sample(cond_rays_o, cond_rays_d, cond_imgs, n_inverse_rays, sample_inds=inds)
                if guide_noises is not None:
                    decoder.injected_noises = guide_noises[k]
                _, loss, _ = self.loss(decoder, code_pred, density_bitfield, target_rgbs, rays_o, rays_d, dt_gamma,
                                       scale_num_ray=target_rgbs.size(1), cfg=self.test_cfg)
                decoder.injected_noises = None
                step_id[0] += 1
                return loss * num_scenes

            noise = data.get("noise", None)
            if noise is None:
                noise = torch.randn((num_scenes, *self.code_size), device=device)
            with self._autocast():
                code = diffusion(self.code_diff_pr(noise), return_loss=False, grad_guide_fn=grad_guide_fn, **kwargs)
        finally:
            for p, r in zip(list(diffusion.parameters()) + list(decoder.parameters()), req):
                p.requires_grad_(r)
            decoder.train(decoder_training_prev)
        return self.code_diff_pr_inv(code.float()), density_grid, density_bitfield

    # ---- fine-tuning with the diffusion prior (diffusion_nerf.py:313-404) ------------------------------------------------
    def val_optim(self, data, code_=None, density_grid=None, density_bitfield=None, show_pbar=False, prior_timesteps=None, prior_noises=None,
                  march_noises=None, density_jitters=None, **kwargs):
        """``n_inverse_steps`` outer iterations of: diffusion-prior loss of the activated code (one UNet forward + backward at a sampled
        timestep) -> its gradient on ``code_`` seeds ``extra_scene_step + 1`` rendering-loss iterations of ``inverse_code`` that share
        one optimizer/scheduler; with ``extra_scene_step == 0`` a single ``loss_decoder`` backward + step instead.

        Extras for parity runs: ``prior_timesteps`` / ``prior_noises`` (one entry per outer step) and ``march_noises`` /
        ``density_jitters`` (one per inner step / per grid refresh, consumed in order) replace the reference's random draws."""
        device = get_module_device(self)
        decoder = self.decoder_ema if self.decoder_use_ema else self.decoder
        diffusion = self.diffusion_ema if self.diffusion_use_ema else self.diffusion
        cond_imgs, cond_intrinsics, cond_poses = data["cond_imgs"], data["cond_intrinsics"], data["cond_poses"]
        num_scenes, num_imgs, h, w, _ = cond_imgs.size()
        cond_rays_o, cond_rays_d = nerf.get_cam_rays(cond_poses, cond_intrinsics, h, w)
        dt_gamma_scale = self.test_cfg.get("dt_gamma_scale", 0.0)
        dt_gamma = dt_gamma_scale / cond_intrinsics[..., :2].mean(dim=(-2, -1))
        if self.image_cond:
            raise NotImplementedError("image-conditioned UNets (concat_cond) are not part of the north-star configs")
        decoder_training_prev = decoder.training
        decoder.train(True)
        extra_scene_step = self.test_cfg.get("extra_scene_step", 0)
        n_inverse_steps = self.test_cfg.get("n_inverse_steps", 100)
        assert n_inverse_steps > 0
        march_noises = iter(march_noises) if march_noises is not None else None
        density_jitters = iter(density_jitters) if density_jitters is not None else None
        try:
            with _requires_grad(diffusion, False), _requires_grad(decoder, False), torch.enable_grad():
                if code_ is None:
                    code_ = self.get_init_code_(num_scenes, cond_imgs.device)
                if density_grid is None:
                    density_grid = self.get_init_density_grid(num_scenes, cond_imgs.device)
                if density_bitfield is None:
                    density_bitfield = self.get_init_density_bitfield(num_scenes, cond_imgs.device)
                code_optimizer = self.build_optimizer(code_, self.test_cfg)
                code_scheduler = self.build_scheduler(code_optimizer, self.test_cfg)
                inner_cfg = dict(self.test_cfg)
                inner_cfg["n_inverse_steps"] = extra_scene_step + 1
                for inverse_step_id in range(n_inverse_steps):
                    code_optimizer.zero_grad()
                    code = self.code_activation(code_)
                    with self._autocast():
                        loss, log_vars = diffusion(self.code_diff_pr(code), return_loss=True, concat_cond=None,
                                                   x_t_detach=self.test_cfg.get("x_t_detach", False), cfg=self.test_cfg,
                                                   timesteps=None if prior_timesteps is None else prior_timesteps[inverse_step_id],
                                                   noise=None if prior_noises is None else prior_noises[inverse_step_id], **kwargs)
                    loss.backward()
                    if extra_scene_step > 0:
                        prior_grad = code_.grad.data.clone()
                        self.inverse_code(decoder, cond_imgs, cond_rays_o, cond_rays_d, dt_gamma=dt_gamma, cfg=inner_cfg, code_=code_,
                                          density_grid=density_grid, density_bitfield=density_bitfield, code_optimizer=code_optimizer,
                                          code_scheduler=code_scheduler, prior_grad=prior_grad, march_noises=march_noises,
                                          density_jitters=density_jitters)
                    else:        # the prior gradient is still in code_.grad; the rendering gradient accumulates onto it
                        code = self.code_activation(code_)
                        if march_noises is not None:
                            decoder.injected_noises = next(march_noises)
                        try:
                            loss_decoder, _, _, _ = self.loss_decoder(decoder, code, density_bitfield, cond_rays_o, cond_rays_d, cond_imgs,
                                                                      dt_gamma, cfg=self.test_cfg)
                        finally:
                            decoder.injected_noises = None
                        loss_decoder.backward()
                        code_optimizer.step()
                        if code_scheduler is not None:
                            code_scheduler.step()
        finally:
            decoder.train(decoder_training_prev)
        return self.code_activation(code_).detach(), density_grid, density_bitfield

    # ---- dispatch (diffusion_nerf.py:406-469), rendering only ----------------------------------------------------------
    def val_step(self, data, **kwargs):
        decoder = self.decoder_ema if self.decoder_use_ema else self.decoder
        with torch.no_grad():
            if "code" in data:
                code, density_grid, density_bitfield = self.load_scene(data, load_density=True)
            elif "cond_imgs" in data:
                mode = self.test_cfg.get("cond_mode", "guide")
                if mode == "guide":
                    with torch.enable_grad():
                        code, density_grid, density_bitfield = self.val_guide(data, **kwargs)
                elif mode == "optim":
                    code, density_grid, density_bitfield = self.val_optim(data, **kwargs)
                elif mode == "guide_optim":
                    optim_kw = {k: kwargs.pop(k) for k in ("prior_timesteps", "prior_noises", "march_noises") if k in kwargs}
                    with torch.enable_grad():
                        code, density_grid, density_bitfield = self.val_guide(data, **kwargs)
                    kwargs.pop("guide_noises", None), kwargs.pop("density_jitters", None)
                    code, density_grid, density_bitfield = self.val_optim(
                        data, code_=self.code_activation.inverse(code).requires_grad_(True), density_grid=density_grid,
                        density_bitfield=density_bitfield, **optim_kw, **kwargs)
                else:
                    raise AttributeError(f"cond_mode={mode!r}")
            else:
                code, density_grid, density_bitfield = self.val_uncond(data, **kwargs)
            pred_imgs = None
            if "test_poses" in data:
                h, w = self.test_cfg.get("img_size", (128, 128))
                image, depth = self.render(decoder, code, density_bitfield, h, w, data["test_intrinsics"], data["test_poses"], cfg=self.test_cfg)
                pred_imgs = (torch.round(image.clamp(0, 1) * 255) / 255).permute(0, 1, 4, 2, 3)
        return dict(log_vars=dict(), num_samples=code.size(0), pred_imgs=pred_imgs, code=code, density_grid=density_grid,
                    density_bitfield=density_bitfield)
