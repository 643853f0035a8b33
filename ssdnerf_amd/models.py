"""``BaseNeRF`` / ``MultiSceneNeRF`` / ``DiffusionNeRF`` under the reference's registry names and constructor keywords, so that
``configs/paper_cfgs/*.py`` build unchanged (reference: lib/models/autodecoders/{base_nerf,multiscene_nerf,diffusion_nerf}.py).

These classes are the thin, reference-shaped SURFACE of the hot path: the method names,
arguments and return values that callers, configs and checkpoints know.  The work behind them lives elsewhere:

  render / density grid        nerf.py, density.py        (fused HIP launches; rays generated in the kernels)
  code activations, losses     codes.py
  fitting codes to images      fitting.py                 (Conditioning, RayBatcher, CodeFitter, GuidanceObjective)
  DDIM / prior loss            diffusion.py               (SamplingPlan, device-resident sampling loop)
  scene cache wire format      scene_cache.py

What stays out of scope: the runner, hooks, EMA updates, datasets, evaluation and visualisation around ``train_step`` /
``val_step``.
"""
from __future__ import annotations

import math
import os
import warnings
from copy import deepcopy
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import nerf
from .codes import attr_path_get, attr_path_set, frozen
from .codes import IdentityCode, MSELoss, NormalizedTanhCode, RegLoss, TanhCode  # noqa: F401  (registered here for config builds)
from .density import get_density as _get_density, update_density_grid
from .fitting import CodeFitter, Conditioning, GuidanceObjective, RayBatcher
from .registry import MODELS, build_module, get_module_device


def _torch_factory(namespace, cfg: Dict):
    """``dict(type='Adam', lr=...)`` -> (torch class, kwargs)"""
    kw = dict(cfg)
    return getattr(namespace, kw.pop("type")), kw


class BaseNeRF(nn.Module):
    def __init__(self, code_size=(3, 8, 64, 64), code_activation=dict(type="TanhCode", scale=1), grid_size=64,
                 decoder=dict(type="TriPlaneDecoder"), decoder_use_ema=False, bg_color=1, pixel_loss=dict(type="MSELoss"), reg_loss=None,
                 update_extra_interval=16, use_lpips_metric=True, init_from_mean=False, init_scale=1e-4, mean_ema_momentum=0.001,
                 mean_scale=1.0, train_cfg=dict(), test_cfg=dict(), pretrained=None):
        super().__init__()
        self.code_size, self.grid_size, self.bg_color = tuple(code_size), grid_size, bg_color
        self.code_activation = build_module(code_activation)
        self.decoder = build_module(decoder)
        self.decoder_use_ema = decoder_use_ema
        if decoder_use_ema:
            self.decoder_ema = deepcopy(self.decoder)
        self.pixel_loss = build_module(pixel_loss)
        self.reg_loss = None if reg_loss is None else build_module(reg_loss)
        self.train_cfg, self.test_cfg = dict(train_cfg or {}), dict(test_cfg or {})
        self.update_extra_interval = update_extra_interval
        self.init_scale, self.mean_ema_momentum, self.mean_scale = init_scale, mean_ema_momentum, mean_scale
        self.init_code = None
        if init_from_mean:
            self.register_buffer("init_code", torch.zeros(self.code_size))
        if pretrained is not None and os.path.isfile(pretrained):
            ckpt = torch.load(pretrained, map_location="cpu")
            self.load_state_dict(ckpt.get("state_dict", ckpt), strict=False)
        self.train_cfg_backup: Dict = {}
        self._remember_overridden()

    # ---- eval-mode attribute overrides: ``test_cfg['override_cfg']`` maps dotted attribute paths to the value they take while the model is
    # in eval mode (the recons configs turn ``diffusion_ema.ddpm_loss.weight_scale`` down to 1); ``train()`` puts the training values back
    def _remember_overridden(self):
        for path in self.test_cfg.get("override_cfg", {}):
            self.train_cfg_backup[path] = attr_path_get(self, path, None)

    def train(self, mode=True):
        overrides = self.test_cfg.get("override_cfg", {})
        if mode:
            for path, value in self.train_cfg_backup.items():
                attr_path_set(self, path, value)
        else:
            for path, value in overrides.items():
                if self.training:                                     # leaving training mode: the current value is the one to come back to
                    self.train_cfg_backup[path] = attr_path_get(self, path)
                attr_path_set(self, path, value)
        return super().train(mode)

    def _modules_for_eval(self):
        return self.decoder_ema if self.decoder_use_ema else self.decoder

    # ---- scene files: {scene_name, param: {code | code_, density_grid (Morton fp16), density_bitfield (u8)}} -----------------------------
    def load_scene(self, data, load_density=False):
        dev = get_module_device(self)
        params = [entry["param"] for entry in data["code"]]
        code = torch.stack([p["code"] if "code" in p else self.code_activation(p["code_"]) for p in params], dim=0).to(dev)
        if not load_density:
            return code, None, None
        return (code, torch.stack([p["density_grid"] for p in params], dim=0).to(dev),
                torch.stack([p["density_bitfield"] for p in params], dim=0).to(dev))

    @staticmethod
    def save_scene(save_dir, code, density_grid, density_bitfield, scene_name):
        os.makedirs(save_dir, exist_ok=True)
        for i, name in enumerate(scene_name):
            param = dict(code=code.data[i].cpu(), density_grid=density_grid.data[i].cpu(), density_bitfield=density_bitfield.data[i].cpu())
            torch.save(dict(scene_name=name, param=param), os.path.join(save_dir, name) + ".pth")

    # ---- fresh per-scene state ------------------------------------------------------------------------------------------------------------
    def get_init_code_(self, num_scenes, device=None):
        """A pre-activation code leaf, (code_size) or (S, code_size): uniform in +-init_scale, or the inverse-activated running mean code."""
        shape = self.code_size if num_scenes is None else (num_scenes, *self.code_size)
        leaf = torch.empty(shape, device=device, dtype=torch.float32, requires_grad=True)
        with torch.no_grad():
            if self.init_code is None:
                leaf.uniform_(-self.init_scale, self.init_scale)
            else:
                leaf.copy_(self.code_activation.inverse(self.init_code * self.mean_scale))
        return leaf

    def get_init_density_grid(self, num_scenes, device=None):
        n = self.grid_size ** 3
        return torch.zeros(n if num_scenes is None else (num_scenes, n), device=device, dtype=torch.float16)

    def get_init_density_bitfield(self, num_scenes, device=None):
        n = self.grid_size ** 3 // 8
        return torch.zeros(n if num_scenes is None else (num_scenes, n), device=device, dtype=torch.uint8)

    @staticmethod
    def build_optimizer(code_, cfg):
        """one torch optimizer over the leaf, or one PER LEAF for a list of per-scene leaves (their states are cached per scene)"""
        cls, kw = _torch_factory(torch.optim, cfg["optimizer"])
        return [cls([leaf], **kw) for leaf in code_] if isinstance(code_, list) else cls([code_], **kw)

    @staticmethod
    def build_scheduler(code_optimizer, cfg):
        if "lr_scheduler" not in cfg:
            return None
        cls, kw = _torch_factory(torch.optim.lr_scheduler, cfg["lr_scheduler"])
        return [cls(opt, **kw) for opt in code_optimizer] if isinstance(code_optimizer, list) else cls(code_optimizer, **kw)

    # ---- density grid -----------------------------------------------------------------------------------------------------------------------
    def update_extra_state(self, decoder, code, density_grid, density_bitfield, iter_density, density_thresh=0.01, decay=0.9, S=128,
                           jitter=None):
        """Full refresh of the occupancy state from the codes, in place: two HIP launches (density.py).  The reference's partial-update branch
        (``iter_density >= 16``) is unreachable from the hot-path configs, which always pass 0."""
        if iter_density >= 16:
            raise NotImplementedError("partial density-grid updates (iter_density >= 16) are not reachable from the hot-path configs")
        with torch.no_grad():
            update_density_grid(decoder, code, density_grid, density_bitfield, density_thresh=density_thresh, decay=decay, jitter=jitter,
                                return_thresh=False)

    def get_density(self, decoder, code, cfg=dict(), jitters=None):
        return _get_density(decoder, code, self.grid_size, density_thresh=cfg.get("density_thresh", 0.01),
                            density_step=cfg.get("density_step", 8), jitters=jitters)

    # ---- rendering loss -------------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def ray_sample(cond_rays_o, cond_rays_d, cond_imgs, n_samples, sample_inds=None):
        """(S, n_samples, 3) rays and target colours out of the (S, V, h, w, 3) arrays: the given pixel indices, or a fresh random subset."""
        return RayBatcher(Conditioning(cond_imgs, cond_rays_o, cond_rays_d, None), n_samples, fixed=False).take(sample_inds)

    @staticmethod
    def get_raybatch_inds(cond_imgs, n_inverse_rays):
        chunks = RayBatcher(Conditioning(cond_imgs, cond_imgs, cond_imgs, None), n_inverse_rays, fixed=True).chunks
        return (None, None) if chunks is None else (chunks, len(chunks))

    def loss(self, decoder, code, density_bitfield, target_rgbs, rays_o, rays_d, dt_gamma=0.0, return_decoder_loss=False,
             scale_num_ray=1.0, cfg=dict(), perturb=True, **kwargs):
        """Render the rays and compare: ``pixel_loss(rgb, target) * 3 * (1 - exp(-loss_coef * scale_num_ray))`` (+ code regulariser)
        (+ the decoder's own regulariser).  Returns (rgb blended with the background, total, parts)."""
        out = decoder(rays_o, rays_d, code, density_bitfield, self.grid_size, dt_gamma=dt_gamma, perturb=perturb, return_loss=return_decoder_loss)
        opacity = out["weights_sum"]
        rgb = out["image"] + self.bg_color * (1 - opacity.unsqueeze(-1))
        ramp = 1 - math.exp(-cfg["loss_coef"] * scale_num_ray) if "loss_coef" in cfg else 1
        parts = dict(pixel_loss=self.pixel_loss(rgb, target_rgbs, **kwargs) * (ramp * 3))
        if self.reg_loss is not None:
            parts["reg_loss"] = self.reg_loss(code, **kwargs)
        if return_decoder_loss and out.get("decoder_reg_loss") is not None:
            parts["decoder_reg_loss"] = out["decoder_reg_loss"]
        total = sum(parts.values())
        return rgb, total, parts

    def loss_decoder(self, decoder, code, density_bitfield, cond_rays_o, cond_rays_d, cond_imgs, dt_gamma=0.0, cfg=dict(), **kwargs):
        """Rendering loss on ``n_decoder_rays`` freshly drawn rays through the TRAIN branch, decoder regulariser included; the logged parts are
        detached 0-dim tensors (the reference turns each into a Python float: one device sync per entry)."""
        was_training = decoder.training
        decoder.train(True)
        try:
            rays_o, rays_d, target = self.ray_sample(cond_rays_o, cond_rays_d, cond_imgs, n_samples=cfg.get("n_decoder_rays", 4096))
            rgb, total, parts = self.loss(decoder, code, density_bitfield, target, rays_o, rays_d, dt_gamma, return_decoder_loss=True,
                                          scale_num_ray=cond_rays_o.shape[1:4].numel(), cfg=cfg, **kwargs)
        finally:
            decoder.train(was_training)
        return total, {k: v.detach() for k, v in parts.items()}, rgb, target

    # ---- inversion -----------------------------------------------------------------------------------------------------------------------------
    def inverse_code(self, decoder, cond_imgs, cond_rays_o, cond_rays_d, dt_gamma=0, cfg=dict(), code_=None, density_grid=None,
                     density_bitfield=None, iter_density=None, code_optimizer=None, code_scheduler=None, prior_grad=None, show_pbar=False,
                     march_noises=None, density_jitters=None):
        """``cfg['n_inverse_steps']`` rendering-loss iterations on ``code_`` in place (``fitting.CodeFitter``), the decoder frozen; missing state
        (codes, grid, bitfield, optimizer, scheduler) is created.  ``prior_grad`` seeds every iteration's gradient.  ``march_noises`` /
        ``density_jitters`` (extra): injected draws, consumed in order.
        -> (activated code, density_grid, density_bitfield, last loss, its parts, last rendered rgb, its targets)"""
        dev = get_module_device(self)
        S = cond_imgs.size(0)
        code_ = self.get_init_code_(S, device=dev) if code_ is None else code_
        density_grid = self.get_init_density_grid(S, dev) if density_grid is None else density_grid
        density_bitfield = self.get_init_density_bitfield(S, dev) if density_bitfield is None else density_bitfield
        if code_optimizer is None:
            assert code_scheduler is None
            code_optimizer = self.build_optimizer(code_, cfg)
        if code_scheduler is None:
            code_scheduler = self.build_scheduler(code_optimizer, cfg)
        cond = Conditioning(cond_imgs, cond_rays_o, cond_rays_d, dt_gamma)
        with frozen(decoder):
            fit = CodeFitter(self, decoder, cond, cfg, code_, density_grid, density_bitfield, code_optimizer, code_scheduler,
                             march_noises=march_noises, density_jitters=density_jitters)
            code, loss, parts, rgb, target = fit.run(cfg.get("n_inverse_steps", 1000), seed_grad=prior_grad)
        return code, density_grid, density_bitfield, loss, parts, rgb, target

    # ---- render ------------------------------------------------------------------------------------------------------------------------------------
    def render(self, decoder, code, density_bitfield, h, w, intrinsics, poses, cfg=dict()):
        return nerf.render(decoder, code, density_bitfield, h, w, intrinsics, poses, grid_size=self.grid_size, bg_color=self.bg_color, cfg=cfg)

    def mean_ema_update(self, code):
        """running mean code of ``init_from_mean=True`` models"""
        if self.init_code is not None:
            from .parallel import reduce_mean
            self.init_code.mul_(1 - self.mean_ema_momentum).add_(reduce_mean(code.detach().mean(dim=0)).data, alpha=self.mean_ema_momentum)

    def train_step(self, data, optimizer, running_status=None):
        raise NotImplementedError("BaseNeRF has no training step of its own; MultiSceneNeRF / DiffusionNeRF do")

    @staticmethod
    def _log_fit(log_vars, rgb, target, code):
        log_vars.update(train_psnr=nerf.eval_psnr(rgb.detach(), target).mean(), code_rms=code.detach().square().flatten(1).mean().sqrt().mean())


@MODELS.register_module()
class MultiSceneNeRF(BaseNeRF):
    """+ the per-scene cache of pre-activation codes and optimizer states (wire format and 16-bit casting rules: ``scene_cache.py``).  The RAM
    cache is sharded over ranks with the ``round(linspace)`` split of the scene sampler, so a rank only holds the scenes it is handed."""

    def __init__(self, *args, cache_size=0, cache_16bit=False, num_file_writers=0, **kwargs):
        super().__init__(*args, **kwargs)
        self.cache_size, self.cache_16bit, self.num_file_writers = cache_size, cache_16bit, num_file_writers
        self.cache, self.cache_loaded, self.file_writers = None, False, None
        if cache_size > 0:
            import torch.distributed as dist
            from .parallel import shard_scenes
            rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
            self.cache = dict.fromkeys(shard_scenes(cache_size, rank, world))

    def _cached_entries(self, data) -> List[Optional[Dict]]:
        if self.cache is None:
            return list(data["code"]) if "code" in data else [None] * len(data["scene_id"])
        if not self.cache_loaded:
            src = self.train_cfg.get("cache_load_from", None)
            files = sorted(os.listdir(src)) if src is not None else []
            if files:
                assert len(files) == self.cache_size
                for sid in self.cache:
                    self.cache[sid] = torch.load(os.path.join(src, files[sid]), map_location="cpu")
            self.cache_loaded = True
        return [self.cache[int(sid)] for sid in data["scene_id"]]

    def load_cache(self, data):
        """-> (per-scene pre-activation code leaves, their optimizers, density_grid (S, H^3), density_bitfield (S, H^3/8)) for
        ``data['scene_id']``: RAM cache (filled once from ``train_cfg['cache_load_from']``) > ``data['code']`` > fresh."""
        from .scene_cache import optimizer_set_state
        dev = get_module_device(self)
        entries = self._cached_entries(data)
        leaves, grids, bits = [], [], []
        for entry in entries:
            if entry is None:
                leaves.append(self.get_init_code_(None, dev))
                grids.append(self.get_init_density_grid(None, dev))
                bits.append(self.get_init_density_bitfield(None, dev))
                continue
            p = entry["param"]
            if "code_" in p:
                leaf = p["code_"].to(dtype=torch.float32, device=dev)
            else:             # a test-time scene file holds the ACTIVATED code only
                warnings.warn("Pre-activation codes not found. Using on-the-fly inversion instead (which could be inconsistent).")
                leaf = self.code_activation.inverse(p["code"].to(dtype=torch.float32, device=dev))
            leaves.append(leaf.requires_grad_(True))
            grids.append(p["density_grid"].to(dev))
            bits.append(p["density_bitfield"].to(dev))
        optimizers = self.build_optimizer(leaves, self.train_cfg)
        for opt, entry in zip(optimizers, entries):
            if entry is not None and "optimizer" in entry:
                optimizer_set_state(opt, entry["optimizer"])
        return leaves, optimizers, torch.stack(grids, dim=0), torch.stack(bits, dim=0)

    def save_cache(self, code_list_, code_optimizers, density_grid, density_bitfield, scene_id, scene_name):
        """Scenes back into the RAM cache (in place where an entry exists) and, with ``train_cfg['save_dir']``, to ``<save_dir>/<name>.pth``
        (fp16 codes + bf16 optimizer moments under ``cache_16bit``).  Tensors handed to the writer threads are private copies."""
        from .scene_cache import _FileWriters, load_tensor_to_dict, optimizer_state_copy, optimizer_state_to, out_dict_to
        code_dtype, state_dtype = code_list_[0].dtype, torch.float32
        if self.cache_16bit:
            code_dtype, state_dtype = (torch.float16 if code_dtype == torch.float32 else code_dtype), torch.bfloat16
        save_dir = self.train_cfg.get("save_dir", None)
        if save_dir is not None:
            os.makedirs(save_dir, exist_ok=True)
            if self.num_file_writers > 0 and self.file_writers is None:
                self.file_writers = _FileWriters(save_dir, self.num_file_writers)
        cast = dict(device="cpu", code_dtype=code_dtype, optimizer_dtype=state_dtype)
        for i, leaf in enumerate(code_list_):
            sid = int(scene_id[i])
            fresh = dict(scene_id=scene_id[i], scene_name=scene_name[i], optimizer=code_optimizers[i].state_dict(),
                         param=dict(code_=leaf.data, density_grid=density_grid[i], density_bitfield=density_bitfield[i]))
            if self.cache is not None:
                held = self.cache[sid]
                if held is None:
                    self.cache[sid] = out_dict_to(fresh, **cast)
                else:
                    held.setdefault("scene_id", fresh["scene_id"])
                    held.setdefault("scene_name", fresh["scene_name"])
                    held["param"].pop("code", None)
                    for key, val in fresh["param"].items():
                        load_tensor_to_dict(held["param"], key, val, device="cpu", dtype=code_dtype)
                    if "optimizer" in held:
                        optimizer_state_copy(fresh["optimizer"], held["optimizer"], device="cpu", dtype=state_dtype)
                    else:
                        held["optimizer"] = optimizer_state_to(fresh["optimizer"], device="cpu", dtype=state_dtype)
            if save_dir is not None:
                record = out_dict_to(fresh, **cast)
                if self.file_writers is not None:
                    # out_dict_to returns the SAME tensor when neither dtype nor device change: clone, or the writer thread races the next step
                    record["param"] = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in record["param"].items()}
                    self.file_writers.put(i, record)
                else:
                    torch.save(record, os.path.join(save_dir, scene_name[i] + ".pth"))

    def _conditioning(self, data, cfg) -> Conditioning:
        return Conditioning.from_batch(data, cfg.get("dt_gamma_scale", 0.0))

    def _joint_step(self, decoder, leaves, code_optimizers, density_grid, density_bitfield, cond, optimizer, log_vars, seed_grads=None, data=None,
                    code=None):
        """the closing iteration of a training step: refresh the grid, rendering loss on ``n_decoder_rays`` rays, backward (on top of the
        seed gradients), step decoder and codes, write the cache back, log"""
        cfg = self.train_cfg
        self.update_extra_state(decoder, code.detach(), density_grid, density_bitfield, 0, density_thresh=cfg.get("density_thresh", 0.01))
        loss, parts, rgb, target = self.loss_decoder(decoder, code, density_bitfield, cond.rays_o, cond.rays_d, cond.images, cond.dt_gamma, cfg=cfg)
        log_vars.update(parts)
        if seed_grads is not None:
            for leaf, g in zip(leaves, seed_grads):
                leaf.grad.copy_(g)
        loss.backward()
        if "decoder" in optimizer:
            optimizer["decoder"].step()
        for opt in code_optimizers:
            opt.step()
        self.save_cache(leaves, code_optimizers, density_grid, density_bitfield, data["scene_id"], data["scene_name"])
        with torch.no_grad():
            if code_optimizers:
                self.mean_ema_update(code)
            self._log_fit(log_vars, rgb, target, code)
        return loss

    def train_step(self, data, optimizer, running_status=None):
        """Auto-decoder step: ``extra_scene_step`` code-only fitting iterations, then one joint iteration of codes + decoder; cache write-back.
        ``optimizer`` = {'decoder': torch optimizer}; the per-scene code optimizers come from the cache.  Log values are 0-dim tensors."""
        cfg = self.train_cfg
        leaves, code_optimizers, grid, bits = self.load_cache(data)
        cond = self._conditioning(data, cfg)
        extra = cfg.get("extra_scene_step", 0)
        if extra > 0:
            self.inverse_code(self.decoder, cond.images, cond.rays_o, cond.rays_d, dt_gamma=cond.dt_gamma, cfg=dict(cfg, n_inverse_steps=extra),
                              code_=leaves, density_grid=grid, density_bitfield=bits, code_optimizer=code_optimizers)
        for opt in (*code_optimizers, optimizer["decoder"]):
            opt.zero_grad()
        code = self.code_activation(torch.stack(leaves, dim=0), update_stats=True)
        log_vars: Dict = {}
        loss = self._joint_step(self.decoder, leaves, code_optimizers, grid, bits, cond, optimizer, log_vars, data=data, code=code)
        log_vars.update(loss=loss.detach())
        return dict(log_vars=log_vars, num_samples=len(data["scene_id"]))


@MODELS.register_module()
class DiffusionNeRF(MultiSceneNeRF):
    def __init__(self, *args, diffusion=dict(type="GaussianDiffusion"), diffusion_use_ema=True, freeze_decoder=True, image_cond=False,
                 code_permute=None, code_reshape=None, autocast_dtype=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.diffusion = build_module(dict(diffusion, train_cfg=self.train_cfg, test_cfg=self.test_cfg))
        self.diffusion_use_ema = diffusion_use_ema
        if diffusion_use_ema:
            self.diffusion_ema = deepcopy(self.diffusion)
        self.freeze_decoder = freeze_decoder
        if freeze_decoder:
            self.decoder.requires_grad_(False)
            if self.decoder_use_ema:
                self.decoder_ema.requires_grad_(False)
        self.image_cond, self.autocast_dtype = image_cond, autocast_dtype
        # latent layout seen by the UNet: optional axis permutation of (planes, channels, h, w), then a reshape -- (3,6,128,128) -> (18,128,128),
        # or the tiled layout (1,2,0,3) -> (6,128,384)
        self.code_permute, self.code_reshape = code_permute, code_reshape
        self.code_reshape_inv = self.code_size if code_permute is None else [self.code_size[a] for a in code_permute]
        self.code_permute_inv = None if code_permute is None else [code_permute.index(a) for a in range(len(code_permute))]
        self._remember_overridden()         # (paths into the diffusion modules exist only now)

    def code_diff_pr(self, code):
        x = code if self.code_permute is None else code.permute(0, *(a + 1 for a in self.code_permute))
        return x if self.code_reshape is None else x.reshape(code.size(0), *self.code_reshape)

    def code_diff_pr_inv(self, code_diff):
        x = code_diff if self.code_reshape is None else code_diff.reshape(code_diff.size(0), *self.code_reshape_inv)
        return x if self.code_permute_inv is None else x.permute(0, *(a + 1 for a in self.code_permute_inv))

    def _autocast(self):
        on = self.autocast_dtype is not None
        return torch.autocast(device_type="cuda", enabled=on, dtype=getattr(torch, self.autocast_dtype) if on else None)

    def _no_image_cond(self):
        if self.image_cond:
            raise NotImplementedError("image-conditioned UNets (concat_cond) are not part of the north-star configs")

    def _eval_diffusion(self):
        return self.diffusion_ema if self.diffusion_use_ema else self.diffusion

    def _start_noise(self, data, num_scenes, device):
        noise = data.get("noise", None)
        return torch.randn((num_scenes, *self.code_size), device=device) if noise is None else noise

    # ---- single-stage training step ----------------------------------------------------------------------------------------------------------------
    def train_step(self, data, optimizer, running_status=None):
        """One SSDNeRF iteration.  Diffusion loss on the activated codes -> step the denoiser(s) (every ``optimizer`` key that starts with
        'diffusion').  The gradient that loss left on the codes then SEEDS ``extra_scene_step`` rendering iterations and the closing joint
        iteration that also steps the decoder; cache write-back.  Without ``train_cfg['optimizer']`` the codes are fixed inputs
        (``data['code']``; second-stage training of the prior alone)."""
        cfg = self.train_cfg
        self._no_image_cond()
        decoder = self.decoder_ema if self.freeze_decoder and self.decoder_use_ema else self.decoder
        fit_codes = "optimizer" in cfg
        if fit_codes:
            leaves, code_optimizers, grid, bits = self.load_cache(data)
            code = self.code_activation(torch.stack(leaves, dim=0), update_stats=True)
        else:
            assert "code" in data
            code, grid, bits = self.load_scene(data, load_density="decoder" in optimizer)
            leaves, code_optimizers = [], []
        prior_opts = [opt for key, opt in optimizer.items() if key.startswith("diffusion")]
        for opt in (*prior_opts, *code_optimizers, *([optimizer["decoder"]] if "decoder" in optimizer else [])):
            opt.zero_grad()
        cond = self._conditioning(data, cfg) if "cond_imgs" in data else None
        with self._autocast():
            loss_prior, log_vars = self.diffusion(self.code_diff_pr(code), concat_cond=None, return_loss=True, x_t_detach=cfg.get("x_t_detach", False),
                                                  cfg=cfg)
        loss_prior.backward()
        for opt in prior_opts:
            opt.step()
        log_vars = dict(log_vars)
        seeds = None
        extra = cfg.get("extra_scene_step", 0)
        if extra > 0:
            assert code_optimizers
            seeds = [leaf.grad.detach().clone() for leaf in leaves]
            _, _, _, _, parts, _, _ = self.inverse_code(decoder, cond.images, cond.rays_o, cond.rays_d, dt_gamma=cond.dt_gamma,
                                                        cfg=dict(cfg, n_inverse_steps=extra), code_=leaves, density_grid=grid, density_bitfield=bits,
                                                        code_optimizer=code_optimizers, prior_grad=seeds)
            log_vars.update({k: v.detach() for k, v in parts.items()})
        if "decoder" in optimizer or code_optimizers:
            if code_optimizers:
                code = self.code_activation(torch.stack(leaves, dim=0))
            loss_fit = self._joint_step(decoder, leaves, code_optimizers, grid, bits, cond, optimizer, log_vars, seed_grads=seeds, data=data, code=code)
            log_vars.update(loss_decoder=loss_fit.detach())
        return dict(log_vars=log_vars, num_samples=len(data["scene_id"]))

    # ---- unconditional sampling -----------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def val_uncond(self, data, show_pbar=False, density_jitters=None, **kwargs):
        """noise -> DDIM over the triplane latents -> scene codes -> their occupancy state (``density_step`` grid refreshes)"""
        prior_timesteps, prior_noises = kwargs.pop("prior_timesteps", None), kwargs.pop("prior_noises", None)
        noise = self._start_noise(data, len(data["scene_id"]), get_module_device(self))
        diffusion = self._eval_diffusion()
        with self._autocast():
            latent = diffusion(self.code_diff_pr(noise), return_loss=False, show_pbar=show_pbar, **kwargs)
        points = latent if isinstance(latent, list) else [latent]          # save_intermediates: the sampler returns its trajectory
        n_refine = self.test_cfg.get("n_inverse_steps", 0)
        codes, grids, bitfields = [], [], []
        for i, point in enumerate(points):
            code = self.code_diff_pr_inv(point.float())
            if n_refine > 0 and i == len(points) - 1:                        # only the final sample is polished under the prior (diffusion_nerf.py:213-231)
                code = self._refine_under_prior(diffusion, code, n_refine, prior_timesteps, prior_noises)
            grid, bits = self.get_density(self._modules_for_eval(), code, cfg=self.test_cfg, jitters=density_jitters)
            codes.append(code); grids.append(grid); bitfields.append(bits)
        if isinstance(latent, list):                                         # one (code, grid, bitfield) per trajectory point, like the reference (:236-237)
            return codes, grids, bitfields
        return codes[-1], grids[-1], bitfields[-1]

    def _refine_under_prior(self, diffusion, code, n_steps, timesteps=None, noises=None):
        """``test_cfg['n_inverse_steps']`` on a batch WITHOUT conditioning views (every recons config sets it, and the reference's val_uncond then
        polishes the sampled codes under the prior alone, lib/models/autodecoders/diffusion_nerf.py:212-229): n optimizer steps on the
        pre-activation code against the diffusion loss, denoiser frozen.  ``timesteps`` / ``noises``: injected draws per step (parity runs)."""
        cfg = self.test_cfg
        with frozen(diffusion), torch.enable_grad():
            leaf = self.code_activation.inverse(code).detach().requires_grad_(True)
            opt = self.build_optimizer(leaf, cfg)
            sch = self.build_scheduler(opt, cfg)
            for k in range(n_steps):
                opt.zero_grad()
                # fp32 here: the reference wraps only the SAMPLING of val_uncond in autocast and evaluates this loss outside it
                # (diffusion_nerf.py:212-229; r03 advisor)
                prior, _ = diffusion(self.code_diff_pr(self.code_activation(leaf)), return_loss=True, cfg=cfg,
                                     timesteps=None if timesteps is None else timesteps[k], noise=None if noises is None else noises[k])
                prior.backward()
                opt.step()
                if sch is not None:
                    sch.step()
        return self.code_activation(leaf).detach()

    # ---- rendering-guided sampling ------------------------------------------------------------------------------------------------------------------
    def val_guide(self, data, guide_noises=None, density_jitters=None, **kwargs):
        """DDIM in which every step's x0 is pulled towards the conditioning views by the gradient of a rendering loss
        (``fitting.GuidanceObjective``).  ``guide_noises`` / ``density_jitters`` (extra): per-step injected march jitter (S, R) and grid jitter
        (H^3, 3) lists replacing the reference's on-device draws."""
        self._no_image_cond()
        diffusion, decoder = self._eval_diffusion(), self._modules_for_eval()
        cond = self._conditioning(data, self.test_cfg)
        was_training = decoder.training
        decoder.train(True)                        # guidance renders through the TRAIN branch
        try:
            with frozen(diffusion, decoder):
                objective = GuidanceObjective(self, decoder, cond, self.test_cfg, march_noises=guide_noises, density_jitters=density_jitters)
                noise = self._start_noise(data, cond.num_scenes, get_module_device(self))
                with self._autocast():
                    latent = diffusion(self.code_diff_pr(noise), return_loss=False, grad_guide_fn=objective, **kwargs)
        finally:
            decoder.train(was_training)
        return self.code_diff_pr_inv(latent.float()), objective.grid, objective.bits

    # ---- fine-tuning under the diffusion prior ------------------------------------------------------------------------------------------------------------
    def val_optim(self, data, code_=None, density_grid=None, density_bitfield=None, show_pbar=False, prior_timesteps=None, prior_noises=None,
                  march_noises=None, density_jitters=None, **kwargs):
        """``n_inverse_steps`` outer iterations of { prior loss of the activated code at a sampled timestep (UNet forward + backward) -> its
        gradient on ``code_`` seeds ``extra_scene_step + 1`` rendering-loss iterations }, one optimizer and one LR schedule across all of them;
        with ``extra_scene_step == 0`` the rendering gradient of one ``loss_decoder`` is added to the prior gradient and a single step is taken.
        Extras for parity runs: ``prior_timesteps`` / ``prior_noises`` (per outer step), ``march_noises`` / ``density_jitters`` (per inner step /
        per grid refresh, consumed in order)."""
        self._no_image_cond()
        cfg = self.test_cfg
        diffusion, decoder = self._eval_diffusion(), self._modules_for_eval()
        cond = self._conditioning(data, cfg)
        dev, S = cond.images.device, cond.num_scenes
        extra, n_outer = cfg.get("extra_scene_step", 0), cfg.get("n_inverse_steps", 100)
        assert n_outer > 0
        march_noises = None if march_noises is None else iter(march_noises)
        density_jitters = None if density_jitters is None else iter(density_jitters)
        was_training = decoder.training
        decoder.train(True)
        try:
            with frozen(diffusion, decoder), torch.enable_grad():
                code_ = self.get_init_code_(S, dev) if code_ is None else code_
                density_grid = self.get_init_density_grid(S, dev) if density_grid is None else density_grid
                density_bitfield = self.get_init_density_bitfield(S, dev) if density_bitfield is None else density_bitfield
                opt = self.build_optimizer(code_, cfg)
                sch = self.build_scheduler(opt, cfg)
                inner_cfg = dict(cfg, n_inverse_steps=extra + 1)
                from .diffusion import _host_noise
                next_noise, rng_after_prefetch, prefetch_ok = None, None, True
                for k in range(n_outer):
                    opt.zero_grad()
                    # (r04 advisor) the prefetch keeps the reference's order of host draws only while nothing else draws from torch's CPU generator between the
                    # prefetch and this point -- true of this library (march jitter: device draws; timesteps: numpy), not of an arbitrary hook or decoder: the
                    # generator's state is compared, and a foreign draw switches the prefetch off for the rest of the loop, with a warning
                    if rng_after_prefetch is not None and not torch.equal(torch.get_rng_state(), rng_after_prefetch):
                        import warnings
                        warnings.warn("val_optim: something drew from torch's CPU generator between the prior-loss noise of this iteration (drawn one iteration "
                                      "early, under queued device work) and its use: the seeded sequence differs from drawing it here.  Prefetch switched off.")
                        prefetch_ok = False
                    rng_after_prefetch = None
                    # the prior loss's noise is a HOST draw (the reference's, seed-reproducible on any device; 4.8 ms for 8 cars latents): iteration k + 1's
                    # is drawn right behind iteration k's UNet launches, while the device works through them -- same generator, same order of draws (nothing
                    # else in this loop draws on the host)
                    noise_k, next_noise = (prior_noises[k], None) if prior_noises is not None else (next_noise, None)
                    with self._autocast():
                        x0_in = self.code_diff_pr(self.code_activation(code_))
                        prior, _ = diffusion(x0_in, return_loss=True, concat_cond=None, x_t_detach=cfg.get("x_t_detach", False), cfg=cfg,
                                             timesteps=None if prior_timesteps is None else prior_timesteps[k], noise=noise_k, **kwargs)
                    prior.backward()
                    if prior_noises is None and k + 1 < n_outer and prefetch_ok:
                        next_noise = _host_noise(x0_in.detach())
                        rng_after_prefetch = torch.get_rng_state()
                    if extra > 0:
                        self.inverse_code(decoder, cond.images, cond.rays_o, cond.rays_d, dt_gamma=cond.dt_gamma, cfg=inner_cfg, code_=code_,
                                          density_grid=density_grid, density_bitfield=density_bitfield, code_optimizer=opt, code_scheduler=sch,
                                          prior_grad=code_.grad.detach().clone(), march_noises=march_noises, density_jitters=density_jitters)
                        continue
                    # no inner loop: the prior gradient is still in code_.grad and the rendering gradient accumulates onto it
                    if march_noises is not None:
                        decoder.injected_noises = next(march_noises)
                    try:
                        fit_loss, _, _, _ = self.loss_decoder(decoder, self.code_activation(code_), density_bitfield, cond.rays_o, cond.rays_d,
                                                              cond.images, cond.dt_gamma, cfg=cfg)
                    finally:
                        decoder.injected_noises = None
                    fit_loss.backward()
                    opt.step()
                    if sch is not None:
                        sch.step()
        finally:
            decoder.train(was_training)
        return self.code_activation(code_).detach(), density_grid, density_bitfield

    # ---- dispatch ------------------------------------------------------------------------------------------------------------------------------------------
    def _scene_from(self, data, kwargs):
        """codes + occupancy for a validation batch: cached scenes, or reconstruction from ``cond_imgs`` per ``test_cfg['cond_mode']``
        ('guide' | 'optim' | 'guide_optim'), or unconditional sampling"""
        if "code" in data:
            return self.load_scene(data, load_density=True)
        if "cond_imgs" not in data:
            return self.val_uncond(data, **kwargs)
        mode = self.test_cfg.get("cond_mode", "guide")
        # injected draws are routed by the half that consumes them: guide_* to val_guide, optim_* / prior_* / march_noises to val_optim
        guide_kw = {k: kwargs.pop(k) for k in ("guide_noises", "density_jitters") if k in kwargs}
        optim_kw = {k: kwargs.pop(k) for k in ("prior_timesteps", "prior_noises", "march_noises") if k in kwargs}
        if "optim_density_jitters" in kwargs:
            optim_kw["density_jitters"] = kwargs.pop("optim_density_jitters")
        if "guide_density_jitters" in kwargs:
            guide_kw["density_jitters"] = kwargs.pop("guide_density_jitters")
        if mode == "guide":
            with torch.enable_grad():
                return self.val_guide(data, **guide_kw, **kwargs)
        if mode == "optim":
            if "density_jitters" in guide_kw and "density_jitters" not in optim_kw:
                optim_kw["density_jitters"] = guide_kw["density_jitters"]
            return self.val_optim(data, **optim_kw, **kwargs)
        if mode == "guide_optim":
            with torch.enable_grad():
                code, grid, bits = self.val_guide(data, **guide_kw, **kwargs)
            return self.val_optim(data, code_=self.code_activation.inverse(code).requires_grad_(True), density_grid=grid, density_bitfield=bits,
                                  **optim_kw, **kwargs)
        raise AttributeError(f"cond_mode={mode!r}")

    def val_step(self, data, **kwargs):
        with torch.no_grad():
            code, grid, bits = self._scene_from(data, kwargs)
            pred = None
            if "test_poses" in data:
                h, w = self.test_cfg.get("img_size", (128, 128))
                image, _ = self.render(self._modules_for_eval(), code, bits, h, w, data["test_intrinsics"], data["test_poses"], cfg=self.test_cfg)
                pred = (torch.round(image.clamp(0, 1) * 255) / 255).permute(0, 1, 4, 2, 3)
        return dict(log_vars=dict(), num_samples=code.size(0), pred_imgs=pred, code=code, density_grid=grid, density_bitfield=bits)
