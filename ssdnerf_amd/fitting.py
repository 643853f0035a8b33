"""Fitting scene codes to posed observations: the machinery behind rendering guidance (``val_guide``), optimisation-based inversion
(``inverse_code``) and the fine-tuning half of ``cond_mode='guide_optim'`` (``val_optim``).  Behaviour: lib/models/autodecoders/
base_nerf.py:231-316, 403-492 and diffusion_nerf.py:241-404; SURVEY.md section 8 rows a15, (f)1.

The reference spreads this over nested closures and long method bodies; here it is four small objects that the model classes compose:

  Conditioning      the observations of a batch of scenes: images, their rays (generated on the device by one HIP pass) and the per-scene
                    cone angle ``dt_gamma``;
  RayBatcher        which pixels each step looks at: disjoint chunks of one random permutation per scene, cycled (or everything, when the
                    views are small), gathered on the device;
  CodeFitter        N rendering-loss iterations on pre-activation code leaves: activation -> (periodic) density refresh -> train-branch
                    render + loss -> backward on top of an optional seed gradient (the diffusion prior's) -> optimizer / scheduler step;
  GuidanceObjective the ``grad_guide_fn`` of rendering-guided DDIM as an object that owns its density grid and step counter.

Everything that draws random numbers accepts injected draws (march jitter, grid jitter), so parity runs reproduce across devices.
The model passes ITSELF in: ``loss`` / ``update_extra_state`` are looked up on it at call time (they are the reference's public hooks).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional, Sequence, Union

import torch

from . import nerf


@dataclass
class Conditioning:
    images: torch.Tensor        # (S, V, h, w, 3)
    rays_o: torch.Tensor        # (S, V, h, w, 3)
    rays_d: torch.Tensor
    dt_gamma: torch.Tensor      # (S,) cone angle of the march: dt_gamma_scale / mean focal length

    @classmethod
    def from_batch(cls, data: Dict, dt_gamma_scale: float) -> "Conditioning":
        imgs, intr, poses = data["cond_imgs"], data["cond_intrinsics"], data["cond_poses"]
        h, w = imgs.shape[2:4]
        o, d = nerf.get_cam_rays(poses, intr, h, w)
        return cls(imgs, o, d, dt_gamma_scale / intr[..., :2].mean(dim=(-2, -1)))

    @property
    def num_scenes(self) -> int:
        return self.images.size(0)

    @property
    def pixels_per_scene(self) -> int:
        return self.images.shape[1:4].numel()


class RayBatcher:
    """``batch(k)`` -> (rays_o, rays_d, target), each (S, R, 3).  With more pixels than ``rays_per_step`` the pixels of every scene are
    permuted once and step k takes chunk k mod n_chunks (``fixed=True``: base_nerf.py:263-274, used by inversion and guidance), or a fresh
    random subset is drawn per call (``fixed=False``: base_nerf.py:231-261 without indices, used by ``loss_decoder``)."""

    def __init__(self, cond: Conditioning, rays_per_step: int, fixed: bool = True):
        self.S, self.P, self.R = cond.num_scenes, cond.pixels_per_scene, rays_per_step
        self.flat = (cond.rays_o.reshape(self.S, self.P, 3), cond.rays_d.reshape(self.S, self.P, 3), cond.images.reshape(self.S, self.P, 3))
        self.chunks: Optional[Sequence[torch.Tensor]] = None
        if fixed and self.P > self.R:
            dev = cond.images.device
            self.chunks = torch.stack([torch.randperm(self.P, device=dev) for _ in range(self.S)], dim=0).split(self.R, dim=1)

    def indices(self, k: int) -> Optional[torch.Tensor]:
        return None if self.chunks is None else self.chunks[k % len(self.chunks)]

    def take(self, inds: Optional[torch.Tensor]):
        if self.P <= self.R:
            return self.flat
        if inds is None:
            dev = self.flat[0].device
            inds = torch.stack([torch.randperm(self.P, device=dev)[:self.R] for _ in range(self.S)], dim=0)
        rows = torch.arange(self.S, device=inds.device)[:, None]
        return tuple(t[rows, inds] for t in self.flat)

    def batch(self, k: int):
        return self.take(self.indices(k))


def _as_iter(x) -> Optional[Iterator]:
    return None if x is None else iter(x)


class CodeFitter:
    """Rendering-loss iterations on code leaves (``code_``: one (S, ...) leaf, or a list of per-scene leaves with one optimizer each).

    One iteration k:  code = activation(code_)  ->  every ``update_extra_interval`` iterations the density grid is refreshed from the detached
    code  ->  ray batch k  ->  ``model.loss`` through the decoder's TRAIN branch  ->  gradients: zeroed, or -- ``seed_grad`` -- overwritten
    with a gradient computed elsewhere (the diffusion prior's), so that the rendering gradient ACCUMULATES on top of it  ->  backward  ->
    optimizer step(s)  ->  scheduler step(s)."""

    def __init__(self, model, decoder, cond: Conditioning, cfg: Dict, code_, density_grid, density_bitfield, optimizers, schedulers=None,
                 march_noises=None, density_jitters=None):
        self.model, self.decoder, self.cond, self.cfg = model, decoder, cond, cfg
        self.code_, self.grid, self.bits = code_, density_grid, density_bitfield
        self.optimizers = list(optimizers) if isinstance(optimizers, (list, tuple)) else [optimizers]
        self.schedulers = [] if schedulers is None else (list(schedulers) if isinstance(schedulers, (list, tuple)) else [schedulers])
        self.batcher = RayBatcher(cond, cfg.get("n_inverse_rays", 4096), fixed=True)
        self.march_noises, self.density_jitters = _as_iter(march_noises), _as_iter(density_jitters)
        self.k = 0
        self.last = None            # (code, loss, loss_dict, rendered rgb, target rgb) of the latest iteration

    def _leaves(self) -> List[torch.Tensor]:
        return self.code_ if isinstance(self.code_, list) else [self.code_]

    def activated(self, **kw) -> torch.Tensor:
        return self.model.code_activation(torch.stack(self.code_, dim=0) if isinstance(self.code_, list) else self.code_, **kw)

    def step(self, seed_grad: Optional[Union[torch.Tensor, Sequence[torch.Tensor]]] = None):
        m, cfg = self.model, self.cfg
        code = self.activated()
        if self.k % m.update_extra_interval == 0:
            m.update_extra_state(self.decoder, code.detach(), self.grid, self.bits, 0, density_thresh=cfg.get("density_thresh", 0.01),
                                 jitter=None if self.density_jitters is None else next(self.density_jitters))
        rays_o, rays_d, target = self.batcher.batch(self.k)
        if self.march_noises is not None:
            self.decoder.injected_noises = next(self.march_noises)
        try:
            rgb, loss, parts = m.loss(self.decoder, code, self.bits, target, rays_o, rays_d, self.cond.dt_gamma,
                                      scale_num_ray=self.cond.pixels_per_scene, cfg=cfg)
        finally:
            self.decoder.injected_noises = None
        if seed_grad is None:
            for opt in self.optimizers:
                opt.zero_grad()
        else:
            seeds = seed_grad if isinstance(self.code_, list) else [seed_grad]
            for leaf, g in zip(self._leaves(), seeds):
                leaf.grad.copy_(g)
        loss.backward()
        for opt in self.optimizers:
            opt.step()
        for sch in self.schedulers:
            sch.step()
        self.k += 1
        self.last = (code.detach(), loss, parts, rgb, target)
        return self.last

    def run(self, n_steps: int, seed_grad=None):
        assert n_steps > 0
        was_training = self.decoder.training
        self.decoder.train(True)                     # the renderer's TRAIN branch: packed march with jitter, differentiable composite
        try:
            for _ in range(n_steps):
                self.step(seed_grad)
        finally:
            self.decoder.train(was_training)
        return self.last


class GuidanceObjective:
    """``loss = GuidanceObjective(...)(x0_pred)``: the rendering loss that steers every DDIM step of ``val_guide``
    (diffusion_nerf.py:282-294).  Per call: x0 -> scene codes, ONE density-grid refresh from them (decay 0.9 against the previous step's
    grid, which this object owns), ray batch k, train-branch render, pixel (+ regularisation) loss, summed over the scenes of the batch."""

    def __init__(self, model, decoder, cond: Conditioning, cfg: Dict, march_noises=None, density_jitters=None):
        self.model, self.decoder, self.cond, self.cfg = model, decoder, cond, cfg
        dev, S, H3 = cond.images.device, cond.num_scenes, model.grid_size ** 3
        self.grid = torch.zeros((S, H3), device=dev)                                   # fp32 here, as in the reference (diffusion_nerf.py:278)
        self.bits = torch.zeros((S, H3 // 8), dtype=torch.uint8, device=dev)
        self.batcher = RayBatcher(cond, cfg.get("n_inverse_rays", 4096), fixed=True)
        self.march_noises, self.density_jitters = march_noises, density_jitters
        self.calls = 0

    def __call__(self, x0_pred: torch.Tensor) -> torch.Tensor:
        m, k = self.model, self.calls
        code = m.code_diff_pr_inv(x0_pred)
        m.update_extra_state(self.decoder, code.detach().float(), self.grid, self.bits, 0, density_thresh=self.cfg.get("density_thresh", 0.01),
                             jitter=None if self.density_jitters is None else self.density_jitters[k])
        rays_o, rays_d, target = self.batcher.batch(k)
        if self.march_noises is not None:
            self.decoder.injected_noises = self.march_noises[k]
        try:
            _, loss, _ = m.loss(self.decoder, code, self.bits, target, rays_o, rays_d, self.cond.dt_gamma, scale_num_ray=target.size(1), cfg=self.cfg)
        finally:
            self.decoder.injected_noises = None
        self.calls += 1
        return loss * self.cond.num_scenes
