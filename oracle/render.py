"""oracle/render.py -- TEST INFRASTRUCTURE.  Reference-shaped host loops on the CPU.

Restates, for one scene at a time and with every random draw injected by the caller:

* ``get_cam_rays``        lib/core/utils/nerf_utils.py:17-61
* ``render_eval``         the eval branch of ``VolumeRenderer.forward``
                          (lib/models/decoders/base_volume_renderer.py:79-123) + the background
                          blend of ``BaseNeRF.render`` (lib/models/autodecoders/base_nerf.py:522-523)
* ``render_train``        the train branch (base_volume_renderer.py:59-77)
* ``update_extra_state`` / ``get_density``
                          full 64^3 branch of lib/models/autodecoders/base_nerf.py:318-401

The C oracle supplies march/composite/morton/packbits; oracle.decoder supplies the decode.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import ops as _ops
from .decoder import point_decode


def get_cam_rays(c2w: torch.Tensor, intrinsics: torch.Tensor, h: int, w: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """c2w (..., 4, 4) or (..., 3, 4), intrinsics (..., 4) -> rays_o, rays_d (..., h, w, 3)."""
    x = torch.linspace(0.5, w - 0.5, w)
    y = torch.linspace(0.5, h - 0.5, h)
    batch = intrinsics.shape[:-1]
    dx = ((x - intrinsics[..., 2:3]) / intrinsics[..., 0:1])[..., None, :].expand(*batch, h, w)
    dy = ((y - intrinsics[..., 3:4]) / intrinsics[..., 1:2])[..., :, None].expand(*batch, h, w)
    dirs = torch.stack([dx, dy, torch.ones_like(dx)], dim=-1)
    rays_d = dirs @ c2w[..., None, :3, :3].transpose(-1, -2)
    rays_o = c2w[..., None, None, :3, 3].expand(rays_d.shape)
    rays_d = F.normalize(rays_d, dim=-1)
    return rays_o.contiguous(), rays_d.contiguous()


def render_eval(params: Dict[str, torch.Tensor], code: torch.Tensor, bitfield: np.ndarray, rays_o: np.ndarray,
                rays_d: np.ndarray, grid_size: int = 64, bound: float = 1.0, min_near: float = 0.2, max_steps: int = 256,
                dt_gamma: float = 0.0, T_thresh: float = 1e-4, bg_color: float = 1.0, ops=None, trace: Optional[dict] = None,
                near_band: float = 2e-6):
    """One scene.  Returns rgb (N,3), depth (N,), weights_sum (N,).  ``trace`` (if given) is filled with the
    integer history of the loop: n_alive and n_step per iteration and the per-ray sample count."""
    o = ops or _ops()
    rays_o = np.ascontiguousarray(rays_o, np.float32).reshape(-1, 3)
    rays_d = np.ascontiguousarray(rays_d, np.float32).reshape(-1, 3)
    N = rays_o.shape[0]
    aabb = np.array([-bound, -bound, -bound, bound, bound, bound], np.float32)
    nears, fars = o.near_far_from_aabb(rays_o, rays_d, aabb, min_near)
    ws = np.zeros(N, np.float32)
    depth = np.zeros(N, np.float32)
    image = np.zeros((N, 3), np.float32)
    rays_alive = np.arange(N, dtype=np.int32)
    rays_t = nears.copy()
    samples = np.zeros(N, np.int64)
    composited = np.zeros(N, np.int64)
    near_cut = np.zeros(N, bool)
    hist: List[Tuple[int, int]] = []
    step = 0
    while step < max_steps:
        n_alive = rays_alive.shape[0]
        if n_alive == 0:
            break
        n_step = min(max(N // n_alive, 1), 8)
        xyzs, dirs, deltas = o.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, bitfield, 1, grid_size,
                                          nears, fars, align=128, dt_gamma=dt_gamma, max_steps=max_steps)
        with torch.no_grad():
            sig, rgb = point_decode(params, code, torch.from_numpy(xyzs), torch.from_numpy(dirs))
        taken = (deltas[: n_alive * n_step, 0].reshape(n_alive, n_step) != 0).sum(1)
        samples[rays_alive] += taken
        if trace is not None:      # samples that ENTER the composite (a ray cut at T < T_thresh leaves the rest of its batch unused): replay of the
            dts = deltas[: n_alive * n_step, 0].reshape(n_alive, n_step)          # kernel's per-ray loop (.cu:875-903), vectorised over the rays
            sg = sig.numpy()[: n_alive * n_step].reshape(n_alive, n_step)
            w_run = ws[rays_alive].copy()
            going = np.ones(n_alive, bool)
            for k in range(n_step):
                going &= dts[:, k] != 0
                alpha = (np.float32(1) - np.exp(-sg[:, k] * dts[:, k], dtype=np.float32)).astype(np.float32)
                T = (np.float32(1) - w_run).astype(np.float32)
                w_run = np.where(going, (w_run + alpha * T).astype(np.float32), w_run)
                composited[rays_alive[going]] += 1
                near_cut[rays_alive[going & (np.abs(T - np.float32(T_thresh)) < near_band)]] = True
                going &= ~(T < np.float32(T_thresh))
        o.composite_rays(n_alive, n_step, rays_alive, rays_t, sig.numpy(), rgb.numpy(), deltas, ws, depth, image, T_thresh)
        hist.append((n_alive, n_step))
        rays_alive = np.ascontiguousarray(rays_alive[rays_alive >= 0])
        step += n_step
    if trace is not None:
        trace["iterations"] = hist
        trace["samples_marched"] = samples
        trace["samples_composited"] = composited      # what the fused kernels count per ray
        trace["near_threshold"] = near_cut            # rays with a termination test within near_band of T_thresh (exp implementations may disagree there)
        trace["nears"], trace["fars"] = nears, fars
    rgb_out = image + bg_color * (1.0 - ws[:, None])
    return rgb_out.astype(np.float32), depth, ws


def render_train(params, code, bitfield, rays_o, rays_d, noises, grid_size=64, bound=1.0, min_near=0.2, max_steps=256,
                 dt_gamma=0.0, T_thresh=1e-4, ops=None):
    """Train branch for one scene, forward only: march_rays_train(force_all_rays, align=128) -> decode -> composite."""
    o = ops or _ops()
    rays_o = np.ascontiguousarray(rays_o, np.float32).reshape(-1, 3)
    rays_d = np.ascontiguousarray(rays_d, np.float32).reshape(-1, 3)
    aabb = np.array([-bound, -bound, -bound, bound, bound, bound], np.float32)
    nears, fars = o.near_far_from_aabb(rays_o, rays_d, aabb, min_near)
    xyzs, dirs, deltas, rays, counter = o.march_rays_train(rays_o, rays_d, bitfield, bound, dt_gamma, max_steps, 1, grid_size,
                                                           nears, fars, noises)
    m = int(counter[0])
    m += 128 - m % 128
    xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    with torch.no_grad():
        sig, rgb = point_decode(params, code, torch.from_numpy(xyzs), torch.from_numpy(dirs))
    ws, depth, image = o.composite_rays_train_forward(sig.numpy(), rgb.numpy(), deltas, rays, T_thresh)
    return dict(weights_sum=ws, depth=depth, image=image, rays=rays, num_points=int(counter[0]), xyzs=xyzs, dirs=dirs,
                deltas=deltas, sigmas=sig.numpy(), rgbs=rgb.numpy())


def grid_cell_coords(grid_size: int) -> np.ndarray:
    """(H^3, 3) int32 in the x-major / z-minor order ``custom_meshgrid(xs, ys, zs)`` produces (base_nerf.py:337-339)."""
    r = np.arange(grid_size, dtype=np.int32)
    xx, yy, zz = np.meshgrid(r, r, r, indexing="ij")
    return np.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], axis=-1)


def update_extra_state(params, code: torch.Tensor, density_grid: np.ndarray, jitter: np.ndarray, grid_size: int = 64,
                       bound: float = 1.0, density_thresh: float = 0.01, decay: float = 0.9, ops=None):
    """Full-refresh branch for ONE scene (num_scenes == 1 in the reference's formulas).

    density_grid: (H^3,) float32 or float16, Morton order, updated in place.
    jitter: (H^3, 3) uniform [0,1) draws (the reference's ``torch.rand_like``), injected for parity.
    Returns the packed bitfield (H^3/8,) uint8 and the threshold used."""
    o = ops or _ops()
    coords = grid_cell_coords(grid_size)
    indices = o.morton3D(coords).astype(np.int64)
    xyzs = (coords.astype(np.float32) - (grid_size - 1) / 2) * (2 * bound / grid_size)
    half = bound / grid_size
    xyzs = (xyzs + (jitter.astype(np.float32) * (2 * half) - half)).astype(np.float32)
    with torch.no_grad():
        sig, _ = point_decode(params, code, torch.from_numpy(xyzs), None, density_only=True)
    tmp = np.full(density_grid.shape, -1, dtype=density_grid.dtype)
    fmax = np.finfo(density_grid.dtype).max
    tmp[indices] = np.minimum(sig.numpy(), fmax).astype(density_grid.dtype)
    valid = (density_grid >= 0) & (tmp >= 0)
    decayed = (density_grid * np.asarray(decay, density_grid.dtype)).astype(density_grid.dtype)
    density_grid[:] = np.where(valid, np.maximum(decayed, tmp), density_grid)
    mean_density = float(np.clip(density_grid, 0, None).astype(np.float32).mean()) if density_grid.dtype == np.float32 else \
        float(torch.from_numpy(density_grid).clamp(min=0).mean())
    thresh = min(mean_density, density_thresh)
    bitfield = o.packbits(density_grid.astype(np.float32), thresh)
    return bitfield, thresh


def get_density(params, code: torch.Tensor, jitters: List[np.ndarray], grid_size: int = 64, density_thresh: float = 0.01,
                dtype=np.float16, ops=None):
    """``BaseNeRF.get_density``: ``len(jitters)`` (8 in the configs) full refreshes with decay 1.0 from a zero grid."""
    grid = np.zeros(grid_size ** 3, dtype=dtype)
    bitfield, thresh = None, None
    for j in jitters:
        bitfield, thresh = update_extra_state(params, code, grid, j, grid_size, density_thresh=density_thresh, decay=1.0, ops=ops)
    return grid, bitfield, thresh
