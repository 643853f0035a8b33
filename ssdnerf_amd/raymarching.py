"""Host-side mirror of the reference's ray-marching operator API (``lib/ops/raymarching/raymarching.py``).

Same eleven callables, same argument meaning, same return values and the same quirks (``align`` pads by a
full ``align`` when already aligned; ``composite_rays`` works in place and returns ``()``; inputs are moved
to the GPU and made contiguous silently; float inputs are computed in fp32), so that the reference's
``VolumeRenderer`` / ``BaseNeRF`` code - and our mirrors of them - run unchanged on top of the MI355X library.
All device work goes through the ``_raymarching`` drop-in backend (ssdnerf_amd/dropin) -> C ABI -> HIP.
"""
from __future__ import annotations

from itertools import groupby
from typing import List, Sequence, Tuple, Union

import torch
from torch.autograd import Function

from .dropin import _raymarching as _backend

__all__ = ["near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "march_rays_train",
           "composite_rays_train", "march_rays", "composite_rays", "batch_near_far_from_aabb", "batch_composite_rays_train"]


def _dev(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_cuda else t.cuda()


def _f32c(t: torch.Tensor) -> torch.Tensor:
    """GPU + fp32 + contiguous: what ``custom_fwd(cast_inputs=torch.float32)`` + ``.contiguous()`` give the reference."""
    t = _dev(t)
    if t.is_floating_point() and t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# ------------------------------------------------------------------------------------------ utils
def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """rays_o/d [N,3] (any leading shape), aabb [6] -> nears, fars [N]   (reference: raymarching.py:20-55)."""
    rays_o, rays_d, aabb = _f32c(rays_o).view(-1, 3), _f32c(rays_d).view(-1, 3), _f32c(aabb)
    n = rays_o.shape[0]
    nears = torch.empty(n, dtype=torch.float32, device=rays_o.device)
    fars = torch.empty(n, dtype=torch.float32, device=rays_o.device)
    _backend.near_far_from_aabb(rays_o, rays_d, aabb, n, min_near, nears, fars)
    return nears, fars


def batch_near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """One launch for all scenes; tensors (S,R,3) or per-scene lists   (reference: raymarching.py:58-82)."""
    if isinstance(rays_o, torch.Tensor):
        assert rays_o.size() == rays_d.size()
        s, r, _ = rays_o.size()
        nears, fars = near_far_from_aabb(rays_o.reshape(s * r, 3), rays_d.reshape(s * r, 3), aabb, min_near)
        return nears.reshape(s, r), fars.reshape(s, r)
    if len(rays_o) == 1:
        nears, fars = near_far_from_aabb(rays_o[0], rays_d[0], aabb, min_near)
        return [nears], [fars]
    sizes = [x.size(0) for x in rays_o]
    nears, fars = near_far_from_aabb(torch.cat(list(rays_o), dim=0), torch.cat(list(rays_d), dim=0), aabb, min_near)
    return nears.split(sizes), fars.split(sizes)


def sph_from_ray(rays_o, rays_d, radius):
    """(theta, phi) in [-1,1]^2 of the far hit with the background sphere   (reference: raymarching.py:85-114)."""
    rays_o, rays_d = _f32c(rays_o).view(-1, 3), _f32c(rays_d).view(-1, 3)
    n = rays_o.shape[0]
    coords = torch.empty(n, 2, dtype=torch.float32, device=rays_o.device)
    _backend.sph_from_ray(rays_o, rays_d, radius, n, coords)
    return coords


def morton3D(coords):
    """int coords [N,3] in [0,1024) -> int32 Morton index [N]   (reference: raymarching.py:117-139)."""
    coords = _dev(coords).int().contiguous()
    n = coords.shape[0]
    indices = torch.empty(n, dtype=torch.int32, device=coords.device)
    _backend.morton3D(coords, n, indices)
    return indices


def morton3D_invert(indices):
    """int32 Morton index [N] -> coords [N,3]   (reference: raymarching.py:142-163)."""
    indices = _dev(indices).int().contiguous()
    n = indices.shape[0]
    coords = torch.empty(n, 3, dtype=torch.int32, device=indices.device)
    _backend.morton3D_invert(indices, n, coords)
    return coords


def packbits(grid, thresh, bitfield=None):
    """grid [C, H^3] (fp32 or fp16) -> uint8 [C*H^3/8], bit i of byte n = grid[8n+i] > thresh   (reference: raymarching.py:166-193)."""
    grid = _dev(grid).contiguous()
    if grid.dtype not in (torch.float32, torch.float16):
        grid = grid.float()
    n = grid.shape[0] * grid.shape[1] // 8
    if bitfield is None:
        bitfield = torch.empty(n, dtype=torch.uint8, device=grid.device)
    _backend.packbits(grid, n, float(thresh), bitfield)
    return bitfield


# ------------------------------------------------------------------------------------------ train
def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                     perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024, noises=None):
    """Variable-length packed samples for a batch of rays (forward only)   (reference: raymarching.py:200-285).

    Returns ``xyzs [m,3], dirs [m,3], deltas [m,2]=(dt, t), rays [N,3]=(ray id, offset, count)``.
    Extra keyword ``noises`` (not in the reference) injects the per-ray jitter instead of drawing
    ``torch.rand`` - RNG streams differ between CUDA, ROCm and the CPU, so parity tests inject it.
    Unlike the reference, slot assignment is deterministic (ray order)."""
    rays_o, rays_d = _f32c(rays_o).view(-1, 3), _f32c(rays_d).view(-1, 3)
    density_bitfield = _dev(density_bitfield).contiguous()
    nears, fars = _f32c(nears), _f32c(fars)
    dev = rays_o.device
    n = rays_o.shape[0]
    m_cap = n * max_steps
    if not force_all_rays and mean_count > 0:
        if align > 0:
            mean_count += align - mean_count % align
        m_cap = mean_count
    xyzs = torch.zeros(m_cap, 3, dtype=torch.float32, device=dev)
    dirs = torch.zeros(m_cap, 3, dtype=torch.float32, device=dev)
    deltas = torch.zeros(m_cap, 2, dtype=torch.float32, device=dev)
    rays = torch.empty(n, 3, dtype=torch.int32, device=dev)
    if step_counter is None:
        step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
    if noises is not None:
        noises = _f32c(noises)
    elif perturb:
        noises = torch.rand(n, dtype=torch.float32, device=dev)
    else:
        noises = torch.zeros(n, dtype=torch.float32, device=dev)
    _backend.march_rays_train(rays_o, rays_d, density_bitfield, bound, dt_gamma, max_steps, n, C, H, m_cap, nears, fars,
                              xyzs, dirs, deltas, rays, step_counter, noises)
    if force_all_rays or mean_count <= 0:
        m = int(step_counter[0].item())  # the reference's D2H sync (raymarching.py:269); the fused path avoids it
        if align > 0:
            m += align - m % align
        xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    return xyzs, dirs, deltas, rays


class _CompositeRaysTrain(Function):
    """Differentiable w.r.t. sigmas and rgbs only; grad of depth is dropped like the reference (raymarching.py:288-338)."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        sigmas, rgbs, deltas = _f32c(sigmas), _f32c(rgbs), _f32c(deltas)
        rays = _dev(rays).contiguous()
        m, n = sigmas.shape[0], rays.shape[0]
        dev = sigmas.device
        weights_sum = torch.empty(n, dtype=torch.float32, device=dev)
        depth = torch.empty(n, dtype=torch.float32, device=dev)
        image = torch.empty(n, 3, dtype=torch.float32, device=dev)
        _backend.composite_rays_train_forward(sigmas, rgbs, deltas, rays, m, n, T_thresh, weights_sum, depth, image)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, image)
        ctx.dims = (m, n, T_thresh)
        return weights_sum, depth, image

    @staticmethod
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        sigmas, rgbs, deltas, rays, weights_sum, image = ctx.saved_tensors
        m, n, T_thresh = ctx.dims
        grad_sigmas = torch.zeros_like(sigmas)
        grad_rgbs = torch.zeros_like(rgbs)
        _backend.composite_rays_train_backward(_f32c(grad_weights_sum), _f32c(grad_image), sigmas, rgbs, deltas, rays, weights_sum,
                                               image, m, n, T_thresh, grad_sigmas, grad_rgbs)
        return grad_sigmas, grad_rgbs, None, None, None


composite_rays_train = _CompositeRaysTrain.apply


def _all_equal(xs) -> bool:
    g = groupby(xs)
    return next(g, True) and not next(g, False)


def batch_composite_rays_train(sigmas, rgbs, deltas, rays, num_points, T_thresh=1e-4):
    """Composite several scenes' packed samples in one launch   (reference: raymarching.py:349-395).
    ``deltas``/``rays`` are per-scene lists; ray ids and point offsets are rebased into the concatenated arrays."""
    s = len(deltas)
    if s == 1:
        ws, depth, image = composite_rays_train(sigmas, rgbs, deltas[0], rays[0], T_thresh)
        return ws[None], depth[None], image[None]
    rebased, counts = [], []
    ray_off = pt_off = 0
    for r, npts in zip(rays, num_points):
        shift = torch.tensor([ray_off, pt_off, 0], dtype=r.dtype, device=r.device)
        rebased.append(r + shift)
        ray_off += r.size(0)
        pt_off += npts
        counts.append(r.size(0))
    ws, depth, image = composite_rays_train(sigmas, rgbs, torch.cat(list(deltas), dim=0), torch.cat(rebased, dim=0), T_thresh)
    if _all_equal(counts):
        return ws.reshape(s, counts[0]), depth.reshape(s, counts[0]), image.reshape(s, counts[0], 3)
    return ws.split(counts, dim=0), depth.split(counts, dim=0), image.split(counts, dim=0)


# ------------------------------------------------------------------------------------------ inference
def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
               align=-1, perturb=False, dt_gamma=0, max_steps=1024, noises=None):
    """<= n_step samples for each of the first n_alive rays of ``rays_alive``, fixed slots   (reference: raymarching.py:402-460)."""
    rays_o, rays_d = _f32c(rays_o).view(-1, 3), _f32c(rays_d).view(-1, 3)
    dev = rays_o.device
    m = n_alive * n_step
    if align > 0:
        m += align - (m % align)
    xyzs = torch.zeros(m, 3, dtype=torch.float32, device=dev)
    dirs = torch.zeros(m, 3, dtype=torch.float32, device=dev)
    deltas = torch.zeros(m, 2, dtype=torch.float32, device=dev)
    if noises is not None:
        noises = _f32c(noises)
    elif perturb:
        noises = torch.rand(n_alive, dtype=torch.float32, device=dev)
    else:
        noises = torch.zeros(n_alive, dtype=torch.float32, device=dev)
    _backend.march_rays(n_alive, n_step, rays_alive, _f32c(rays_t), rays_o, rays_d, bound, dt_gamma, max_steps, C, H,
                        _dev(density_bitfield).contiguous(), _f32c(near), _f32c(far), xyzs, dirs, deltas, noises)
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    """In place on rays_alive / rays_t / weights_sum / depth / image; returns ``()``   (reference: raymarching.py:463-489)."""
    _backend.composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, _f32c(sigmas), _f32c(rgbs), _f32c(deltas), weights_sum,
                            depth, image)
    return tuple()
