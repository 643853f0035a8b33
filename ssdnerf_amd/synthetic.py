"""Seeded synthetic inputs for the hot path (no dataset, checkpoint or network is available).

Everything is generated on the CPU with an explicit ``torch.Generator`` so that the same bytes are
produced in this container, on the GPU box and on every rank.

* ``spiral_poses``      -- 251 camera-to-world matrices on a sphere of radius 2.6 looking at the
                           origin, z-up, camera +z forward / +y down: the convention of the SRN
                           spiral the reference renders (``demo/camera_spiral_cars``; pose
                           translation / 0.5, ``lib/datasets/shapenet_srn.py:149-156``).  Generated
                           analytically -- the reference's pose files are not copied.
* ``cars_intrinsics``   -- ``[fx, fy, cx, cy] = [131.25, 131.25, 64, 64]`` for 128x128.
* ``make_decoder_params`` -- tiny-MLP weights with the shapes of the cars config
                           (``configs/paper_cfgs/ssdnerf_cars_uncond.py:39-50``), Xavier-uniform like
                           ``triplane_decoder.py:97-102`` plus a deliberate "shape pathway" so that
                           density is object-like instead of uniform fog.
* ``make_triplane``     -- a ``(3, 6, 128, 128)`` code in the ``TanhCode(scale=2)`` range whose
                           channel 0 carries three soft silhouettes (visual hull of a car-sized box)
                           and whose other channels are smooth noise.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

CODE_SIZE = (3, 6, 128, 128)
GRID_SIZE = 64
IMG_SIZE = 128


def cars_intrinsics(h: int = IMG_SIZE, w: int = IMG_SIZE) -> torch.Tensor:
    s = h / 128.0
    return torch.tensor([131.25 * s, 131.25 * s, 64.0 * s, 64.0 * s], dtype=torch.float32)


def spiral_poses(num: int = 251, radius: float = 2.6, turns: float = 4.0) -> torch.Tensor:
    """(num, 4, 4) c2w.  Elevation sweeps 80deg..-5deg while azimuth makes ``turns`` revolutions."""
    i = torch.arange(num, dtype=torch.float64)
    frac = i / max(num - 1, 1)
    elev = torch.deg2rad(80.0 - 85.0 * frac)
    azim = 2.0 * math.pi * turns * frac
    eye = torch.stack([radius * torch.cos(elev) * torch.cos(azim),
                       radius * torch.cos(elev) * torch.sin(azim),
                       radius * torch.sin(elev)], dim=-1)
    fwd = -eye / eye.norm(dim=-1, keepdim=True)                      # camera +z
    up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64).expand_as(fwd)
    right = torch.cross(fwd, up, dim=-1)
    right = right / right.norm(dim=-1, keepdim=True)                 # camera +x
    down = torch.cross(fwd, right, dim=-1)                           # camera +y (image y grows downwards)
    c2w = torch.zeros(num, 4, 4, dtype=torch.float64)
    c2w[:, :3, 0] = right
    c2w[:, :3, 1] = down
    c2w[:, :3, 2] = fwd
    c2w[:, :3, 3] = eye
    c2w[:, 3, 3] = 1.0
    return c2w.float()


def _xavier_uniform(gen: torch.Generator, out_f: int, in_f: int) -> torch.Tensor:
    a = math.sqrt(6.0 / (in_f + out_f))
    return (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * a


def make_decoder_params(seed: int = 2021) -> Dict[str, torch.Tensor]:
    """State-dict keyed exactly like the reference's ``TriPlaneDecoder`` (``base_net.0.weight`` ...)."""
    g = torch.Generator().manual_seed(seed)
    p = {
        "base_net.0.weight": _xavier_uniform(g, 64, 18), "base_net.0.bias": torch.zeros(64),
        "density_net.0.weight": _xavier_uniform(g, 1, 64), "density_net.0.bias": torch.zeros(1),
        "dir_net.0.weight": _xavier_uniform(g, 64, 16) * 0.5, "dir_net.0.bias": torch.zeros(64),
        "color_net.0.weight": _xavier_uniform(g, 3, 64), "color_net.0.bias": torch.zeros(3),
    }
    p["base_net.0.bias"] = (torch.rand(64, generator=g) * 2 - 1) * 0.1
    # shape pathway: hidden unit 0 = 2*(f[c=0,xy] + f[c=0,xz] + f[c=0,yz] - 4); feature index = c*3 + plane
    w0 = torch.zeros(18)
    w0[0:3] = 2.0
    p["base_net.0.weight"][0] = w0
    p["base_net.0.bias"][0] = -8.0
    p["density_net.0.weight"] *= 0.5
    p["density_net.0.weight"][0, 0] = 3.0
    p["density_net.0.bias"][0] = -8.1   # sigma ~ 40 inside (alpha ~ 0.4 per step), ~3e-4 outside
    return p


def _smooth_noise(gen: torch.Generator, chans: int, size: int, cells: int) -> torch.Tensor:
    low = torch.randn(1, chans, cells, cells, generator=gen)
    return F.interpolate(low, size=(size, size), mode="bicubic", align_corners=False)[0]


def make_triplane(seed: int = 2021, variant: str = "object") -> torch.Tensor:
    """(3, 6, 128, 128) fp32 code in [-2, 2].  ``variant='uniform'`` is the worst-case fog scene."""
    g = torch.Generator().manual_seed(seed)
    n_pl, n_ch, h, w = CODE_SIZE
    if variant == "uniform":
        return (torch.rand(CODE_SIZE, generator=g) * 4 - 2).float()
    code = torch.empty(CODE_SIZE)
    for p in range(n_pl):
        code[p] = torch.tanh(_smooth_noise(g, n_ch, h, 8) * 0.8) * 2
    # soft silhouettes of a car-sized box with per-scene jittered half extents (x, y, z)
    ext = torch.tensor([0.62, 0.30, 0.24]) * (0.85 + 0.3 * torch.rand(3, generator=g))
    lin = (torch.arange(h, dtype=torch.float32) + 0.5) / h * 2 - 1   # texel centres, align_corners=False
    # plane p samples (u, v): xy->(x,y), xz->(x,z), yz->(y,z); grid x = width axis, grid y = height axis
    axes = [(0, 1), (0, 2), (1, 2)]
    wobble = _smooth_noise(g, 3, h, 6) * 0.04
    for p, (au, av) in enumerate(axes):
        u = lin[None, :].expand(h, w)
        v = lin[:, None].expand(h, w)
        du = (u.abs() - ext[au]) / 0.03
        dv = (v.abs() - ext[av]) / 0.03
        inside = -torch.maximum(du, dv) + wobble[p] / 0.03
        code[p, 0] = torch.tanh(inside) * 2
    return code.clamp(-2, 2).float().contiguous()


def make_scene_batch(num_scenes: int, seed: int = 2021, variant: str = "object") -> torch.Tensor:
    return torch.stack([make_triplane(seed + s, variant) for s in range(num_scenes)], dim=0)
