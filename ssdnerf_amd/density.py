"""Density-grid maintenance: the full-refresh branch of ``BaseNeRF.update_extra_state`` and ``get_density``
(reference: lib/models/autodecoders/base_nerf.py:318-401) as two HIP launches per refresh - fused
{cell centre + jitter -> gather -> density MLP -> max-EMA into the Morton grid -> mean} and packbits with the
threshold ``min(mean, density_thresh)`` taken ON DEVICE (the reference syncs to the host for it)."""
from __future__ import annotations

from typing import Optional

import torch

from . import _cabi as C
from .decoders import TriPlaneDecoder, pack_triplanes


def update_density_grid(decoder: TriPlaneDecoder, code: torch.Tensor, density_grid: torch.Tensor, density_bitfield: torch.Tensor,
                        density_thresh: float = 0.01, decay: float = 0.9, jitter: Optional[torch.Tensor] = None,
                        planes: Optional[torch.Tensor] = None, return_thresh: bool = True):
    """In place on ``density_grid`` (S,H^3; fp16 or fp32, Morton order) and ``density_bitfield`` (S,H^3/8 uint8).
    ``jitter`` (H^3,3) in [0,1) is the reference's ``torch.rand_like`` draw, shared by all scenes; None draws it here."""
    assert decoder.fused_supported(code), "update_density_grid needs the fused-decode configuration"
    s, h3 = density_grid.shape
    h = round(h3 ** (1 / 3))
    assert h ** 3 == h3
    dev = code.device
    if planes is None:
        planes = pack_triplanes(code.detach(), decoder.plane_dtype)
    if jitter is None:
        jitter = torch.rand(h3, 3, dtype=torch.float32, device=dev)
    jitter = jitter.float().contiguous()
    mean = torch.zeros(1, dtype=torch.float32, device=dev)
    _, _, hp, wp, _ = planes.shape
    C.check(C.lib().ssdnerf_density_grid_update(C.ptr(planes), C.dtype_code(planes), C.u32(hp), C.u32(wp), C.ptr(decoder.packed_params()),
                                                C.u32(s), C.u32(h), C.f32(decoder.bound), C.ptr(jitter), C.f32(decay), C.ptr(density_grid),
                                                C.dtype_code(density_grid), C.ptr(mean), C.stream()), "density_grid_update")
    if density_grid.dtype == torch.float16:
        mean = mean.half().float()          # torch.mean of an fp16 grid is rounded to fp16 before the min() (base_nerf.py:382-386)
    C.check(C.lib().ssdnerf_packbits_dev_thresh(C.ptr(density_grid), C.dtype_code(density_grid), C.u32(s * h3 // 8), C.ptr(mean),
                                                C.f32(density_thresh), C.ptr(density_bitfield), C.stream()), "packbits_dev_thresh")
    if return_thresh:
        return torch.minimum(mean, torch.tensor(density_thresh, device=dev))[0]
    return None


def get_density(decoder: TriPlaneDecoder, code: torch.Tensor, grid_size: int = 64, density_thresh: float = 0.01, density_step: int = 8,
                jitters=None, grid_dtype=torch.float16):
    """``BaseNeRF.get_density``: ``density_step`` full refreshes with decay 1.0 from a zero grid (base_nerf.py:391-401)."""
    s = code.size(0)
    dev = code.device
    grid = torch.zeros(s, grid_size ** 3, dtype=grid_dtype, device=dev)
    bits = torch.zeros(s, grid_size ** 3 // 8, dtype=torch.uint8, device=dev)
    planes = pack_triplanes(code.detach(), decoder.plane_dtype)
    for i in range(density_step):
        update_density_grid(decoder, code, grid, bits, density_thresh=density_thresh, decay=1.0,
                            jitter=None if jitters is None else jitters[i], planes=planes, return_thresh=False)
    return grid, bits
