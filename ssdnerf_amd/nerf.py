"""Scene-level render entry points: the slice of ``BaseNeRF`` that sits on the hot path
(reference: lib/models/autodecoders/base_nerf.py:494-533 ``render``, :551-553 output quantisation,
lib/core/utils/nerf_utils.py:17-61 ``get_cam_rays``), host orchestration in Python on PyTorch-ROCm."""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from .decoders import TriPlaneDecoder, pack_triplanes


def get_ray_directions(h: int, w: int, intrinsics: torch.Tensor, norm: bool = False, device=None) -> torch.Tensor:
    """intrinsics (*,4) = [fx, fy, cx, cy] -> pixel-centre directions (*, h, w, 3) in camera space."""
    batch = intrinsics.shape[:-1]
    x = torch.linspace(0.5, w - 0.5, w, device=device)
    y = torch.linspace(0.5, h - 0.5, h, device=device)
    dx = ((x - intrinsics[..., 2:3]) / intrinsics[..., 0:1])[..., None, :].expand(*batch, h, w)
    dy = ((y - intrinsics[..., 3:4]) / intrinsics[..., 1:2])[..., :, None].expand(*batch, h, w)
    d = torch.stack([dx, dy, torch.ones_like(dx)], dim=-1)
    return F.normalize(d, dim=-1) if norm else d


def get_rays(directions: torch.Tensor, c2w: torch.Tensor, norm: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    rays_d = directions @ c2w[..., None, :3, :3].transpose(-1, -2)
    rays_o = c2w[..., None, None, :3, 3].expand(rays_d.shape)
    if norm:
        rays_d = F.normalize(rays_d, dim=-1)
    return rays_o, rays_d


def get_cam_rays(c2w: torch.Tensor, intrinsics: torch.Tensor, h: int, w: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """c2w (S,V,4,4), intrinsics (S,V,4) -> rays_o, rays_d (S,V,h,w,3), directions normalised after rotation.

    GPU tensors go through one HIP kernel (csrc/raygen.hip: a pass that writes 24 B per ray instead of ~10 eager ops over the full
    arrays); CPU tensors take the tensor-op form of the reference (nerf_utils.py:57-61), which is also what the tests compare with."""
    if c2w.is_cuda:
        from . import _cabi as C
        batch = tuple(c2w.shape[:-2])
        pose = c2w.detach().to(torch.float32).reshape(-1, 16).contiguous()
        intr = intrinsics.detach().to(torch.float32).expand(*batch, 4).reshape(-1, 4).contiguous()
        rays_o = torch.empty(*batch, h, w, 3, dtype=torch.float32, device=c2w.device)
        rays_d = torch.empty_like(rays_o)
        C.check(C.lib().ssdnerf_cam_rays(C.ptr(pose), C.ptr(intr), C.u32(pose.size(0)), C.u32(h), C.u32(w), C.ptr(rays_o), C.ptr(rays_d), C.stream()),
                "cam_rays")
        return rays_o, rays_d
    directions = get_ray_directions(h, w, intrinsics, norm=False, device=intrinsics.device)
    return get_rays(directions, c2w, norm=True)


@torch.no_grad()
def render(decoder: TriPlaneDecoder, code: torch.Tensor, density_bitfield: torch.Tensor, h: int, w: int, intrinsics: torch.Tensor,
           poses: torch.Tensor, grid_size: int = 64, bg_color: float = 1.0, cfg: Optional[Dict] = None,
           planes: Optional[torch.Tensor] = None, rays: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, return_u8: bool = False,
           defer_overflow_check: bool = False, next_batch: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None):
    """``BaseNeRF.render``: (S,V) views of S scenes -> image (S,V,h,w,3) blended with ``bg_color``, depth (S,V,h,w).

    ``next_batch`` (extra, r06; streaming loops over cached scenes -- evaluation, bench.py): ``(density_bitfield, intrinsics, poses)`` of the NEXT ``render`` call on this
    decoder.  Stage A of that render (cull + survivor march: it needs the bitfield and the cameras, not the planes) is launched now, on a second stream and into its own
    workspace and output tensors, and runs beside THIS render's shading kernel (``TriPlaneDecoder.render_packed(prefetch=...)``); the next call recognises its inputs
    -- the same tensors, unchanged -- and launches only its shading kernel.  Those inputs must be complete on the current stream at this call; a next call with other
    inputs renders normally; ``finish_render`` after the last call of a loop forgets a stage nobody will use.  Only the uncond form (``dt_gamma_scale`` 0) takes it.

    ``defer_overflow_check`` (extra, r05): the one host read per call -- did a ray reach the ``max_steps`` cap, so that the batch has to be redone through the
    stepwise path? -- normally sits between this call's launches and the next call's (a host sync: ~0.12 ms per 6 ms render, and the GPU idles while the
    next call's Python runs).  With the flag set, the device flag is copied to pinned host memory asynchronously and examined when the NEXT ``render`` on this
    decoder has queued its own launches (or in ``finish_render(decoder)``): by then the copy has long landed, nothing waits.  If the flag was raised, the
    batch is redone THEN, into the SAME output tensors.  Contract: the tensors returned by a deferred call are final once the next ``render`` /
    ``finish_render`` on the decoder has returned -- a caller that streams batches (evaluation loops, bench.py) calls ``finish_render`` after the last one -- and
    until then the call's INPUTS (code, bitfield, poses, intrinsics, rays) must not be modified in place: a redo reads them again (they and the outputs stay
    referenced by the decoder until the check has run).

    ``planes`` / ``rays`` let callers that render the same scenes or cameras repeatedly keep the packed planes /
    ray arrays resident instead of rebuilding them (they are pure functions of ``code`` / ``poses, intrinsics``).
    ``return_u8`` (extra): also return the image quantised as ``eval_and_viz`` does (base_nerf.py:551-553), (S,V,h,w,3) uint8 -- written by
    the render kernels next to the float image on the camera-fed path, one ``quantize_u8`` pass otherwise."""
    cfg = cfg or {}
    was_training = decoder.training
    if was_training:                                                  # (nn.Module.train walks the module tree: ~20 us a call, exposed between two renders)
        decoder.train(False)
    dt_gamma_scale = cfg.get("dt_gamma_scale", 0.0)
    s, v = poses.shape[:2]
    if dt_gamma_scale == 0:
        # the uncond configs: the cone angle is 0 for every scene, known on the host -> the constant-step kernel form (shade_mfma.hip MODE 2)
        dt_gamma = [0.0] * s
    else:
        dt_gamma = (dt_gamma_scale * 2 / (intrinsics[..., 0] + intrinsics[..., 1]).mean(dim=-1)).reshape(-1)      # (S,), stays on the device
    max_render_rays = cfg.get("max_render_rays", -1)
    chunked = 0 < max_render_rays < v * h * w
    if planes is None and decoder.render_mode == "fused" and decoder.fused_supported(code):
        planes = pack_triplanes(code, decoder.plane_dtype)
    overflow = []                                                     # device flags of the fused launches: ONE host read after the last chunk
    if planes is not None and rays is None and not chunked and os.environ.get("SSDNERF_RENDER_ARRAYS", "0") != "1":   # (=1: debugging aid, materialise the rays)
        # the whole batch in one fused launch pair, rays generated in the kernels from (poses, intrinsics): no (S,V,h,w,3) arrays at all
        prefetch = None
        if next_batch is not None:                                    # (density_bitfield, intrinsics, poses) of the NEXT render call: its stage A runs beside this one's shading kernel
            n_bits, n_intr, n_poses = next_batch
            if dt_gamma_scale == 0 and tuple(n_poses.shape[:2]) == (s, v):
                prefetch = dict(cams=(n_poses, n_intr.expand(s, v, 4), h, w), density_bitfield=n_bits)
        out = decoder.render_packed(planes, None, None, density_bitfield, grid_size, dt_gamma, 1e-4, bg_color=bg_color,
                                    check_overflow=False, cams=(poses, intrinsics.expand(s, v, 4), h, w), want_u8=return_u8, prefetch=prefetch)
        overflow.append(decoder.last_render_stats["overflow"])
        image, depth = out["image"], out["depth"]
        image_u8 = out.get("image_u8")
    else:
        rays_o, rays_d = get_cam_rays(poses, intrinsics, h, w) if rays is None else rays
        rays_o = rays_o.reshape(s, v * h * w, 3)
        rays_d = rays_d.reshape(s, v * h * w, 3)
        chunks_o = rays_o.split(max_render_rays, dim=1) if chunked else [rays_o]
        chunks_d = rays_d.split(max_render_rays, dim=1) if chunked else [rays_d]
        images, depths = [], []
        gammas, image_u8 = None, None
        for o, d in zip(chunks_o, chunks_d):
            if planes is not None:
                # dt_gamma stays on the device (the reference calls .item() per scene, base_volume_renderer.py:112)
                out = decoder.render_packed(planes, o, d, density_bitfield, grid_size, dt_gamma, 1e-4, bg_color=bg_color,
                                            check_overflow=False)
                overflow.append(decoder.last_render_stats["overflow"])
                rgb = out["image"]                                    # already a dense (S,N,3) tensor: no stack copy
            else:
                gammas = gammas or [float(g) for g in (dt_gamma if isinstance(dt_gamma, list) else dt_gamma.tolist())]
                out = decoder(o, d, code, density_bitfield, grid_size, dt_gamma=gammas, perturb=False)
                ws = torch.stack(out["weights_sum"], dim=0)
                rgb = torch.stack(out["image"], dim=0) + bg_color * (1 - ws.unsqueeze(-1))
            images.append(rgb)
            depths.append(out["depth"] if isinstance(out["depth"], torch.Tensor) else torch.stack(out["depth"], dim=0))
        image = torch.cat(images, dim=1) if len(images) > 1 else images[0]
        depth = torch.cat(depths, dim=1) if len(depths) > 1 else depths[0]
    deferred = defer_overflow_check and len(overflow) == 1 and (not return_u8 or image_u8 is not None)
    # an earlier deferred call's flag: its copy was queued a whole render ago
    _settle_deferred(decoder)
    if deferred:
        st = decoder.__dict__.setdefault("_deferred_overflow", {"ring": [], "i": 0, "pending": None})
        if len(st["ring"]) < 2:
            st["ring"].append(torch.empty((), dtype=torch.int32, pin_memory=True))
        host = st["ring"][st["i"] & 1]
        st["i"] += 1
        host.copy_(overflow[0].reshape(()), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st["pending"] = dict(host=host, event=ev, image=image, depth=depth, image_u8=image_u8 if return_u8 else None,
                             args=(code, density_bitfield, h, w, intrinsics, poses, grid_size, bg_color, rays, dt_gamma, s))
        overflow = []
    # ONE host read per render call (the reference syncs once per loop iteration); a single launch's flag is read as it is -- no stack / sum kernels
    # between the render and the read, which sits on the critical path of back-to-back renders (profiles/r04/l_host_gap_*.txt)
    if overflow and int((overflow[0] if len(overflow) == 1 else torch.stack([f.reshape(()) for f in overflow]).sum()).item()) != 0:
        # a ray reached the reference loop's global step cap (max_steps occupied samples), where the reference's answer depends on its n_step
        # schedule: redo the batch through the reference-shaped stepwise path, which is exact by construction (same rule as
        # TriPlaneDecoder._forward_eval_fused).  One sync per render call; the reference syncs once per loop iteration.
        image, depth = _stepwise_redo(decoder, code, density_bitfield, h, w, intrinsics, poses, grid_size, bg_color, rays, dt_gamma, s)
        image_u8 = None
    image = image.reshape(s, v, h, w, 3)
    depth = depth.reshape(s, v, h, w)
    if was_training:
        decoder.train(True)
    if return_u8:
        return image, depth, (quantize_u8(image) if image_u8 is None else image_u8.reshape(s, v, h, w, 3))
    return image, depth


def _stepwise_redo(decoder, code, density_bitfield, h, w, intrinsics, poses, grid_size, bg_color, rays, dt_gamma, s):
    """the batch through the reference-shaped stepwise path (exact at the global step cap by construction): image (S, N, 3) with background, depth (S, N)"""
    rays_o, rays_d = get_cam_rays(poses, intrinsics, h, w) if rays is None else rays
    gammas = [float(g) for g in (dt_gamma if isinstance(dt_gamma, list) else dt_gamma.tolist())]
    out = decoder._forward_eval_stepwise(list(rays_o.reshape(s, -1, 3)), list(rays_d.reshape(s, -1, 3)), code, density_bitfield,
                                         [grid_size] * s if isinstance(grid_size, int) else grid_size, gammas, False, 1e-4)
    ws = torch.stack(out["weights_sum"], dim=0)
    return torch.stack(out["image"], dim=0) + bg_color * (1 - ws.unsqueeze(-1)), torch.stack(out["depth"], dim=0)


@torch.no_grad()
def _settle_deferred(decoder) -> bool:
    """examine the flag of the decoder's last deferred render; redo that batch into its own output tensors if it was raised.  True if a batch was redone."""
    st = decoder.__dict__.get("_deferred_overflow")
    if not st or st["pending"] is None:
        return False
    p, st["pending"] = st["pending"], None
    p["event"].synchronize()                                          # (recorded a render ago: no wait in a streaming loop)
    if int(p["host"].item()) == 0:
        return False
    image, depth = _stepwise_redo(decoder, *p["args"])
    p["image"].copy_(image.reshape(p["image"].shape))
    p["depth"].copy_(depth.reshape(p["depth"].shape))
    if p["image_u8"] is not None:
        p["image_u8"].copy_(quantize_u8(p["image"]).reshape(p["image_u8"].shape))
    return True


def finish_render(decoder) -> bool:
    """settle the last ``render(..., defer_overflow_check=True)`` on this decoder (see there) and forget a stage A that ``next_batch`` launched ahead for a render
    that will not come.  Returns True if that batch had to be redone."""
    decoder.drop_prefetch()
    return _settle_deferred(decoder)


def quantize_u8(image: torch.Tensor) -> torch.Tensor:
    """clamp [0,1] and round to k/255 as ``eval_and_viz`` does (base_nerf.py:551-553), kept as uint8 (what is all-gathered)."""
    if image.is_cuda and image.dtype == torch.float32 and image.is_contiguous():
        import ctypes
        from . import _cabi as C
        out = torch.empty(image.shape, dtype=torch.uint8, device=image.device)
        C.check(C.lib().ssdnerf_quantize_u8(C.ptr(image), ctypes.c_uint64(image.numel()), C.ptr(out), C.stream()), "quantize_u8")
        return out
    return torch.round(image.clamp(0, 1) * 255).to(torch.uint8)


def eval_psnr(img1: torch.Tensor, img2: torch.Tensor, max_val: float = 1.0, eps: float = 1e-6) -> torch.Tensor:
    """PSNR per batch element with the reference's epsilon (lib/core/evaluation/metrics.py:52-55)."""
    import math
    mse = (img1 - img2).square().flatten(1).mean(dim=-1)
    return 10 * (2 * math.log10(max_val) - torch.log10(mse + eps))


# ---------------------------------------------------------------------------------------------- density volume / mesh
@torch.no_grad()
def extract_fields(bound_min, bound_max, resolution: int, query_func, S: int = 128, device=None) -> torch.Tensor:
    """Sample ``query_func`` on the ``resolution``^3 lattice spanning [bound_min, bound_max] (x-major, ``custom_meshgrid`` order) in
    S^3 chunks (lib/core/utils/nerf_utils.py:64-79).  The volume is assembled ON THE DEVICE and returned as a tensor - the reference
    copies every chunk to the host; callers that want numpy call ``.cpu().numpy()`` once."""
    xs_all = [torch.linspace(float(bound_min[a]), float(bound_max[a]), resolution, device=device) for a in range(3)]
    u = torch.zeros(resolution, resolution, resolution, dtype=torch.float32, device=device)
    for xi, xs in enumerate(xs_all[0].split(S)):
        for yi, ys in enumerate(xs_all[1].split(S)):
            for zi, zs in enumerate(xs_all[2].split(S)):
                xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], dim=-1)
                u[xi * S: xi * S + len(xs), yi * S: yi * S + len(ys), zi * S: zi * S + len(zs)] = query_func(pts).reshape(len(xs), len(ys), len(zs))
    return u


@torch.no_grad()
def extract_density_volume(decoder: TriPlaneDecoder, code_single: torch.Tensor, resolution: int = 256) -> torch.Tensor:
    """Density on the lattice ``extract_geometry`` marches over: the box grown by 0.1 on every side, sigma forced to 0 outside the AABB
    (nerf_utils.py:98-112).  One fused density decode per 128^3 chunk on the GPU."""
    aabb = decoder.aabb.to(code_single.device)

    def query(pts):
        sigma = decoder.point_density_decode(pts[None], code_single[None])[0].flatten()
        out = (pts < aabb[:3]).any(dim=-1) | (pts > aabb[3:]).any(dim=-1)
        return sigma.masked_fill(out, 0)

    return extract_fields(aabb[:3] - 0.1, aabb[3:] + 0.1, resolution, query, device=code_single.device)


def extract_geometry(decoder: TriPlaneDecoder, code_single: torch.Tensor, resolution: int = 256, threshold: float = 10, backend: str = "native"):
    """``vertices, triangles`` of the density iso-surface in world coordinates (nerf_utils.py:82-112).  The marching-cubes step is PyMCubes in
    the reference; here it is native and runs on the GPU over the volume the fused density decode has just assembled there (``mesh.py``,
    csrc/marching_cubes.hip): no 256^3 copy to the host.  ``backend='mcubes'`` calls PyMCubes instead where it is installed (same vertex rule;
    the triangulation of ambiguous cells may differ).  Returns numpy arrays like the reference (vertices float, triangles int)."""
    u = extract_density_volume(decoder, code_single, resolution)
    if backend == "mcubes":
        import mcubes
        vertices, triangles = mcubes.marching_cubes(u.cpu().numpy(), threshold)
    else:
        from .mesh import marching_cubes
        v, t = marching_cubes(u, threshold)
        vertices, triangles = v.cpu().numpy().astype("float64"), t.cpu().numpy()
    b_min = (decoder.aabb[:3] - 0.1).cpu().numpy()
    b_max = (decoder.aabb[3:] + 0.1).cpu().numpy()
    return vertices / (resolution - 1.0) * (b_max - b_min)[None, :] + b_min[None, :], triangles
