"""Drop-in replacement for the reference's pybind11 module ``_raymarching``.

The reference binds ten functions taking ``at::Tensor`` by value (lib/ops/raymarching/src/bindings.cpp:5-18,
signatures lib/ops/raymarching/src/raymarching.h:7-18) and imports them by module name
(``import _raymarching as _backend``, lib/ops/raymarching/raymarching.py:10-13).  Putting this directory on
``sys.path`` makes that very import resolve to the MI355X library: same function names, same positional
arguments, same in-place/caller-allocates contract - each call forwards raw device pointers to the C ABI
(include/ssdnerf_hip.h) on torch's CURRENT stream (the reference uses the legacy default stream).
"""
import torch

from ssdnerf_amd import _cabi as C

_ws_cache = {}


def _workspace(nbytes, device):
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _f32(*ts):
    for t in ts:
        if t.dtype != torch.float32:
            raise RuntimeError(f"expected a float32 tensor, got {t.dtype} (the reference wrappers cast to fp32 before the call)")
        if not t.is_contiguous():
            raise RuntimeError("expected a contiguous tensor")
    C.require_cuda(*ts)


def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
    _f32(rays_o, rays_d, aabb, nears, fars)
    C.check(C.lib().ssdnerf_near_far_from_aabb(C.ptr(rays_o), C.ptr(rays_d), C.ptr(aabb), C.u32(N), C.f32(min_near), C.ptr(nears),
                                               C.ptr(fars), C.stream()), "near_far_from_aabb")


def sph_from_ray(rays_o, rays_d, radius, N, coords):
    _f32(rays_o, rays_d, coords)
    C.check(C.lib().ssdnerf_sph_from_ray(C.ptr(rays_o), C.ptr(rays_d), C.f32(radius), C.u32(N), C.ptr(coords), C.stream()), "sph_from_ray")


def morton3D(coords, N, indices):
    C.require_cuda(coords, indices)
    assert coords.dtype == torch.int32 and indices.dtype == torch.int32
    C.check(C.lib().ssdnerf_morton3D(C.ptr(coords.contiguous()), C.u32(N), C.ptr(indices), C.stream()), "morton3D")


def morton3D_invert(indices, N, coords):
    C.require_cuda(coords, indices)
    assert coords.dtype == torch.int32 and indices.dtype == torch.int32
    C.check(C.lib().ssdnerf_morton3D_invert(C.ptr(indices.contiguous()), C.u32(N), C.ptr(coords), C.stream()), "morton3D_invert")


def packbits(grid, N, density_thresh, bitfield):
    C.require_cuda(grid, bitfield)
    assert bitfield.dtype == torch.uint8 and grid.is_contiguous()
    C.check(C.lib().ssdnerf_packbits(C.ptr(grid), C.dtype_code(grid), C.u32(N), C.f32(density_thresh), C.ptr(bitfield), C.stream()), "packbits")


def march_rays_train(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C_, H, M, nears, fars, xyzs, dirs, deltas, rays, counter, noises):
    _f32(rays_o, rays_d, nears, fars, xyzs, dirs, deltas, noises)
    C.require_cuda(grid, rays, counter)
    need = C.lib().ssdnerf_march_rays_train_workspace(C.u32(N))
    ws = _workspace(need, rays_o.device)
    C.check(C.lib().ssdnerf_march_rays_train(C.ptr(rays_o), C.ptr(rays_d), C.ptr(grid), C.f32(bound), C.f32(dt_gamma), C.u32(max_steps),
                                             C.u32(N), C.u32(C_), C.u32(H), C.u32(M), C.ptr(nears), C.ptr(fars), C.ptr(xyzs), C.ptr(dirs),
                                             C.ptr(deltas), C.ptr(rays), C.ptr(counter), C.ptr(noises), C.ptr(ws),
                                             C.ctypes.c_size_t(ws.numel()), C.stream()), "march_rays_train")


def composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image):
    _f32(sigmas, rgbs, deltas, weights_sum, depth, image)
    C.check(C.lib().ssdnerf_composite_rays_train_forward(C.ptr(sigmas), C.ptr(rgbs), C.ptr(deltas), C.ptr(rays), C.u32(M), C.u32(N),
                                                         C.f32(T_thresh), C.ptr(weights_sum), C.ptr(depth), C.ptr(image), C.stream()),
            "composite_rays_train_forward")


def composite_rays_train_backward(grad_weights_sum, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, M, N, T_thresh,
                                  grad_sigmas, grad_rgbs):
    _f32(grad_weights_sum, grad_image, sigmas, rgbs, deltas, weights_sum, image, grad_sigmas, grad_rgbs)
    C.check(C.lib().ssdnerf_composite_rays_train_backward(C.ptr(grad_weights_sum), C.ptr(grad_image), C.ptr(sigmas), C.ptr(rgbs),
                                                          C.ptr(deltas), C.ptr(rays), C.ptr(weights_sum), C.ptr(image), C.u32(M), C.u32(N),
                                                          C.f32(T_thresh), C.ptr(grad_sigmas), C.ptr(grad_rgbs), C.stream()),
            "composite_rays_train_backward")


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C_, H, grid, nears, fars, xyzs, dirs,
               deltas, noises):
    _f32(rays_t, rays_o, rays_d, nears, fars, xyzs, dirs, deltas, noises)
    C.require_cuda(rays_alive, grid)
    C.check(C.lib().ssdnerf_march_rays(C.u32(n_alive), C.u32(n_step), C.ptr(rays_alive), C.ptr(rays_t), C.ptr(rays_o), C.ptr(rays_d),
                                       C.f32(bound), C.f32(dt_gamma), C.u32(max_steps), C.u32(C_), C.u32(H), C.ptr(grid), C.ptr(nears),
                                       C.ptr(fars), C.ptr(xyzs), C.ptr(dirs), C.ptr(deltas), C.ptr(noises), C.stream()), "march_rays")


def composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image):
    _f32(rays_t, sigmas, rgbs, deltas, weights_sum, depth, image)
    C.require_cuda(rays_alive)
    C.check(C.lib().ssdnerf_composite_rays(C.u32(n_alive), C.u32(n_step), C.f32(T_thresh), C.ptr(rays_alive), C.ptr(rays_t), C.ptr(sigmas),
                                           C.ptr(rgbs), C.ptr(deltas), C.ptr(weights_sum), C.ptr(depth), C.ptr(image), C.stream()),
            "composite_rays")
