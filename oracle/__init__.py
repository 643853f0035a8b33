"""oracle/ -- TEST INFRASTRUCTURE.  CPU restatement of the reference's hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  ``ssdnerf_amd`` (the product) never does, and fails loudly without its HIP library.

Layout
------
raymarching_oracle.c / shencoder_oracle.c   plain-C restatement of the reference's two CUDA
                                            extensions (each function cites reference file:line)
decoder.py                                  triplane gather + tiny MLP, PyTorch-CPU (the gather
                                            arithmetic of the reference lives in ATen's grid_sample)
render.py                                   reference-shaped host loops (VolumeRenderer eval/train
                                            branch, density grid, camera rays)
diffusion.py                                DDIM schedule / sampler restatement + UNet (torch CPU)
build_ref.sh, ref_glue.cpp, ref_shim/       recipe that compiles the REFERENCE'S OWN .cu files for
                                            the CPU into oracle/_ref/ (git-ignored) -- the pin

Parity status: the reference ships no tests or golden vectors, so the pin is
(a) oracle/_ref = the reference's kernels themselves executed on the CPU, and (b) golden vectors
under tests/golden/ produced by importing the reference's Python modules (tests/golden/make_golden.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIB_PATH = os.path.join(_BUILD, "libssdnerf_oracle.so")
_SOURCES = ["raymarching_oracle.c", "shencoder_oracle.c"]

_lib: Optional[ctypes.CDLL] = None


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (-ffp-contract=off: see the arithmetic contract)."""
    srcs = [os.path.join(_HERE, s) for s in _SOURCES]
    if not force and os.path.exists(_LIB_PATH) and all(
            os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return _LIB_PATH
    os.makedirs(_BUILD, exist_ok=True)
    cmd = ["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-fno-fast-math",
           "-mfma", "-o", _LIB_PATH] + srcs + ["-lm"]
    subprocess.check_call(cmd)
    return _LIB_PATH


def build_ref() -> None:
    """Compile the reference's own kernels for the CPU (no-op when /root/reference is absent)."""
    subprocess.check_call(["bash", os.path.join(_HERE, "build_ref.sh")])


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def set_threads(n: int) -> None:
    """Cap the oracle's OpenMP team size (the GPU box has 256 host cores; the per-iteration work is small)."""
    l = lib()
    l.orc_set_threads.restype = None
    l.orc_set_threads(ctypes.c_int(int(n)))


def ref_lib(tag: str = "fma") -> Optional[ctypes.CDLL]:
    """The reference's kernels compiled for the CPU, or None when oracle/_ref was never built."""
    p = os.path.join(_HERE, "_ref", f"libssdnerf_ref_{tag}.so")
    return ctypes.CDLL(p) if os.path.exists(p) else None


# ----------------------------------------------------------------------------------------------
# numpy front-end: identical call signatures for the oracle ("orc_") and the compiled reference
# ("ref_"), so tests can run the same inputs through both.
# ----------------------------------------------------------------------------------------------
def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_u32 = ctypes.c_uint32
_cf = ctypes.c_float


class Ops:
    """Array-level API over either library.  prefix='orc' (oracle) or 'ref' (compiled reference)."""

    def __init__(self, cdll: ctypes.CDLL, prefix: str):
        self._l = cdll
        self._pre = prefix

    def _fn(self, name):
        f = getattr(self._l, f"{self._pre}_{name}")
        f.restype = None
        return f

    def near_far_from_aabb(self, rays_o, rays_d, aabb, min_near=0.2):
        rays_o, rays_d, aabb = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3), _f32(aabb)
        n = rays_o.shape[0]
        nears, fars = np.empty(n, np.float32), np.empty(n, np.float32)
        self._fn("near_far_from_aabb")(_p(rays_o), _p(rays_d), _p(aabb), _u32(n), _cf(min_near), _p(nears), _p(fars))
        return nears, fars

    def sph_from_ray(self, rays_o, rays_d, radius):
        rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
        n = rays_o.shape[0]
        coords = np.empty((n, 2), np.float32)
        self._fn("sph_from_ray")(_p(rays_o), _p(rays_d), _cf(radius), _u32(n), _p(coords))
        return coords

    def morton3D(self, coords):
        coords = _i32(coords).reshape(-1, 3)
        out = np.empty(coords.shape[0], np.int32)
        self._fn("morton3D")(_p(coords), _u32(coords.shape[0]), _p(out))
        return out

    def morton3D_invert(self, indices):
        indices = _i32(indices).reshape(-1)
        out = np.empty((indices.shape[0], 3), np.int32)
        self._fn("morton3D_invert")(_p(indices), _u32(indices.shape[0]), _p(out))
        return out

    def packbits(self, grid, thresh):
        grid = _f32(grid).reshape(-1)
        n = grid.shape[0] // 8
        out = np.empty(n, np.uint8)
        self._fn("packbits")(_p(grid), _u32(n), _cf(thresh), _p(out))
        return out

    def march_rays_train(self, rays_o, rays_d, bitfield, bound, dt_gamma, max_steps, C, H, nears, fars, noises,
                         M=None, counter=None):
        rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
        n = rays_o.shape[0]
        M = n * max_steps if M is None else M
        xyzs, dirs = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32)
        deltas = np.zeros((M, 2), np.float32)
        rays = np.empty((n, 3), np.int32)
        counter = np.zeros(2, np.int32) if counter is None else counter
        bitfield, nears, fars, noises = _u8(bitfield), _f32(nears), _f32(fars), _f32(noises)
        self._fn("march_rays_train")(_p(rays_o), _p(rays_d), _p(bitfield), _cf(bound), _cf(dt_gamma), _u32(max_steps),
                                     _u32(n), _u32(C), _u32(H), _u32(M), _p(nears), _p(fars), _p(xyzs), _p(dirs),
                                     _p(deltas), _p(rays), _p(counter), _p(noises))
        return xyzs, dirs, deltas, rays, counter

    def composite_rays_train_forward(self, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        sigmas, rgbs, deltas, rays = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
        M, n = sigmas.shape[0], rays.shape[0]
        ws, depth, image = np.empty(n, np.float32), np.empty(n, np.float32), np.empty((n, 3), np.float32)
        self._fn("composite_rays_train_forward")(_p(sigmas), _p(rgbs), _p(deltas), _p(rays), _u32(M), _u32(n),
                                                 _cf(T_thresh), _p(ws), _p(depth), _p(image))
        return ws, depth, image

    def composite_rays_train_backward(self, grad_ws, grad_image, sigmas, rgbs, deltas, rays, ws, image, T_thresh=1e-4):
        sigmas, rgbs, deltas, rays = _f32(sigmas), _f32(rgbs), _f32(deltas), _i32(rays)
        M, n = sigmas.shape[0], rays.shape[0]
        gs, gc = np.zeros(M, np.float32), np.zeros((M, 3), np.float32)
        self._fn("composite_rays_train_backward")(_p(_f32(grad_ws)), _p(_f32(grad_image)), _p(sigmas), _p(rgbs), _p(deltas),
                                                  _p(rays), _p(_f32(ws)), _p(_f32(image)), _u32(M), _u32(n), _cf(T_thresh),
                                                  _p(gs), _p(gc))
        return gs, gc

    def march_rays(self, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, bitfield, C, H, nears, fars,
                   align=-1, dt_gamma=0.0, max_steps=1024, noises=None):
        rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
        M = n_alive * n_step
        if align > 0:
            M += align - (M % align)  # adds a full `align` when already aligned (raymarching.py:437-438)
        xyzs, dirs = np.zeros((M, 3), np.float32), np.zeros((M, 3), np.float32)
        deltas = np.zeros((M, 2), np.float32)
        noises = np.zeros(n_alive, np.float32) if noises is None else _f32(noises)
        self._fn("march_rays")(_u32(n_alive), _u32(n_step), _p(_i32(rays_alive)), _p(_f32(rays_t)), _p(rays_o), _p(rays_d),
                               _cf(bound), _cf(dt_gamma), _u32(max_steps), _u32(C), _u32(H), _p(_u8(bitfield)),
                               _p(_f32(nears)), _p(_f32(fars)), _p(xyzs), _p(dirs), _p(deltas), _p(noises))
        return xyzs, dirs, deltas

    def composite_rays(self, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                       T_thresh=1e-4):
        """In place on rays_alive (int32), rays_t, weights_sum, depth, image (all must be C-contiguous arrays)."""
        for a, dt in ((rays_alive, np.int32), (rays_t, np.float32), (weights_sum, np.float32), (depth, np.float32),
                      (image, np.float32)):
            assert a.dtype == dt and a.flags["C_CONTIGUOUS"]
        self._fn("composite_rays")(_u32(n_alive), _u32(n_step), _cf(T_thresh), _p(rays_alive), _p(rays_t), _p(_f32(sigmas)),
                                   _p(_f32(rgbs)), _p(_f32(deltas)), _p(weights_sum), _p(depth), _p(image))

    def sh_encode_forward(self, inputs, degree, calc_grad_inputs=False):
        inputs = _f32(inputs).reshape(-1, 3)
        b, c2 = inputs.shape[0], degree * degree
        out = np.empty((b, c2), np.float32)
        dy_dx = np.empty((b, 3 * c2), np.float32) if calc_grad_inputs else np.empty(1, np.float32)
        self._fn("sh_encode_forward")(_p(inputs), _p(out), _u32(b), _u32(3), _u32(degree), ctypes.c_int(int(calc_grad_inputs)),
                                      _p(dy_dx))
        return (out, dy_dx) if calc_grad_inputs else out

    def sh_encode_backward(self, grad, inputs, degree, dy_dx):
        inputs, grad, dy_dx = _f32(inputs).reshape(-1, 3), _f32(grad), _f32(dy_dx)
        gi = np.zeros_like(inputs)
        self._fn("sh_encode_backward")(_p(grad), _p(inputs), _u32(inputs.shape[0]), _u32(3), _u32(degree), _p(dy_dx), _p(gi))
        return gi


def ops() -> Ops:
    return Ops(lib(), "orc")


def ref_ops(tag: str = "fma") -> Optional[Ops]:
    l = ref_lib(tag)
    return Ops(l, "ref") if l is not None else None
