"""ctypes binding of libssdnerf_hip.so (the C ABI declared in include/ssdnerf_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; the library receives raw
device pointers.  There is NO fallback: if the shared library is missing or fails to load, importing
any operator raises immediately (the product path never routes through the CPU oracle).
"""
from __future__ import annotations

import ctypes  # re-exported as C.ctypes for the drop-in modules
import os
from typing import Optional

import torch

# SSDNERF_HIP_LIB: another build of the SAME library (e.g. one compiled with experimental -D flags into .variants/) for A/B runs
_LIB_PATH = os.environ.get("SSDNERF_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libssdnerf_hip.so")
_lib: Optional[ctypes.CDLL] = None

ABI_VERSION = 3
F32, F16 = 0, 1

EXPORTS = [
    "ssdnerf_last_error", "ssdnerf_abi_version", "ssdnerf_near_far_from_aabb", "ssdnerf_sph_from_ray", "ssdnerf_morton3D",
    "ssdnerf_morton3D_invert", "ssdnerf_packbits", "ssdnerf_march_rays_train_workspace", "ssdnerf_march_rays_train",
    "ssdnerf_march_rays_train_batch_workspace", "ssdnerf_march_rays_train_batch_count", "ssdnerf_march_rays_train_batch_write",
    "ssdnerf_composite_rays_train_forward", "ssdnerf_composite_rays_train_backward", "ssdnerf_march_rays", "ssdnerf_composite_rays",
    "ssdnerf_sh_encode_forward", "ssdnerf_sh_encode_backward", "ssdnerf_triplane_pack", "ssdnerf_point_decode", "ssdnerf_point_decode_backward_workspace",
    "ssdnerf_point_decode_backward",
    "ssdnerf_render_rays_fused", "ssdnerf_render_rays_fused_batch", "ssdnerf_render_queue_workspace", "ssdnerf_render_first_hit",
    "ssdnerf_render_shade_queue", "ssdnerf_render_shade_queue_mfma", "ssdnerf_render_first_hit_cams", "ssdnerf_render_shade_queue_mfma_cams", "ssdnerf_density_grid_update", "ssdnerf_packbits_dev_thresh", "ssdnerf_ddim_step_v",
    "ssdnerf_group_norm_workspace", "ssdnerf_group_norm_backward_workspace", "ssdnerf_group_norm_nhwc", "ssdnerf_group_norm_nhwc_runs", "ssdnerf_group_norm_nhwc_backward", "ssdnerf_group_norm_nhwc_backward_cat", "ssdnerf_bias_residual_nhwc",
    "ssdnerf_conv2d_nhwc_bf16_supported", "ssdnerf_conv2d_nhwc_bf16_plan", "ssdnerf_conv2d_nhwc_bf16", "ssdnerf_conv2d_nhwc_f32x2", "ssdnerf_conv2d_nhwc_f32x2_plan", "ssdnerf_attention_qkv_bf16", "ssdnerf_attention_qkv_f32", "ssdnerf_attention_qkv_f32_lse", "ssdnerf_attention_qkv_f32_backward", "ssdnerf_cam_rays", "ssdnerf_quantize_u8",
    "ssdnerf_marching_cubes_count", "ssdnerf_marching_cubes_emit", "ssdnerf_conv2d_nhwc_f32x2_presplit_supported", "ssdnerf_conv2d_nhwc_f32x2_presplit", "ssdnerf_split_f32_nhwc",
]


def lib_path() -> str:
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"libssdnerf_hip.so not found at {_LIB_PATH}: build it with `python -m ssdnerf_amd.build` "
                "(there is no CPU fallback for the product path)")
        l = ctypes.CDLL(_LIB_PATH)
        l.ssdnerf_last_error.restype = ctypes.c_char_p
        l.ssdnerf_abi_version.restype = ctypes.c_int
        l.ssdnerf_march_rays_train_workspace.restype = ctypes.c_size_t
        l.ssdnerf_march_rays_train_workspace.argtypes = [ctypes.c_uint32]
        l.ssdnerf_render_queue_workspace.restype = ctypes.c_size_t
        l.ssdnerf_render_queue_workspace.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        l.ssdnerf_march_rays_train_batch_workspace.restype = ctypes.c_size_t
        l.ssdnerf_march_rays_train_batch_workspace.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        l.ssdnerf_point_decode_backward_workspace.restype = ctypes.c_size_t
        l.ssdnerf_point_decode_backward_workspace.argtypes = [ctypes.c_uint32] * 4
        l.ssdnerf_group_norm_workspace.restype = ctypes.c_size_t
        l.ssdnerf_group_norm_workspace.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        l.ssdnerf_group_norm_backward_workspace.restype = ctypes.c_size_t
        l.ssdnerf_group_norm_backward_workspace.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
        if l.ssdnerf_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libssdnerf_hip.so ABI {l.ssdnerf_abi_version()} != expected {ABI_VERSION}: rebuild")
        _lib = l
    return _lib


_SYNC_CHECK = os.environ.get("SSDNERF_SYNC_CHECK", "0") == "1"      # debugging aid: synchronise after every library call and name it first


def check(status: int, what: str) -> None:
    if status != 0:
        raise RuntimeError(f"{what} failed ({status}): {lib().ssdnerf_last_error().decode()}")
    if _SYNC_CHECK and not torch.cuda.is_current_stream_capturing():
        import sys
        print(f"[ssdnerf sync-check] {what}", file=sys.stderr, flush=True)
        torch.cuda.synchronize()


def ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def u32(v) -> ctypes.c_uint32:
    return ctypes.c_uint32(int(v))


def f32(v) -> ctypes.c_float:
    return ctypes.c_float(float(v))


def dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float16:
        return F16
    raise RuntimeError(f"unsupported dtype {t.dtype} (fp32 / fp16 only)")


def require_cuda(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("ssdnerf_amd operators need tensors on the GPU (cuda:N == HIP device N)")
