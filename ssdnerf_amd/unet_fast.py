"""Inference executor for ``DenoisingUnetMod`` (reference: lib/models/architecture/ddpm/denoising.py:191-216 forward,
lib/models/architecture/ddpm/modules.py:12-129 blocks).

The DDIM loop calls the UNet 50-75 times per batch of scenes with fixed shapes and no autograd.  The module tree in
``unet.py`` stays the source of truth (state-dict keys, training, the guided path that needs gradients); this executor is
what ``DenoisingUnetMod.forward`` runs under ``torch.no_grad()`` on the GPU:

* activations stay **channel-last** in the compute dtype from the first convolution to the last (no NCHW<->NHWC
  transposes around MIOpen's NHWC kernels, no autocast casts: weights are converted once);
* every GroupNorm (+ scale/shift from the time embedding, + SiLU) is the fused HIP kernel pair of ``csrc/groupnorm.hip``
  (C ABI ``ssdnerf_group_norm_nhwc``) -- 2 reads + 1 write of the activation instead of the eager chain's 4-7 kernels;
* the 22 per-block projections of the time embedding are one GEMM per forward;
* attention consumes the channel-last activation as ``[B, T, C]`` directly (qkv / proj are plain GEMMs, no reshapes of the
  big tensors), per-head layout ``[q | k | v]`` as in modules.py:40-42;
* the whole forward is captured once per (batch, dtype) into a hipGraph and replayed (``torch.cuda.CUDAGraph``), removing
  ~600 launches' worth of host latency per step.

Weights are re-packed when any parameter's version counter changes (optimizer step, ``load_state_dict``).
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import _cabi as C

_GN_DTYPE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}

#: while an executor traces its first forward this is a list that every LIBRARY call inside the block functions (MIOpen convolution, library
#: GEMM for a 1x1 projection, scaled_dot_product_attention) appends its name to; ``FastUnet.library_fallbacks`` is its length (0 = every
#: convolution / projection / attention of the forward ran on the hand-written kernels)
_FALLBACK_LOG: Optional[list] = None


def _lib_call(what: str):
    if _FALLBACK_LOG is not None:
        _FALLBACK_LOG.append(what)


def group_norm_nhwc(x: torch.Tensor, groups: int, gamma: torch.Tensor, beta: torch.Tensor, scale_shift: Optional[torch.Tensor], eps: float,
                    act: bool, workspace: torch.Tensor, out: Optional[torch.Tensor] = None, pre_bias: Optional[torch.Tensor] = None,
                    workspace_is_zero: bool = False, stats_ready: bool = False, x2: Optional[torch.Tensor] = None,
                    runs: Optional[Tuple[torch.Tensor, Optional[torch.Tensor]]] = None, split_out: bool = False) -> torch.Tensor:
    """``x``: (B, C, H, W) tensor in channels_last memory format, or (B, T, C) contiguous.  ``scale_shift``: fp32 view (B, 2C) whose
    rows may be strided.  ``pre_bias``: fp32 (C,) added to x before the norm.  ``stats_ready``: ``workspace`` already holds the sums
    (written by the producing convolution's epilogue).  ``x2``: normalise the channel concatenation [x | x2] without building it
    (4-D only).  ``runs`` = (runs of x, runs of x2 or None): statistics per run of 4 channels (fp64 (B, C/4, 2) each, what the convolutions
    write with ``gn_groups = Cout // 4``) instead of ``workspace``; only the normalisation pass runs.  ``split_out`` (fp32, C % 32 == 0): the
    result is written PRE-SPLIT for ``conv2d_nhwc_f32x2_presplit`` (its bytes are not fp32 values any more; same shape and size).
    Returns a tensor of the input's shape/strides (of the concatenation's shape with ``x2``)."""
    act_bits = int(bool(act)) | (2 if split_out else 0)
    if x.dim() == 4:
        B, C1, H, W = x.shape
        HW = H * W
        if not x.is_contiguous(memory_format=torch.channels_last):
            raise RuntimeError("group_norm_nhwc: 4-D input must be channels_last")
        Cc = C1
        if x2 is not None:
            if x2.dim() != 4 or x2.shape[0] != B or x2.shape[2:] != x.shape[2:] or x2.dtype != x.dtype or not x2.is_contiguous(memory_format=torch.channels_last):
                raise RuntimeError("group_norm_nhwc: x2 must match x in batch, spatial size, dtype and layout")
            Cc = C1 + x2.shape[1]
    else:
        B, HW, Cc = x.shape
        C1 = Cc
        if x2 is not None or not x.is_contiguous():
            raise RuntimeError("group_norm_nhwc: 3-D input must be a single contiguous (B, T, C) tensor")
    if out is not None:
        y = out
    elif x2 is None:
        y = torch.empty_like(x)
    else:
        y = torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    ss_stride = 0
    if scale_shift is not None:
        assert scale_shift.dtype == torch.float32 and scale_shift.shape == (B, 2 * Cc) and scale_shift.stride(1) == 1
        ss_stride = scale_shift.stride(0)
    if runs is not None:
        assert pre_bias is None and (x2 is None) == (runs[1] is None)
        C.check(C.lib().ssdnerf_group_norm_nhwc_runs(C.ptr(x), C.ptr(x2), C.u32(C1), _GN_DTYPE[x.dtype], C.u32(B), C.u32(HW), C.u32(Cc), C.u32(groups), C.ptr(gamma),
                                                      C.ptr(beta), C.ptr(scale_shift), C.u32(ss_stride), C.f32(eps), act_bits, C.ptr(runs[0]), C.ptr(runs[1]),
                                                      C.ptr(y), C.stream()), "group_norm_nhwc_runs")
        return y
    C.check(C.lib().ssdnerf_group_norm_nhwc(C.ptr(x), C.ptr(x2), C.u32(C1), _GN_DTYPE[x.dtype], C.u32(B), C.u32(HW), C.u32(Cc), C.u32(groups), C.ptr(pre_bias), C.ptr(gamma),
                                             C.ptr(beta), C.ptr(scale_shift), C.u32(ss_stride), C.f32(eps), act_bits, C.ptr(workspace),
                                             2 if stats_ready else int(bool(workspace_is_zero)), C.ptr(y), C.stream()),
            "group_norm_nhwc")
    return y


def group_norm_backward_workspace_doubles(B: int, groups: int) -> int:
    """doubles of zeroed workspace one ``group_norm_nhwc_backward`` call takes (r06: the statistics pass adds to several copies of the sums)"""
    return int(C.lib().ssdnerf_group_norm_backward_workspace(C.u32(B), C.u32(groups))) // 8


def group_norm_nhwc_backward(x: torch.Tensor, dy: torch.Tensor, groups: int, gamma: torch.Tensor, beta: torch.Tensor, scale_shift: Optional[torch.Tensor],
                             eps: float, act: bool, fwd_sums: torch.Tensor, workspace: Optional[torch.Tensor] = None, split_out: bool = False,
                             sums_are_runs: bool = False) -> torch.Tensor:
    """d/dx of ``group_norm_nhwc`` (single source, no pre_bias) for frozen gamma / beta / scale_shift (csrc/groupnorm.hip, k_gn_bwd_*).
    ``x``, ``dy``: (B, C, H, W) channels_last, same dtype; ``fwd_sums``: the forward's workspace (fp64, B * groups * 2).  ``split_out`` (fp32,
    C % 32 == 0): dx is written PRE-SPLIT for the backward-data convolution that consumes it (``conv2d_nhwc_f32x2_presplit``).  ``sums_are_runs``: ``fwd_sums`` is the
    run-level statistics tensor (B, C / 4, 2) that ``group_norm_nhwc(runs=...)`` read (no reduction to groups on the host side)."""
    if x.dim() != 4 or not x.is_contiguous(memory_format=torch.channels_last) or dy.shape != x.shape or dy.dtype != x.dtype \
            or not dy.is_contiguous(memory_format=torch.channels_last):
        raise RuntimeError("group_norm_nhwc_backward: x and dy must be channels_last 4-D tensors of the same shape and dtype")
    B, Cc, H, W = x.shape
    ss_stride = 0
    if scale_shift is not None:
        assert scale_shift.dtype == torch.float32 and scale_shift.shape == (B, 2 * Cc) and scale_shift.stride(1) == 1
        ss_stride = scale_shift.stride(0)
    dx = torch.empty_like(x)
    need = group_norm_backward_workspace_doubles(B, groups)
    ws = workspace if workspace is not None else torch.zeros(need, dtype=torch.float64, device=x.device)   # (all zero on entry)
    if ws.numel() < need or ws.dtype != torch.float64:
        raise RuntimeError(f"group_norm_nhwc_backward: the workspace must hold {need} zeroed doubles (group_norm_backward_workspace_doubles)")
    C.check(C.lib().ssdnerf_group_norm_nhwc_backward(C.ptr(x), C.ptr(dy), _GN_DTYPE[x.dtype], C.u32(B), C.u32(H * W), C.u32(Cc), C.u32(groups), C.ptr(gamma),
                                                      C.ptr(beta), C.ptr(scale_shift), C.u32(ss_stride), C.f32(eps), int(bool(act)) | (2 if split_out else 0) | (4 if sums_are_runs else 0), C.ptr(fwd_sums), C.ptr(ws), 1,
                                                      C.ptr(dx), C.stream()), "group_norm_nhwc_backward")
    return dx


def group_norm_nhwc_backward_cat(x: torch.Tensor, x2: torch.Tensor, dy: torch.Tensor, groups: int, gamma: torch.Tensor, beta: torch.Tensor,
                                 scale_shift: Optional[torch.Tensor], eps: float, act: bool, fwd_sums: torch.Tensor, fwd_sums2: Optional[torch.Tensor] = None,
                                 workspace: Optional[torch.Tensor] = None):
    """``group_norm_nhwc_backward`` for a norm over the never-built concatenation [x | x2] (``group_norm_nhwc(x, ..., x2=x2)``): returns (dx, dx2), two dense
    channels_last tensors.  ``fwd_sums2``: x2's run-level statistics (then ``fwd_sums`` is x's, both (B, C_i / 4, 2)); None: ``fwd_sums`` holds per-group sums."""
    ok = lambda t: t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)
    if not (ok(x) and ok(x2) and ok(dy)) or x.dtype != x2.dtype or x.dtype != dy.dtype or x.shape[0] != x2.shape[0] or x.shape[2:] != x2.shape[2:] \
            or dy.shape != (x.size(0), x.size(1) + x2.size(1), x.size(2), x.size(3)):
        raise RuntimeError("group_norm_nhwc_backward_cat: x, x2 and dy must be channels_last 4-D tensors of one dtype, dy over the concatenated channels")
    B, C1, H, W = x.shape
    Cc = C1 + x2.size(1)
    ss_stride = 0
    if scale_shift is not None:
        assert scale_shift.dtype == torch.float32 and scale_shift.shape == (B, 2 * Cc) and scale_shift.stride(1) == 1
        ss_stride = scale_shift.stride(0)
    dx, dx2 = torch.empty_like(x), torch.empty_like(x2)
    need = group_norm_backward_workspace_doubles(B, groups)
    ws = workspace if workspace is not None else torch.zeros(need, dtype=torch.float64, device=x.device)
    if ws.numel() < need or ws.dtype != torch.float64:
        raise RuntimeError(f"group_norm_nhwc_backward_cat: the workspace must hold {need} zeroed doubles")
    C.check(C.lib().ssdnerf_group_norm_nhwc_backward_cat(C.ptr(x), C.ptr(x2), C.u32(C1), C.ptr(dy), _GN_DTYPE[x.dtype], C.u32(B), C.u32(H * W), C.u32(Cc), C.u32(groups),
                                                          C.ptr(gamma), C.ptr(beta), C.ptr(scale_shift), C.u32(ss_stride), C.f32(eps),
                                                          int(bool(act)) | (4 if fwd_sums2 is not None else 0), C.ptr(fwd_sums), C.ptr(fwd_sums2), C.ptr(ws), 1, C.ptr(dx), C.ptr(dx2),
                                                          C.stream()), "group_norm_nhwc_backward_cat")
    return dx, dx2


def bias_residual_nhwc(x: torch.Tensor, bias: Optional[torch.Tensor], residual: Optional[torch.Tensor], gn_sums: Optional[torch.Tensor] = None,
                       gn_groups: int = 0) -> torch.Tensor:
    """In place ``x += bias[c] + residual`` for a channels_last (B, C, H, W) or contiguous (B, T, C) tensor (fp32 bias, residual of x's
    dtype and layout); ``gn_sums`` (zeroed fp64 (B, groups, 2)) receives the GroupNorm sums of the result."""
    if x.dim() == 4:
        B, Cc, H, W = x.shape
        HW = H * W
        ok = x.is_contiguous(memory_format=torch.channels_last) and (residual is None or residual.is_contiguous(memory_format=torch.channels_last))
    else:
        B, HW, Cc = x.shape
        ok = x.is_contiguous() and (residual is None or residual.is_contiguous())
    if not ok or (residual is not None and (residual.shape != x.shape or residual.dtype != x.dtype)):
        raise RuntimeError("bias_residual_nhwc: input (and residual) must be channel-last with matching shape and dtype")
    C.check(C.lib().ssdnerf_bias_residual_nhwc(C.ptr(x), _GN_DTYPE[x.dtype], C.u32(B), C.u32(HW), C.u32(Cc), C.ptr(bias), C.ptr(residual), C.ptr(x),
                                                C.ptr(gn_sums), C.u32(gn_groups), C.stream()), "bias_residual_nhwc")
    return x


def conv2d_nhwc_bf16(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, stride: int = 1,
                     upsample: bool = False, gn_sums: Optional[torch.Tensor] = None, gn_groups: int = 0, tile_hint: int = 0,
                     splitk_ws: Optional[torch.Tensor] = None, splits_hint: int = 0, x2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Hand-written implicit-GEMM convolution (csrc/conv_igemm.hip).  ``x`` (B, Cin, H, W) bf16 channels_last, ``w`` (Cout, Cin, k, k) bf16
    channels_last, ``bias`` fp32, ``residual`` like the output.  ``upsample``: convolve the nearest-2x upsampling of x without building it.
    ``splitk_ws``: all-zero fp32 scratch (left all zero) that lets small layers be cut along K.  ``x2``: convolve the channel
    concatenation [x | x2] without building it."""
    if x.dtype != torch.bfloat16 or w.dtype != torch.bfloat16:
        raise RuntimeError("conv2d_nhwc_bf16: bf16 tensors only")
    if not x.is_contiguous(memory_format=torch.channels_last) or not w.is_contiguous(memory_format=torch.channels_last):
        raise RuntimeError("conv2d_nhwc_bf16: input and weight must be channels_last")
    B, Cin1, H, W = x.shape
    Cin = Cin1
    if x2 is not None:
        if x2.dtype != torch.bfloat16 or not x2.is_contiguous(memory_format=torch.channels_last) or x2.shape[0] != B or x2.shape[2:] != x.shape[2:]:
            raise RuntimeError("conv2d_nhwc_bf16: x2 must match x in batch, spatial size, dtype and layout")
        Cin = Cin1 + x2.shape[1]
    Cout, Cin_w, k, k2 = w.shape
    if Cin_w != Cin or k != k2:
        raise RuntimeError("conv2d_nhwc_bf16: weight shape does not match the input")
    Hv, Wv = (2 * H, 2 * W) if upsample else (H, W)
    pad = k // 2
    Ho, Wo = (Hv + 2 * pad - k) // stride + 1, (Wv + 2 * pad - k) // stride + 1
    y = torch.empty((B, Cout, Ho, Wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    if residual is not None and (residual.shape != y.shape or residual.dtype != y.dtype or not residual.is_contiguous(memory_format=torch.channels_last)):
        raise RuntimeError("conv2d_nhwc_bf16: residual must match the output's shape, dtype and layout")
    C.check(C.lib().ssdnerf_conv2d_nhwc_bf16(C.ptr(x), C.ptr(x2), C.u32(Cin1), C.ptr(w), C.ptr(bias), C.ptr(residual), C.ptr(y), C.u32(B), C.u32(H), C.u32(W), C.u32(Cin), C.u32(Cout),
                                              C.u32(k), C.u32(stride), C.u32(int(upsample)), C.ptr(gn_sums), C.u32(gn_groups), int(tile_hint), C.ptr(splitk_ws),
                                              ctypes.c_size_t(0 if splitk_ws is None else splitk_ws.numel() * splitk_ws.element_size()), int(splits_hint),
                                              C.stream()),
            "conv2d_nhwc_bf16")
    return y


def split_bf16x2(w: torch.Tensor):
    """fp32 -> (hi, lo) bf16 pair with hi + lo == w to 16 significand bits (round to nearest for both terms)."""
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return hi, lo


def split_bf16x2_adjacent(w: torch.Tensor):
    """``split_bf16x2`` of a (Cout, Cin, k, k) weight as two channels_last views of ONE buffer, lo directly behind hi: what the two-group fp32
    kernel needs (csrc/conv_igemm.hip, k_conv_pp_bf16<ROWS, F32>: one buffer descriptor serves both terms)."""
    hi, lo = split_bf16x2(w)
    buf = torch.empty((2, w.shape[0], w.shape[2], w.shape[3], w.shape[1]), dtype=torch.bfloat16, device=w.device)
    buf[0].copy_(hi.permute(0, 2, 3, 1))
    buf[1].copy_(lo.permute(0, 2, 3, 1))
    return buf[0].permute(0, 3, 1, 2), buf[1].permute(0, 3, 1, 2)


def presplit_supported(x: torch.Tensor, cout: int, k: int, with_stats: bool = False) -> int:
    """whether the k x k / stride 1 convolution of the channels_last fp32 tensor ``x`` to ``cout`` channels is a layer a kernel takes on PRE-SPLIT
    activations (``conv2d_nhwc_f32x2_presplit``; 1: the two-group row kernel, 2: the generic DMA-ring kernel's PS form, 0: none): the norm that
    produces x then writes its result with ``split_out=True``"""
    if not (x.is_cuda and x.dtype == torch.float32):
        return 0
    return int(C.lib().ssdnerf_conv2d_nhwc_f32x2_presplit_supported(
        C.u32(x.size(0)), C.u32(x.size(2)), C.u32(x.size(3)), C.u32(x.size(1)), C.u32(cout), C.u32(k), int(with_stats)))


def nhwc_pixel_stride(x: torch.Tensor) -> int:
    """floats from pixel to pixel if the (B, C, H, W) tensor is channel-last over a possibly WIDER channel axis (a dense channels_last tensor, or the
    channel slice autograd returns for one input of a concatenation), else 0"""
    if x.dim() != 4:
        return 0
    B, Cc, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    if sc != 1 or sw < Cc or sh != W * sw or (B > 1 and sb != H * W * sw):
        return 0
    return int(sw)


def split_f32_nhwc(x: torch.Tensor) -> torch.Tensor:
    """channel-last fp32 (B, C, H, W), C % 32 == 0 -> the carrier tensor of its PRE-SPLIT form (what ``group_norm_nhwc(..., split_out=True)`` writes), for a
    convolution operand that does not come out of a norm (csrc/groupnorm.hip, k_split_f32).  ``x`` may be a channel slice of a wider channel-last tensor
    (``nhwc_pixel_stride``): it is read in place, the dense copy is never made."""
    stride = nhwc_pixel_stride(x)
    if x.dtype != torch.float32 or stride == 0 or x.size(1) % 32 != 0 or stride % 4 != 0 or x.data_ptr() % 16 != 0:
        raise RuntimeError("split_f32_nhwc: channel-last fp32 (B, C, H, W) with C % 32 == 0 (dense, or a 16-byte aligned channel slice)")
    y = torch.empty((x.size(0), x.size(1), x.size(2), x.size(3)), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    C.check(C.lib().ssdnerf_split_f32_nhwc(C.ptr(x), C.ptr(y), ctypes.c_uint64(x.size(0) * x.size(2) * x.size(3)), C.u32(x.size(1)), ctypes.c_uint64(stride),
                                           C.stream()), "split_f32_nhwc")
    return y


def conv2d_nhwc_f32x2_presplit(x_split: torch.Tensor, w_hi: torch.Tensor, w_lo: torch.Tensor, bias: Optional[torch.Tensor] = None,
                               residual: Optional[torch.Tensor] = None, gn_sums: Optional[torch.Tensor] = None, gn_groups: int = 0,
                               splitk_ws: Optional[torch.Tensor] = None, tile_hint: int = 0, splits_hint: int = 0) -> torch.Tensor:
    """``conv2d_nhwc_f32x2`` (1 x 1 or 3 x 3, stride 1) of an activation that ``group_norm_nhwc(..., split_out=True)`` wrote pre-split (csrc/conv_igemm.hip,
    k_conv_pp_bf16<ROWS, F32, PS> -- bit-identical to the on-the-fly split -- or k_conv_igemm_bf16<..., PS> on the small layers): the operand split is out
    of the K loop.  ``w_hi`` / ``w_lo`` from ``split_bf16x2_adjacent``; ``splitk_ws``: the all-zero scratch of ``shared_splitk_ws``."""
    B, Cin, H, W = x_split.shape
    Cout, k = w_hi.shape[0], w_hi.shape[2]
    if not x_split.is_contiguous(memory_format=torch.channels_last) or x_split.dtype != torch.float32 or tuple(w_hi.shape[1:]) != (Cin, k, k) or k not in (1, 3):
        raise RuntimeError("conv2d_nhwc_f32x2_presplit: (B, Cin, H, W) channels_last carrier tensor and a (Cout, Cin, k, k) weight pair, k 1 | 3")
    y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x_split.device, memory_format=torch.channels_last)
    if residual is not None and (residual.shape != y.shape or residual.dtype != y.dtype or not residual.is_contiguous(memory_format=torch.channels_last)):
        raise RuntimeError("conv2d_nhwc_f32x2_presplit: residual must match the output's shape, dtype and layout")
    C.check(C.lib().ssdnerf_conv2d_nhwc_f32x2_presplit(C.ptr(x_split), C.ptr(w_hi), C.ptr(w_lo), C.ptr(bias), C.ptr(residual), C.ptr(y), C.u32(B), C.u32(H), C.u32(W),
                                                        C.u32(Cin), C.u32(Cout), C.u32(k), C.ptr(gn_sums), C.u32(gn_groups), int(tile_hint), int(splits_hint), C.ptr(splitk_ws),
                                                        ctypes.c_size_t(0 if splitk_ws is None else splitk_ws.numel() * splitk_ws.element_size()), C.stream()),
            "conv2d_nhwc_f32x2_presplit")
    return y


def conv2d_nhwc_f32x2(x: torch.Tensor, w_hi: torch.Tensor, w_lo: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                      stride: int = 1, upsample: bool = False, gn_sums: Optional[torch.Tensor] = None, gn_groups: int = 0, tile_hint: int = 0,
                      x2: Optional[torch.Tensor] = None, splits_hint: int = 0, splitk_ws: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 convolution with fp32-class (bf16 x 2) products on the matrix cores (csrc/conv_igemm.hip, k_conv_igemm_f32x2).  ``x`` (B, Cin, H, W)
    fp32 channels_last, ``w_hi`` / ``w_lo`` = ``split_bf16x2(weight)`` in channels_last, ``residual`` / result fp32 channels_last."""
    if x.dtype != torch.float32 or w_hi.dtype != torch.bfloat16 or w_lo.dtype != torch.bfloat16:
        raise RuntimeError("conv2d_nhwc_f32x2: fp32 input, bf16 weight pair")
    if not (x.is_contiguous(memory_format=torch.channels_last) and w_hi.is_contiguous(memory_format=torch.channels_last) and w_lo.is_contiguous(memory_format=torch.channels_last)):
        raise RuntimeError("conv2d_nhwc_f32x2: input and weights must be channels_last")
    B, Cin1, H, W = x.shape
    Cin = Cin1
    if x2 is not None:
        if x2.dtype != torch.float32 or not x2.is_contiguous(memory_format=torch.channels_last) or x2.shape[0] != B or x2.shape[2:] != x.shape[2:]:
            raise RuntimeError("conv2d_nhwc_f32x2: x2 must match x in batch, spatial size, dtype and layout")
        Cin = Cin1 + x2.shape[1]
    Cout, Cin_w, k, k2 = w_hi.shape
    if Cin_w != Cin or k != k2 or w_lo.shape != w_hi.shape:
        raise RuntimeError("conv2d_nhwc_f32x2: weight shape does not match the input")
    Hv, Wv = (2 * H, 2 * W) if upsample else (H, W)
    pad = k // 2
    Ho, Wo = (Hv + 2 * pad - k) // stride + 1, (Wv + 2 * pad - k) // stride + 1
    plan = C.lib().ssdnerf_conv2d_nhwc_f32x2_plan(C.u32(B * Ho * Wo), C.u32(Cin), C.u32(Cout), C.u32(k), int(tile_hint), int(splits_hint))
    ws_bytes = 0 if splitk_ws is None else splitk_ws.numel() * splitk_ws.element_size()
    # split-K accumulates into ``splitk_ws`` (all zero, left all zero) or, without a large enough one, into the output: hand that over zeroed
    split = (plan >> 8) > 1 and ws_bytes < B * Ho * Wo * Cout * 4
    y = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if split:
        y.zero_()
    if residual is not None and (residual.shape != y.shape or residual.dtype != y.dtype or not residual.is_contiguous(memory_format=torch.channels_last)):
        raise RuntimeError("conv2d_nhwc_f32x2: residual must match the output's shape, dtype and layout")
    C.check(C.lib().ssdnerf_conv2d_nhwc_f32x2(C.ptr(x), C.ptr(x2), C.u32(Cin1), C.ptr(w_hi), C.ptr(w_lo), C.ptr(bias), C.ptr(residual), C.ptr(y), C.u32(B), C.u32(H),
                                               C.u32(W), C.u32(Cin), C.u32(Cout), C.u32(k), C.u32(stride), C.u32(int(upsample)), C.ptr(gn_sums), C.u32(gn_groups),
                                               int(tile_hint), int(splits_hint), int(split), C.ptr(splitk_ws), ctypes.c_size_t(ws_bytes), C.stream()), "conv2d_nhwc_f32x2")
    return y


def attention_qkv_bf16(qkv: torch.Tensor, heads: int) -> torch.Tensor:
    """Hand-written MFMA flash attention (csrc/attention.hip).  ``qkv`` (B, T, 3C) bf16 contiguous, per-head channel order [q | k | v];
    returns (B, T, C) bf16."""
    B, T, C3 = qkv.shape
    Cc = C3 // 3
    if qkv.dtype != torch.bfloat16 or not qkv.is_contiguous():
        raise RuntimeError("attention_qkv_bf16: contiguous bf16 input only")
    out = torch.empty((B, T, Cc), dtype=torch.bfloat16, device=qkv.device)
    C.check(C.lib().ssdnerf_attention_qkv_bf16(C.ptr(qkv), C.ptr(out), C.u32(B), C.u32(T), C.u32(heads), C.u32(Cc // heads), C.stream()), "attention_qkv_bf16")
    return out


def attention_qkv_f32(qkv: torch.Tensor, heads: int) -> torch.Tensor:
    """fp32 form of the same kernel: ``qkv`` (B, T, 3C) fp32, fp32-class products on the bf16 matrix cores (q, k, v and the probabilities each
    split into a bf16 pair, hi*hi + hi*lo + lo*hi accumulated in fp32 -- the arithmetic class of the fp32 convolutions); returns (B, T, C) fp32."""
    B, T, C3 = qkv.shape
    Cc = C3 // 3
    if qkv.dtype != torch.float32 or not qkv.is_contiguous():
        raise RuntimeError("attention_qkv_f32: contiguous fp32 input only")
    out = torch.empty((B, T, Cc), dtype=torch.float32, device=qkv.device)
    C.check(C.lib().ssdnerf_attention_qkv_f32(C.ptr(qkv), C.ptr(out), C.u32(B), C.u32(T), C.u32(heads), C.u32(Cc // heads), C.stream()), "attention_qkv_f32")
    return out


def attention_qkv_f32_with_lse(qkv: torch.Tensor, heads: int):
    """``attention_qkv_f32`` that also returns the rows' log-sum-exp (log2 domain, (B, heads, T) fp32) for ``attention_qkv_f32_backward``."""
    B, T, C3 = qkv.shape
    Cc = C3 // 3
    if qkv.dtype != torch.float32 or not qkv.is_contiguous():
        raise RuntimeError("attention_qkv_f32_with_lse: contiguous fp32 input only")
    out = torch.empty((B, T, Cc), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((B, heads, T), dtype=torch.float32, device=qkv.device)
    C.check(C.lib().ssdnerf_attention_qkv_f32_lse(C.ptr(qkv), C.ptr(out), C.ptr(lse), C.u32(B), C.u32(T), C.u32(heads), C.u32(Cc // heads), C.stream()),
            "attention_qkv_f32_lse")
    return out, lse


def attention_qkv_f32_backward(qkv: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, lse: torch.Tensor, heads: int) -> torch.Tensor:
    """d loss / d qkv of ``attention_qkv_f32`` (csrc/attention.hip: k_attn_bwd_D, k_attn_bwd_dq, k_attn_bwd_dkv); everything (B, T, .) fp32 contiguous."""
    B, T, C3 = qkv.shape
    Cc = C3 // 3
    if any(t.dtype != torch.float32 or not t.is_contiguous() for t in (qkv, out, dout, lse)) or out.shape != (B, T, Cc) or dout.shape != out.shape:
        raise RuntimeError("attention_qkv_f32_backward: contiguous fp32 tensors of matching shapes only")
    dqkv = torch.empty_like(qkv)
    ws = torch.empty(B * heads * T, dtype=torch.float32, device=qkv.device)
    C.check(C.lib().ssdnerf_attention_qkv_f32_backward(C.ptr(qkv), C.ptr(out), C.ptr(dout), C.ptr(lse), C.ptr(dqkv), C.ptr(ws), C.u32(B), C.u32(T), C.u32(heads),
                                                        C.u32(Cc // heads), C.stream()), "attention_qkv_f32_backward")
    return dqkv


def attention_supported(dtype, T: int, ch: int) -> bool:
    """shapes the hand-written attention kernels take (csrc/attention.hip): any T, head width a multiple of 8 up to 128"""
    return dtype in (torch.bfloat16, torch.float32) and ch % 8 == 0 and 8 <= ch <= 128 and T >= 1


def attention_qkv(qkv: torch.Tensor, heads: int) -> torch.Tensor:
    return attention_qkv_bf16(qkv, heads) if qkv.dtype == torch.bfloat16 else attention_qkv_f32(qkv, heads)


#: hipGraph capture mode of the executor and of the gradient path (unet._GraphedGrad)
CAPTURE_MODE = os.environ.get("SSDNERF_GRAPH_CAPTURE_MODE", "thread_local")


class _Conv:
    """A convolution split into its bias-less GEMM part (``mm``) and an fp32 bias that the *consumer* folds in: the following
    GroupNorm (``pre_bias``) or the residual epilogue -- the library convolution would spend a pass of its own on it.

    ``pad_in`` / ``pad_out``: zero-pad the input / output channels of the weights up to a multiple of 8 (one 16-byte bf16 chunk: the granularity of
    the hand-written implicit GEMM since r03; r02 needed 64) so that the two edge layers of the UNet (18 -> 128 stem, 128 -> 18 head;
    denoising.py:116-118,178-187) run on it as well: the executor hands the stem a zero-padded input and slices the head's output (the extra
    products are exact zeros)."""
    __slots__ = ("w", "w_lo", "bias", "stride", "padding", "fold", "own", "cin", "cout")
    SPLITK_BYTES = 16 << 20

    def __init__(self, conv, dtype, pad_in: bool = False, pad_out: bool = False):
        weight = conv.weight.detach()
        if weight.dim() == 3:                                        # nn.Conv1d 1x1 projection (attention qkv / proj) == a 1x1 convolution
            weight = weight[..., None]
            stride, padding, dilation, k, groups = (1, 1), (0, 0), (1, 1), (1, 1), conv.groups
        else:
            stride, padding, dilation, k, groups = conv.stride, conv.padding, conv.dilation, conv.kernel_size, conv.groups
        assert groups == 1
        self.cout, self.cin = int(weight.shape[0]), int(weight.shape[1])
        bias = conv.bias.detach().float() if conv.bias is not None else None
        pin = (-self.cin) % 8 if pad_in else 0
        pout = (-self.cout) % 8 if pad_out else 0
        if pin or pout:
            weight = F.pad(weight, (0, 0, 0, 0, 0, pin, 0, pout))
            if bias is not None and pout:
                bias = F.pad(bias, (0, pout))
        self.w = weight.to(dtype).contiguous(memory_format=torch.channels_last)
        self.bias = bias.contiguous() if bias is not None else None
        self.stride, self.padding = tuple(stride), tuple(padding)
        self.fold = weight.shape[0] % 8 == 0                          # the 16-byte vector kernels need C % 8 == 0
        # the hand-written MFMA implicit GEMM (csrc/conv_igemm.hip) takes every layer whose (padded) channel counts are multiples of 8: bf16 as
        # is, fp32 with fp32-class products (weights pre-split into a bf16 pair; the library's fp32 convolution is kept for everything else)
        self.own = bool(dtype in (torch.bfloat16, torch.float32) and self.w.is_cuda and k[0] == k[1] and stride[0] == stride[1]
                        and tuple(padding) == (k[0] // 2, k[0] // 2) and tuple(dilation) == (1, 1)
                        and C.lib().ssdnerf_conv2d_nhwc_bf16_supported(int(weight.shape[1]), int(weight.shape[0]), k[0], stride[0], 0))
        self.w_lo = None
        if self.own and dtype == torch.float32 and _Conv.F32X2:
            self.w_lo = split_bf16x2_adjacent(weight.float())
            self.w = self.w_lo[0]                                   # (the fp32 copy is not needed; shape queries go through the hi term)
        elif dtype == torch.float32:
            self.own = False
        if not self.own and (pin or pout):                          # no kernel for it after all: keep the layer's true shape for the library
            self.__init__(conv, dtype)

    splitk_ws_by_device: dict = {}                  # device index -> all-zero fp32 scratch for the small layers' split-K (shared_splitk_ws)
    F32X2 = os.environ.get("SSDNERF_UNET_F32X2", "1") != "0"      # fp32 executor: own bf16 x 2 convolution (default) or the library's fp32 one

    def takes_presplit(self, x, with_stats=False) -> bool:
        """the norm that feeds this convolution may hand its result over pre-split (fp32 executor, large 3 x 3 layers)"""
        return bool(self.w_lo is not None and self.own and _Conv.PRESPLIT and self.stride[0] == 1 and int(self.w.shape[1]) == int(x.size(1))
                    and presplit_supported(x, int(self.w.shape[0]), int(self.w.shape[2]), with_stats))

    PRESPLIT = os.environ.get("SSDNERF_UNET_PRESPLIT", "1") != "0"   # 0: the fp32 two-group kernel splits its pixels on the fly again (A/B runs)

    def igemm(self, x, bias=None, residual=None, upsample=False, gn_sums=None, gn_groups=0, x2=None, presplit=False):
        if presplit:
            return conv2d_nhwc_f32x2_presplit(x, self.w_lo[0], self.w_lo[1], bias, residual, gn_sums, gn_groups, splitk_ws=shared_splitk_ws(x.device))
        if self.w_lo is not None:
            return conv2d_nhwc_f32x2(x, self.w_lo[0], self.w_lo[1], bias, residual, self.stride[0], upsample, gn_sums, gn_groups, x2=x2,
                                     splitk_ws=shared_splitk_ws(x.device))
        return conv2d_nhwc_bf16(x, self.w, bias, residual, self.stride[0], upsample, gn_sums, gn_groups, splitk_ws=shared_splitk_ws(x.device), x2=x2)

    def _w_lib(self):                                               # weight for the library path of a block whose other convolutions do not fit the own kernel
        return self.w if self.w_lo is None else (self.w_lo[0].float() + self.w_lo[1].float()).contiguous(memory_format=torch.channels_last)

    def mm(self, x):
        _lib_call(f"conv2d {tuple(self.w.shape)}")
        return F.conv2d(x, self._w_lib(), None, self.stride, self.padding)

    def __call__(self, x):
        if self.own:
            return self.igemm(x, self.bias)
        if self.bias is None:
            return self.mm(x)
        if not self.fold:
            _lib_call(f"conv2d {tuple(self.w.shape)}")
            return F.conv2d(x, self.w, self.bias.to(self.w.dtype), self.stride, self.padding)
        return bias_residual_nhwc(self.mm(x), self.bias, None)


class _like:
    """shape / dtype / device stand-in for a (B, C, H, W) tensor with another channel count (the concatenation a norm is about to produce)"""
    __slots__ = ("_x", "_c")

    def __init__(self, x, channels):
        self._x, self._c = x, channels

    is_cuda = property(lambda self: self._x.is_cuda)
    dtype = property(lambda self: self._x.dtype)

    def size(self, i):
        return self._c if i == 1 else self._x.size(i)


def shared_splitk_ws(device) -> Optional[torch.Tensor]:
    """The all-zero fp32 scratch the split-K convolutions of ``device`` reduce through (every call leaves it all zero again).  One buffer PER
    DEVICE, created on first use and never dropped: captured hipGraphs and executors of that device hold its pointer (r03 advisor: one
    process-global buffer that was re-created whenever the device changed left graphs of the first device with a freed pointer and could hand a
    kernel another GPU's memory).  Users of one device share it, so they must be ordered on one stream -- the executor's graph replays and the
    autograd path's eager launches both run on torch's current stream."""
    dev = torch.device(device)
    if dev.type != "cuda":
        return None                                                 # (CPU tensors only reach here under the CPU suite's stand-in kernels)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    ws = _Conv.splitk_ws_by_device.get(idx)
    if ws is None:
        ws = _Conv.splitk_ws_by_device[idx] = torch.zeros(_Conv.SPLITK_BYTES // 4, dtype=torch.float32, device=torch.device("cuda", idx))
    return ws


class _GN:
    __slots__ = ("groups", "gamma", "beta", "eps")

    def __init__(self, gn: torch.nn.GroupNorm):
        self.groups, self.eps = gn.num_groups, gn.eps
        self.gamma, self.beta = gn.weight.detach().float().contiguous(), gn.bias.detach().float().contiguous()


class FastUnet:
    """``FastUnet(net)(x_t, t)`` == ``net(x_t, t)`` (inference, no labels / concat conditioning)."""

    capture_by_default = True

    def __init__(self, net, dtype: torch.dtype = torch.float32, use_graph: Optional[bool] = None):
        from .unet import DenoisingResBlockMod, MultiHeadAttentionMod, DenoisingDownsampleMod, DenoisingUpsampleMod
        if net.num_classes != 0 or net.concat_cond_channels != 0:
            raise RuntimeError("FastUnet: label / concat conditioning is not on the hot path (use the module forward)")
        self.net, self.dtype = net, dtype
        self.use_graph = self.capture_by_default if use_graph is None else use_graph
        self.device = next(net.parameters()).device
        self._types = (DenoisingResBlockMod, MultiHeadAttentionMod, DenoisingDownsampleMod, DenoisingUpsampleMod)
        self._graphs: Dict[Tuple[int, int, int], tuple] = {}
        self.fallback_log: Optional[list] = None
        self.library_fallbacks: Optional[int] = None
        self._pack()

    # ------------------------------------------------------------------------------------------------ weights
    @staticmethod
    def param_version(net):
        """Identity of the weights the packed copies were made from: storage pointer, version counter, device and dtype of every parameter.
        In-place updates through autograd-visible ops (optimizer steps, ``copy_``, ``load_state_dict``) bump the version; ``module.to()`` /
        ``.half()`` / ``p.data = ...`` swap the storage; all of these re-pack.  Writes THROUGH ``p.data`` (``p.data.copy_()``, ``p.data.mul_()``)
        bypass both -- PyTorch gives no hook for them -- so code that updates weights that way (some EMA loops) must call
        ``DenoisingUnetMod.invalidate_fast_cache()`` afterwards."""
        return hash(tuple((p.data_ptr(), p._version, p.device.index, p.dtype) for p in net.parameters()))

    def invalidate(self):
        """Drop the packed weights and captured graphs; the next call re-packs from the module's current parameters."""
        self.version = None
        self._retire_graphs()
        self.fallback_log = None

    def _retire_graphs(self):
        """Drop the captured graphs (their static buffers and private pools go back to the allocator)."""
        self._graphs.clear()

    def _pack(self):
        net, dt = self.net, self.dtype
        Res, Att, Down, Up = self._types
        self.version = self.param_version(net)
        te = net.time_embedding
        self.te = (te.dim, te.max_period, te.blocks[0].weight.detach().float(), te.blocks[0].bias.detach().float(),
                   te.blocks[2].weight.detach().float(), te.blocks[2].bias.detach().float())
        emb_w: List[torch.Tensor] = []
        emb_b: List[torch.Tensor] = []
        self._emb_off = 0
        self._n_gn = 1                                                      # the output head's norm

        def res(m):
            assert m.norm_with_embedding.use_scale_shift, "FastUnet implements the scale-shift norm the hot-path configs use"
            assert len(m.conv_2) == 2, "dropout is inactive at inference and must not be configured > 0 here"
            lin = m.norm_with_embedding.embedding_layer[1]
            off = self._emb_off
            emb_w.append(lin.weight.detach().float()); emb_b.append(lin.bias.detach().float())
            self._emb_off += lin.out_features
            conv2, shortcut = _Conv(m.conv_2[-1], dt), _Conv(m.shortcut, dt) if m.learnable_shortcut else None
            out_bias = conv2.bias if shortcut is None or shortcut.bias is None else (conv2.bias + shortcut.bias)
            self._n_gn += 2
            return ("res", _GN(m.conv_1[0]), _Conv(m.conv_1[2], dt), _GN(m.norm_with_embedding.norm), (off, lin.out_features), conv2, shortcut, out_bias)

        def att(m):
            assert m.groups == 1
            self._n_gn += 1
            return ("att", _GN(m.norm), m.num_heads, _Conv(m.qkv, dt), _Conv(m.proj, dt))

        def seq(block):
            ops = []
            for layer in block:
                if isinstance(layer, Res):
                    ops.append(res(layer))
                elif isinstance(layer, Att):
                    ops.append(att(layer))
                elif isinstance(layer, Down):
                    assert isinstance(layer.downsample, torch.nn.Conv2d)
                    ops.append(("conv", _Conv(layer.downsample, dt)))
                elif isinstance(layer, Up):
                    ops.append(("up", _Conv(layer.conv, dt) if layer.with_conv else None))
                elif isinstance(layer, torch.nn.Conv2d):
                    ops.append(("conv", _Conv(layer, dt)))
                else:
                    raise RuntimeError(f"FastUnet: unsupported layer {type(layer).__name__}")
            return ops

        self.in_ops = [seq(b) for b in net.in_blocks]
        stem = net.in_blocks[0][0]
        if len(net.in_blocks[0]) == 1 and isinstance(stem, torch.nn.Conv2d):     # 18 -> 128: input channels zero-padded to 24 (see _Conv)
            self.in_ops[0] = [("conv", _Conv(stem, dt, pad_in=True))]
        self.stem_cin = self.in_ops[0][0][1].w.shape[1] if self.in_ops[0][0][0] == "conv" and self.in_ops[0][0][1].own else None
        self.mid_ops = seq(net.mid_blocks)
        self.out_ops = [seq(b) for b in net.out_blocks]
        self.head = (_GN(net.out.gn), _Conv(net.out.conv, dt, pad_out=True))
        self.emb_w, self.emb_b = torch.cat(emb_w, 0).contiguous(), torch.cat(emb_b, 0).contiguous()
        if self.device.type == "cuda":
            shared_splitk_ws(self.device)
        self._ws, self._ws_by_batch = None, {}
        self._max_c = max(int(p.shape[0]) for p in net.parameters() if p.dim() == 4)

    # ------------------------------------------------------------------------------------------------ blocks
    # Every block function takes and returns (activation, stats): ``stats`` is a slice of the statistics arena that already holds the sums of
    # the activation PER RUN OF 4 CHANNELS (fp64 (B, C/4, 2), written by the epilogue of the convolution that produced it: r03 -- the producer
    # then need not know how its consumer groups the channels, and a skip tensor's sums serve the encoder's next norm AND the decoder's
    # concatenated one), or None (the consuming norm then runs its own statistics pass).
    def _stats_slice(self, batch, channels=256):
        n = batch * (channels // 4) * 2                                     # a slice of the pre-zeroed statistics arena
        ws = self._ws[self._ws_next:self._ws_next + n]
        self._ws_next += n
        assert ws.numel() == n, "GroupNorm statistics arena exhausted"
        return ws

    def _gn(self, x, gn: _GN, ss, act, pre_bias=None, stats=None, x2=None, stats2=None, split=False):
        Cc = (x.size(1) if x.dim() == 4 else x.size(2)) + (x2.size(1) if x2 is not None else 0)
        if stats is not None and (x2 is None or stats2 is not None) and (Cc // gn.groups) % 4 == 0:
            return group_norm_nhwc(x, gn.groups, gn.gamma, gn.beta, ss, gn.eps, act, stats, x2=x2, runs=(stats, stats2), split_out=split)
        return group_norm_nhwc(x, gn.groups, gn.gamma, gn.beta, ss, gn.eps, act, self._stats_slice(x.size(0)), pre_bias=pre_bias, workspace_is_zero=True,
                               x2=x2, split_out=split)

    def _can_fuse_stats(self, conv: _Conv, x, upsample=False):
        if not conv.own:
            return False
        hw = x.size(2) * x.size(3) * (4 if upsample else 1) // (conv.stride[0] * conv.stride[1])
        cout, cin, k = conv.w.size(0), conv.w.size(1), conv.w.size(2)
        if cout % 4 != 0:
            return False                                                    # csrc/conv_igemm.hip: statistics per 4-channel half chunk
        if hw > 128 * 256:
            return False                                                    # every output tile of a sample adds to the same few dozen fp64 addresses: beyond ~500
                                                                            # tiles per sample (the tiled layout's 128 x 384 levels) a separate pass is faster
        if conv.w_lo is not None:                                           # fp32 kernel (mirrors ssdnerf_conv2d_nhwc_f32x2's choice of tile and split)
            plan = C.lib().ssdnerf_conv2d_nhwc_f32x2_plan(C.u32(x.size(0) * hw), C.u32(cin), C.u32(cout), C.u32(k), 0, 0)
            if plan >> 8 != 1:
                return True                                                 # split along K: the finishing pass takes the statistics
            return hw % (128 if (plan & 0xff) == 1 else 64) == 0
        plan = C.lib().ssdnerf_conv2d_nhwc_bf16_plan(C.u32(x.size(0) * hw), C.u32(cin), C.u32(cout), C.u32(k), 0, 1, 0)
        if plan >> 8 != 1:
            return True                                                     # a split-K layer: its finishing pass takes the statistics
        return hw % (256 if (plan & 0xff) == 4 else 128 if (plan & 0xff) == 1 else 64) == 0   # unsplit: the M tile must lie inside one sample

    def _conv_stats(self, conv: _Conv, x, bias=None, residual=None, upsample=False, x2=None, presplit=False):
        """An own convolution with the run-level statistics of its output where the kernel can take them (else None)."""
        st = self._stats_slice(x.size(0), conv.w.size(0)) if self._can_fuse_stats(conv, x, upsample) else None
        return conv.igemm(x, bias, residual, upsample=upsample, gn_sums=st, gn_groups=conv.w.size(0) // 4 if st is not None else 0, x2=x2, presplit=presplit), st

    def _res(self, x, stats, op, ss_all, x2=None, stats2=None):
        """One residual block.  ``x2``: the block's input is the concatenation [x | x2] (decoder half) and is never built."""
        _, gn1, conv1, gn2, (off, n), conv2, shortcut, out_bias = op
        ss = ss_all[:, off:off + n]
        own = conv1.own and conv2.own and (shortcut is None or shortcut.own)
        if x2 is not None and not (own and shortcut is not None and x.size(1) % 8 == 0 and x2.size(1) % 8 == 0):
            x, x2, stats, stats2 = torch.cat([x, x2], dim=1).contiguous(memory_format=torch.channels_last), None, None, None
        if own:
            # r04 (fp32 executor): the norms in front of the LARGE 3 x 3 layers write their result pre-split for the two-group kernel (bit-identical
            # products; the operand split leaves the convolution's K loop, where every pixel was split six times)
            cin1 = x.size(1) + (x2.size(1) if x2 is not None else 0)
            ps1 = conv1.w_lo is not None and conv1.takes_presplit(_like(x, cin1), with_stats=True)
            g1 = self._gn(x, gn1, None, True, stats=stats, x2=x2, stats2=stats2, split=ps1)
            h, st1 = self._conv_stats(conv1, g1, conv1.bias, presplit=ps1)                  # conv + bias (+ statistics for gn2)
            ps2 = conv2.w_lo is not None and conv2.takes_presplit(h, with_stats=True)
            g2 = self._gn(h, gn2, ss, True, stats=st1, split=ps2)
            skip = shortcut.igemm(x, x2=x2) if shortcut is not None else x                  # the shortcut's bias rides in out_bias
            return self._conv_stats(conv2, g2, out_bias, skip, presplit=ps2)                # conv + bias + skip (+ statistics for whatever reads it next)
        g1 = self._gn(x, gn1, None, True, stats=stats, x2=x2, stats2=stats2)
        h = conv1.mm(g1)
        h = conv2.mm(self._gn(h, gn2, ss, True, pre_bias=conv1.bias))
        return bias_residual_nhwc(h, out_bias, shortcut.mm(x) if shortcut is not None else x), None   # + b_conv2 (+ b_shortcut) + skip

    def _att(self, x, stats, op):
        _, gn, heads, qkv_conv, proj_conv = op
        B, Cc, H, W = x.shape
        T, ch = H * W, Cc // heads
        xt = x.permute(0, 2, 3, 1).reshape(B, T, Cc)                       # a view: channels_last storage is already [B][T][C]
        own = qkv_conv.own and proj_conv.own and attention_supported(self.dtype, T, ch)
        ps = own and qkv_conv.w_lo is not None and qkv_conv.takes_presplit(x)        # (fp32 executor: the qkv projection multiplies pre-split operands)
        xn = self._gn(xt, gn, None, False, stats=stats, split=ps)           # (B, T, C)
        if own:
            # the whole block on the hand-written kernels: 1x1 projections on the implicit GEMM (bias, residual and the next norm's statistics in
            # its epilogue), softmax(QK^T)V on the MFMA flash-attention kernel
            qkv = qkv_conv.igemm(xn.view(B, H, W, Cc).permute(0, 3, 1, 2), qkv_conv.bias, presplit=ps)   # (B, 3C, H, W) channels_last
            a = attention_qkv(qkv.permute(0, 2, 3, 1).reshape(B, T, 3 * Cc), heads)                  # (B, T, C)
            return self._conv_stats(proj_conv, a.view(B, H, W, Cc).permute(0, 3, 1, 2), proj_conv.bias, x)
        _lib_call(f"attention block C={Cc} T={T} (linear, sdpa, linear)")
        wqkv, wproj = qkv_conv._w_lib()[:, :, 0, 0], proj_conv._w_lib()[:, :, 0, 0]
        qkv = F.linear(xn, wqkv.to(xn.dtype), qkv_conv.bias.to(xn.dtype))   # (B, T, 3C), channel = head*3ch + {q,k,v}*ch + i
        if qkv.is_cuda and attention_supported(self.dtype, T, ch):
            a = attention_qkv(qkv, heads)
        else:
            q, k, v = qkv.view(B, T, heads, 3, ch).permute(3, 0, 2, 1, 4)   # each (B, heads, T, ch)
            a = F.scaled_dot_product_attention(q, k, v, scale=1.0 / math.sqrt(ch)).permute(0, 2, 1, 3).reshape(B, T, Cc)
        h = bias_residual_nhwc(F.linear(a, wproj.to(a.dtype), proj_conv.bias.to(a.dtype)), None, xt)      # h + x
        return h.view(B, H, W, Cc).permute(0, 3, 1, 2), None                # back to a channels_last (B, C, H, W) view

    def _run(self, ops, h, ss_all, stats=None, x2=None, stats2=None):
        for op in ops:
            kind = op[0]
            if kind == "res":
                h, stats = self._res(h, stats, op, ss_all, x2=x2, stats2=stats2)
                x2 = stats2 = None
            elif kind == "att":
                h, stats = self._att(h, stats, op)
            elif kind == "conv":
                h, stats = self._conv_stats(op[1], h, op[1].bias) if op[1].own else (op[1](h), None)
            elif kind == "up":
                if op[1] is not None and op[1].own:
                    h, stats = self._conv_stats(op[1], h, op[1].bias, upsample=True)       # the upsampled tensor is never built
                else:
                    _lib_call("interpolate (+ library convolution)")
                    h = F.interpolate(h, scale_factor=2, mode="nearest")
                    if op[1] is not None:
                        h = op[1](h)
                    stats = None
        return h, stats

    def _time_embedding(self, t):
        dim, max_period, w0, b0, w2, b2 = self.te
        half = dim // 2
        freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=t.device) / half)
        args = t[:, None].float() * freqs[None]
        e = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        if dim % 2:
            e = torch.cat([e, torch.zeros_like(e[:, :1])], dim=-1)
        return F.linear(F.silu(F.linear(e, w0, b0)), w2, b2)

    def _forward(self, x_t, t):
        net = self.net
        if net.use_rescale_timesteps:
            t = t.float() * (1000.0 / net.num_timesteps)
        self._ws.zero_()                                                    # all GroupNorm statistics of this forward, one memset
        self._ws_next = 0
        emb = self._time_embedding(t)
        ss_all = F.linear(F.silu(emb), self.emb_w, self.emb_b)             # every block's [scale | shift], fp32, one GEMM
        if self.stem_cin is not None and self.stem_cin != x_t.size(1):      # stem on the own kernel: hand it a zero-padded channel-last input
            h = torch.zeros((x_t.size(0), self.stem_cin) + tuple(x_t.shape[2:]), dtype=self.dtype, device=x_t.device).contiguous(memory_format=torch.channels_last)
            h[:, :x_t.size(1)] = x_t
        else:
            h = x_t.to(self.dtype).contiguous(memory_format=torch.channels_last)
        hs, stats = [], None
        for ops in self.in_ops:
            h, stats = self._run(ops, h, ss_all, stats)
            hs.append((h, stats))
        h, stats = self._run(self.mid_ops, h, ss_all, stats)
        for ops in self.out_ops:
            assert ops[0][0] == "res"
            skip, skip_stats = hs.pop()
            h, stats = self._run(ops, h, ss_all, stats, x2=skip, stats2=skip_stats)   # torch.cat([h, skip]) happens inside the block's kernels
        gn, conv = self.head
        out = conv(self._gn(h, gn, None, True, stats=stats))
        return out[:, :net.out_channels].float().contiguous()               # NCHW fp32, what the DDIM update consumes (drops the head's padding)

    # ------------------------------------------------------------------------------------------------ entry
    def _ensure_ws(self, B):
        ws = self._ws_by_batch.get(B)                                       # one arena per batch size: captured graphs keep pointing at theirs
        n = (2 * self._n_gn + 16) * B * (max(256, self._max_c) // 4) * 2    # one slice per statistics producer (run level) / per norm without one
        if ws is None or ws.numel() != n:
            ws = self._ws_by_batch[B] = torch.zeros(n, dtype=torch.float64, device=self.device)
        self._ws = ws

    def session(self, x_t: torch.Tensor, t: torch.Tensor) -> "UnetSession":
        """Static-buffer handle on the captured forward for inputs shaped like ``x_t`` / ``t``: a sampling loop writes the latent into
        ``session.x`` (in place), the timesteps into ``session.t``, calls ``session.run()`` and reads ``session.y`` -- no per-step copies in or
        out of the graph's buffers (``__call__`` pays three: x, t and the output clone)."""
        if not self.use_graph:
            raise RuntimeError("FastUnet.session needs graph capture")
        if self.version is None or self.param_version(self.net) != self.version or tuple(x_t.shape) not in self._graphs:
            self(x_t, t)                                                    # packs / captures (one forward, first use only)
        g, sx, st, sy = self._graphs[tuple(x_t.shape)]
        return UnetSession(self, g, sx, st, sy)

    @torch.no_grad()
    def __call__(self, x_t: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        if self.version is None or self.param_version(self.net) != self.version:
            self._pack()
            self._retire_graphs()
        self._ensure_ws(x_t.size(0))
        if self.fallback_log is None:                                       # first forward: record which ops (if any) went to a library
            global _FALLBACK_LOG
            _FALLBACK_LOG = []
            try:
                y = self._forward(x_t, t)
            finally:
                self.fallback_log, _FALLBACK_LOG = _FALLBACK_LOG, None
            self.library_fallbacks = len(self.fallback_log)
            if not self.use_graph:
                return y
        if not self.use_graph:
            return self._forward(x_t, t)
        key = tuple(x_t.shape)
        entry = self._graphs.get(key)
        if entry is None:
            sx, st = x_t.clone(), t.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                                  # warm-up: MIOpen / hipBLASLt pick kernels and workspaces
                for _ in range(2):
                    self._forward(sx, st)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            # (thread-local capture mode: what OTHER threads do -- the collective library's watchdog polling its events in an N > 1 job -- must not invalidate
            #  a capture of this thread's stream; this thread issues only capturable calls, as the warm-up above has just shown)
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                sy = self._forward(sx, st)
            entry = self._graphs[key] = (g, sx, st, sy)
        g, sx, st, sy = entry
        sx.copy_(x_t); st.copy_(t)
        g.replay()
        return sy.clone()


class UnetSession:
    """See ``FastUnet.session``.  Valid until the executor re-packs (weights changed); ``run`` checks."""
    __slots__ = ("ex", "graph", "x", "t", "y", "version")

    def __init__(self, ex: FastUnet, graph, x, t, y):
        self.ex, self.graph, self.x, self.t, self.y, self.version = ex, graph, x, t, y, ex.version

    def run(self) -> torch.Tensor:
        if self.ex.version != self.version or self.ex.param_version(self.ex.net) != self.version:
            raise RuntimeError("UnetSession: the network's weights changed since this session was opened; open a new one")
        self.graph.replay()
        return self.y
