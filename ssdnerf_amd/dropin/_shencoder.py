"""Drop-in replacement for the reference's pybind11 module ``_shencoder``
(lib/ops/shencoder/src/bindings.cpp:5-8; imported by name at lib/ops/shencoder/sphere_harmonics.py:9-12).
Same checks as the reference (device / contiguous / floating, shencoder.cu:403-413,422-435), fp32 only."""
import torch

from ssdnerf_amd import _cabi as C


def _check(*ts):
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError("must be a CUDA tensor")
        if not t.is_contiguous():
            raise RuntimeError("must be a contiguous tensor")
        if t.dtype != torch.float32:
            raise RuntimeError(f"must be a float32 tensor (got {t.dtype}); the reference wrapper casts to fp32 first")


def sh_encode_forward(inputs, outputs, B, D, C_, calc_grad_inputs, dy_dx):
    _check(inputs, outputs, dy_dx)
    C.check(C.lib().ssdnerf_sh_encode_forward(C.ptr(inputs), C.ptr(outputs), C.u32(B), C.u32(D), C.u32(C_), C.ctypes.c_int(int(bool(calc_grad_inputs))),
                                              C.ptr(dy_dx), C.stream()), "sh_encode_forward")


def sh_encode_backward(grad, inputs, B, D, C_, dy_dx, grad_inputs):
    _check(grad, inputs, dy_dx, grad_inputs)
    C.check(C.lib().ssdnerf_sh_encode_backward(C.ptr(grad), C.ptr(inputs), C.u32(B), C.u32(D), C.u32(C_), C.ptr(dy_dx), C.ptr(grad_inputs),
                                               C.stream()), "sh_encode_backward")
