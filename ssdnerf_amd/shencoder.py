"""Mirror of ``lib/ops/shencoder/sphere_harmonics.py``: ``SHEncoder`` module + ``sh_encode`` autograd function
on top of the ``_shencoder`` drop-in backend (HIP kernel, degree <= 8, analytic Jacobian)."""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.autograd import Function

from .dropin import _shencoder as _backend


class _SHEncode(Function):
    @staticmethod
    def forward(ctx, inputs, degree, calc_grad_inputs=False):
        inputs = inputs.float().contiguous()          # the reference forces fp32 (sphere_harmonics.py:17)
        b, d = inputs.shape
        out_dim = degree ** 2
        outputs = torch.empty(b, out_dim, dtype=torch.float32, device=inputs.device)
        dy_dx = torch.empty(b, d * out_dim if calc_grad_inputs else 1, dtype=torch.float32, device=inputs.device)
        if not calc_grad_inputs:
            dy_dx = dy_dx.reshape(-1)[:1]
        _backend.sh_encode_forward(inputs, outputs, b, d, degree, calc_grad_inputs, dy_dx)
        ctx.save_for_backward(inputs, dy_dx)
        ctx.dims = (b, d, degree)
        ctx.calc_grad_inputs = calc_grad_inputs
        return outputs

    @staticmethod
    def backward(ctx, grad):
        if not ctx.calc_grad_inputs:
            return None, None, None
        inputs, dy_dx = ctx.saved_tensors
        b, d, degree = ctx.dims
        grad_inputs = torch.zeros_like(inputs)
        _backend.sh_encode_backward(grad.float().contiguous(), inputs, b, d, degree, dy_dx, grad_inputs)
        return grad_inputs, None, None


sh_encode = _SHEncode.apply


class SHEncoder(nn.Module):
    """``SHEncoder(input_dim=3, degree=4)``: real spherical harmonics of a direction (reference: sphere_harmonics.py:61-87)."""

    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim = input_dim
        self.degree = degree
        self.output_dim = degree ** 2
        assert self.input_dim == 3, "SH encoder only support input dim == 3"
        assert 0 < self.degree <= 8, "SH encoder only supports degree in [1, 8]"

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        inputs = inputs / size
        prefix = list(inputs.shape[:-1])
        flat = inputs.reshape(-1, self.input_dim)
        out = sh_encode(flat, self.degree, flat.requires_grad)
        return out.reshape(prefix + [self.output_dim])
