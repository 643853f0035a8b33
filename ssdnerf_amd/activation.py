"""Mirror of ``lib/ops/activation.py``: ``TruncExp`` = exp forward (fp32), clamped-exp backward."""
import torch
import torch.nn as nn
from torch.autograd import Function


class _TruncExp(Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.exp(x.float())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return g * y.clamp(min=1e-6, max=1e6)


trunc_exp = _TruncExp.apply


class TruncExp(nn.Module):
    @staticmethod
    def forward(x):
        return _TruncExp.apply(x)
