"""Wiring of the inference executor (ssdnerf_amd/unet_fast.py) against the module forward, on CPU.

The HIP kernels cannot run here, so the executor's two native calls are replaced by torch stand-ins *in this test only*;
everything else (channel-last plumbing, the batched time-embedding GEMM and its per-block slices, attention layout, skip
concatenation, weight re-packing on parameter change) is the product code.  The kernel itself is checked in
tests/test_unet_fast_gpu.py."""
import pytest
import torch
import torch.nn.functional as F

import ssdnerf_amd  # noqa: F401
from ssdnerf_amd import unet_fast
from ssdnerf_amd.registry import MODULES


def _gn_standin(x, groups, gamma, beta, scale_shift, eps, act, workspace, out=None, pre_bias=None, workspace_is_zero=False, stats_ready=False, x2=None):
    if x2 is not None:
        x = torch.cat([x, x2], dim=1)
    xc = x if x.dim() == 4 else x.transpose(1, 2)                       # (B, C, ...)
    xc = xc.float()
    if pre_bias is not None:
        xc = xc + pre_bias.reshape((1, -1) + (1,) * (xc.dim() - 2))
    y = F.group_norm(xc, groups, gamma, beta, eps)
    if scale_shift is not None:
        c = xc.size(1)
        sc, sh = scale_shift[:, :c], scale_shift[:, c:]
        shape = (xc.size(0), c) + (1,) * (xc.dim() - 2)
        y = y * (1 + sc.reshape(shape)) + sh.reshape(shape)
    if act:
        y = F.silu(y)
    y = y.to(x.dtype)
    return y.contiguous(memory_format=torch.channels_last) if x.dim() == 4 else y.transpose(1, 2).contiguous()


def _bias_residual_standin(x, bias, residual, gn_sums=None, gn_groups=0):
    if bias is not None:
        x += bias.to(x.dtype)[None, :, None, None]
    if residual is not None:
        x += residual
    return x


@pytest.fixture(autouse=True)
def _standins(monkeypatch):
    monkeypatch.setattr(unet_fast, "group_norm_nhwc", _gn_standin)
    monkeypatch.setattr(unet_fast, "bias_residual_nhwc", _bias_residual_standin)


def _small_unet(seed=0):
    net = MODULES.build(dict(type="DenoisingUnetMod", image_size=16, in_channels=6, base_channels=32, channels_cfg=[1, 2, 2], resblocks_per_downsample=2,
                             dropout=0.0, use_scale_shift_norm=True, downsample_conv=True, upsample_conv=True, num_heads=4, attention_res=[8, 4])).eval()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    return net


def test_executor_matches_module_forward():
    net = _small_unet()
    x = torch.randn(2, 6, 16, 16, generator=torch.Generator().manual_seed(1))
    t = torch.tensor([999, 19])
    with torch.no_grad():
        want = net(x, t)                                                  # CPU tensors -> eager module path
        got = unet_fast.FastUnet(net, dtype=torch.float32, use_graph=False)(x, t)
    assert got.shape == want.shape and got.dtype == torch.float32
    assert torch.allclose(got, want, atol=2e-4, rtol=2e-4), (got - want).abs().max()


def test_executor_repacks_after_parameter_update():
    net = _small_unet()
    ex = unet_fast.FastUnet(net, dtype=torch.float32, use_graph=False)
    x = torch.randn(1, 6, 16, 16, generator=torch.Generator().manual_seed(2))
    t = torch.tensor([500])
    with torch.no_grad():
        a = ex(x, t)
        net.out.conv.weight.mul_(2.0); net.out.conv.bias.mul_(2.0)
        b = ex(x, t)
        assert torch.allclose(b, net(x, t), atol=2e-4, rtol=2e-4)
    assert torch.allclose(b, 2 * a, atol=1e-4, rtol=1e-4)


def test_executor_refuses_conditioning():
    net = MODULES.build(dict(type="DenoisingUnetMod", image_size=8, in_channels=4, base_channels=32, channels_cfg=[1], resblocks_per_downsample=1,
                             use_scale_shift_norm=True, num_classes=3, attention_res=[]))
    with pytest.raises(RuntimeError):
        unet_fast.FastUnet(net)
