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


def _gn_standin(x, groups, gamma, beta, scale_shift, eps, act, workspace, out=None, pre_bias=None, workspace_is_zero=False, stats_ready=False, x2=None,
                runs=None, split_out=False):
    assert not split_out                                                 # (pre-split outputs only exist for the GPU's large fp32 layers)
    if x2 is not None:
        x = torch.cat([x, x2], dim=1)
    xc = x if x.dim() == 4 else x.transpose(1, 2)                       # (B, C, ...)
    xc = xc.float()
    if pre_bias is not None:
        xc = xc + pre_bias.reshape((1, -1) + (1,) * (xc.dim() - 2))
    if runs is not None:
        # the statistics come from the producer's epilogue (fp64 sums per run of 4 channels): normalise with THEM, so that wrong runs show up
        B, Cc = xc.shape[:2]
        r = (runs[0] if runs[1] is None else torch.cat([runs[0].view(B, -1, 2), runs[1].view(B, -1, 2)], dim=1)).view(B, groups, Cc // (4 * groups), 2).sum(2)
        n = xc[0].numel() // groups
        mean = r[..., 0] / n
        rstd = (r[..., 1] / n - mean * mean + eps).rsqrt()
        shape = (B, groups) + (1,) * (xc.dim() - 1)
        y = ((xc.reshape(B, groups, -1) - mean[..., None]) * rstd[..., None]).to(torch.float32).reshape(xc.shape)
        cshape = (1, Cc) + (1,) * (xc.dim() - 2)
        y = y * gamma.reshape(cshape) + beta.reshape(cshape)
    else:
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


def test_executor_repacks_after_storage_swaps_and_explicit_invalidation():
    """Weight-cache invalidation beyond version counters: ``p.data = ...`` and ``module.to()/.double()`` swap the storage (detected through the
    storage pointer / dtype), ``load_state_dict`` invalidates explicitly, and a write THROUGH ``p.data`` -- invisible to PyTorch's counters --
    is picked up after ``invalidate_fast_cache()``."""
    net = _small_unet(3)
    x = torch.randn(1, 6, 16, 16, generator=torch.Generator().manual_seed(5))
    t = torch.tensor([321])
    with torch.no_grad():
        net.fast_inference = False
        ex = net._fast_executor(torch.float32)
        ex.use_graph = False
        a = ex(x, t)
        # 1. storage swap without a version bump
        net.out.conv.weight.data = net.out.conv.weight.data * 2.0
        net.out.conv.bias.data = net.out.conv.bias.data * 2.0
        assert torch.allclose(ex(x, t), 2 * a, atol=1e-4, rtol=1e-4)
        # 2. load_state_dict
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        sd["out.conv.weight"] *= 0.5; sd["out.conv.bias"] *= 0.5
        net.load_state_dict(sd)
        assert torch.allclose(ex(x, t), a, atol=1e-4, rtol=1e-4)
        # 3. a write through .data is NOT seen (documented) until the cache is invalidated by hand
        net.out.conv.weight.data.mul_(3.0); net.out.conv.bias.data.mul_(3.0)
        net.invalidate_fast_cache()
        assert torch.allclose(ex(x, t), 3 * a, atol=2e-4, rtol=2e-4)


def test_executor_refuses_conditioning():
    net = MODULES.build(dict(type="DenoisingUnetMod", image_size=8, in_channels=4, base_channels=32, channels_cfg=[1], resblocks_per_downsample=1,
                             use_scale_shift_norm=True, num_classes=3, attention_res=[]))
    with pytest.raises(RuntimeError):
        unet_fast.FastUnet(net)


def _conv_f32x2_standin(x, w_hi, w_lo, bias=None, residual=None, stride=1, upsample=False, gn_sums=None, gn_groups=0, tile_hint=0, x2=None, splits_hint=0, splitk_ws=None):
    assert x.is_contiguous(memory_format=torch.channels_last) and w_hi.dtype == torch.bfloat16 and w_hi.is_contiguous(memory_format=torch.channels_last)
    assert stride in (1, 2) and not upsample and x2 is None and x.size(1) % 8 == 0 and w_hi.size(0) % 8 == 0      # (r05: stride 2 and zero-padded channel counts reach the kernel too)
    y = F.conv2d(x, w_hi.float() + w_lo.float(), bias, stride=stride, padding=w_hi.shape[-1] // 2)
    if residual is not None:
        assert residual.is_contiguous(memory_format=torch.channels_last)
        y = y + residual
    if gn_sums is not None:                                                 # what the kernel's epilogue leaves: sums per run of 4 output channels (added to zeros)
        B, Cc = y.shape[:2]
        assert gn_groups == Cc // 4 and gn_sums.dtype == torch.float64 and float(gn_sums.abs().max()) == 0
        r = y.double().reshape(B, Cc // 4, -1)
        gn_sums.view(B, Cc // 4, 2).add_(torch.stack([r.sum(-1), (r * r).sum(-1)], dim=-1))
    return y.contiguous(memory_format=torch.channels_last)


def test_input_gradient_convs_use_the_same_kernel_forward_and_backward(monkeypatch):
    """The differentiable path (guidance / val_optim: gradient w.r.t. the input, frozen weights): eligible convolutions run forward and
    backward-data through conv2d_nhwc_f32x2 (stand-in here), the backward with flipped / in-out-swapped weights; the input gradient
    must equal plain autograd through the module."""
    from ssdnerf_amd import unet
    net = MODULES.build(dict(type="DenoisingUnetMod", image_size=16, in_channels=6, base_channels=64, channels_cfg=[1, 2], resblocks_per_downsample=1,
                             dropout=0.0, use_scale_shift_norm=True, downsample_conv=True, upsample_conv=True, num_heads=4, attention_res=[8])).eval()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    x0 = torch.randn(2, 6, 16, 16, generator=g)
    t = torch.tensor([700, 30])
    probe = torch.randn(2, 6, 16, 16, generator=g)

    def grad_of(x):
        x = x.clone().requires_grad_(True)
        y = net(x, t)
        return y.detach(), torch.autograd.grad((y * probe).sum(), x)[0]

    net.requires_grad_(False)
    y_ref, g_ref = grad_of(x0)                                            # CPU tensors: every conv is nn.Conv2d's own forward
    calls = []
    unet._Conv2d.library_calls = 0
    monkeypatch.setattr(unet, "_device_ok", lambda x: True)
    monkeypatch.setattr(unet, "GRAD_GN", False)                     # this test is about the convolutions only (the fused norms have their own below)
    monkeypatch.setattr(unet, "GRAD_ATT", False)
    monkeypatch.setattr(unet_fast, "conv2d_nhwc_f32x2", lambda *a, **k: (calls.append(a[1].shape), _conv_f32x2_standin(*a, **k))[1])
    y, gx = grad_of(x0)
    n_fwd = sum(1 for m in net.modules() if isinstance(m, unet._Conv2d))    # r05: EVERY convolution -- the 6 -> 64 stem, the 64 -> 6 head and the stride-2 layer through
    n_general = sum(1 for m in net.modules() if isinstance(m, unet._Conv2d) and (m.stride != (1, 1) or m.in_channels % 8 or m.out_channels % 8))
    assert n_general == 3 and unet._Conv2d.library_calls == 0               # `_ConvGeneralFn` (zero-padded channels, zero-inserted dy), the others through `_ConvF32x2Fn`
    assert n_fwd >= 11 and len(calls) == 2 * n_fwd                        # each convolution: one forward launch + one backward-data launch
    assert torch.allclose(y, y_ref, atol=1e-4, rtol=1e-4), (y - y_ref).abs().max()
    assert float((gx - g_ref).abs().max()) <= 2e-4 * float(g_ref.abs().max())
    # weights that need their own gradient, autocast, and no-grad inputs all stay on the library path
    del calls[:]
    net.requires_grad_(True)
    grad_of(x0)
    net.requires_grad_(False)
    with torch.no_grad():
        net(x0, t)
    assert calls == []
    # the operand cache follows parameter updates
    conv = next(m for m in net.modules() if isinstance(m, unet._Conv2d) and m.in_channels % 64 == 0 and m.kernel_size == (3, 3))
    a = conv._split_pair(True)
    assert conv._split_pair(True)[0] is a[0]
    with torch.no_grad():
        conv.weight.mul_(2.0)
    b = conv._split_pair(True)
    assert b[0] is not a[0] and torch.equal(b[0].float(), (conv.weight.detach().flip(2, 3).transpose(0, 1)).to(torch.bfloat16).float())


def test_tiled_triplane_unet_shapes_module_oracle_and_executor():
    """The tiled layout of configs/new_cfgs/ssdnerf_cars_recons1v_tiled.py: (3,6,h,w) codes laid side by side as a 6-channel h x 3w image,
    ``image_size`` an int although the input is not square, GroupNorm groups that do not divide into 4-channel vectors, head widths the
    MFMA attention kernel does not cover.  Module forward vs the functional oracle, and the inference executor (SDPA / library fallbacks)."""
    from oracle import diffusion as OD
    net = MODULES.build(dict(type="DenoisingUnetMod", image_size=16, in_channels=6, base_channels=20, channels_cfg=[1, 2, 2], resblocks_per_downsample=1,
                             dropout=0.0, use_scale_shift_norm=True, downsample_conv=True, upsample_conv=True, num_heads=4, attention_res=[8, 4],
                             norm_cfg=dict(type="GN", num_groups=4))).eval()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    x = torch.randn(2, 6, 16, 48, generator=g)
    t = torch.tensor([800, 10])
    with torch.no_grad():
        want = net(x, t)
        ref = OD.unet_forward(net.state_dict(), x, t, image_size=16, base_channels=20, channels_cfg=(1, 2, 2), resblocks_per_downsample=1,
                              num_heads=4, attention_res=(8, 4), norm_groups=4)
        got = unet_fast.FastUnet(net, dtype=torch.float32, use_graph=False)(x, t)
    assert want.shape == (2, 6, 16, 48)
    assert torch.allclose(want, ref, atol=2e-5, rtol=1e-4), (want - ref).abs().max()
    assert torch.allclose(got, want, atol=2e-4, rtol=2e-4), (got - want).abs().max()


def _gn_backward_standin(x, dy, groups, gamma, beta, scale_shift, eps, act, fwd_sums, workspace=None, split_out=False, sums_are_runs=False):
    assert not split_out                                            # (the pre-split dx exists on the GPU only)
    assert x.is_contiguous(memory_format=torch.channels_last) and dy.is_contiguous(memory_format=torch.channels_last) and fwd_sums.dtype == torch.float64
    with torch.enable_grad():
        xx = x.detach().clone().requires_grad_(True)
        y = F.group_norm(xx, groups, gamma, beta, eps)
        if scale_shift is not None:
            c = xx.size(1)
            y = y * (1 + scale_shift[:, :c, None, None]) + scale_shift[:, c:, None, None]
        if act:
            y = F.silu(y)
        (gx,) = torch.autograd.grad((y * dy).sum(), xx)
    return gx.contiguous(memory_format=torch.channels_last)


def test_input_gradient_norms_fused_channel_last(monkeypatch):
    """Input-gradient path wiring (the default; SSDNERF_UNET_GRAD_GN / _ATT=0 switch it off): in it every residual block runs GN+SiLU -> conv -> GN*(1+scale)+shift+SiLU -> conv
    through _GroupNormActFn (forward group_norm_nhwc, backward group_norm_nhwc_backward; torch stand-ins here) and stays channel-last
    between the matrix-core convolutions; output and input gradient equal the plain module's."""
    from ssdnerf_amd import unet
    net = MODULES.build(dict(type="DenoisingUnetMod", image_size=16, in_channels=6, base_channels=64, channels_cfg=[1, 2], resblocks_per_downsample=1,
                             dropout=0.1, use_scale_shift_norm=True, downsample_conv=True, upsample_conv=True, num_heads=4, attention_res=[8])).eval()
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    net.requires_grad_(False)
    x0 = torch.randn(2, 6, 16, 16, generator=g)
    t = torch.tensor([400, 990])
    probe = torch.randn(2, 6, 16, 16, generator=g)

    def grad_of():
        x = x0.clone().requires_grad_(True)
        y = net(x, t)
        return y.detach(), torch.autograd.grad((y * probe).sum(), x)[0]

    y_ref, g_ref = grad_of()
    fwd, bwd, layouts = [], [], []
    monkeypatch.setattr(unet, "_device_ok", lambda x: True)
    monkeypatch.setattr(unet, "GRAD_GN", True)
    monkeypatch.setattr(unet, "GRAD_ATT", True)
    monkeypatch.setattr(unet_fast, "conv2d_nhwc_f32x2", lambda *a, **k: (layouts.append(a[0].is_contiguous(memory_format=torch.channels_last)), _conv_f32x2_standin(*a, **k))[1])
    from_epilogue = []
    monkeypatch.setattr(unet_fast, "group_norm_nhwc", lambda *a, **k: (fwd.append(a[6]), from_epilogue.append(k.get("runs") is not None), _gn_standin(*a, **k))[2])
    monkeypatch.setattr(unet_fast, "group_norm_nhwc_backward", lambda *a, **k: (bwd.append(1), _gn_backward_standin(*a, **k))[1])
    y, gx = grad_of()
    n_res = sum(1 for m in net.modules() if isinstance(m, unet.DenoisingResBlockMod))
    n_att = sum(1 for m in net.modules() if isinstance(m, unet.MultiHeadAttentionMod))
    # two fused norms per residual block + the output head (with SiLU), one plain norm per attention block (channel-last attention path)
    assert n_att >= 2 and len(fwd) == 2 * n_res + 1 + n_att and sum(fwd) == 2 * n_res + 1 and len(bwd) == len(fwd)
    assert all(layouts)
    # r04: the second norm of every block reads conv_1's epilogue statistics, and so does every norm whose input a fused convolution produced
    # (block outputs, the up-sampling convolutions, the skip concatenations of two such tensors): more than the second norms alone
    assert sum(from_epilogue) > n_res, (sum(from_epilogue), len(from_epilogue))          # (this small net follows most blocks with attention: 9 of 21; the cars UNet: see DESIGN.md)
    assert torch.allclose(y, y_ref, atol=1e-4, rtol=1e-4), (y - y_ref).abs().max()
    assert float((gx - g_ref).abs().max()) <= 2e-4 * float(g_ref.abs().max())
    # dropout active (training mode) keeps the eager block
    del fwd[:]
    net.train()
    net(x0.clone().requires_grad_(True), t)
    net.eval()
    assert len(fwd) == 1 + n_att                                               # the output head and the attention blocks (no dropout there)


def test_nhwc_pixel_stride_recognises_dense_tensors_and_channel_slices():
    """``unet_fast.nhwc_pixel_stride``: floats from pixel to pixel of a channel-last tensor over a possibly wider channel axis (the slices autograd returns for the
    inputs of a concatenation), 0 for anything else -- what decides whether the split pass may read a gradient in place."""
    wide = torch.randn(2, 48, 5, 7).contiguous(memory_format=torch.channels_last)
    assert unet_fast.nhwc_pixel_stride(wide) == 48
    assert unet_fast.nhwc_pixel_stride(wide[:, :32]) == 48 and unet_fast.nhwc_pixel_stride(wide[:, 32:]) == 48
    assert unet_fast.nhwc_pixel_stride(wide[:1, 16:]) == 48                      # (one sample: the batch stride does not matter)
    assert unet_fast.nhwc_pixel_stride(wide.contiguous()) == 0                   # NCHW
    assert unet_fast.nhwc_pixel_stride(wide[:, :, ::2]) == 0                     # rows skipped
    assert unet_fast.nhwc_pixel_stride(wide[:, ::2]) == 0                        # channels skipped
    assert unet_fast.nhwc_pixel_stride(wide[0]) == 0                             # not 4-D
    a, b = wide[:, :32].clone(memory_format=torch.contiguous_format).requires_grad_(True), torch.randn(2, 16, 5, 7, requires_grad=True)
    cat = torch.cat([a.contiguous(memory_format=torch.channels_last), b.contiguous(memory_format=torch.channels_last)], dim=1)
    ga, gb = torch.autograd.grad(cat, (a, b), wide)                              # the case it exists for
    assert cat.is_contiguous(memory_format=torch.channels_last)
