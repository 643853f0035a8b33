"""The decode gradient (triplane gather + tiny MLP, d loss / d planes) that the device kernels k_decode_bwd_feat / _bin / _sum compute, checked on the
CPU: ssdnerf_amd/csrc/decode_bwd_math.h is plain C, a gcc build of it (tests/host/decode_bwd_host.c) is compared with PyTorch autograd
through the oracle's decode (grid_sample + Linear + SiLU + TruncExp + Sigmoid).  The device kernel compiles the same header."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("host") / "decode_bwd_host.so")
    subprocess.run(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fPIC", "-shared", os.path.join(ROOT, "tests", "host", "decode_bwd_host.c"),
                    "-o", so, "-lm"], check=True)
    lib = ctypes.CDLL(so)
    lib.decode_bwd_points.restype = None
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _run(host, planes, P, xyzs, shs, g_sig, g_rgb, sat=0.001):
    hp, wp = planes.shape[1:3]
    gp = np.zeros_like(planes)
    feats = np.zeros((xyzs.shape[0], 18), np.float32)
    host.decode_bwd_points(_p(planes), ctypes.c_uint32(hp), ctypes.c_uint32(wp), _p(P), _p(xyzs), _p(shs), ctypes.c_uint32(xyzs.shape[0]),
                           ctypes.c_float(sat), _p(g_sig), _p(g_rgb), _p(gp), _p(feats))
    return gp, feats


@pytest.mark.parametrize("hw", [(32, 32), (16, 24)])
def test_host_build_of_the_kernel_arithmetic_matches_autograd(host, hw):
    from oracle import guidance as OG
    from oracle.decoder import gather_point_code, sh_encode
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd.decoders import pack_mlp_params
    h, w = hw
    g = torch.Generator().manual_seed(7)
    params = S.make_decoder_params()
    code = (torch.randn(3, 6, h, w, generator=g) * 0.8).requires_grad_(True)
    n = 3000
    xyzs = torch.rand(n, 3, generator=g) * 2.4 - 1.2            # some points beyond the border: clamped gather, gradient to the edge texels
    xyzs[:8] = torch.tensor([[-1.0, 1.0, 0.0], [1.0, -1.0, 1.0], [0.0, 0.0, 0.0], [1 - 1e-7, 1 - 1e-7, -1 + 1e-7], [-1.5, 0.3, 2.0],
                             [0.999, -0.999, 0.5], [1 / 3, -1 / 3, 0.123456], [-0.5, 0.25, -0.75]])
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    g_sig = torch.randn(n, generator=g) * 0.1
    g_rgb = torch.randn(n, 3, generator=g)
    sigma, rgb = OG.decode_autograd(params, code, xyzs, dirs)
    # TruncExp's backward clamps exp() to [1e-6, 1e6] (lib/ops/activation.py:17-20); decode_autograd uses plain exp -> same inside the range
    assert float(sigma.detach().max()) < 1e6 and float(sigma.detach().min()) > 1e-6
    (want,) = torch.autograd.grad((sigma * g_sig).sum() + (rgb * g_rgb).sum(), code)
    planes = np.zeros((3, h, w, 8), np.float32)
    planes[..., :6] = code.detach().permute(0, 2, 3, 1).numpy()
    P = pack_mlp_params(params, "cpu").numpy()
    shs = sh_encode(dirs).numpy().astype(np.float32)
    gp, feats = _run(host, planes, P, xyzs.numpy(), shs, g_sig.numpy(), g_rgb.numpy().copy())
    np.testing.assert_allclose(feats, gather_point_code(code.detach(), xyzs).numpy(), rtol=0, atol=5e-6)
    got = torch.from_numpy(gp[..., :6]).permute(0, 3, 1, 2)
    assert float(np.abs(gp[..., 6:]).max()) == 0.0
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    # density head only (the density-grid / sigma-only callers)
    (want_s,) = torch.autograd.grad((OG.decode_autograd(params, code, xyzs, dirs)[0] * g_sig).sum(), code)
    gp_s, _ = _run(host, planes, P, xyzs.numpy(), None, g_sig.numpy(), None)
    assert float((torch.from_numpy(gp_s[..., :6]).permute(0, 3, 1, 2) - want_s).abs().max()) <= 2e-5 * float(want_s.abs().max())


def test_truncated_exp_gradient_clamp(host):
    """d sigma / d (pre-activation) is clamp(exp(.), 1e-6, 1e6): with a huge density bias the gradient saturates at 1e6 per unit."""
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd.decoders import pack_mlp_params
    params = {k: v.clone() for k, v in S.make_decoder_params().items()}
    params["density_net.0.bias"] = params["density_net.0.bias"] + 40.0
    planes = np.zeros((3, 8, 8, 8), np.float32)
    planes[..., :6] = np.random.default_rng(0).normal(size=(3, 8, 8, 6)).astype(np.float32) * 0.3
    xyz = np.array([[0.1, -0.2, 0.3]], np.float32)
    gp_big, _ = _run(host, planes, pack_mlp_params(params, "cpu").numpy(), xyz, None, np.array([1.0], np.float32), None)
    params["density_net.0.bias"] = params["density_net.0.bias"] - 40.0
    P0 = pack_mlp_params(params, "cpu").numpy()
    gp_ref, _ = _run(host, planes, P0, xyz, None, np.array([1.0], np.float32), None)
    # same direction, magnitude ratio = 1e6 / exp(sa0)
    from oracle.decoder import point_decode
    code = torch.from_numpy(planes[..., :6]).permute(0, 3, 1, 2).contiguous()
    sig0, _ = point_decode(params, code, torch.from_numpy(xyz), None, density_only=True)
    ratio = gp_big.sum() / gp_ref.sum()
    assert ratio == pytest.approx(1e6 / float(sig0[0]), rel=1e-4)


def test_host_build_reproduces_the_reference_train_branch_gradient(host):
    """The reference's own train-branch render (its Python + its kernels on the CPU, tests/golden/render_train_64.npz): upstream gradients from
    the oracle's composite backward, pushed through the host build of the decode-backward arithmetic, land on the fixture's d loss / d code."""
    from oracle import guidance as OG, ops as _ops, render as R
    from oracle.decoder import sh_encode
    from ssdnerf_amd import synthetic as S
    from ssdnerf_amd.decoders import pack_mlp_params
    gold = os.path.join(ROOT, "tests", "golden")
    f, rays = np.load(os.path.join(gold, "render_train_64.npz")), np.load(os.path.join(gold, "cam_rays_64.npz"))
    params, code = S.make_decoder_params(), S.make_triplane()
    g = torch.Generator().manual_seed(7)
    jit = [torch.rand(64 ** 3, 3, generator=g).numpy() for _ in range(2)]
    _, bits, _ = R.get_density(params, code, jit, density_thresh=0.1)
    sub = f["ray_subset"]
    ro = np.ascontiguousarray(rays["rays_o"][:, sub].reshape(-1, 3), np.float32)
    rd = np.ascontiguousarray(rays["rays_d"][:, sub].reshape(-1, 3), np.float32)
    o = _ops()
    nears, fars = o.near_far_from_aabb(ro, rd, np.array([-1, -1, -1, 1, 1, 1], np.float32), 0.2)
    xyzs, dirs, deltas, rec, counter = o.march_rays_train(ro, rd, bits, 1.0, 0.0038095, 256, 1, 64, nears, fars, np.zeros(ro.shape[0], np.float32))
    m = int(counter[0])
    mp = m + 128 - m % 128
    xyzs, dirs, deltas = np.ascontiguousarray(xyzs[:mp]), np.ascontiguousarray(dirs[:mp]), deltas[:mp]
    with torch.no_grad():
        sig, rgb = OG.decode_autograd(params, code, torch.from_numpy(xyzs), torch.from_numpy(dirs))
    sig, rgb = sig.requires_grad_(True), rgb.requires_grad_(True)
    ws, _, image = OG._CompositeTrain.apply(sig, rgb, deltas, rec, 1e-4)
    loss = ((image + (1 - ws.unsqueeze(-1)) - torch.from_numpy(f["target"])) ** 2).mean() * 20.0
    assert abs(float(loss.detach()) - float(f["loss"])) < 1e-5
    gs, gc = torch.autograd.grad(loss, [sig, rgb])
    planes = np.zeros((3, 128, 128, 8), np.float32)
    planes[..., :6] = code.permute(0, 2, 3, 1).numpy()
    gp, _ = _run(host, planes, pack_mlp_params(params, "cpu").numpy(), xyzs, sh_encode(torch.from_numpy(dirs)).numpy().astype(np.float32),
                 np.ascontiguousarray(gs.numpy()), np.ascontiguousarray(gc.numpy()))
    got = torch.from_numpy(gp[..., :6]).permute(0, 3, 1, 2)[:, :, ::16, ::16].numpy()
    assert float(np.abs(got - f["grad_code_sample"]).max()) <= 1e-5 * float(f["grad_code_absmax"])
