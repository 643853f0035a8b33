"""Input gradient of the fused GroupNorm (+ scale/shift, + SiLU) that the device kernels k_gn_bwd_stats / k_gn_bwd_apply compute, checked on the
CPU: ssdnerf_amd/csrc/gn_bwd_math.h is plain C; a gcc build (tests/host/gn_bwd_host.c) is compared with PyTorch autograd through
F.group_norm * (1 + scale) + shift -> SiLU.  The device kernels compile the same header."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("host") / "gn_bwd_host.so")
    subprocess.run(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fPIC", "-shared", os.path.join(ROOT, "tests", "host", "gn_bwd_host.c"), "-o", so, "-lm"],
                   check=True)
    lib = ctypes.CDLL(so)
    lib.gn_bwd.restype = None
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("C,G,scale_shift,act", [(32, 8, True, True), (32, 8, False, True), (20, 4, True, False), (64, 32, False, False)])
def test_host_build_of_the_kernel_arithmetic_matches_autograd(host, C, G, scale_shift, act):
    g = torch.Generator().manual_seed(C + G)
    B, H, W = 3, 6, 10
    x = (torch.randn(B, C, H, W, generator=g) * 1.7 + 0.4).requires_grad_(True)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g) * 0.3
    ss = torch.randn(B, 2 * C, generator=g) * 0.5 if scale_shift else None
    dy = torch.randn(B, C, H, W, generator=g)
    y = F.group_norm(x, G, gamma, beta, 1e-5)
    if ss is not None:
        y = y * (1 + ss[:, :C, None, None]) + ss[:, C:, None, None]
    if act:
        y = F.silu(y)
    (want,) = torch.autograd.grad((y * dy).sum(), x)
    xn = np.ascontiguousarray(x.detach().permute(0, 2, 3, 1).numpy())        # channel-last, as the kernels see it
    dyn = np.ascontiguousarray(dy.permute(0, 2, 3, 1).numpy())
    dx = np.zeros_like(xn)
    host.gn_bwd(_p(xn), _p(dyn), ctypes.c_uint32(B), ctypes.c_uint32(H * W), ctypes.c_uint32(C), ctypes.c_uint32(G), _p(gamma.numpy()), _p(beta.numpy()),
                _p(None if ss is None else ss.numpy()), ctypes.c_float(1e-5), ctypes.c_int(int(act)), _p(dx))
    got = torch.from_numpy(dx).permute(0, 3, 1, 2)
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
