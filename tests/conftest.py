import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a host without a HIP device: skip the gpu-marked tests instead of failing them."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs an MI355X (no HIP device visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def orc():
    import oracle
    return oracle.ops()


@pytest.fixture(scope="session")
def ref():
    """The reference's own kernels compiled for the CPU (oracle/_ref). Built on demand when
    /root/reference is present; tests that need it skip when it is absent AND was never built."""
    import oracle
    if oracle.ref_ops("fma") is None and os.path.isdir("/root/reference"):
        oracle.build_ref()
    r = oracle.ref_ops("fma")
    if r is None:
        pytest.skip("oracle/_ref not built (reference sources absent)")
    return r
