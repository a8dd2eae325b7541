import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference (test infrastructure only)."""
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def zl():
    """The product: ctypes view of libzlz4_amd.so (HIP, no fallback)."""
    import zig_lz4_amd
    zig_lz4_amd.lib()
    return zig_lz4_amd


@pytest.fixture(scope="session")
def gpu(zl):
    import torch
    assert torch.cuda.is_available(), "gpu-marked test started without a GPU"
    assert zl.device_available(), "libzlz4_amd.so found no gfx950 device"
    return torch.device("cuda:0")
