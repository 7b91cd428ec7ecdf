import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope='session')
def hip_lib():
    import __graft_entry__ as g
    g.build_hip()
    from chsimpy_amd import _lib
    return _lib.load()


@pytest.fixture(scope='session')
def gpu(hip_lib):
    # gpu-marked tests call through the C ABI and must fail, not skip, without a device
    return hip_lib
