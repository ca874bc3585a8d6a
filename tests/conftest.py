import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import ctypes
        lib = os.path.join(ROOT, "multigrid_petsc_amd", "libmgk.so")
        if not os.path.exists(lib):
            return False
        return ctypes.CDLL(lib).mgk_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def mgk():
    """One kernel-ABI context for the whole GPU session (GPU tests run in ONE process)."""
    from multigrid_petsc_amd.mgk import Mgk
    m = Mgk(0)   # raises loudly if the library or the device is missing: no CPU fallback
    yield m
    m.close()
