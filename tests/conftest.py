import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def ob():
    """The CPU oracle (test infrastructure)."""
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def native():
    """libpatolette_amd.so through ctypes; building it needs hipcc, loading it needs no GPU."""
    from patolette_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    _native.lib()
    return _native


@pytest.fixture(scope="session")
def gpu(native):
    L = native.lib()
    if L.patolette_amd_device_count() <= 0:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box (there is no CPU fallback)")
    return L
