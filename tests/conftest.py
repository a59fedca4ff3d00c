import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    from tests.helpers import load_golden

    cache = {}

    def get(group):
        if group not in cache:
            cache[group] = load_golden(group)
        return cache[group]

    return get


@pytest.fixture
def smoother_flags():
    """Debug switch of libvbmp_hip.so that forces one of the alternative device forms of a kernel
    (0x10/0x20 K9 lane-/row-per-series, 0x40/0x80 K1 one-wave/block, 0x100 K3 VALU form); reset afterwards."""
    import ctypes
    from pyvbmp_amd import _lib
    lib = _lib.load()
    lib.vbmp_debug_set_flags.argtypes = [ctypes.c_int]
    lib.vbmp_debug_set_flags.restype = None
    yield lib.vbmp_debug_set_flags
    lib.vbmp_debug_set_flags(0)
