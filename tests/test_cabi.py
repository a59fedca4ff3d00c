"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/vbmp_hip.h declares (no compute calls here); the ctypes table in pyvbmp_amd/_lib.py covers
exactly the same set."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vbmp_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(vbmp_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from pyvbmp_amd import _lib
    return _lib.load()


def test_header_declares_something():
    syms = declared_symbols()
    assert "vbmp_niw_ss_update_f64" in syms and "vbmp_abi_version" in syms and len(syms) >= 7


def test_every_declared_symbol_is_exported(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/vbmp_hip.h but not exported by libvbmp_hip.so"


def test_ctypes_table_matches_header(lib):
    from pyvbmp_amd import _lib
    table = {f"{b}_{s}" for b in _lib.SYMBOLS for s in _lib.DTYPES} | {"vbmp_abi_version"}
    assert table == set(declared_symbols())
    assert lib.vbmp_abi_version() == _lib.ABI_VERSION


def test_bad_arguments_are_rejected_without_a_gpu(lib):
    # argument validation happens on the host before any HIP call
    null = ctypes.c_void_p(0)
    assert lib.vbmp_spd_inv_logdet_f64(null, 16, null, null, 5, 4, null, null) == -1
    assert lib.vbmp_spd_inv_logdet_f64(null, 16, null, null, 0, 4, null, null) == 0  # empty batch: nothing to do
    assert lib.vbmp_spd_inv_logdet_f32(ctypes.c_void_p(16), 16, ctypes.c_void_p(16), null, 5, 65, null, null) == -1
    some = ctypes.c_void_p(16)
    assert lib.vbmp_rows_affine_f64(some, 10, 65, some, null, 3, some, null) == -1  # k beyond VBMP_ROWS_MAX_DIM
    assert lib.vbmp_rows_affine_f64(null, 10, 4, some, null, 3, some, null) == -1
    assert lib.vbmp_rows_affine_f32(null, 0, 4, null, null, 3, null, null) == 0


def test_product_refuses_cpu_tensors(lib):
    from pyvbmp_amd import _lib, ops
    with pytest.raises(_lib.VbmpHipError):
        ops.spd_inv_logdet(torch.eye(3, dtype=torch.float64).unsqueeze(0))


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under pyvbmp_amd/ may reference it."""
    pkg = os.path.join(ROOT, "pyvbmp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports oracle"
