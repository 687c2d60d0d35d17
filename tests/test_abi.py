"""CPU checks of the C-ABI boundary: the library loads and exports every symbol include/diffreg_hip.h declares."""
import ctypes
import os
import re

from tests.conftest import ROOT

HEADER = os.path.join(ROOT, "include", "diffreg_hip.h")
DEBUG_HEADER = os.path.join(ROOT, "include", "diffreg_hip_debug.h")     # diagnostics: not part of the drop-in boundary
LIB = os.path.join(ROOT, "diff-reg_amd", "diffreg_hip", "libdiffreg_hip.so")


def declared_symbols(debug=True):
    txt = open(HEADER).read() + (open(DEBUG_HEADER).read() if debug else "")
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dr_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build first: python __graft_entry__.py build"
    lib = ctypes.CDLL(LIB)
    syms = declared_symbols()
    assert "dr_sinkhorn_f32" in syms and "dr_version" in syms
    for s in syms:
        assert hasattr(lib, s), "missing export " + s


def test_python_binding_covers_header():
    from diffreg_hip import lib
    assert sorted(lib.SIGNATURES) == declared_symbols()
    assert lib.raw().dr_version() == lib.ABI_VERSION
    assert lib.raw().dr_strerror(-1) == b"invalid argument"


def test_no_cpu_fallback():
    import pytest
    import torch
    from diffreg_hip import lib
    with pytest.raises(RuntimeError):
        lib.sinkhorn(torch.zeros(1, 4, 4), torch.tensor(1.0), 3)
