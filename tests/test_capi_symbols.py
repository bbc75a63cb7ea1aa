"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/mdt_hip.h
declares (no compute calls without a GPU).  Also: the product never reaches into oracle/."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as g

ROOT = g.ROOT


@pytest.fixture(scope="module")
def lib():
    g.build_hip()
    return ctypes.CDLL(g.LIB)


def declared_functions():
    src = open(os.path.join(ROOT, "include", "mdt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mdt_[a-z_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    pkg = g.load_package()
    assert set(declared_functions()) == set(pkg._lib.SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for name in declared_functions():
        assert hasattr(lib, name), name
    lib.mdt_version.restype = ctypes.c_int
    assert lib.mdt_version() == 310


def test_struct_layout_matches_header():
    pkg = g.load_package()
    # int64 x3, int32 x8, int64, double, void*, int64 x2, void**, double x2 : 24 + 32 + 8 + 8 + 8 + 16 + 8 + 16
    assert ctypes.sizeof(pkg._lib.MdtArgs) == 120


def test_product_does_not_touch_the_oracle():
    """the oracle is test infrastructure: nothing under the package may import, link or open it"""
    pat = re.compile(r"oracle|libmdt_oracle", re.I)
    for dirpath, _, files in os.walk(g.PKG_DIR):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), os.path.join(dirpath, f)
    out = os.popen(f"ldd {g.LIB}").read()
    assert "oracle" not in out


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pkg = g.load_package()
    import numpy as np
    with pytest.raises(pkg.MdtRuntimeError):
        pkg.ess(np.random.default_rng(0).standard_normal((100, 4, 2)))
