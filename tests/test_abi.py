"""CPU: the C-ABI library loads and exports every symbol include/dgtd.h declares (no compute calls),
and the ctypes signature table stays in lock-step with the header."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "dgtd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|int64_t|const char\*)\s+(dgtd_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = [a.strip() for a in m.group(3).split(",") if a.strip() and a.strip() != "void"]
        out[m.group(2)] = len(args)
    return out


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    return ge


def test_library_exports_every_declared_symbol(built):
    import dgtd
    lib = ctypes.CDLL(dgtd._lib.LIB_PATH)
    decl = header_functions()
    assert len(decl) >= 8
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in dgtd.h but not exported"
    assert lib.dgtd_version() >= 100


def test_ctypes_table_matches_header(built):
    import dgtd
    decl = header_functions()
    assert set(decl) == set(dgtd._lib.SIGNATURES), set(decl) ^ set(dgtd._lib.SIGNATURES)
    for name, n in decl.items():
        assert len(dgtd._lib.SIGNATURES[name][1]) == n, name


def test_product_fails_loudly_without_device_or_library(built, tmp_path, monkeypatch):
    import torch
    import dgtd
    with pytest.raises(dgtd._lib.DgtdError):
        dgtd.ops.layer_norm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64), 1e-6)  # CPU tensor: refused, no fallback
    monkeypatch.setattr(dgtd._lib, "LIB_PATH", str(tmp_path / "missing.so"))
    monkeypatch.setattr(dgtd._lib, "_lib", None)
    with pytest.raises(ImportError):
        dgtd._lib.load()
