"""CPU: the C-ABI library loads and exports exactly what include/natac.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from nucleoatac_amd import _lib
    return _lib


def header_functions():
    h = open(os.path.join(ROOT, "include", "natac.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(natac_[a-z0-9_]+)\s*\(", h)))


def test_header_and_binding_agree(lib):
    names = header_functions()
    assert len(names) >= 25
    assert sorted(lib.SIGNATURES.keys()) == names


def test_library_exports_every_symbol(lib):
    so = ctypes.CDLL(lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(so, name), name
    hdr_version = int(re.search(r"#define NATAC_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "natac.h")).read()).group(1))
    assert so.natac_abi_version() == hdr_version == lib.ABI_VERSION


def test_enums_match_header(lib):
    h = open(os.path.join(ROOT, "include", "natac.h")).read()
    for name, val in re.findall(r"(NATAC_[TGK]_[A-Z_]+)\s*=\s*(\d+)", h):
        if name.endswith("_COUNT"):
            continue
        py = name[len("NATAC_"):]
        assert getattr(lib, py) == int(val), name


def test_errors_are_reported_not_thrown(lib):
    """without a GPU the context cannot be created: a negative code + message, never a crash / fallback"""
    L = lib.load()
    n = ctypes.c_int(-1)
    rc = L.natac_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    rc = L.natac_ctx_create(0, ctypes.byref(h))
    assert rc < 0 and not h.value
    assert len(L.natac_last_error()) > 0
    from nucleoatac_amd.device import Context
    with pytest.raises(lib.NatacError):
        Context(0)


def test_no_oracle_import_in_product():
    """the product package never imports the CPU oracle (it is test infrastructure)"""
    pkg = os.path.join(ROOT, "nucleoatac_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)
