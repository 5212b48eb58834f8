"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/isbfsar.h
declares (no compute calls: there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from isbfsar_amd.build import build
    return build(verbose=False)


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "isbfsar.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(isb_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built_lib):
    import ctypes
    lib = ctypes.CDLL(built_lib)
    syms = _declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/isbfsar.h but not exported"


def test_ctypes_signatures_cover_header(built_lib):
    from isbfsar_amd import _lib
    assert set(_declared_symbols()) == set(_lib.SIGNATURES), set(_declared_symbols()) - set(_lib.SIGNATURES)
    h = _lib.lib()
    for name in _lib.SIGNATURES:           # an entry point without argtypes truncates int pointers
        assert getattr(h, name).argtypes is not None, name
    assert h.isb_version() == 1
    assert isinstance(h.isb_device_count(), int)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from isbfsar_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.IsbError, match="no CPU fallback"):
        _lib.lib()


def test_no_gpu_error_path(built_lib):
    """Without a device, create() must return an error code + message, not crash."""
    import ctypes as C
    from isbfsar_amd import _lib
    h = _lib.lib()
    if h.isb_device_count() > 0:
        pytest.skip("a GPU is visible")
    cfg = _lib.isb_ar_cfg(16, 30, 5, 0, 0, 0)
    out = C.c_void_p()
    rc = h.isb_ar_create(C.byref(cfg), C.byref(out))
    assert rc < 0 and h.isb_last_error()
    bad = _lib.isb_ar_cfg(1, 30, 5, 0, 0, 0)
    assert h.isb_ar_create(C.byref(bad), C.byref(out)) == -1


def test_product_does_not_import_oracle():
    """The product package must never route through oracle/ (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "isbfsar_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, f)
