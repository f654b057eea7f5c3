"""The C-ABI library loads and exports every symbol include/*.h declares (no compute: CPU box)."""
import ctypes
import os
import re

import pytest

from scrna_seq_qannealing_clustering_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    names = set()
    for hdr in sorted(os.listdir(os.path.join(ROOT, "include"))):
        src = open(os.path.join(ROOT, "include", hdr)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_library_is_built():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"


def test_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 16
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), "libmi_sa.so does not export %s" % name
    assert sorted(_lib.EXPORTS) == names            # the Python binding covers the whole header


def test_load_declares_signatures_and_reports_errors():
    lib = _lib.load()
    assert lib.mi_abi_version() == 1
    # argument validation happens before any device work, so this is safe without a GPU
    h = ctypes.c_void_p()
    rc = lib.mi_sa_problem_create_dense_f32(None, 4, 0.0, 0, ctypes.byref(h))
    assert rc == -1 and b"NULL" in lib.mi_last_error()
    with pytest.raises(_lib.MiSaError) as ei:
        _lib.check(rc)
    assert ei.value.code == -1


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmi_sa.so")
    with pytest.raises(RuntimeError, match="no CPU fallback|not built"):
        _lib.load()
