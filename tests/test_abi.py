"""CPU checks of the drop-in boundary: libtristage.so loads and exports exactly
the symbols include/tristage.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "tristage.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ts_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = _header_symbols()
    for must in ("ts_index_create", "ts_index_add", "ts_index_search", "ts_index_ntotal",
                 "ts_index_destroy", "ts_maxsim", "ts_merge_topk", "ts_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from tristage_rag_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libtristage.so not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _header_symbols():
        assert hasattr(lib, name), f"{name} declared in tristage.h but not exported"
    # and the ctypes table binds exactly that set
    assert sorted(_lib.SIGNATURES) == _header_symbols()
    assert _lib.load().ts_abi_version() == 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from tristage_rag_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libtristage.so"))
    with pytest.raises(ImportError):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tristage-rag_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle-free", ""), f"{f} mentions the oracle"
