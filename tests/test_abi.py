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
    # header, binding and library agree on the ABI version (no literal here: the constant has to move with the ABI)
    assert _lib.load().ts_abi_version() == _lib.header_abi_version() >= 2


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


def test_argument_errors_are_reported_without_a_gpu():
    """Argument validation happens before any HIP call: negative status + a thread-local message
    (the reference's counterpart: Python exceptions, SURVEY.md 8b 'Error convention')."""
    from tristage_rag_amd import _lib
    lib = _lib.load()
    off = (ctypes.c_int32 * 2)(0, 1)
    bad = (ctypes.c_int32 * 2)(1, 0)
    out = ctypes.c_void_p(1)   # never dereferenced: the calls fail first
    # null query pointer
    assert lib.ts_maxsim_indexed_batch(None, off, 1, out, out, out, off, 8, _lib.TS_BF16, 0, out, 0, None) == _lib.TS_ERR_INVALID
    assert b"maxsim_indexed_batch" in lib.ts_last_error()
    # decreasing offsets
    assert lib.ts_maxsim_indexed_batch(out, bad, 1, out, out, out, off, 8, _lib.TS_BF16, 0, out, 0, None) == _lib.TS_ERR_INVALID
    assert b"non-decreasing" in lib.ts_last_error()
    # bad mode / dtype
    assert lib.ts_maxsim(out, 1, out, out, 1, 8, 7, 0, out, 0, None) == _lib.TS_ERR_INVALID
    assert lib.ts_maxsim_indexed(out, 1, out, out, out, 1, 8, _lib.TS_F16, 5, out, 0, None) == _lib.TS_ERR_INVALID
    # nothing to do is not an error
    assert lib.ts_maxsim_indexed_batch(out, off, 0, out, out, out, off, 8, _lib.TS_BF16, 0, out, 0, None) == _lib.TS_OK
    assert lib.ts_maxsim(out, 1, out, out, 0, 8, _lib.TS_F32, 0, out, 0, None) == _lib.TS_OK
    with pytest.raises(Exception):
        _lib.check(_lib.TS_ERR_INVALID)


def test_per_device_kernel_attribute_table():
    """hipFuncSetAttribute is per device: the table that guards it (ts_common.h TsDeviceOnce) runs each
    (kernel, device) action exactly once under racing host threads, retries a failed action and never
    marks an untracked device.  Exercised through the library's own self-test (no GPU involved)."""
    from tristage_rag_amd import _lib
    lib = _lib.load()
    for threads, devices in ((1, 1), (16, 8), (5, 64), (32, 3)):
        assert lib.ts_selftest_device_once(threads, devices) == _lib.TS_OK, _lib.last_error()
    assert lib.ts_selftest_device_once(0, 4) == _lib.TS_ERR_INVALID
    assert lib.ts_selftest_device_once(4, 65) == _lib.TS_ERR_INVALID


def test_scores_entry_point_validates_arguments_without_a_gpu():
    from tristage_rag_amd import _lib
    lib = _lib.load()
    assert lib.ts_index_scores(None, None, 1, 0, None, 32, None) == _lib.TS_ERR_INVALID
    assert lib.ts_maxsim_release_scratch(-1) == _lib.TS_OK          # nothing allocated: nothing to free


def test_forward_kernel_entry_points_validate_arguments_without_a_gpu():
    """ts_add_layernorm / ts_add_prenorm / ts_embed_layernorm / ts_attention_varlen / ts_rope_inplace / ts_geglu and
    ts_bm25_search_batch: bad arguments and unsupported shapes are reported before any HIP call; empty inputs are fine."""
    from tristage_rag_amd import _lib
    lib = _lib.load()
    p = ctypes.c_void_p(4096)            # aligned, never dereferenced
    odd = ctypes.c_void_p(4100)          # not 16-byte aligned
    f = ctypes.c_float(1e-5)
    BF, F16, F32 = _lib.TS_BF16, _lib.TS_F16, _lib.TS_F32
    for fn in (lib.ts_add_layernorm, lib.ts_add_prenorm):
        assert fn(None, BF, None, p, p, f, 4, 64, p, None, BF, 0, None) == _lib.TS_ERR_INVALID          # no input
        assert fn(p, 9, None, p, p, f, 4, 64, p, None, BF, 0, None) == _lib.TS_ERR_INVALID             # bad dtype
        assert fn(p, BF, None, p, p, f, 4, 64, None, None, BF, 0, None) == _lib.TS_ERR_INVALID         # no output
        assert fn(p, BF, None, p, p, f, 4, 66, p, None, BF, 0, None) == _lib.TS_ERR_UNSUPPORTED        # H % 4
        assert fn(p, BF, None, p, p, f, 4, 4096, p, None, BF, 0, None) == _lib.TS_ERR_UNSUPPORTED      # H > 2048
        assert fn(odd, BF, None, p, p, f, 4, 64, p, None, BF, 0, None) == _lib.TS_ERR_UNSUPPORTED      # alignment
        assert fn(p, BF, None, p, p, f, 0, 64, p, None, BF, 0, None) == _lib.TS_OK                     # no rows
    assert lib.ts_embed_layernorm(None, p, None, p, p, p, p, p, f, 4, 64, p, None, BF, 0, None) == _lib.TS_ERR_INVALID
    assert lib.ts_embed_layernorm(p, p, None, p, p, p, p, None, f, 4, 62, p, None, BF, 0, None) == _lib.TS_ERR_UNSUPPORTED
    assert lib.ts_embed_layernorm(p, p, None, p, p, p, p, None, f, 0, 64, p, None, BF, 0, None) == _lib.TS_OK
    one = ctypes.c_float(0.125)
    assert lib.ts_attention_varlen(None, p, 2, 64, 4, 32, BF, one, 0, None, None, None, p, 0, None) == _lib.TS_ERR_INVALID
    assert lib.ts_attention_varlen(p, p, 2, 64, 4, 32, F32, one, 0, None, None, None, p, 0, None) == _lib.TS_ERR_INVALID     # 16-bit types only
    assert lib.ts_attention_varlen(p, p, 2, 64, 4, 32, BF, one, -1, None, None, None, p, 0, None) == _lib.TS_ERR_INVALID     # negative window
    assert lib.ts_attention_varlen(p, p, 2, 64, 4, 48, BF, one, 0, None, None, None, p, 0, None) == _lib.TS_ERR_UNSUPPORTED  # head dimension
    assert lib.ts_attention_varlen(p, p, 2, 4096, 4, 64, BF, one, 0, None, None, None, p, 0, None) == _lib.TS_ERR_UNSUPPORTED  # K, V^T beyond LDS
    assert lib.ts_attention_varlen(p, p, 70000, 64, 4, 32, BF, one, 0, None, None, None, p, 0, None) == _lib.TS_ERR_UNSUPPORTED  # grid limit
    assert b"attention_varlen" in lib.ts_last_error()
    assert lib.ts_attention_varlen(p, p, 2, 64, 4, 32, BF, one, 0, p, None, None, p, 0, None) == _lib.TS_ERR_INVALID    # cos without sin
    assert lib.ts_attention_varlen(p, p, 0, 64, 4, 32, BF, one, 0, None, None, None, p, 0, None) == _lib.TS_OK
    assert lib.ts_rope_inplace(None, BF, p, p, 2, 64, 4, 32, 0, None) == _lib.TS_ERR_INVALID
    assert lib.ts_rope_inplace(p, F32, p, p, 2, 64, 4, 32, 0, None) == _lib.TS_ERR_INVALID
    assert lib.ts_rope_inplace(p, BF, p, p, 2, 64, 4, 36, 0, None) == _lib.TS_ERR_UNSUPPORTED
    assert lib.ts_rope_inplace(p, BF, p, p, 0, 64, 4, 32, 0, None) == _lib.TS_OK
    assert lib.ts_geglu(None, BF, 4, 64, p, 0, None) == _lib.TS_ERR_INVALID
    assert lib.ts_geglu(p, BF, 4, 60, p, 0, None) == _lib.TS_ERR_UNSUPPORTED
    assert lib.ts_geglu(p, BF, 0, 64, p, 0, None) == _lib.TS_OK
    assert lib.ts_bm25_search_batch(None, p, p, 1, 5, p, p, p, None) == _lib.TS_ERR_INVALID
