"""ctypes binding of libtristage.so (the C ABI declared in include/tristage.h).

The HIP library IS the product path: if it is missing or cannot be loaded this
module raises — there is no CPU or PyTorch fallback behind it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_float, c_int32, c_int64, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# TRISTAGE_LIB: load another build of the same library (kernel-variant A/B runs)
LIB_PATH = os.environ.get("TRISTAGE_LIB") or os.path.join(_HERE, "libtristage.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

TS_OK = 0
TS_ERR_INVALID, TS_ERR_HIP, TS_ERR_OOM, TS_ERR_EMPTY, TS_ERR_UNSUPPORTED = -1, -2, -3, -4, -5
TS_F32, TS_F16, TS_BF16 = 0, 1, 2
TS_METRIC_INNER_PRODUCT = 0
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "tristage.h")


def header_abi_version() -> int:
    """TS_ABI_VERSION of include/tristage.h — the version this binding was written against."""
    import re
    m = re.search(r"^#define\s+TS_ABI_VERSION\s+(\d+)", open(HEADER_PATH).read(), flags=re.M)
    if not m:
        raise ImportError(f"TS_ABI_VERSION not found in {HEADER_PATH}")
    return int(m.group(1))


TS_FLAG_HOST_PTR, TS_FLAG_NO_FILTER, TS_FLAG_NORMALIZE, TS_FLAG_ASYNC, TS_FLAG_PIPELINE, TS_FLAG_CLASSIC, TS_FLAG_ONE_LAUNCH = 1, 2, 4, 8, 16, 32, 64

# name -> (restype, argtypes); mirrors include/tristage.h one to one
SIGNATURES = {
    "ts_index_create": (c_int32, [c_int32, c_int32, c_int32, c_int32, POINTER(c_void_p)]),
    "ts_index_destroy": (c_int32, [c_void_p]),
    "ts_index_reset": (c_int32, [c_void_p]),
    "ts_index_reserve": (c_int32, [c_void_p, c_int64]),
    "ts_index_add": (c_int32, [c_void_p, c_void_p, c_int64, c_int32, c_uint32, c_void_p]),
    "ts_index_search": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p,
                                  c_void_p, c_uint32, c_void_p]),
    "ts_index_scores": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int64, c_void_p]),
    "ts_index_ntotal": (c_int64, [c_void_p]),
    "ts_index_dim": (c_int32, [c_void_p]),
    "ts_index_dtype": (c_int32, [c_void_p]),
    "ts_index_set_id_offset": (c_int32, [c_void_p, c_int64]),
    "ts_index_reconstruct": (c_int32, [c_void_p, c_int64, c_int64, c_void_p, c_uint32, c_void_p]),
    "ts_index_last_search_info": (c_int32, [c_void_p, POINTER(c_int64)]),
    "ts_index_last_ticket": (c_int64, [c_void_p]),
    "ts_index_filter_path": (c_int32, [c_void_p, c_int32]),
    "ts_index_finish": (c_int32, [c_void_p, c_void_p, POINTER(c_int64), c_int32, POINTER(c_int32)]),
    "ts_index_set_profiling": (c_int32, [c_void_p, c_int32]),
    "ts_index_get_timings": (c_int32, [c_void_p, POINTER(ctypes.c_double), POINTER(c_int64), c_int32]),
    "ts_index_read_probe": (c_int32, [c_void_p, c_int32, POINTER(ctypes.c_double), POINTER(ctypes.c_double),
                                      POINTER(c_int64), c_void_p]),
    "ts_merge_topk": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                c_int32, c_void_p]),
    "ts_merge_topk_strided": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int64, c_int64,
                                        c_void_p, c_void_p, c_int32, c_void_p]),
    "ts_maxsim": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                            c_int32, c_void_p, c_int32, c_void_p]),
    "ts_maxsim_indexed": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                    c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "ts_maxsim_indexed_batch": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "ts_bm25_create": (c_int32, [c_int32, POINTER(c_void_p)]),
    "ts_bm25_destroy": (c_int32, [c_void_p]),
    "ts_bm25_set_index": (c_int32, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, ctypes.c_double]),
    "ts_bm25_search": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                 POINTER(c_int32), c_void_p]),
    "ts_bm25_search_batch": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                       c_void_p, c_void_p]),
    "ts_add_layernorm": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int32, c_void_p,
                                   c_void_p, c_int32, c_int32, c_void_p]),
    "ts_embed_layernorm": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                     c_int64, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "ts_add_prenorm": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int32, c_void_p,
                                 c_void_p, c_int32, c_int32, c_void_p]),
    "ts_attention_varlen": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_int32,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "ts_rope_inplace": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "ts_geglu": (c_int32, [c_void_p, c_int32, c_int64, c_int32, c_void_p, c_int32, c_void_p]),
    "ts_linear_tile_weight": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "ts_linear_act": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_int32, c_void_p, c_int32,
                                c_void_p]),
    "ts_linear_add_layernorm": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int32, c_int64,
                                          c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p]),
    "ts_mlp_add_layernorm": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int32,
                                       c_int64, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p]),
    "ts_maxsim_release_scratch": (c_int32, [c_int32]),
    "ts_selftest_device_once": (c_int32, [c_int32, c_int32]),
    "ts_last_error": (c_char_p, []),
    "ts_abi_version": (c_int32, []),
}

_lib = None


class TriStageNativeError(RuntimeError):
    """A libtristage.so call failed (message from ts_last_error())."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libtristage error {code}: {message}")
        self.code = code
        self.message = message


def build(force: bool = False) -> str:
    """Compile libtristage.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.run(["make", "-C", CSRC_DIR], check=True, stdout=subprocess.PIPE,
                   stderr=subprocess.STDOUT)
    return LIB_PATH


def load() -> ctypes.CDLL:
    """Load the library and bind every symbol of the header. Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so
    # (SONAME libamdhip64.so.7) and resolves it by file name, so it must be
    # mapped BEFORE libtristage.so — whose DT_NEEDED libamdhip64.so.7 then binds
    # to that same copy.  Loaded the other way round the process ends up with two
    # runtimes and the second one sees no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {CSRC_DIR}` "
            "(or __graft_entry__.build()); there is no fallback path")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    want = header_abi_version()
    if lib.ts_abi_version() != want:
        raise ImportError(f"libtristage.so has ABI version {lib.ts_abi_version()}, include/tristage.h (and this binding) "
                          f"version {want}: rebuild with `make -C {CSRC_DIR}`")
    _lib = lib
    return lib


def last_error() -> str:
    msg = load().ts_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(code: int) -> None:
    if code != TS_OK:
        raise TriStageNativeError(code, last_error())
