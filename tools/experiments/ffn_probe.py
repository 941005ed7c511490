#!/usr/bin/env python3
"""ts_ffn_up_gelu (an experiment build, FFN_LIB=path/to/lib.so) against F.linear + F.gelu: parity and time."""
import ctypes, json, os, sys
import torch
import torch.nn.functional as F

lib = ctypes.CDLL(os.environ["FFN_LIB"])
lib.ts_ffn_up_gelu.restype = ctypes.c_int32
lib.ts_ffn_up_gelu.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32,
                               ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
lib.ts_last_error.restype = ctypes.c_char_p
TS_BF16 = 2

def ffn_up_gelu(x, w, b):
    out = torch.empty((x.shape[0], w.shape[0]), dtype=x.dtype, device=x.device)
    st = lib.ts_ffn_up_gelu(x.data_ptr(), w.data_ptr(), b.data_ptr(), TS_BF16, x.shape[0], w.shape[0], x.shape[1], out.data_ptr(), 0,
                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0, lib.ts_last_error()
    return out

def timeit(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

g = torch.Generator(device="cuda").manual_seed(0)
shapes = [(172032, 384, 1536), (65536, 768, 3072), (25600, 1024, 4096)]
if os.environ.get("FFN_ONE"): shapes = shapes[:1]
for M, K, N in shapes:
    x = (torch.randn(M, K, device="cuda", generator=g) * 0.8).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = (torch.randn(N, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    ref = F.gelu(F.linear(x, w, b))
    got = ffn_up_gelu(x, w, b)
    rec = {"M": M, "K": K, "N": N, "max_abs_diff": float((got.float() - ref.float()).abs().max()),
           "mismatch_frac": float((got != ref).float().mean()),
           "fused_ms": round(timeit(lambda: ffn_up_gelu(x, w, b)), 4),
           "linear_ms": round(timeit(lambda: F.linear(x, w, b)), 4),
           "linear_gelu_ms": round(timeit(lambda: F.gelu(F.linear(x, w, b))), 4)}
    print(json.dumps(rec), flush=True)
