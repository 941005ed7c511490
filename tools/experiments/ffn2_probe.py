#!/usr/bin/env python3
"""ts_ffn2 (scan-structured FFN-up + GELU, FFN_LIB=path/to/lib.so) against F.linear (+ F.gelu): parity and time."""
import ctypes, json, os, sys
import torch
import torch.nn.functional as F

lib = ctypes.CDLL(os.environ["FFN_LIB"])
lib.ts_ffn2.restype = ctypes.c_int32
lib.ts_ffn2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                        ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
lib.ts_last_error.restype = ctypes.c_char_p

def tile(w):      # [N, K] -> the scan's corpus layout [N/32][K/16][h][r][8]
    N, K = w.shape
    return w.view(N // 32, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()

def ffn2(wt, x, b, N, qh, gelu):
    out = torch.empty((x.shape[0], N), dtype=x.dtype, device=x.device)
    st = lib.ts_ffn2(wt.data_ptr(), x.data_ptr(), b.data_ptr(), 2, x.shape[0], N, x.shape[1], qh, int(gelu), out.data_ptr(), 0,
                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0, lib.ts_last_error()
    return out

def timeit(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps, 4)

g = torch.Generator(device="cuda").manual_seed(0)
for M, K, N, gelu in [(172032, 384, 1536, True), (172032, 384, 1152, False), (172032, 384, 384, False), (65536, 768, 3072, True)]:
    x = (torch.randn(M, K, device="cuda", generator=g) * 0.8).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = (torch.randn(N, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    wt = tile(w)
    ref = F.gelu(F.linear(x, w, b)) if gelu else F.linear(x, w, b)
    rec = {"M": M, "K": K, "N": N, "gelu": gelu, "library_ms": timeit(lambda: F.gelu(F.linear(x, w, b)) if gelu else F.linear(x, w, b))}
    for qh in (4, 3, 2):
        if (K // 16) * qh * 1024 > 160 * 1024:
            continue
        got = ffn2(wt, x, b, N, qh, gelu)
        rec[f"qh{qh}_ms"] = timeit(lambda: ffn2(wt, x, b, N, qh, gelu))
        rec[f"qh{qh}_maxdiff"] = float((got.float() - ref.float()).abs().max())
        rec[f"qh{qh}_mismatch"] = float((got != ref).float().mean())
    print(json.dumps(rec), flush=True)
