#!/usr/bin/env python3
"""ts_ffn_up_gelu against F.linear + F.gelu at the encoder shapes: parity and time."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tristage_rag_amd.index import ffn_up_gelu

def timeit(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

g = torch.Generator(device="cuda").manual_seed(0)
for M, K, N in [(172032, 384, 1536), (65536, 768, 3072), (25600, 1024, 4096), (1000, 384, 1536)]:
    x = (torch.randn(M, K, device="cuda", generator=g) * 0.8).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = (torch.randn(N, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    ref = F.gelu(F.linear(x, w, b))
    got = ffn_up_gelu(x, w, b)
    err = float((got.float() - ref.float()).abs().max())
    rec = {"M": M, "K": K, "N": N, "max_abs_diff": err, "ref_max": float(ref.abs().max()),
           "mismatch_frac": float((got != ref).float().mean()),
           "fused_ms": round(timeit(lambda: ffn_up_gelu(x, w, b)), 4),
           "linear_ms": round(timeit(lambda: F.linear(x, w, b)), 4),
           "linear_gelu_ms": round(timeit(lambda: F.gelu(F.linear(x, w, b))), 4)}
    rec["fused_TFLOPs"] = round(2 * M * K * N / rec["fused_ms"] / 1e9, 1)
    print(json.dumps(rec), flush=True)
