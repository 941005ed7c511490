#!/usr/bin/env python3
"""ts_attention_varlen against torch's SDPA (masked / unmasked) at stage-3 shapes: kernel time with HIP events."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tristage_rag_amd.index import attention_varlen

def timeit(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for B, L, nh, dh, lo in [(1024, 168, 12, 32, 60), (1024, 168, 12, 32, 168), (1024, 256, 12, 32, 100), (256, 512, 16, 64, 200), (1024, 64, 12, 32, 20)]:
    H = nh * dh
    qkv = torch.randn(B, L, 3 * H, device=dev, dtype=torch.bfloat16, generator=g)
    lens = torch.randint(lo, L + 1, (B,), generator=g, device=dev).to(torch.int32)
    out = torch.zeros(B, L, H, device=dev, dtype=torch.bfloat16)
    q, k, v = (qkv.view(B, L, 3, nh, dh)[:, :, i].transpose(1, 2) for i in range(3))
    mask = (torch.arange(L, device=dev)[None, :] < lens[:, None])[:, None, None, :]
    rec = {"B": B, "L": L, "heads": nh, "dh": dh, "valid_tokens": int(lens.sum()), "padded_tokens": B * L,
           "varlen_ms": round(timeit(lambda: attention_varlen(qkv, lens, nh, out=out)), 4),
           "sdpa_masked_ms": round(timeit(lambda: F.scaled_dot_product_attention(q, k, v, attn_mask=mask)), 4),
           "sdpa_unmasked_ms": round(timeit(lambda: F.scaled_dot_product_attention(q, k, v)), 4)}
    # algorithmic bytes: q, k, v of the valid tokens read once, the output written once
    rec["algorithmic_MB"] = round(rec["valid_tokens"] * H * 2 * 4 / 1e6, 1)
    rec["varlen_GBps"] = round(rec["algorithmic_MB"] / rec["varlen_ms"], 1)
    print(json.dumps(rec), flush=True)
