#!/usr/bin/env python3
"""Does scaled_dot_product_attention take jagged (variable-length) batches on this ROCm build, and how fast?"""
import json, os, sys, time
import torch
import torch.nn.functional as F
B, Lmax, nh, dh = 1024, 168, 12, 32
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
lens = torch.randint(60, Lmax + 1, (B,), generator=g, device=dev)
offs = torch.zeros(B + 1, dtype=torch.int64, device=dev)
offs[1:] = torch.cumsum(lens, 0)
T = int(offs[-1])
out = {"tokens_valid": T, "tokens_padded": B * Lmax}
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
try:
    vals = [torch.randn(T, nh, dh, device=dev, dtype=torch.bfloat16) for _ in range(3)]
    q, k, v = (torch.nested.nested_tensor_from_jagged(x, offs).transpose(1, 2) for x in vals)
    o = F.scaled_dot_product_attention(q, k, v)
    out["njt_sdpa_ms"] = round(timeit(lambda: F.scaled_dot_product_attention(q, k, v)), 3)
    # check against padded + mask
    qp = torch.zeros(B, nh, Lmax, dh, device=dev, dtype=torch.bfloat16); kp = torch.zeros_like(qp); vp = torch.zeros_like(qp)
    for b in range(0, B, 97):
        a, e = int(offs[b]), int(offs[b + 1])
        qp[b, :, : e - a] = vals[0][a:e].transpose(0, 1); kp[b, :, : e - a] = vals[1][a:e].transpose(0, 1); vp[b, :, : e - a] = vals[2][a:e].transpose(0, 1)
    mask = (torch.arange(Lmax, device=dev)[None, :] < lens[:, None])[:, None, None, :]
    ref = F.scaled_dot_product_attention(qp, kp, vp, attn_mask=mask)
    ov = o.transpose(1, 2).values()
    err = 0.0
    for b in range(0, B, 97):
        a, e = int(offs[b]), int(offs[b + 1])
        err = max(err, float((ov[a:e].transpose(0, 1).float() - ref[b, :, : e - a].float()).abs().max()))
    out["njt_vs_masked_max_abs_diff"] = err
    out["padded_masked_ms"] = round(timeit(lambda: F.scaled_dot_product_attention(qp, kp, vp, attn_mask=mask)), 3)
except Exception as e:
    out["njt_error"] = repr(e)[:300]
print(json.dumps(out))
