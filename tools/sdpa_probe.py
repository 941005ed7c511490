#!/usr/bin/env python3
"""Where does the cross-encoder forward spend its time?  (stage 3, MiniLM-L6 shape, 1024 pairs x 168 tokens, bf16)
Times the HF model forward and the attention call alone: SDPA with the padding mask (what transformers passes),
without any mask (the flash path), and with the mask folded into the dot product (one extra head column:
q' = [q, 1], k' = [k, 0 | -1e4]) so that the mask-free kernel can be used with padded batches."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    from tristage_rag_amd.encoders import load_backbone
    B, L, Hh, dh = 1024, 168, 12, 32
    dev = "cuda"
    out = {}
    q = torch.randn(B, Hh, L, dh, device=dev, dtype=torch.bfloat16)
    k, v = torch.randn_like(q), torch.randn_like(q)
    lens = torch.randint(120, L + 1, (B,), device=dev)
    mask = (torch.arange(L, device=dev)[None, :] < lens[:, None])
    am = mask[:, None, None, :]
    out["sdpa_bool_mask_ms"] = round(timeit(lambda: F.scaled_dot_product_attention(q, k, v, attn_mask=am)), 3)
    fm = torch.zeros(B, 1, 1, L, device=dev, dtype=torch.bfloat16).masked_fill(~am, float("-inf"))
    out["sdpa_float_mask_ms"] = round(timeit(lambda: F.scaled_dot_product_attention(q, k, v, attn_mask=fm)), 3)
    out["sdpa_no_mask_ms"] = round(timeit(lambda: F.scaled_dot_product_attention(q, k, v)), 3)
    # mask folded into the contraction
    qa = torch.zeros(B, Hh, L, 40, device=dev, dtype=torch.bfloat16)
    ka, va = torch.zeros_like(qa), torch.zeros_like(qa)
    qa[..., :dh], ka[..., :dh], va[..., :dh] = q, k, v
    qa[..., dh] = 1.0
    ka[..., dh] = torch.where(mask, 0.0, -1e4)[:, None, :].to(torch.bfloat16)
    out["sdpa_folded_mask_ms"] = round(timeit(lambda: F.scaled_dot_product_attention(qa, ka, va, scale=dh ** -0.5)), 3)
    ref = F.scaled_dot_product_attention(q, k, v, attn_mask=am).float()
    got = F.scaled_dot_product_attention(qa, ka, va, scale=dh ** -0.5)[..., :dh].float()
    out["folded_vs_masked_max_abs_diff"] = float((ref - got)[mask[:, None, :, None].expand_as(ref)].abs().max())
    for name in ("flash_sdp_enabled", "mem_efficient_sdp_enabled", "math_sdp_enabled"):
        out[name] = bool(getattr(torch.backends.cuda, name)())
    # the model forward itself
    tok, model, _ = load_backbone("random:minilm", "/tmp/ts_models", "seqcls", num_labels=1)
    model.to(dev).eval()
    ids = torch.randint(1000, 20000, (B, L), device=dev)
    m = mask.long()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out["hf_forward_masked_ms"] = round(timeit(lambda: model(input_ids=ids, attention_mask=m)), 3)
        out["hf_forward_allones_ms"] = round(timeit(lambda: model(input_ids=ids, attention_mask=torch.ones_like(m))), 3)
        out["hf_forward_nomask_ms"] = round(timeit(lambda: model(input_ids=ids)), 3)
    out["attn_implementation"] = getattr(model.config, "_attn_implementation", None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
