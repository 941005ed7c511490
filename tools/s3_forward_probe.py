#!/usr/bin/env python3
"""One stage-3 forward (MiniLM-L6 shape, 1024 pairs of ~168 tokens, bf16) in isolation: wall and GPU time per
forward of the written-out classifier; run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tristage_rag_amd.encoders import CrossEncoderModel

B, L = 1024, 168
ce = CrossEncoderModel("random:minilm", device="cuda", use_amp=True)
g = torch.Generator(device="cuda").manual_seed(0)
lens = torch.randint(140, L + 1, (B,), generator=g, device="cuda").to(torch.int32)
lens[0] = L
t = torch.arange(L, device="cuda")[None, :]
mask = (t < lens[:, None]).to(torch.int64)
ids = torch.randint(5, 30000, (B, L), generator=g, device="cuda") * mask
enc = {"input_ids": ids, "attention_mask": mask, "token_type_ids": torch.zeros_like(ids), "lengths": lens}
out = {"B": B, "L": L, "valid_tokens": int(lens.sum())}
for name, attn, ln, fo in (("hip_attention+hip_layernorm", True, True, True), ("the same, projection and LayerNorm as two kernels", True, True, False),
                           ("torch_attention+hip_layernorm", False, True, False), ("torch_both", False, False, False)):
    lean = ce._lean_model()
    lean.fused_attention, lean.fused_layernorm, lean.fused_output_layernorm = attn, ln, fo
    if os.environ.get("S3_GELU_IN_UP"):
        lean.gelu_in_down = False
    for _ in range(3):
        ce.logits_from_ids(enc)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for _ in range(10):
        ce.logits_from_ids(enc)
    b.record(); torch.cuda.synchronize()
    out[name] = {"wall_ms": round((time.perf_counter() - t0) * 100, 3), "gpu_ms": round(a.elapsed_time(b) / 10, 3)}
    if os.environ.get("S3_ONLY_FIRST"):
        break
print(json.dumps(out))
