#!/usr/bin/env python3
"""The stage-2 token encoder (ModernBERT-base shape, bf16) in isolation: 512 documents of up to 128 tokens through the
written-out forward; run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tristage_rag_amd.encoders import lean_encoder_for, load_backbone

B, L = 512, 128
tok, model, _ = load_backbone("random:modernbert", "/tmp/ts_models", "base")
model.to("cuda").eval()
lean = lean_encoder_for(model, torch.bfloat16)
g = torch.Generator(device="cuda").manual_seed(0)
lens = torch.randint(64, L + 1, (B,), generator=g, device="cuda").to(torch.int32)
lens[0] = L
mask = (torch.arange(L, device="cuda")[None, :] < lens[:, None]).to(torch.int64)
ids = torch.randint(5, 30000, (B, L), generator=g, device="cuda") * mask
for _ in range(3):
    lean(ids, mask, None, lengths=lens)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); a.record()
for _ in range(10):
    lean(ids, mask, None, lengths=lens)
b.record(); torch.cuda.synchronize()
print(json.dumps({"B": B, "L": L, "valid_tokens": int(lens.sum()), "wall_ms": round((time.perf_counter() - t0) * 100, 3),
                  "gpu_ms": round(a.elapsed_time(b) / 10, 3)}))
