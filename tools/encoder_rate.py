#!/usr/bin/env python3
"""Encoder forwards (PyTorch-ROCm, SURVEY.md 8d "evidence only"): tokens/s and model FLOP/s of the
three transformer forwards of the path with randomly initialised models of the reference's
architectures, bf16 autocast.  FLOPs = 2 x (non-embedding parameters) x tokens (attention's
quadratic term left out, so the figure is a lower bound)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tristage_rag_amd.encoders import load_backbone


def rate(name, head, batch, seqlen, reps=10, lean=False):
    tok, model, _ = load_backbone(name, "/tmp/ts_models", head, **({"num_labels": 1} if head == "seqcls" else {}))
    model.to("cuda").eval()
    if lean:   # the written-out forward with the library's HIP kernels between the GEMMs (tristage_rag_amd.encoders)
        from tristage_rag_amd.encoders import LeanBertClassifier, lean_encoder_for
        fwd = LeanBertClassifier(model, torch.bfloat16) if head == "seqcls" else lean_encoder_for(model, torch.bfloat16)
        hf = model
        model = lambda input_ids, attention_mask: fwd(input_ids, attention_mask)
        model.named_parameters, model.parameters = hf.named_parameters, hf.parameters
    emb = sum(p.numel() for n, p in model.named_parameters() if "embed" in n)
    params = sum(p.numel() for p in model.parameters()) - emb
    ids = torch.randint(1000, 20000, (batch, seqlen), device="cuda")
    mask = torch.ones_like(ids)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(3):
            model(input_ids=ids, attention_mask=mask)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            model(input_ids=ids, attention_mask=mask)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    toks = batch * seqlen
    return {"model": name, "head": head, "forward": "lean" if lean else "transformers", "batch": batch, "seq_len": seqlen, "non_embedding_params_M": round(params / 1e6, 1),
            "ms_per_forward": round(dt * 1e3, 3), "tokens_per_s": round(toks / dt), "model_TFLOPs": round(2 * params * toks / dt / 1e12, 1),
            "frac_of_2.5PF_bf16": round(2 * params * toks / dt / 2.5e15, 4)}


if __name__ == "__main__":
    out = [rate("random:bert", "base", 64, 128),             # stage-1 document encoding (BERT-base shape)
           rate("random:bert", "base", 64, 128, lean=True),
           rate("random:bert", "base", 512, 128),
           rate("random:bert", "base", 512, 128, lean=True),
           rate("random:minilm", "seqcls", 1024, 168, lean=True),
           rate("random:modernbert", "base", 64, 128, lean=True),
           rate("random:modernbert", "base", 512, 128),
           rate("random:modernbert", "base", 512, 128, lean=True),
           rate("random:modernbert", "base", 1, 16, lean=True),
           rate("random:xlmr-large", "seqcls", 100, 256, reps=5, lean=True),
           rate("random:modernbert", "base", 64, 128),       # stage-2 token store build (ModernBERT-base shape)
           rate("random:modernbert", "base", 1, 16),         # stage-2 query forward, batch 1
           rate("random:minilm", "seqcls", 1024, 168),       # stage-3 rerank_many forward
           rate("random:minilm", "seqcls", 128, 192),        # stage-3 per-query forward (graph bucket)
           rate("random:xlmr-large", "seqcls", 100, 256, reps=5),   # configs[4] stage 3: bge-reranker-large shape, one query's 100 pairs
           rate("random:xlmr-large", "seqcls", 1024, 64, reps=3)]   # the same model on search_many-sized batches of short pairs
    for o in out:
        print(json.dumps(o))
