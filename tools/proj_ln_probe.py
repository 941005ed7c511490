#!/usr/bin/env python3
"""ts_linear_add_layernorm on the stage-3 (MiniLM-L6) shapes, M = 157 539 tokens (a packed batch of 1024 reranking pairs):
BertSelfOutput (K = 384) and BertOutput (K = 1536) as one kernel against the two-kernel paths they replace
(ts_linear_act / the library GEMM, then ts_add_layernorm).  One JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tristage_rag_amd.index import TiledLinear, add_layernorm

M = int(os.environ.get("PROBE_M", 157539))
N = 384
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n


out = {"M": M, "N": N}
res = torch.randn((M, N), generator=g, device=dev)
gamma = 1.0 + 0.1 * torch.randn((N,), generator=g, device=dev)
beta = 0.1 * torch.randn((N,), generator=g, device=dev)
for name, K in (("self_output_K384", 384), ("output_K1536", 1536)):
    w = (torch.randn((N, K), generator=g, device=dev) * 0.05).bfloat16()
    b = (torch.randn((N,), generator=g, device=dev) * 0.1).bfloat16()
    x = torch.randn((M, K), generator=g, device=dev).bfloat16()
    tl = TiledLinear(w, b, with_layernorm=True)
    y32, ylp = tl.add_layernorm(x, res, gamma, beta, 1e-12)
    e32, elp = add_layernorm(F.linear(x, w, b), res, gamma, beta, 1e-12, lp_dtype=torch.bfloat16)
    t_fused = timed(lambda: tl.add_layernorm(x, res, gamma, beta, 1e-12))
    t_lib = timed(lambda: F.linear(x, w, b))
    t_two_lib = timed(lambda: add_layernorm(F.linear(x, w, b), res, gamma, beta, 1e-12, lp_dtype=torch.bfloat16))
    t_ln = timed(lambda: add_layernorm(elp, res, gamma, beta, 1e-12, lp_dtype=torch.bfloat16))
    rec = {"fused_ms": round(t_fused, 4), "library_gemm_ms": round(t_lib, 4), "add_layernorm_ms": round(t_ln, 4),
           "library_gemm_then_add_layernorm_ms": round(t_two_lib, 4),
           "max_abs_diff_vs_two_kernels": float((y32 - e32).abs().max()),
           "fused_TFLOPs": round(2.0 * M * N * K / t_fused / 1e9, 1),
           "fused_HBM_GBps": round(M * (2 * K + 4 * N + 6 * N) / t_fused / 1e6, 1)}
    if K <= 384:
        t1 = TiledLinear(w, b)
        rec["ts_linear_act_then_add_layernorm_ms"] = round(timed(lambda: add_layernorm(t1(x), res, gamma, beta, 1e-12, lp_dtype=torch.bfloat16)), 4)
    out[name] = rec
# the whole feed-forward block: up (+ GELU) and down + residual + LayerNorm, the GELU in the up projection's epilogue or
# folded into the down kernel's row staging
H = N
w1 = (torch.randn((4 * H, H), generator=g, device=dev) * 0.05).bfloat16()
b1 = (torch.randn((4 * H,), generator=g, device=dev) * 0.1).bfloat16()
w2 = (torch.randn((H, 4 * H), generator=g, device=dev) * 0.03).bfloat16()
b2 = (torch.randn((H,), generator=g, device=dev) * 0.1).bfloat16()
x0 = torch.randn((M, H), generator=g, device=dev).bfloat16()
up, down = TiledLinear(w1, b1), TiledLinear(w2, b2, with_layernorm=True)
a = down.add_layernorm(up(x0), res, gamma, beta, 1e-12, gelu_input=True)
b = down.add_layernorm(up(x0, gelu=True), res, gamma, beta, 1e-12)
out["feed_forward_block"] = {
    "gelu_in_up_epilogue_ms": round(timed(lambda: down.add_layernorm(up(x0, gelu=True), res, gamma, beta, 1e-12)), 4),
    "gelu_in_down_staging_ms": round(timed(lambda: down.add_layernorm(up(x0), res, gamma, beta, 1e-12, gelu_input=True)), 4),
    "up_without_activation_ms": round(timed(lambda: up(x0)), 4), "up_with_gelu_ms": round(timed(lambda: up(x0, gelu=True)), 4),
    "bit_identical": bool(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]))}
from tristage_rag_amd.index import mlp_add_layernorm
c = mlp_add_layernorm(up, down, x0, res, gamma, beta, 1e-12)
out["feed_forward_block"]["one_kernel_ms"] = round(timed(lambda: mlp_add_layernorm(up, down, x0, res, gamma, beta, 1e-12)), 4)
out["feed_forward_block"]["one_kernel_bit_identical"] = bool(torch.equal(c[0], b[0]) and torch.equal(c[1], b[1]))
up_out = up(x0)
out["feed_forward_block"]["down_with_gelu_input_ms"] = round(timed(lambda: down.add_layernorm(up_out, res, gamma, beta, 1e-12, gelu_input=True)), 4)
out["feed_forward_block"]["down_ms"] = round(timed(lambda: down.add_layernorm(up_out, res, gamma, beta, 1e-12)), 4)
print(json.dumps(out))
