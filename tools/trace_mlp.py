#!/usr/bin/env python3
"""Per-wave phase timeline of ONE ts_mlp_add_layernorm launch (needs a -DTS_TUNING -DML_TRACE build in TRISTAGE_LIB).
Stamps: 0 entry, 1 x image share written, 2 past the first barrier, 3 up(0) multiplied, 4 chunk 0 activated and packed,
5 chunk 0 image complete (past barrier B), 6 output staged (all chunks done), 7 exit."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tristage_rag_amd import _lib
from tristage_rag_amd.index import TiledLinear, mlp_add_layernorm
H, I, M = 384, 1536, 157539
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
w1 = (torch.randn((I, H), generator=g, device=dev) * 0.05).bfloat16()
b1 = (torch.randn((I,), generator=g, device=dev) * 0.1).bfloat16()
w2 = (torch.randn((H, I), generator=g, device=dev) * 0.03).bfloat16()
b2 = (torch.randn((H,), generator=g, device=dev) * 0.1).bfloat16()
x = torch.randn((M, H), generator=g, device=dev).bfloat16()
res = torch.randn((M, H), generator=g, device=dev)
gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
up, down = TiledLinear(w1, b1), TiledLinear(w2, b2, with_layernorm=True)
for _ in range(3):
    mlp_add_layernorm(up, down, x, res, gamma, beta, 1e-12)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (4096 * 8))()
assert lib.ts_debug_ml_trace(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
t = t[t[:, 7] > 0]
t0 = t[:, 0].min()
names = ["entry", "x image share written", "past the first barrier", "up(0) multiplied", "chunk 0 activated + packed",
         "chunk 0 image complete", "output staged", "exit"]
rel = (t - t[:, :1]) / 100.0
print(f"{t.shape[0]} waves traced (the first 341 workgroups)")
for i, n in enumerate(names):
    c = rel[:, i]
    print(f"  {i} {n:30s} since entry: min {c.min():7.2f}  median {np.median(c):7.2f}  p95 {np.percentile(c, 95):7.2f}  max {c.max():7.2f} us")
ent = (t[:, 0] - t0) / 100.0
print(f"  entry times of the traced waves: median {np.median(ent):.1f} us, max {ent.max():.1f} us; last exit {((t[:, 7] - t0) / 100.0).max():.1f} us")
