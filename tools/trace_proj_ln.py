#!/usr/bin/env python3
"""Per-wave phase timeline of ONE ts_linear_add_layernorm launch (needs a -DTS_TUNING -DPL_TRACE build in TRISTAGE_LIB).
Stamps: 0 entry, 1 chunk-0 share loaded and written, 2 past the first barrier, 3 compute: chunk 0 multiplied / loader:
chunk 1 written, 4 past the last chunk's barrier, 5 compute: output staged, 6 past the staging barrier, 7 exit."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tristage_rag_amd import _lib
from tristage_rag_amd.index import TiledLinear
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (384, 1536)
M = 157539
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
w = (torch.randn((N, K), generator=g, device=dev) * 0.05).bfloat16()
b = (torch.randn((N,), generator=g, device=dev) * 0.1).bfloat16()
x = torch.randn((M, K), generator=g, device=dev).bfloat16()
res = torch.randn((M, N), generator=g, device=dev)
gamma, beta = torch.ones(N, device=dev), torch.zeros(N, device=dev)
tl = TiledLinear(w, b, with_layernorm=True)
for _ in range(3):
    tl.add_layernorm(x, res, gamma, beta, 1e-12)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (4096 * 8))()
assert lib.ts_debug_pl_trace(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
nblk = N // 32
wave = np.arange(4096) % 16
t0 = t[t[:, 0] > 0][:, 0].min()
names = ["entry", "chunk 0 share written", "past barrier 0", "chunk 0 multiplied | chunk 1 written", "past the last chunk's barrier",
         "output staged", "past the staging barrier", "exit"]
for role, sel in (("compute waves", (wave < nblk) & (t[:, 7] > 0)), ("loader waves", (wave >= nblk) & (wave < nblk + 4) & (t[:, 7] > 0))):
    tt = t[sel]
    rel = (tt - tt[:, :1]) / 100.0
    print(f"N={N} K={K} {role}: {tt.shape[0]} traced (the first 256 workgroups)")
    for i, n in enumerate(names):
        ok = tt[:, i] > 0
        if not ok.any():
            continue
        c = rel[ok, i]
        print(f"  {i} {n:38s} since entry: min {c.min():7.2f}  median {np.median(c):7.2f}  p95 {np.percentile(c, 95):7.2f}  max {c.max():7.2f} us")
ent = (t[t[:, 0] > 0][:, 0] - t0) / 100.0
print(f"  entry times of the traced waves: median {np.median(ent):.1f} us, max {ent.max():.1f} us; last exit {((t[t[:, 7] > 0][:, 7] - t0) / 100.0).max():.1f} us")
