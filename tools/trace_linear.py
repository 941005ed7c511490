#!/usr/bin/env python3
"""Per-wave phase timeline of ONE ts_linear_act launch (needs a -DTS_TUNING -DFS_TRACE build in TRISTAGE_LIB).
Stamps: 0 entry, 1 ring issued, 2 x image written to LDS, 3 after the barrier, 5 first unit's k loop done,
4 first unit done (epilogue included), 7 second unit's piece staged and the next requested, 6 exit."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tristage_rag_amd import _lib
from tristage_rag_amd.index import TiledLinear
N, K, act = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1152, 384, 0)
M = 172032
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
w = (torch.randn((N, K), generator=g, device=dev) * 0.05).bfloat16()
b = (torch.randn((N,), generator=g, device=dev) * 0.1).bfloat16()
x = torch.randn((M, K), generator=g, device=dev).bfloat16()
tl = TiledLinear(w, b)
for _ in range(3):
    tl(x, gelu=bool(act))
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (4096 * 8))()
assert lib.ts_debug_fs_trace(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
t = t[t[:, 6] > 0]
t0 = t[:, 0].min()
names = ["entry", "ring issued", "x image in LDS", "after barrier", "unit 1 done", "unit 1 k loop done", "exit", "unit 2 piece staged + requested"]
order = [0, 1, 2, 3, 5, 4, 7, 6]
print(f"N={N} K={K} act={act}: {t.shape[0]} waves traced (the first 512 workgroups)")
rel = (t - t[:, :1]) / 100.0          # us since the wave's own entry
for i in order:
    n = names[i]
    c = rel[:, i]
    if (t[:, i] <= 0).any():
        c = c[t[:, i] > 0]
        if not len(c):
            continue
    print(f"  {i} {n:18s} since entry: min {c.min():7.2f}  median {np.median(c):7.2f}  p95 {np.percentile(c, 95):7.2f}  max {c.max():7.2f} us")
ent = (t[:, 0] - t0) / 100.0
print(f"  entry times of the traced waves: median {np.median(ent):.1f} us, max {ent.max():.1f} us; last exit {((t[:, 6] - t0) / 100.0).max():.1f} us")
