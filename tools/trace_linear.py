#!/usr/bin/env python3
"""Per-wave phase timeline of ONE ts_linear_act launch (needs a -DTS_TUNING -DFS_TRACE build in TRISTAGE_LIB).
Stamps, compute waves: 0 entry, 1 image share written, 2 past the barrier, 3 block 1 multiplied, 4 block 1 in its slot,
5 block 2 multiplied, 6 block 2 in its slot, 7 exit; storer waves: 3 / 4 = first / second pair of blocks read out of the slots."""
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
wave = np.arange(4096) % 12
t0 = t[t[:, 0] > 0][:, 0].min()
names = ["entry", "image share written", "past the barrier", "block 1 multiplied | pair 1 read", "block 1 in its slot | pair 2 read",
         "block 2 multiplied", "block 2 in its slot", "exit"]
print(f"N={N} K={K} act={act} (the first 341 workgroups)")
for role, sel in (("compute waves", (wave < 8) & (t[:, 7] > 0)), ("storer waves", (wave >= 8) & (t[:, 7] > 0))):
    tt = t[sel]
    rel = (tt - tt[:, :1]) / 100.0
    print(f" {role}: {tt.shape[0]} traced")
    for i, n in enumerate(names):
        ok = tt[:, i] > 0
        if not ok.any():
            continue
        c = rel[ok, i]
        print(f"  {i} {n:36s} since entry: min {c.min():7.2f}  median {np.median(c):7.2f}  p95 {np.percentile(c, 95):7.2f}  max {c.max():7.2f} us")
ent = (t[t[:, 0] > 0][:, 0] - t0) / 100.0
print(f"  entry times of the traced waves: median {np.median(ent):.1f} us, max {ent.max():.1f} us; last exit {((t[t[:, 7] > 0][:, 7] - t0) / 100.0).max():.1f} us")
