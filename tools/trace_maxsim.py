#!/usr/bin/env python3
"""Per-wave phase timeline of ONE single-query MaxSim launch (needs a -DTS_TUNING -DM16_TRACE build
in TRISTAGE_LIB).  Phases: 0 entry, 1 prefix sums + barrier, 2 slice found and ring issued, 3 query image in LDS
(+ barrier), 4 query norms, 5 first tile done, 6 slice done, 7 records flushed."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tristage_rag_amd import _lib
from tristage_rag_amd.index import maxsim_indexed
docs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(7)
lens_all = torch.randint(64, 193, (100_000,), generator=g, device=dev, dtype=torch.int64)
starts_all = torch.cumsum(lens_all, 0) - lens_all
store = torch.randn((int(lens_all.sum()), 768), generator=g, device=dev, dtype=torch.float32).bfloat16()
q = torch.randn((32, 768), generator=g, device=dev, dtype=torch.float32).bfloat16()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (4096 * 8))()
for rep in range(4):
    pk = torch.randperm(100_000, generator=g, device=dev)[:docs]
    out = maxsim_indexed(q, store, starts_all[pk].contiguous(), lens_all[pk].to(torch.int32).contiguous())
    torch.cuda.synchronize()
assert lib.ts_debug_m16_trace(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)[:2048]
t = t[t[:, 0] > 0]
act = t[:, 7] > 0
t0 = t[:, 0].min()
us = (t - t0) / 100.0
names = ["entry", "prefix+barrier", "slice+ring issued", "q image in LDS", "q norms", "first tile", "slice done", "flushed"]
print(f"docs={docs}: {int(act.sum())} active waves of {t.shape[0]} launched")
for i, n in enumerate(names):
    col = us[act, i] if i >= 3 else us[:, i]
    print(f"  {i} {n:20s} min {col.min():7.2f}  median {np.median(col):7.2f}  p95 {np.percentile(col, 95):7.2f}  max {col.max():7.2f} us")
