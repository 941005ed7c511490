#!/usr/bin/env python3
"""PCIe-inclusive rate of the stage-1 search: queries handed over as HOST float32 arrays and
results returned as HOST arrays (the FAISS-style call), 10 M x 768 fp16 corpus, batch 64,
top-1000.  Reported in DESIGN.md §6; never bench.py's `value`."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tristage_rag_amd.index import FlatIPIndex

n, d, k, B = 10_000_000, 768, 1000, 64
idx = FlatIPIndex(d, dtype="f16")
idx.reserve(n)
g = torch.Generator(device="cuda").manual_seed(1)
for c in range(n // 500_000):
    x = torch.randn((500_000, d), generator=g, device="cuda")
    idx.add((x / (x.norm(dim=1, keepdim=True) + 1e-8)).half())
q = np.random.default_rng(0).standard_normal((B, d)).astype(np.float32)
q /= np.linalg.norm(q, axis=1, keepdims=True)
for _ in range(3):
    idx.search(q, k)
t0 = time.perf_counter()
for _ in range(20):
    D, I = idx.search(q, k)
dt = (time.perf_counter() - t0) / 20
qd = torch.from_numpy(q).cuda().half()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    idx.search(qd, k)
torch.cuda.synchronize()
dd = (time.perf_counter() - t0) / 20
print(json.dumps({"host_io_ms_per_batch": round(dt * 1e3, 4), "host_io_qps": round(B / dt, 1),
                  "device_resident_ms_per_batch": round(dd * 1e3, 4), "device_resident_qps": round(B / dd, 1)}))
