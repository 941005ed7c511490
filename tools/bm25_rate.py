#!/usr/bin/env python3
"""BM25 (the lexical half of stage 1, on by default in the reference): GPU search time per query
vs the host implementation, on a synthetic Zipf corpus."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tristage_rag_amd.stage1_retriever import BM25Index
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rng = np.random.default_rng(0)
V = 50_000
p = 1.0 / np.arange(1, V + 1); p /= p.sum()
vocab = np.array([f"w{i}" for i in range(V)])
t0 = time.perf_counter()
docs = [" ".join(vocab[rng.choice(V, size=int(rng.integers(30, 90)), p=p)]) for _ in range(n)]
print(f"corpus {n} docs built in {time.perf_counter()-t0:.1f} s")
queries = [" ".join(vocab[rng.choice(V, size=8, p=p)]) for _ in range(50)]
gpu = BM25Index(gpu_device=0)
t0 = time.perf_counter(); gpu.fit(docs); print(f"fit + upload: {time.perf_counter()-t0:.1f} s")
gpu.search(queries[0], 300)
t0 = time.perf_counter()
res = [gpu.search(q, 300) for q in queries]
tg = (time.perf_counter() - t0) / len(queries)
host = BM25Index(gpu_device=None)
host.__dict__.update({k: v for k, v in gpu.__dict__.items() if k not in ("_gpu", "gpu_device")})
t0 = time.perf_counter()
ref = [host.search(q, 300) for q in queries[:3]]
th = (time.perf_counter() - t0) / 3
assert all(a == b for a, b in zip(res[:3], ref)), "GPU and host BM25 differ"
print(f"GPU {tg*1e3:.3f} ms / query   host {th*1e3:.1f} ms / query   (top-300, 8-term queries, bit-identical)")
