#!/usr/bin/env python3
"""Timeline of ONE launch of the one-launch stage-1 search (needs a -DTS_TUNING -DFZ_TRACE build in TRISTAGE_LIB).
Scan waves: 0 entry, 1 query image built, 2 first block done, 3 thresholds seen, 4 last block done, 5 parked tiles
re-filtered, 6 staging flushed.  Threshold workgroups: 0 entry, 1 arrival hint seen, 2 sample complete (first query),
3 first threshold published, 4 done."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tristage_rag_amd import _lib
from tristage_rag_amd.index import FlatIPIndex
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
d = 768
dev = torch.device("cuda", 0)
idx = FlatIPIndex(d, dtype="f16")
idx.reserve(rows)
g = torch.Generator(device=dev).manual_seed(1)
for r0 in range(0, rows, 250_000):
    n = min(250_000, rows - r0)
    x = torch.randn((n, d), generator=g, device=dev)
    idx.add((x / x.norm(dim=1, keepdim=True)).half())
q = torch.randn((64, d), generator=g, device=dev)
q = (q / q.norm(dim=1, keepdim=True)).half()
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * (4096 * 8))()
for rep in range(4):
    idx.search(q, 1000)
    torch.cuda.synchronize()
print(idx.last_search_info())
assert lib.ts_debug_fused_trace(buf) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
scan = t[:3000]
scan = scan[scan[:, 0] > 0]
tau = t[3000:3300]
tau = tau[tau[:, 0] > 0]
t0 = min(scan[:, 0].min(), tau[:, 0].min())
def show(name, a, cols, labels):
    print(f"{name}: {len(a)} rows")
    for c, lab in zip(cols, labels):
        v = (a[:, c] - t0) / 100.0
        v = v[a[:, c] > 0]
        if len(v):
            print(f"  {c} {lab:28s} min {v.min():8.2f}  median {np.median(v):8.2f}  p95 {np.percentile(v, 95):8.2f}  max {v.max():8.2f} us  (n={len(v)})")
show("scan waves", scan, [0, 1, 2, 7, 3, 4, 5, 6], ["entry", "query image built", "first block done", "second block done", "thresholds seen", "last block done", "parked tiles re-filtered", "staging flushed"])
show("threshold waves", tau, range(5), ["entry", "arrival hint seen", "sample complete", "first threshold published", "done"])
