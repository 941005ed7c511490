#!/usr/bin/env python3
"""What a write-only, a read-only and a copy stream reach on this box (torch elementwise kernels over 8 GiB):
the ceilings the write-heavy kernels of the encoder forwards (add_layernorm: 6 B read + 6 B written per element; the
projections' 16-bit outputs) are priced against in DESIGN.md 4.7."""
import json
import torch
n = 2 * 1024 ** 3                         # 2 Gi float32 = 8 GiB
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
gb = n * 4 / 1e9
out = {"buffer_GB": round(gb, 2)}
t = timed(lambda: a.fill_(1.0)); out["write_only_fill_TBps"] = round(gb / t / 1e3, 3)
t = timed(lambda: a.zero_()); out["write_only_memset_TBps"] = round(gb / t / 1e3, 3)
t = timed(lambda: a.sum()); out["read_only_sum_TBps"] = round(gb / t / 1e3, 3)
t = timed(lambda: b.copy_(a)); out["copy_read_plus_write_TBps"] = round(2 * gb / t / 1e3, 3)
t = timed(lambda: torch.add(a, 1.0, out=b)); out["add_read_plus_write_TBps"] = round(2 * gb / t / 1e3, 3)
print(json.dumps(out))
