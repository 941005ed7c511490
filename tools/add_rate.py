import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from tristage_rag_amd.index import FlatIPIndex
for dt_in, dt_st, norm in ((torch.float16, "f16", False), (torch.float32, "f16", True), (torch.bfloat16, "bf16", False)):
    x = torch.randn((500_000, 768), device="cuda", dtype=torch.float32).to(dt_in)
    idx = FlatIPIndex(768, dtype=dt_st)
    idx.reserve(4_000_000)
    idx.add(x, normalize=norm) if norm else idx.add(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        idx.add(x, normalize=norm) if norm else idx.add(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 6
    print(f"add 500k x 768 {dt_in} -> {dt_st} normalize={norm}: {dt*1e3:.3f} ms  ({x.numel()*x.element_size()/dt/1e9:.0f} GB/s in)")
    idx.close()
