#!/usr/bin/env python3
"""ts_linear_act on the stage-3 (MiniLM-L6) projection shapes, M = 172 032 tokens: time per call against
F.linear (+ gelu), for the library in TRISTAGE_LIB (a -DTS_TUNING build honours TS_FS_QH = rows per workgroup / 32)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tristage_rag_amd.index import TiledLinear

M = int(os.environ.get("PROBE_M", 172032))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
out = {"M": M, "TS_FS_QH": os.environ.get("TS_FS_QH")}
for name, N, K, act in (("qkv", 1152, 384, 0), ("attn_out", 384, 384, 0), ("up_gelu", 1536, 384, 1)):
    w = (torch.randn((N, K), generator=g, device=dev) * 0.05).bfloat16()
    b = (torch.randn((N,), generator=g, device=dev) * 0.1).bfloat16()
    x = torch.randn((M, K), generator=g, device=dev).bfloat16()
    tl = TiledLinear(w, b)
    ref = F.linear(x, w, b)
    ref = F.gelu(ref) if act else ref
    y = tl(x, gelu=bool(act))
    err = float((y.float() - ref.float()).abs().max())

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return a.elapsed_time(e) / n
    t_hip = timed(lambda: tl(x, gelu=bool(act)))
    t_lib = timed(lambda: F.gelu(F.linear(x, w, b)) if act else F.linear(x, w, b))
    flops = 2.0 * M * N * K
    out[name] = {"hip_ms": round(t_hip, 4), "lib_ms": round(t_lib, 4), "hip_TFLOPs": round(flops / t_hip / 1e9, 1),
                 "max_abs_err_vs_lib": err}
print(json.dumps(out))
