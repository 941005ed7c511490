# one-off randomized soak of the projection kernels (run by hand: python tests/soak_linear.py SEED; not collected by pytest):
# ts_linear_act against torch within two 16-bit steps; ts_linear_add_layernorm and ts_mlp_add_layernorm against the kernel
# sequences they replace, bit for bit — random row counts (around tile and grid multiples), block counts, reduction lengths,
# with / without bias, residual, beta, both 16-bit types.
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn.functional as F
from tristage_rag_amd.index import TiledLinear, add_layernorm, mlp_add_layernorm
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
g = torch.Generator(device="cuda").manual_seed(int(rng.integers(1 << 30)))
bad = 0
def rows():
    base = int(rng.choice([1, 31, 32, 33, 95, 96, 97, 191, 193, 4097, 96 * 256 - 1, 96 * 256 + 1, 64 * 256 + 5, 70000]))
    return max(1, base + int(rng.integers(-3, 4)) * int(rng.integers(0, 2)))
for trial in range(40):
    tdt = torch.bfloat16 if trial % 2 else torch.float16
    step = 2.0 ** (-8 if tdt == torch.bfloat16 else -11)
    M = rows()
    kind = trial % 4
    if kind in (0, 1):      # ts_linear_act
        K = int(rng.choice([128, 256, 384])); N = 32 * int(rng.integers(1, 49))
        x = (torch.randn((M, K), generator=g, device="cuda") * 0.8).to(tdt)
        w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(tdt)
        b = (torch.randn((N,), generator=g, device="cuda") * 0.1).to(tdt) if rng.integers(0, 2) else None
        lin = TiledLinear(w, b)
        for gelu in (False, True):
            ref = F.linear(x, w, b); ref = F.gelu(ref) if gelu else ref
            got = lin(x, gelu=gelu)
            err = float((got.float() - ref.float()).abs().max())
            if not err <= 2 * step * max(1.0, float(ref.abs().max())):
                bad += 1; print("linear_act mismatch", tdt, M, K, N, gelu, err)
    elif kind == 2:         # ts_linear_add_layernorm == linear_act + add_layernorm
        K = int(rng.choice([384, 768, 1536])); N = 32 * int(rng.integers(2, 13))
        x = (torch.randn((M, K), generator=g, device="cuda") * 0.8).to(tdt)
        w = (torch.randn((N, K), generator=g, device="cuda") * 0.04).to(tdt)
        b = (torch.randn((N,), generator=g, device="cuda") * 0.1).to(tdt) if rng.integers(0, 2) else None
        res = torch.randn((M, N), generator=g, device="cuda") if rng.integers(0, 2) else None
        gamma = 1.0 + 0.1 * torch.randn((N,), generator=g, device="cuda")
        beta = 0.1 * torch.randn((N,), generator=g, device="cuda") if rng.integers(0, 2) else None
        lin = TiledLinear(w, b, with_layernorm=True)
        y32, ylp = lin.add_layernorm(x, res, gamma, beta, 1e-12)
        e32, elp = add_layernorm(lin(x), res, gamma, beta, 1e-12, lp_dtype=tdt)
        # (N <= 128: ts_add_layernorm keeps one chunk per lane there and the compiler contracts that instance differently: 1e-6)
        same = (torch.equal(y32, e32) and torch.equal(ylp, elp)) if N > 128 else float((y32 - e32).abs().max()) <= 2e-6
        if not same:
            bad += 1; print("linear_add_layernorm mismatch", tdt, M, K, N, float((y32 - e32).abs().max()))
    else:                   # ts_mlp_add_layernorm == linear_act(gelu) + linear_add_layernorm
        H, I = 384, 384 * int(rng.integers(1, 5))
        x = (torch.randn((M, H), generator=g, device="cuda") * float(rng.choice([0.3, 0.8, 3.0]))).to(tdt)
        w1 = (torch.randn((I, H), generator=g, device="cuda") * 0.06).to(tdt)
        b1 = (torch.randn((I,), generator=g, device="cuda") * 0.1).to(tdt) if rng.integers(0, 2) else None
        w2 = (torch.randn((H, I), generator=g, device="cuda") * 0.03).to(tdt)
        b2 = (torch.randn((H,), generator=g, device="cuda") * 0.1).to(tdt) if rng.integers(0, 2) else None
        res = torch.randn((M, H), generator=g, device="cuda") if rng.integers(0, 2) else None
        gamma = 1.0 + 0.1 * torch.randn((H,), generator=g, device="cuda")
        beta = 0.1 * torch.randn((H,), generator=g, device="cuda") if rng.integers(0, 2) else None
        up, down = TiledLinear(w1, b1), TiledLinear(w2, b2, with_layernorm=True)
        m32, mlp = mlp_add_layernorm(up, down, x, res, gamma, beta, 1e-12)
        w32, wlp = down.add_layernorm(up(x, gelu=True), res, gamma, beta, 1e-12)
        if not (torch.equal(m32, w32) and torch.equal(mlp, wlp)):
            bad += 1; print("mlp_add_layernorm mismatch", tdt, M, I, float((m32 - w32).abs().max()))
torch.cuda.synchronize()
print("soak done, mismatches:", bad)
