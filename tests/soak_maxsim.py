# one-off randomized soak of the streaming MaxSim against the oracle (run by hand: python tests/soak_maxsim.py SEED; not collected by pytest)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import oracle
from tristage_rag_amd.index import maxsim_indexed, maxsim_indexed_batch, maxsim
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for trial in range(60):
    dtype = ["bf16", "f16", "f32"][trial % 3]
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    H = int(rng.choice([8, 16, 24, 40, 64, 96, 128, 200, 256, 384, 512, 768, 1024, 1536]))
    n_store = int(rng.integers(1, 300))
    lens = rng.integers(0, int(rng.choice([3, 40, 193, 400])), size=n_store)
    store = oracle.quantize(rng.standard_normal((int(lens.sum()) + 1, H)).astype(np.float32), dtype)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    nq = int(rng.integers(1, 6))
    lqs = rng.integers(1, int(rng.choice([8, 33, 70, 200])), size=nq)
    ncs = rng.integers(0, n_store + 1, size=nq)
    picks = [rng.integers(0, n_store, size=c) for c in ncs]
    qs = [oracle.quantize(rng.standard_normal((L, H)).astype(np.float32), dtype) for L in lqs]
    q_off = np.concatenate([[0], np.cumsum(lqs)]); c_off = np.concatenate([[0], np.cumsum(ncs)])
    pk = np.concatenate(picks).astype(np.int64) if c_off[-1] else np.zeros(0, np.int64)
    ts = torch.from_numpy(store).cuda().to(tdt)
    mode = "maxsim" if trial % 2 else "colbert"
    got = maxsim_indexed_batch(torch.from_numpy(np.concatenate(qs)).cuda().to(tdt), q_off, ts,
                               torch.from_numpy(starts[pk]).cuda(), torch.from_numpy(lens[pk].astype(np.int32)).cuda(), c_off, mode=mode).cpu().numpy()
    tol = 4e-6 if dtype == "f32" else 2e-5
    for j in range(nq):
        docs = [store[starts[i]: starts[i] + lens[i]] for i in picks[j]]
        if not docs: continue
        want = oracle.maxsim_scores(qs[j], docs, mode)
        err = np.abs(got[c_off[j]:c_off[j + 1]] - want).max()
        if not err < tol:
            bad += 1; print("MISMATCH", trial, dtype, H, lqs[j], len(docs), mode, err)
print("soak done, mismatches:", bad)
