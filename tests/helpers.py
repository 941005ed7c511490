"""Shared helpers for the parity tests (oracle = checker, HIP path = subject)."""
import numpy as np

from oracle import oracle


def make_corpus(n, d, seed=1234, dtype="f16"):
    """Synthetic corpus per SURVEY.md §8d: standard normal rows, x/(|x|+1e-8),
    then rounded to the storage dtype (returned as float32 values)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d), dtype=np.float32)
    x = x / (np.linalg.norm(x, axis=1, keepdims=True) + 1e-8)
    return oracle.quantize(x.astype(np.float32), dtype)


check_topk = oracle.check_topk                 # the parity rule lives beside the oracle (smoke() uses it too)
check_topk_sparse = oracle.check_topk_sparse
