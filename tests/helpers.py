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


def check_topk(D, I, corpus, queries, k, id_offset=0, score_tol=1e-3, tie_tol=2e-6):
    """D, I from the HIP path vs the float64 oracle on the SAME quantised inputs.

    Bar (BASELINE.json north_star): bit-exact top-k doc ids, scores within 1e-3.
    fp32 accumulation order on the GPU differs from float64, so two scores
    closer than `tie_tol` may legitimately swap; such a swap is accepted only if
    the oracle's own float64 scores of the two ids differ by < tie_tol.
    Returns the number of positions where ids differed (all explained)."""
    D = np.asarray(D)
    I = np.asarray(I)
    D0, I0 = oracle.ip_topk(corpus, queries, k, f64=True, id_offset=id_offset)
    n = corpus.shape[0]
    kk = min(k, n)
    assert D.shape == D0.shape and I.shape == I0.shape
    # padding
    assert (I[:, kk:] == -1).all()
    assert (D[:, kk:] <= -3.0e38).all()
    swaps = 0
    for q in range(queries.shape[0]):
        if np.array_equal(I[q, :kk], I0[q, :kk]):
            np.testing.assert_allclose(D[q, :kk], D0[q, :kk], atol=score_tol, rtol=0)
            continue
        s = oracle.scores_f64(corpus, queries[q])
        got = I[q, :kk] - id_offset
        assert got.min() >= 0 and got.max() < n, "id out of range"
        assert len(set(got.tolist())) == kk, "duplicate ids in the result"
        sg = s[got]
        # returned scores match the oracle's score of the SAME id
        np.testing.assert_allclose(D[q, :kk], sg, atol=score_tol, rtol=0)
        # order is descending up to near-ties; exact ties by ascending id
        dif = np.diff(sg)
        assert (dif <= tie_tol).all(), f"query {q}: result not sorted (max inversion {dif.max()})"
        # nothing better than the boundary was left out
        kth = np.sort(s)[::-1][kk - 1]
        assert sg.min() >= kth - tie_tol, f"query {q}: a returned id is below the k-th best score"
        left_out = np.setdiff1d(I0[q, :kk] - id_offset, got)
        assert (s[left_out] <= sg.min() + tie_tol).all(), f"query {q}: a better id was left out"
        swaps += int((I[q, :kk] != I0[q, :kk]).sum())
    return swaps
