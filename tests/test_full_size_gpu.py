"""Properties checked at BASELINE.json's full stage-1 shape (per-GPU shard of the
10M x 768 fp16 config = 1.25M rows; plus the 1-GPU 10M case when memory allows),
where a float64 oracle over everything would take too long:
  * a planted document (query itself, scaled) is returned first with its exact id;
  * results are sorted, ids unique and in range;
  * the filter path and the exact dense path agree bit for bit;
  * a sub-sample of queries is checked against the oracle in full (1.25 M rows), or — at 10 M and
    50 M rows — against an independent fp32 reference with every id difference explained by a
    float64 near-tie (helpers.check_topk_sparse; no "at most N mismatches" allowances)."""
import numpy as np
import pytest

from helpers import check_topk, check_topk_sparse
from oracle import oracle

pytestmark = pytest.mark.gpu


def _gen_on_gpu(torch, n, d, seed, dtype):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn((n, d), generator=g, device="cuda", dtype=torch.float32)
    x = x / (x.norm(dim=1, keepdim=True) + 1e-8)
    return x.to(dtype)


def test_shard_of_headline_config():
    import torch
    from tristage_rag_amd.index import FlatIPIndex
    n, d, k, B = 1_250_000, 768, 1000, 64
    idx = FlatIPIndex(d, dtype="f16")
    idx.reserve(n)
    chunk = 250_000
    keep = []
    for c in range(n // chunk):
        x = _gen_on_gpu(torch, chunk, d, 1234 + c, torch.float16)
        idx.add(x)
        keep.append(x)
    q = _gen_on_gpu(torch, B, d, 4321, torch.float16)
    planted = [(5, 123_456), (17, 1_249_999), (63, 0)]
    # plant: row := query (unit norm) -> score 1.0, far above every random row (~0.2 max)
    corpus = torch.cat(keep, 0)
    del keep
    for qi, row in planted:
        corpus[row] = q[qi]
    idx.reset()
    idx.add(corpus)
    assert idx.ntotal == n
    D, I = idx.search(q, k)
    assert idx.last_search_info()["path"] == "filter"
    Dn, In = D.cpu().numpy(), I.cpu().numpy()
    for qi, row in planted:
        assert In[qi, 0] == row and abs(Dn[qi, 0] - 1.0) < 1e-3
    assert (np.diff(Dn, axis=1) <= 0).all()
    assert In.min() >= 0 and In.max() < n
    assert all(len(set(r.tolist())) == k for r in In)
    D2, I2 = idx.search(q, k, exact_dense=True)
    assert torch.equal(I, I2) and torch.equal(D, D2)
    # full oracle check on 4 of the 64 queries
    sel = [0, 5, 31, 63]
    c32 = corpus.float().cpu().numpy()
    q32 = q.float().cpu().numpy()
    check_topk(Dn[sel], In[sel], c32, q32[sel], k)
    idx.close()


def test_headline_config_10m_x_768_on_one_gpu():
    """The 1-GPU bench workload itself (10 M x 768 fp16, B=64, k=1000), checked against an
    independent fp32 torch reference (rocBLAS GEMM + torch.topk on the same quantised data)
    for a few queries, plus the size-independent properties."""
    import torch
    from tristage_rag_amd.index import FlatIPIndex
    n, d, k, B = 10_000_000, 768, 1000, 64
    free, _ = torch.cuda.mem_get_info()
    if free < 60e9:
        pytest.skip("needs ~45 GB of free HBM")
    idx = FlatIPIndex(d, dtype="f16")
    idx.reserve(n)
    chunk = 500_000
    blocks = []
    for c in range(n // chunk):
        x = _gen_on_gpu(torch, chunk, d, 1234 + c, torch.float16)
        blocks.append(x)
        idx.add(x)
    q = _gen_on_gpu(torch, B, d, 4321, torch.float16)
    D, I = idx.search(q, k)
    assert idx.last_search_info()["path"] == "filter"
    assert bool((D[:, 1:] <= D[:, :-1]).all()) and int(I.min()) >= 0 and int(I.max()) < n
    assert all(len(set(r.tolist())) == k for r in I.cpu().numpy()[:8])
    D2, I2 = idx.search(q, k, exact_dense=True)                 # chunked materialise + radix select
    assert torch.equal(I, I2) and torch.equal(D, D2)
    # independent reference for 4 queries
    sel = [0, 21, 42, 63]
    qs = q[sel].float()
    ref = torch.cat([qs @ b.float().T for b in blocks], dim=1)  # [4, n] fp32
    Dr, Ir = torch.topk(ref, k, dim=1)
    def fetch(ids):                                              # corpus rows by global id
        t = torch.as_tensor(ids, device="cuda")
        out = torch.empty((len(ids), d), dtype=torch.float32, device="cuda")
        for c, b in enumerate(blocks):
            m = (t >= c * chunk) & (t < (c + 1) * chunk)
            if bool(m.any()):
                out[m] = b[t[m] - c * chunk].float()
        return out.cpu().numpy()

    qn = q.float().cpu().numpy()
    for j, qi in enumerate(sel):                                 # every id difference explained by a float64 near-tie
        check_topk_sparse(D[qi].cpu().numpy(), I[qi].cpu().numpy(), Ir[j].cpu().numpy(), fetch, qn[qi])
        assert torch.allclose(D[qi], Dr[j], atol=1e-3)
    idx.close()


def test_cfg4_corpus_50m_x_1024_bf16_on_one_gpu():
    """BASELINE.json configs[4]'s WHOLE corpus (50 M x 1024 bf16 = 102.4 GB) resident on one
    MI355X (288 GB): 64-bit addressing of the tiled corpus (16-byte unit 2^32 is row
    33,554,432), planted rows at both ends and across that boundary, sortedness / uniqueness,
    filter path == exact dense path, and an independent fp32 torch reference (blocks
    regenerated from their seeds) for two queries."""
    import torch
    from tristage_rag_amd.index import FlatIPIndex
    n, d, k, B = 50_000_000, 1024, 1000, 64
    free, _ = torch.cuda.mem_get_info()
    if free < 150e9:
        pytest.skip("needs ~110 GB of free HBM")
    q = _gen_on_gpu(torch, B, d, 4321, torch.bfloat16)
    planted = {0: 0, 7: 33_554_431, 8: 33_554_432, 21: 41_000_001, 63: n - 1}
    chunk = 500_000
    idx = FlatIPIndex(d, dtype="bf16")
    idx.reserve(n)

    def block(c):
        x = _gen_on_gpu(torch, chunk, d, 99_000 + c, torch.bfloat16)
        for qi, row in planted.items():
            if c * chunk <= row < (c + 1) * chunk:
                x[row - c * chunk] = q[qi]
        return x

    for c in range(n // chunk):
        idx.add(block(c))
    assert idx.ntotal == n
    D, I = idx.search(q, k)
    assert idx.last_search_info()["path"] == "filter"
    for qi, row in planted.items():
        assert int(I[qi, 0]) == row and abs(float(D[qi, 0]) - 1.0) < 1e-3
    assert bool((D[:, 1:] <= D[:, :-1]).all()) and int(I.min()) >= 0 and int(I.max()) < n
    assert all(len(set(r.tolist())) == k for r in I.cpu().numpy()[:8])
    D2, I2 = idx.search(q, k, exact_dense=True)
    assert torch.equal(I, I2) and torch.equal(D, D2)
    sel = [8, 40]
    qs = q[sel].float()
    ref = torch.cat([qs @ block(c).float().T for c in range(n // chunk)], dim=1)   # [2, n] fp32
    Dr, Ir = torch.topk(ref, k, dim=1)
    def fetch(ids):                                              # rows by global id: their blocks are regenerated
        ids = np.asarray(ids)
        out = np.empty((len(ids), d), dtype=np.float32)
        for c in np.unique(ids // chunk):
            m = (ids // chunk) == c
            out[m] = block(int(c))[torch.as_tensor(ids[m] - c * chunk, device="cuda")].float().cpu().numpy()
        return out

    qn = q.float().cpu().numpy()
    for j, qi in enumerate(sel):                                 # every id difference explained by a float64 near-tie
        # north_star's bar, 1e-3 (bf16 products are exact in fp32; only the summation order differs: ~1e-6)
        check_topk_sparse(D[qi].cpu().numpy(), I[qi].cpu().numpy(), Ir[j].cpu().numpy(), fetch, qn[qi],
                          score_tol=1e-3)
        assert torch.allclose(D[qi], Dr[j], atol=1e-3)
    idx.close()


def test_one_launch_scan_above_22m_rows_stays_on_the_filter_path():
    """above ~22 M rows the sample needed four rounds per scan wave, but a wave can deliver at
    most three before it blocks for the thresholds — scan and threshold waves timed each other out (40-60 ms) and
    every batch fell back to the dense path; the waves that left early also put the arrival counter behind its
    goal for good.  plan_fused now caps the sample at three rounds: the forced one-launch search of a 24 M-row
    corpus must take the filter path (no fallback), repeatedly, and equal the five-launch path bit for bit."""
    import torch
    from tristage_rag_amd.index import FlatIPIndex
    n, d, k, B = 24_000_000, 128, 100, 64
    free, _ = torch.cuda.mem_get_info()
    if free < 20e9:
        pytest.skip("needs ~10 GB of free HBM")
    idx = FlatIPIndex(d, dtype="f16")
    idx.reserve(n)
    chunk = 2_000_000
    for c in range(n // chunk):
        idx.add(_gen_on_gpu(torch, chunk, d, 7000 + c, torch.float16))
    q = _gen_on_gpu(torch, B, d, 4321, torch.float16)
    D0, I0 = idx.search(q, k, classic=True)
    assert idx.last_search_info()["path"] == "filter" and not idx.last_search_info()["one_launch"]
    for rep in range(3):        # a launch that gave up would also slow down or break the NEXT one on its workspace set
        D1, I1 = idx.search(q, k, one_launch=True)
        info = idx.last_search_info()
        assert info["path"] == "filter" and info["one_launch"], (rep, info)
        assert torch.equal(I1, I0) and torch.equal(D1, D0)
    outs = [idx.search(q, k, one_launch=True, async_=True) for _ in range(6)]    # every workspace set, back to back
    assert idx.finish() == []                                                   # no batch needed the exact redo
    for Da, Ia in outs:
        assert torch.equal(Ia, I0) and torch.equal(Da, D0)
    idx.close()
