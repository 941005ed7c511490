"""Properties checked at BASELINE.json's full stage-1 shape (per-GPU shard of the
10M x 768 fp16 config = 1.25M rows; plus the 1-GPU 10M case when memory allows),
where a float64 oracle over everything would take too long:
  * a planted document (query itself, scaled) is returned first with its exact id;
  * results are sorted, ids unique and in range;
  * the filter path and the exact dense path agree bit for bit;
  * a sub-sample of queries is checked against the oracle in full."""
import numpy as np
import pytest

from helpers import check_topk
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
        assert In[qi, 0] == row and abs(Dn[qi, 0] - 1.0) < 2e-3
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
