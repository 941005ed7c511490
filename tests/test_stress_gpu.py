"""Randomised soak of the stage-1 entry points: many (N, d, k, B, dtype) combinations and
submission modes against the oracle, plus a long pipelined stream of batches whose results
must equal the synchronous ones bit for bit (guards the event/stream choreography)."""
import numpy as np
import pytest

from helpers import check_topk, make_corpus

pytestmark = pytest.mark.gpu


def test_random_shapes_against_oracle():
    import torch
    from tristage_rag_amd.index import FlatIPIndex
    rng = np.random.default_rng(2024)
    for trial in range(24):
        dtype = ["f16", "bf16", "f32"][trial % 3]
        d = int(rng.choice([8, 33, 64, 100, 128, 384, 768]))
        n = int(rng.choice([1, 31, 32, 33, 1000, 4096, 33_000, 45_017, 90_000]))
        k = int(rng.choice([1, 5, 64, 257, 1000]))
        B = int(rng.choice([1, 2, 31, 32, 33, 64, 65, 100]))
        if n * d > 40_000_000:
            n = 40_000_000 // d
        corpus = make_corpus(n, d, seed=trial, dtype=dtype)
        if n > 10:
            corpus[rng.integers(0, n, size=n // 10)] = corpus[0]          # sprinkle exact ties
        queries = make_corpus(B, d, seed=1000 + trial, dtype=dtype)
        idx = FlatIPIndex(d, dtype=dtype)
        cut = int(rng.integers(0, n + 1))
        if cut:
            idx.add(corpus[:cut])
        if cut < n:
            idx.add(torch.from_numpy(corpus[cut:]).cuda())                # host + device appends
        assert idx.ntotal == n
        if trial % 2:
            D, I = idx.search(queries, k)
        else:
            Dt, It = idx.search(torch.from_numpy(queries).cuda(), k, async_=True, inputs_ready=bool(trial % 4))
            idx.finish()
            D, I = Dt.cpu().numpy(), It.cpu().numpy()
        check_topk(D, I, corpus, queries, k)
        idx.close()


def test_long_pipelined_stream_is_deterministic():
    import torch
    from tristage_rag_amd.index import FlatIPIndex
    corpus = make_corpus(400_000, 128, seed=77, dtype="f16")
    idx = FlatIPIndex(128, dtype="f16")
    idx.add(corpus)
    qs = [torch.from_numpy(make_corpus(64, 128, seed=500 + i, dtype="f16")).cuda().half() for i in range(8)]
    torch.cuda.synchronize()
    want = [idx.search(q, 500) for q in qs]
    for rep in range(4):
        outs = [idx.search(qs[i % 8], 500, async_=True, inputs_ready=True) for i in range(120)]   # > one finish() window
        idx.finish()
        for i, (D, I) in enumerate(outs):
            assert torch.equal(I, want[i % 8][1]) and torch.equal(D, want[i % 8][0]), (rep, i)
    check_topk(want[0][0].cpu().numpy()[:3], want[0][1].cpu().numpy()[:3], corpus, qs[0].float().cpu().numpy()[:3], 500)
    idx.close()
