"""Randomised soak of the stage-1 entry points: many (N, d, k, B, dtype) combinations and
submission modes against the oracle, plus a long pipelined stream of batches whose results
must equal the synchronous ones bit for bit (guards the event/stream choreography)."""
import numpy as np
import pytest

from helpers import check_topk, make_corpus
from oracle import oracle

pytestmark = pytest.mark.gpu


def test_random_shapes_against_oracle():
    import torch
    from tristage_rag_amd.index import FlatIPIndex
    import os
    seed = int(os.environ.get("TS_STRESS_SEED", "2024"))     # other seeds: one-off soaks, e.g. TS_STRESS_SEED=7
    rng = np.random.default_rng(seed)
    for trial in range(int(os.environ.get("TS_STRESS_TRIALS", "24"))):
        dtype = ["f16", "bf16", "f32"][trial % 3]
        d = int(rng.choice([8, 33, 64, 100, 128, 384, 768]))
        n = int(rng.choice([1, 31, 32, 33, 1000, 4096, 33_000, 45_017, 90_000]))
        k = int(rng.choice([1, 5, 64, 257, 1000]))
        B = int(rng.choice([1, 2, 31, 32, 33, 64, 65, 100]))
        if n * d > 40_000_000:
            n = 40_000_000 // d
        corpus = make_corpus(n, d, seed=seed * 131 + trial, dtype=dtype)
        if n > 10:
            corpus[rng.integers(0, n, size=n // 10)] = corpus[0]          # sprinkle exact ties
        queries = make_corpus(B, d, seed=seed * 131 + 1000 + trial, dtype=dtype)
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
        outs = [idx.search(qs[i % 8], 500, async_=True, inputs_ready=True) for i in range(150)]   # > one finish() window (240 passes of 32)
        idx.finish()
        for i, (D, I) in enumerate(outs):
            assert torch.equal(I, want[i % 8][1]) and torch.equal(D, want[i % 8][0]), (rep, i)
    check_topk(want[0][0].cpu().numpy()[:3], want[0][1].cpu().numpy()[:3], corpus, qs[0].float().cpu().numpy()[:3], 500)
    idx.close()


def test_concurrent_host_threads():
    """Several host threads at once (ctypes releases the GIL): two index handles searched
    concurrently and the stateless MaxSim entry points sharing one stream's scratch buffer while
    it grows.  Every result must equal its single-threaded value."""
    import threading
    import torch
    from tristage_rag_amd.index import FlatIPIndex, maxsim_indexed
    from helpers import make_corpus
    rng = np.random.default_rng(3)
    d = 128
    idxs, qs, want = [], [], []
    for t in range(2):
        c = make_corpus(60_000 + 7 * t, d, seed=50 + t, dtype="f16")
        idx = FlatIPIndex(d, dtype="f16")
        idx.add(torch.from_numpy(c).cuda().half())
        q = torch.from_numpy(make_corpus(64, d, seed=60 + t, dtype="f16")).cuda().half()
        idxs.append(idx); qs.append(q); want.append(idx.search(q, 100))
    H = 128
    lens = rng.integers(1, 150, size=6000)
    store = torch.from_numpy(rng.standard_normal((int(lens.sum()), H)).astype(np.float32)).cuda().bfloat16()
    starts = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)[:-1]])).cuda()
    tl = torch.from_numpy(lens.astype(np.int32)).cuda()
    q2 = torch.from_numpy(rng.standard_normal((20, H)).astype(np.float32)).cuda().bfloat16()
    sizes = [50, 700, 3000, 6000]
    ms_want = {n: maxsim_indexed(q2, store, starts[:n], tl[:n]) for n in sizes}
    torch.cuda.synchronize()
    errors = []

    def search_worker(t):
        try:
            for _ in range(20):
                D, I = idxs[t].search(qs[t], 100)
                assert torch.equal(I, want[t][1]) and torch.equal(D, want[t][0])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def maxsim_worker(order):
        try:
            for _ in range(15):
                for n in order:
                    assert torch.equal(maxsim_indexed(q2, store, starts[:n], tl[:n]), ms_want[n])
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=search_worker, args=(t,)) for t in range(2)]
    threads += [threading.Thread(target=maxsim_worker, args=(o,)) for o in (sizes, sizes[::-1])]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for idx in idxs:
        idx.close()


def test_one_handle_many_host_threads():
    """SURVEY.md 8b threading contract: ts_index_search on ONE handle from several host threads with distinct
    streams and output buffers.  Filter path, exact dense path (k above the filter limit), the exact FALLBACK of
    a failing filter (20 000 exact ties overflow the candidate list; the redo re-reads the call's own query
    image), host-pointer calls and ts_index_scores all run at once; every result equals its single-threaded
    value bit for bit."""
    import threading
    import torch
    from tristage_rag_amd.index import FlatIPIndex
    d, n = 128, 200_000
    corpus = make_corpus(n, d, seed=91, dtype="f16")
    base = make_corpus(64, d, seed=92, dtype="f16")
    q33 = make_corpus(33, d, seed=93, dtype="f16")
    # 20 000 copies of one row that is (nearly) orthogonal to the 97 queries of the two filter-path jobs:
    # for them it scores ~0 and never reaches a threshold, for the query equal to it it is 20 000 exact ties
    qmat, _ = np.linalg.qr(np.concatenate([base, q33]).T.astype(np.float64))
    v = np.random.default_rng(90).standard_normal(d)
    v -= qmat @ (qmat.T @ v)
    dup = oracle.quantize((v / np.linalg.norm(v)).astype(np.float32)[None, :], "f16")[0]
    corpus[30_000:50_000] = dup                             # massive exact ties
    idx = FlatIPIndex(d, dtype="f16")
    idx.add(torch.from_numpy(corpus).cuda().half())
    tie_q = base.copy()
    tie_q[3] = dup                                          # this query's top 20 000 scores are equal
    jobs = [dict(q=base, k=100), dict(q=q33, k=1000),
            dict(q=tie_q, k=500), dict(q=make_corpus(64, d, seed=94, dtype="f16"), k=3000),
            dict(q=make_corpus(5, d, seed=95, dtype="f16"), k=10, host=True),
            dict(q=make_corpus(40, d, seed=96, dtype="f16"), scores=True)]
    want, paths = [], []
    for j in jobs:                                          # single-threaded reference values
        if j.get("scores"):
            want.append(idx.scores(torch.from_numpy(j["q"]).cuda().half()).clone())
            paths.append("scores")
        elif j.get("host"):
            want.append(idx.search(j["q"], j["k"]))
            paths.append(idx.last_search_info()["path"])
        else:
            D, I = idx.search(torch.from_numpy(j["q"]).cuda().half(), j["k"])
            want.append((D.clone(), I.clone()))
            paths.append(idx.last_search_info()["path"])
    assert paths[:4] == ["filter", "filter", "filter+dense-fallback", "dense"], paths
    check_topk(want[2][0].cpu().numpy()[:5], want[2][1].cpu().numpy()[:5], corpus, tie_q[:5], 500)
    torch.cuda.synchronize()
    errors = []

    def worker(t):
        try:
            j = jobs[t % len(jobs)]
            w = want[t % len(jobs)]
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                qd = None if j.get("host") else torch.from_numpy(j["q"]).cuda().half()
                for rep in range(12):
                    if j.get("scores"):
                        assert torch.equal(idx.scores(qd), w)
                    elif j.get("host"):
                        D, I = idx.search(j["q"], j["k"])
                        assert np.array_equal(I, w[1]) and np.array_equal(D, w[0])
                    else:
                        D = torch.empty((qd.shape[0], j["k"]), dtype=torch.float32, device="cuda")
                        I = torch.empty((qd.shape[0], j["k"]), dtype=torch.int64, device="cuda")
                        idx.search(qd, j["k"], out=(D, I))
                        assert torch.equal(I, w[1]) and torch.equal(D, w[0]), (t, rep)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(9)]   # 9 threads > 4 workspace sets
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    idx.close()
