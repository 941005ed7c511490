"""GPU parity: the HIP stage-1 path (through the C ABI) vs the CPU oracle on
identical, storage-quantised inputs.  Bar: bit-exact top-k ids, scores within
1e-3 (BASELINE.json north_star); accepted id swaps are only those the float64
oracle itself cannot separate (helpers.check_topk)."""
import json
import os

import numpy as np
import pytest

from helpers import check_topk, make_corpus
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def _index(d, dtype, rows=None):
    from tristage_rag_amd.index import FlatIPIndex
    idx = FlatIPIndex(d, dtype=dtype)
    if rows is not None:
        idx.add(rows)
    return idx


@pytest.mark.parametrize("dtype", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("n,d,k,B", [(1000, 96, 10, 5), (5183, 384, 100, 64), (33, 40, 7, 1),
                                      (4097, 128, 1000, 33)])
def test_dense_path_matches_oracle(dtype, n, d, k, B):
    corpus = make_corpus(n, d, seed=1234, dtype=dtype)
    queries = make_corpus(B, d, seed=4321, dtype=dtype)
    idx = _index(d, dtype, corpus)
    assert idx.ntotal == n
    D, I = idx.search(queries, k)
    assert idx.last_search_info()["path"] == "dense"
    check_topk(D, I, corpus, queries, k)
    idx.close()


def test_more_than_64_queries_and_k_larger_than_n():
    corpus = make_corpus(50, 64, dtype="f16")
    queries = make_corpus(150, 64, seed=5, dtype="f16")
    idx = _index(64, "f16", corpus)
    D, I = idx.search(queries, 100)          # k > N: FAISS pads with -1
    check_topk(D, I, corpus, queries, 100)
    assert (I[:, 50:] == -1).all()
    idx.close()


def test_k_beyond_the_select_kernels_and_big_async_batches_have_a_slow_path():
    """limits that used to refuse.  k > 16384 on more than 16384 rows (the select kernels hold
    16384 keys in LDS): every score from the HIP dense scan + one stable device sort — exact, ties by ascending id,
    FAISS padding beyond ntotal; numpy and tensor queries, an id offset (a shard).  More than 256 queries in one
    asynchronous call: sliced."""
    import torch
    n, d = 40_000, 32
    base = make_corpus(n // 2, d, seed=4, dtype="f16")
    corpus = np.concatenate([base, base])                  # every row twice: exact ties far apart in id
    queries = make_corpus(3, d, seed=6, dtype="f16")
    idx = _index(d, "f16", corpus)
    for k in (20_000, 45_000):
        D, I = idx.search(queries, k)
        D0, I0 = oracle.ip_topk(corpus, queries, k)
        kk = min(k, n)
        assert (I[:, kk:] == -1).all() and (I0[:, kk:] == -1).all()
        check_topk(D[:, :kk], I[:, :kk], corpus, queries, kk)
        for q in range(3):                                 # duplicated rows: the smaller id first
            pos = {int(i): r for r, i in enumerate(I[q, :kk])}
            twins = [(i, i + n // 2) for i in range(0, n // 2, 997) if i in pos and i + n // 2 in pos]
            assert twins and all(pos[a] < pos[b] for a, b in twins)
    idx.set_id_offset(1000)
    Dt, It = idx.search(torch.from_numpy(queries).cuda().half(), 20_000)
    assert int(It.min()) == 1000 and int(It.max()) == 1000 + n - 1
    idx.set_id_offset(0)
    # 300 queries in ONE asynchronous call
    big = make_corpus(300, d, seed=8, dtype="f16")
    want_D, want_I = idx.search(big, 50)
    qb = torch.from_numpy(big).cuda().half()
    Da, Ia = idx.search(qb, 50, async_=True)
    assert idx.finish() == []
    assert np.array_equal(Ia.cpu().numpy(), want_I) and np.array_equal(Da.cpu().numpy(), want_D)
    idx.close()


def test_exact_ties_are_ordered_by_ascending_id():
    rng = np.random.default_rng(3)
    base = make_corpus(64, 128, seed=9, dtype="f16")
    corpus = base[rng.integers(0, 64, size=3000)]      # heavy duplication
    queries = make_corpus(8, 128, seed=10, dtype="f16")
    idx = _index(128, "f16", corpus)
    D, I = idx.search(queries, 500)
    D0, I0 = oracle.ip_topk(corpus, queries, 500)
    # within a run of equal GPU scores the ids must ascend
    for q in range(8):
        for r in range(1, 500):
            if D[q, r] == D[q, r - 1]:
                assert I[q, r] > I[q, r - 1]
    check_topk(D, I, corpus, queries, 500)
    idx.close()


@pytest.mark.parametrize("dtype,n,d,k,B", [("f16", 100_000, 128, 100, 64), ("bf16", 70_001, 256, 10, 7),
                                            ("f16", 300_000, 64, 1000, 64), ("f32", 65_536, 64, 50, 32)])
def test_filter_path_matches_oracle_and_dense_path(dtype, n, d, k, B):
    corpus = make_corpus(n, d, seed=1234, dtype=dtype)
    queries = make_corpus(B, d, seed=4321, dtype=dtype)
    idx = _index(d, dtype, corpus)
    D, I = idx.search(queries, k)
    info = idx.last_search_info()
    assert info["path"] == "filter", info
    assert k <= info["max_candidates"] <= 16384
    check_topk(D, I, corpus, queries, k)
    D2, I2 = idx.search(queries, k, exact_dense=True)   # same kernel arithmetic, no filter
    assert idx.last_search_info()["path"] == "dense"
    assert np.array_equal(I, I2)
    assert np.array_equal(D, D2)
    idx.close()


def test_filter_falls_back_when_candidates_overflow():
    # every row identical: every score ties, the candidate lists overflow and the
    # exact dense path must take over (ids 0..k-1 by the tie rule)
    row = make_corpus(1, 128, seed=1, dtype="f16")
    corpus = np.repeat(row, 40_000, axis=0)
    queries = make_corpus(3, 128, seed=2, dtype="f16")
    idx = _index(128, "f16", corpus)
    D, I = idx.search(queries, 20)
    assert idx.last_search_info()["path"] == "filter+dense-fallback"
    assert (I == np.arange(20)[None, :]).all()
    idx.close()


def test_incremental_add_unaligned_and_reconstruct():
    d = 72
    parts = [make_corpus(n, d, seed=s, dtype="bf16") for n, s in ((45, 1), (1000, 2), (7, 3), (1, 4))]
    idx = _index(d, "bf16")
    for p in parts:
        idx.add(p)
    corpus = np.concatenate(parts, 0)
    assert idx.ntotal == corpus.shape[0]
    np.testing.assert_array_equal(idx.reconstruct_n(0, idx.ntotal), corpus)
    queries = make_corpus(9, d, seed=77, dtype="bf16")
    D, I = idx.search(queries, 30)
    check_topk(D, I, corpus, queries, 30)
    idx.reset()
    assert idx.ntotal == 0
    with pytest.raises(ValueError, match="No documents indexed"):
        idx.search(queries, 5)
    idx.close()


def test_add_rounds_float32_rows_to_storage_dtype_and_normalizes():
    rng = np.random.default_rng(0)
    raw = rng.standard_normal((300, 100)).astype(np.float32) * 3.0
    idx = _index(100, "f16")
    idx.add(raw, normalize=True)     # x/(|x|+1e-8) on device, reference :285-288
    want = oracle.quantize(oracle.normalize_embeddings(raw).astype(np.float32), "f16")
    got = idx.reconstruct_n()
    # device fp32 norm vs numpy's: at most one fp16 ulp apart
    np.testing.assert_allclose(got, want, atol=2 ** -11, rtol=0)
    assert (got != want).mean() < 0.01
    idx.close()


def test_torch_tensor_interface_and_id_offset(torch_mod):
    torch = torch_mod
    corpus = make_corpus(2000, 128, dtype="f16")
    queries = make_corpus(16, 128, seed=8, dtype="f16")
    idx = _index(128, "f16")
    idx.add(torch.from_numpy(corpus).cuda().half())
    idx.set_id_offset(1_000_000_000_000)
    D, I = idx.search(torch.from_numpy(queries).cuda().half(), 25)
    assert D.is_cuda and I.dtype == torch.int64
    check_topk(D.cpu().numpy(), I.cpu().numpy(), corpus, queries, 25, id_offset=1_000_000_000_000)
    idx.close()


def test_merge_topk_matches_oracle(torch_mod):
    torch = torch_mod
    rng = np.random.default_rng(5)
    R, B, k = 8, 64, 1000
    s = np.sort(rng.standard_normal((R, B, k)).astype(np.float32), axis=2)[:, :, ::-1].copy()
    ids = np.stack([np.sort(rng.choice(10**6, size=(B, k)), axis=1) + r * 10**6 for r in range(R)]).astype(np.int64)
    s[1, :, 10:20] = s[0, :, 10:20]                       # cross-list exact score ties
    s[1] = np.sort(s[1], axis=1)[:, ::-1]
    s[7, 3, 500:] = -3.4028234663852886e38                # a short list: padded tail
    ids[7, 3, 500:] = -1
    from tristage_rag_amd.index import merge_topk
    D, I = merge_topk(torch.from_numpy(s).cuda(), torch.from_numpy(ids).cuda())
    D0, I0 = oracle.merge_topk(s, ids, k)
    assert np.array_equal(I.cpu().numpy(), I0)
    np.testing.assert_array_equal(D.cpu().numpy(), D0)


@pytest.mark.parametrize("dtype", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("mode", ["maxsim", "colbert"])
def test_maxsim_kernel_matches_oracle(torch_mod, dtype, mode):
    torch = torch_mod
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    rng = np.random.default_rng(21)
    H, Lq = 96, 19
    lens = [1, 31, 32, 33, 64, 100, 192, 7, 0, 150]
    q = oracle.quantize(rng.standard_normal((Lq, H)).astype(np.float32), dtype)
    docs = [oracle.quantize(rng.standard_normal((L, H)).astype(np.float32), dtype) for L in lens]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    from tristage_rag_amd.index import maxsim
    out = maxsim(torch.from_numpy(q).cuda().to(tdt), torch.from_numpy(np.concatenate(docs, 0)).cuda().to(tdt),
                 torch.from_numpy(off).cuda(), mode=mode).cpu().numpy()
    want = oracle.maxsim_scores(q, docs, mode)
    # bar is 1e-3; the exact-f32 MFMA gets ~1e-7, the 16-bit MFMA path (f32 accumulate) ~1e-6
    np.testing.assert_allclose(out, want, atol=2e-6 if dtype == "f32" else 1e-5, rtol=0)
    assert out[8] == 0.0


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("H,Lq", [(768, 5), (768, 32), (768, 33), (768, 64), (768, 65), (384, 150),
                                  (104, 40), (2048, 70), (4096, 9), (128, 192)])
def test_streaming_maxsim_shapes(torch_mod, dtype, H, Lq):
    """The HBM-streaming MaxSim (ts_maxsim16.hip): one / two query tiles per pass, several
    passes, hidden sizes with and without padded k steps, the fallback for sizes whose query
    image does not fit LDS (H = 4096); ragged candidates incl. empty ones, both entry points
    and both scoring modes."""
    torch = torch_mod
    from tristage_rag_amd.index import maxsim, maxsim_indexed
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    rng = np.random.default_rng(H + Lq)
    n = 1000 if (H == 768 and Lq in (5, 64)) else 150
    lens = rng.integers(0, 193, size=n)
    lens[:6] = [0, 1, 32, 33, 192, 64]
    store = oracle.quantize(rng.standard_normal((int(lens.sum()) + 64, H)).astype(np.float32), dtype)
    store[5] = 0.0                                                       # an all-zero token row
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]]) + 3
    q = oracle.quantize(rng.standard_normal((Lq, H)).astype(np.float32), dtype)
    docs = [store[starts[i]: starts[i] + lens[i]] for i in range(n)]
    tq, ts = torch.from_numpy(q).cuda().to(tdt), torch.from_numpy(store).cuda().to(tdt)
    for mode in ("maxsim", "colbert"):
        want = oracle.maxsim_scores(q, docs, mode)
        got = maxsim_indexed(tq, ts, torch.from_numpy(starts).cuda(), torch.from_numpy(lens.astype(np.int32)).cuda(),
                             mode=mode).cpu().numpy()
        np.testing.assert_allclose(got, want, atol=1e-5, rtol=0)
        assert got[0] == 0.0
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        packed = torch.from_numpy(np.concatenate(docs, 0)).cuda().to(tdt)
        got2 = maxsim(tq, packed, torch.from_numpy(off).cuda(), mode=mode).cpu().numpy()
        np.testing.assert_allclose(got2, want, atol=1e-5, rtol=0)


def test_maxsim_kernel_on_reference_golden_cases(torch_mod):
    torch = torch_mod
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kat.json")))
    from tristage_rag_amd.index import maxsim
    for case in kat["maxsim"]:
        q = torch.tensor(case["q"], dtype=torch.float32).cuda()
        d = torch.tensor(case["d"], dtype=torch.float32).cuda()
        off = torch.tensor([0, d.shape[0]], dtype=torch.int32).cuda()
        for mode in ("maxsim", "colbert"):
            got = float(maxsim(q, d, off, mode=mode)[0])
            assert got == pytest.approx(case[mode], abs=2e-6)      # reference: torch fp32 on CPU


def test_long_query_and_odd_hidden_size(torch_mod):
    torch = torch_mod
    rng = np.random.default_rng(2)
    H, Lq = 50, 192
    q = rng.standard_normal((Lq, H)).astype(np.float32)
    docs = [rng.standard_normal((L, H)).astype(np.float32) for L in (5, 70)]
    off = np.array([0, 5, 75], np.int32)
    from tristage_rag_amd.index import maxsim
    out = maxsim(torch.from_numpy(q).cuda(), torch.from_numpy(np.concatenate(docs, 0)).cuda(),
                 torch.from_numpy(off).cuda()).cpu().numpy()
    np.testing.assert_allclose(out, oracle.maxsim_scores(q, docs), atol=2e-6, rtol=0)


def test_async_searches_match_sync_and_repair_failures(torch_mod):
    torch = torch_mod
    corpus = make_corpus(120_000, 128, dtype="f16")
    idx = _index(128, "f16", corpus)
    qs = [torch.from_numpy(make_corpus(64, 128, seed=100 + i, dtype="f16")).cuda().half() for i in range(5)]
    sync = [idx.search(q, 200) for q in qs]
    outs = [idx.search(q, 200, async_=True) for q in qs]      # enqueued back to back
    assert idx.finish() == []                                  # nothing needed a redo
    for (D, I), (D0, I0) in zip(outs, sync):
        assert torch.equal(I, I0) and torch.equal(D, D0)
    assert idx.last_search_info()["path"] == "filter"
    idx.close()
    # an index where the filter cannot prove exactness (all scores tie): finish() repairs in place
    row = make_corpus(1, 128, seed=1, dtype="f16")
    idx = _index(128, "f16", np.repeat(row, 40_000, axis=0))
    q = torch.from_numpy(make_corpus(4, 128, seed=2, dtype="f16")).cuda().half()
    D, I = idx.search(q, 20, async_=True)
    D2, I2 = idx.search(q, 20, async_=True)
    assert len(idx.finish()) == 2
    assert (I.cpu().numpy() == np.arange(20)[None, :]).all() and torch.equal(I, I2)
    idx.close()


def test_ndcg_at_10_parity_on_scifact_shaped_synthetic_task():
    """BASELINE.json: nDCG@10 within +-0.002 of the reference's exact CPU search.  No MTEB data
    or weights exist offline, so the task is synthetic with the SciFact shape (5183 docs, 300
    queries, d=384, fp32 embeddings as the reference's FAISS index holds them)."""
    from tristage_rag_amd.evaluation import ndcg_at_k
    rng = np.random.default_rng(11)
    n, nq, d = 5183, 300, 384
    docs = make_corpus(n, d, seed=5, dtype="f32")
    rel = rng.integers(0, n, size=nq)
    queries = docs[rel] + 0.9 * make_corpus(nq, d, seed=6, dtype="f32")     # noisy copies of the relevant doc
    queries = (queries / np.linalg.norm(queries, axis=1, keepdims=True)).astype(np.float32)
    qrels = {f"q{i}": {f"d{int(rel[i])}": 1, f"d{int((rel[i] * 7 + 1) % n)}": 1} for i in range(nq)}
    idx = _index(d, "f32", docs)
    D, I = idx.search(queries, 100)                       # cfg1: stage-1 only, top-100
    D0, I0 = oracle.ip_topk(docs, queries, 100)
    run = lambda DD, II: {f"q{i}": {f"d{int(j)}": float(s) for j, s in zip(II[i], DD[i])} for i in range(nq)}
    got, want = ndcg_at_k(qrels, run(D, I), 10), oracle.ndcg_at_k(qrels, run(D0, I0), 10)
    assert 0.3 < want < 1.0
    assert abs(got - want) <= 0.002
    check_topk(D, I, docs, queries, 100)                  # ids identical except float64-inseparable near-ties (the rule, no count)
    idx.close()


@pytest.mark.parametrize("n,d,k,B,dtype", [(32768 + 5, 32, 1, 33, "f16"), (40_001, 100, 7, 64, "bf16"),
                                            (50_000, 1024, 64, 5, "bf16")])
def test_filter_path_edge_shapes(n, d, k, B, dtype):
    corpus = make_corpus(n, d, seed=21, dtype=dtype)
    corpus[123] = 0.0                                      # an all-zero row scores exactly 0
    queries = make_corpus(B, d, seed=22, dtype=dtype)
    queries[0] = -queries[0]
    idx = _index(d, dtype, corpus)
    D, I = idx.search(queries, k)
    assert idx.last_search_info()["path"] == "filter"
    check_topk(D, I, corpus, queries, k)
    idx.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_maxsim_indexed_reads_token_store_in_place(torch_mod, dtype):
    torch = torch_mod
    from tristage_rag_amd.index import maxsim, maxsim_indexed
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16}[dtype]
    rng = np.random.default_rng(4)
    H, Lq = 128, 11
    lens = rng.integers(1, 190, size=300)
    store = oracle.quantize(rng.standard_normal((int(lens.sum()) + 50, H)).astype(np.float32), dtype)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]]) + 17           # not at row 0
    starts = np.minimum(starts, store.shape[0] - lens)
    q = oracle.quantize(rng.standard_normal((Lq, H)).astype(np.float32), dtype)
    pick = rng.permutation(300)[:120]                                    # arbitrary candidate order
    got = maxsim_indexed(torch.from_numpy(q).cuda().to(tdt), torch.from_numpy(store).cuda().to(tdt),
                         torch.from_numpy(starts[pick]).cuda(), torch.from_numpy(lens[pick].astype(np.int32)).cuda())
    docs = [store[starts[i]: starts[i] + lens[i]] for i in pick]
    np.testing.assert_allclose(got.cpu().numpy(), oracle.maxsim_scores(q, docs),
                               atol=2e-6 if dtype == "f32" else 1e-5, rtol=0)


def test_maxsim_more_candidates_than_one_launch_takes(torch_mod):
    """> 4096 candidates: the streaming kernel is launched per chunk of 4096 (single query), and
    the batched entry point falls back to per-query calls; tiny (0..4 token) candidates, so a
    wave's slice crosses several candidates and empty ones sit between them."""
    torch = torch_mod
    from tristage_rag_amd.index import maxsim_indexed, maxsim_indexed_batch
    rng = np.random.default_rng(77)
    H, Lq, n = 64, 21, 9000
    lens = rng.integers(0, 5, size=n)                      # 0..4 tokens: one (partial) tile each
    store = oracle.quantize(rng.standard_normal((int(lens.sum()) + 4, H)).astype(np.float32), "bf16")
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    q = oracle.quantize(rng.standard_normal((Lq, H)).astype(np.float32), "bf16")
    docs = [store[starts[i]: starts[i] + lens[i]] for i in range(n)]
    want = oracle.maxsim_scores(q, docs)
    tq, ts = torch.from_numpy(q).cuda().bfloat16(), torch.from_numpy(store).cuda().bfloat16()
    t_s, t_l = torch.from_numpy(starts).cuda(), torch.from_numpy(lens.astype(np.int32)).cuda()
    got = maxsim_indexed(tq, ts, t_s, t_l).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=1e-5, rtol=0)
    assert (got[lens == 0] == 0.0).all()
    got2 = maxsim_indexed_batch(tq, [0, Lq], ts, t_s, t_l, [0, n]).cpu().numpy()
    np.testing.assert_array_equal(got2, got)
    # two queries x 4000 one-tile candidates in one launch
    got3 = maxsim_indexed_batch(torch.cat([tq, tq]), [0, Lq, 2 * Lq], ts, torch.cat([t_s[:4000], t_s[:4000]]),
                                torch.cat([t_l[:4000], t_l[:4000]]), [0, 4000, 8000]).cpu().numpy()
    np.testing.assert_array_equal(got3[:4000], got[:4000])
    np.testing.assert_array_equal(got3[4000:], got[:4000])
    # 150 queries (more than one launch's 64), 3..20 candidates each
    nq = 150
    lq = rng.integers(1, 40, size=nq)
    nc = rng.integers(3, 21, size=nq)
    qq = oracle.quantize(rng.standard_normal((int(lq.sum()), H)).astype(np.float32), "bf16")
    qo, co = np.concatenate([[0], np.cumsum(lq)]), np.concatenate([[0], np.cumsum(nc)])
    pk = rng.integers(0, n, size=int(nc.sum()))
    got4 = maxsim_indexed_batch(torch.from_numpy(qq).cuda().bfloat16(), qo, ts, t_s[pk], t_l[pk], co).cpu().numpy()
    for j in (0, 63, 64, 65, 127, 128, 149):
        dj = [docs[i] for i in pk[co[j]:co[j + 1]]]
        np.testing.assert_allclose(got4[co[j]:co[j + 1]], oracle.maxsim_scores(qq[qo[j]:qo[j + 1]], dj), atol=1e-5, rtol=0)
    with pytest.raises(ValueError):
        maxsim_indexed_batch(tq, [0, Lq], ts, t_s, t_l, [0, n - 1])            # starts/lens vs offsets
    with pytest.raises(ValueError):
        maxsim_indexed_batch(tq, [1, Lq], ts, t_s, t_l, [0, n])                # offsets must start at 0


@pytest.mark.parametrize("dtype,H,lqs", [("bf16", 768, (5, 32, 17, 1)), ("f16", 384, (70, 9, 33, 150)),
                                         ("bf16", 96, (12, 40)), ("f32", 64, (7, 20))])
def test_maxsim_batch_of_queries_in_one_launch(torch_mod, dtype, H, lqs):
    """ts_maxsim_indexed_batch: ragged queries and candidate lists (incl. an empty list and
    empty candidates) against the oracle, and bit-identical to per-query calls."""
    torch = torch_mod
    from tristage_rag_amd.index import maxsim_indexed, maxsim_indexed_batch
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    rng = np.random.default_rng(len(lqs) * H)
    n_store = 400
    lens = rng.integers(0, 193, size=n_store)
    store = oracle.quantize(rng.standard_normal((int(lens.sum()) + 8, H)).astype(np.float32), dtype)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    ncand = [300, 0, 57, 1][: len(lqs)]
    picks = [rng.permutation(n_store)[:c] for c in ncand]
    qs = [oracle.quantize(rng.standard_normal((L, H)).astype(np.float32), dtype) for L in lqs]
    q_off = np.concatenate([[0], np.cumsum(lqs)]).astype(np.int32)
    c_off = np.concatenate([[0], np.cumsum(ncand)]).astype(np.int32)
    pk = np.concatenate(picks).astype(np.int64)
    tq = torch.from_numpy(np.concatenate(qs, 0)).cuda().to(tdt)
    ts = torch.from_numpy(store).cuda().to(tdt)
    t_starts, t_lens = torch.from_numpy(starts[pk]).cuda(), torch.from_numpy(lens[pk].astype(np.int32)).cuda()
    for mode in ("maxsim", "colbert"):
        got = maxsim_indexed_batch(tq, q_off, ts, t_starts, t_lens, c_off, mode=mode)
        for j, q in enumerate(qs):
            a, b = c_off[j], c_off[j + 1]
            docs = [store[starts[i]: starts[i] + lens[i]] for i in picks[j]]
            if docs:
                np.testing.assert_allclose(got[a:b].cpu().numpy(), oracle.maxsim_scores(q, docs, mode),
                                           atol=2e-6 if dtype == "f32" else 1e-5, rtol=0)
                one = maxsim_indexed(tq[q_off[j]:q_off[j + 1]], ts, t_starts[a:b], t_lens[a:b], mode=mode)
                # max and the per-tile arithmetic do not depend on how tiles are sliced over waves
                assert torch.equal(one, got[a:b])


@pytest.mark.parametrize("k,path", [(2048, "filter"), (3000, "dense")])
def test_largest_k_on_each_path(k, path):
    corpus = make_corpus(120_000, 64, seed=31, dtype="f16")
    queries = make_corpus(3, 64, seed=32, dtype="f16")
    idx = _index(64, "f16", corpus)
    D, I = idx.search(queries, k)
    assert idx.last_search_info()["path"] == path
    check_topk(D, I, corpus, queries, k)
    idx.close()


def test_empty_query_batch_and_bad_arguments(torch_mod):
    torch = torch_mod
    idx = _index(64, "f16", make_corpus(100, 64, dtype="f16"))
    D, I = idx.search(np.zeros((0, 64), np.float32), 5)
    assert D.shape == (0, 5) and I.shape == (0, 5)
    with pytest.raises(ValueError):
        idx.search(np.zeros((2, 63), np.float32), 5)           # wrong dimension
    with pytest.raises(ValueError):
        idx.search(np.zeros((2, 64), np.float32), 0)           # k must be positive
    with pytest.raises(ValueError):
        idx.add(np.zeros((2, 65), np.float32))
    with pytest.raises(ValueError):
        idx.search(torch.zeros((2, 64)), 5, async_=True)       # async needs device tensors
    from tristage_rag_amd._lib import TriStageNativeError
    from tristage_rag_amd.index import FlatIPIndex
    with pytest.raises(TriStageNativeError, match="needs .* bytes of LDS"):
        FlatIPIndex(4096, dtype="f16")                          # beyond the LDS-resident query image
    idx.close()


def test_stage1_index_persistence_roundtrip_on_gpu(tmp_path):
    from tristage_rag_amd.encoders import SentenceEncoder
    from tristage_rag_amd.stage1_retriever import Stage1Config, Stage1Retriever
    enc = SentenceEncoder("random:tiny", device="cuda")
    docs = [f"document number {i} about topic {i % 7}" for i in range(300)]
    cfg = dict(model_name="random:tiny", device="cuda", cache_dir=str(tmp_path / "m"), index_dir=str(tmp_path / "i"),
               use_fp16=False, index_dtype="f16")
    a = Stage1Retriever(Stage1Config(**cfg), model=enc)
    a.add_documents(docs)
    want = a.search("document about topic 3", top_k=10)
    a.save_index()
    b = Stage1Retriever(Stage1Config(**cfg), model=enc)
    b.load_index()
    assert b.faiss_index.ntotal == 300 and b.documents == docs
    got = b.search("document about topic 3", top_k=10)
    assert [(r["doc_id"], r["score"]) for r in got] == [(r["doc_id"], r["score"]) for r in want]


def test_pipelined_async_searches_match_sync(torch_mod):
    """TS_FLAG_PIPELINE: many batches in flight on the internal streams, alternating workspace
    sets, mixed with plain async and synchronous calls — results must equal the synchronous ones."""
    torch = torch_mod
    corpus = make_corpus(150_000, 128, dtype="f16")
    idx = _index(128, "f16", corpus)
    qs = [torch.from_numpy(make_corpus(64 if i % 3 else 17, 128, seed=200 + i, dtype="f16")).cuda().half()
          for i in range(12)]
    torch.cuda.synchronize()
    want = [idx.search(q, 100) for q in qs]
    for rounds in range(3):
        outs = []
        for i, q in enumerate(qs):
            if i == 7:
                outs.append(idx.search(q, 100, async_=True))                 # plain async in between
            else:
                outs.append(idx.search(q, 100, async_=True, inputs_ready=True))
        mid = idx.search(qs[0], 100)                                          # a synchronous call while others are pending
        assert idx.finish() == []
        assert torch.equal(mid[1], want[0][1])
        for (D, I), (D0, I0) in zip(outs, want):
            assert torch.equal(I, I0) and torch.equal(D, D0)
    # outputs written into caller buffers that are immediately reused by the caller's stream
    buf_D = torch.empty((64, 100), dtype=torch.float32, device="cuda")
    buf_I = torch.empty((64, 100), dtype=torch.int64, device="cuda")
    acc = []
    for q in [q for q in qs if q.shape[0] == 64][:4]:
        idx.search(q, 100, async_=True, inputs_ready=True, out=(buf_D, buf_I))
        acc.append(buf_I.clone())            # consumer on the caller's stream, ordered after the search
    idx.finish()
    exp = [w[1] for q, w in zip(qs, want) if q.shape[0] == 64][:4]
    for a, b in zip(acc, exp):
        assert torch.equal(a, b)
    idx.close()


def test_k_equals_n_returns_every_row_in_canonical_order():
    corpus = make_corpus(1000, 96, seed=41, dtype="f16")
    corpus[500] = corpus[3]                                  # one exact tie
    queries = make_corpus(4, 96, seed=42, dtype="f16")
    idx = _index(96, "f16", corpus)
    D, I = idx.search(queries, 1000)                         # k = N (SURVEY.md §8d adversarial set)
    assert all(sorted(r.tolist()) == list(range(1000)) for r in I)
    check_topk(D, I, corpus, queries, 1000)
    idx.close()


@pytest.mark.parametrize("dtype,n,d,k,B,qdt", [("f16", 400_000, 128, 100, 64, "same"), ("bf16", 300_001, 256, 10, 7, "f32"),
                                                ("f16", 1_000_000, 64, 1000, 64, "same"), ("f32", 500_000, 64, 50, 33, "same"),
                                                ("f16", 262_144, 100, 257, 40, "f32"), ("bf16", 2_000_000, 96, 1000, 64, "same")])
def test_one_launch_scan_equals_five_launch_path_and_oracle(torch_mod, dtype, n, d, k, B, qdt):
    """The one-launch search (ts_fused.hip: query image, threshold histogram and scan+filter in ONE kernel) against
    the five-launch filter path (bit-identical results: both are exact) and the oracle; synchronous,
    asynchronous and pipelined submission; queries in the storage dtype and in float32."""
    torch = torch_mod
    corpus = make_corpus(n, d, seed=1234, dtype=dtype)
    corpus[n // 3: n // 3 + 50] = corpus[5]                      # a few exact ties
    queries = make_corpus(B, d, seed=4321, dtype=dtype)
    idx = _index(d, dtype, torch.from_numpy(corpus).cuda())
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[dtype]
    q = torch.from_numpy(queries).cuda()
    q = q.float() if qdt == "f32" else q.to(tdt)
    D, I = idx.search(q, k)
    info = idx.last_search_info()
    assert info["path"] == "filter" and info["one_launch"], info
    assert k <= info["max_candidates"] <= 16384
    Dc, Ic = idx.search(q, k, classic=True)
    assert idx.last_search_info()["path"] == "filter" and not idx.last_search_info()["one_launch"]
    assert torch.equal(I, Ic) and torch.equal(D, Dc)
    check_topk(D.cpu().numpy()[:6], I.cpu().numpy()[:6], corpus, queries[:6], k)
    for ready, force in ((False, False), (True, False), (True, True)):   # async; pipelined (five launches by default); pipelined one-launch
        outs = [idx.search(q, k, async_=True, inputs_ready=ready, one_launch=force) for _ in range(7)]
        assert idx.finish() == []
        assert idx.last_search_info()["one_launch"] == (force or not ready)
        for Da, Ia in outs:
            assert torch.equal(Ia, I) and torch.equal(Da, D)
    Dh, Ih = idx.search(queries, k)                              # host pointers
    assert np.array_equal(Ih, I.cpu().numpy())
    idx.close()


def test_one_launch_scan_falls_back_exactly_on_massive_ties(torch_mod):
    torch = torch_mod
    n, d, k = 600_000, 64, 300
    corpus = make_corpus(n, d, seed=7, dtype="f16")
    corpus[100_000:125_000] = corpus[3]                          # 25 000 exact ties: the candidate list overflows
    queries = make_corpus(16, d, seed=8, dtype="f16")
    queries[2] = corpus[3]
    idx = _index(d, "f16", torch.from_numpy(corpus).cuda().half())
    D, I = idx.search(torch.from_numpy(queries).cuda().half(), k)
    info = idx.last_search_info()
    assert info["path"] == "filter+dense-fallback" and info["one_launch"]
    check_topk(D.cpu().numpy(), I.cpu().numpy(), corpus, queries, k)
    q2 = torch.from_numpy(make_corpus(16, d, seed=9, dtype="f16")).cuda().half()
    D2, I2 = idx.search(q2, k)                                   # the workspace is clean again afterwards
    assert idx.last_search_info()["path"] == "filter" and idx.last_search_info()["one_launch"]
    Dc, Ic = idx.search(q2, k, classic=True)
    assert torch.equal(I2, Ic) and torch.equal(D2, Dc)
    idx.close()


@pytest.mark.parametrize("n,d,k,B,qdt", [(5000, 768, 100, 33, "f32"), (70_001, 600, 50, 70, "f32"), (200_000, 768, 1000, 64, "f32"),
                                          (33, 520, 7, 1, "f32"), (150_000, 704, 10, 5, "f16")])
def test_fp32_storage_split_scan_matches_oracle(torch_mod, n, d, k, B, qdt):
    """TS_F32 storage with 512 < d <= 768 runs the bf16x3 split scan (ts_scan_f32s.hip: every fp32 value as three
    bf16 terms, six 16-bit MFMAs per k step) instead of the exact-f32 MFMA: dense path, filter path (>= 32768 rows),
    more than 32 queries (two passes), reconstruct and ts_index_scores on the same layout — against the float64
    oracle with the same near-tie rule as every other stage-1 test (the dropped terms are < 2e-7 for unit rows)."""
    torch = torch_mod
    corpus = make_corpus(n, d, seed=1234, dtype="f32")
    if n > 100:
        corpus[n // 2: n // 2 + 20] = corpus[3]                   # exact ties
    queries = make_corpus(B, d, seed=4321, dtype="f32")
    idx = _index(d, "f32", torch.from_numpy(corpus).cuda())
    assert np.array_equal(idx.reconstruct_n(0, n), corpus)      # fp32 storage keeps all 32 bits
    q = torch.from_numpy(queries).cuda()
    if qdt == "f16":
        queries = oracle.quantize(queries, "f16")
        q = torch.from_numpy(queries).cuda().half()
    D, I = idx.search(q, k)
    check_topk(D.cpu().numpy(), I.cpu().numpy(), corpus, queries, k)
    if n >= 32768:
        assert idx.last_search_info()["path"] == "filter" and not idx.last_search_info()["one_launch"]
        D2, I2 = idx.search(q, k, exact_dense=True)
        assert torch.equal(I, I2) and torch.equal(D, D2)
    S = idx.scores(q[: min(B, 3)])
    for b in range(min(B, 3)):
        np.testing.assert_allclose(S[b].cpu().numpy(), oracle.scores_f64(corpus, queries[b]), atol=3e-6)
    idx.close()
