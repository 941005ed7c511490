"""Pins the CPU oracle to outputs of the reference's own code
(tests/golden/reference_kat.json, made by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kat.json")))


def test_bm25_matches_reference():
    b = KAT["bm25"]
    idx = oracle.BM25Index()
    idx.fit(b["documents"])
    assert idx.doc_lens == b["doc_lens"]
    assert idx.avg_doc_len == pytest.approx(b["avg_doc_len"], rel=0, abs=0)
    assert {k: idx.idf[k] for k in sorted(idx.idf)} == pytest.approx(b["idf"], rel=1e-15)
    for q, toks, want in zip(b["queries"], b["tokenize"], b["search_top5"]):
        assert idx.tokenize(q) == toks
        got = idx.search(q, 5)
        assert [i for i, _ in got] == [i for i, _ in want]          # stable tie order
        assert [s for _, s in got] == pytest.approx([s for _, s in want], rel=1e-15)


def test_bm25_after_a_second_fit_matches_reference():
    """The reference's fit() appends to the lists of the previous fit (src/stage1_retriever.py:56-80): the oracle keeps
    that, and the reference's own outputs after fit(first) + fit(first + second) pin it."""
    b = KAT["bm25_refit"]
    idx = oracle.BM25Index()
    idx.fit(list(b["first"]))
    idx.fit(list(b["first"]) + list(b["second"]))
    assert idx.doc_lens == b["doc_lens"] and idx.corpus_size == b["corpus_size"]
    assert idx.avg_doc_len == b["avg_doc_len"]
    assert {k: idx.idf[k] for k in sorted(idx.idf)} == pytest.approx(b["idf"], rel=1e-15)
    for q, want in zip(b["queries"], b["search_top6"]):
        got = idx.search(q, 6)
        assert [i for i, _ in got] == [i for i, _ in want]
        assert [s for _, s in got] == pytest.approx([s for _, s in want], rel=1e-15)


def test_survey_known_answers():
    # SURVEY.md §8c, captured independently from the same reference functions
    idx = oracle.BM25Index()
    idx.fit(KAT["bm25"]["documents"])
    got = idx.search("neural networks attention", 5)
    assert [i for i, _ in got] == [3, 4, 0, 1, 2]
    assert got[0][1] == pytest.approx(3.930759, abs=1e-6)
    assert got[1][1] == pytest.approx(1.507427, abs=1e-6)
    rrf = oracle.reciprocal_rank_fusion([(0, .9), (1, .8), (2, .7)], [(2, 3.0), (0, 1.0), (4, .5)])
    assert [i for i, _ in rrf] == [0, 2, 1, 4]
    assert [s for _, s in rrf] == pytest.approx([0.0325224749, 0.0322664585, 0.0161290323, 0.0158730159], abs=1e-9)


def test_fusion_matches_reference():
    for case in KAT["fusion"]:
        dense = [tuple(x) for x in case["dense"]]
        bm25 = [tuple(x) for x in case["bm25"]]
        rrf = oracle.reciprocal_rank_fusion(dense, bm25)
        assert [[i, s] for i, s in rrf] == case["rrf"]
        w = oracle.weighted_fusion(dense, bm25)
        assert [i for i, _ in w] == [i for i, _ in case["weighted"]]
        assert [s for _, s in w] == pytest.approx([s for _, s in case["weighted"]], rel=1e-15)


def test_normalize_matches_reference():
    n = KAT["normalize"]
    x = np.array(n["x"], dtype=np.float32)
    y = oracle.normalize_embeddings(x)
    assert str(y.dtype) == n["y_dtype"]
    np.testing.assert_array_equal(y.astype(np.float64), np.array(n["y"]))
    # and the C restatement agrees to float32 rounding
    yc = np.empty_like(x)
    oracle.lib().oracle_normalize(x.ctypes.data, x.shape[0], x.shape[1], yc.ctypes.data)
    np.testing.assert_allclose(yc, y, atol=1e-7, rtol=0)


@pytest.mark.parametrize("mode", ["maxsim", "colbert"])
def test_maxsim_matches_reference(mode):
    for case in KAT["maxsim"]:
        q = np.array(case["q"], dtype=np.float32)
        d = np.array(case["d"], dtype=np.float32)
        want = case[mode]  # torch fp32 on CPU
        assert oracle.maxsim_numpy(q, d, mode) == pytest.approx(want, abs=2e-6)
        got = oracle.maxsim_scores(q, [d], mode)[0]
        assert float(got) == pytest.approx(want, abs=2e-6)


def test_maxsim_empty_document_scores_zero():
    q = np.ones((3, 8), np.float32)
    out = oracle.maxsim_scores(q, [np.zeros((0, 8), np.float32), np.ones((2, 8), np.float32)])
    assert out[0] == 0.0 and out[1] == pytest.approx(1.0, abs=1e-6)


def test_minmax_and_adaptive_batch_match_reference():
    for case in KAT["minmax"]:
        assert oracle.minmax_normalize(case["in"]) == pytest.approx(case["out"], abs=0)
    for case in KAT["adaptive_batch"]:
        texts = [] if case["words"] is None else [" ".join(["w"] * case["words"])] * 3
        assert oracle.adaptive_batch_size(texts, case["batch_size"]) == case["out"]
    a = np.array([2.0, -1.0, 0.5])
    out = np.empty(3)
    oracle.lib().oracle_minmax(a.ctypes.data, 3, out.ctypes.data)
    assert out.tolist() == [1.0, 0.0, 0.5]


def test_cosine_matches_reference():
    c = KAT["cosine"]
    got = oracle.cosine_similarity(np.array(c["q"]), np.array(c["D"]))
    np.testing.assert_allclose(got, np.array(c["out"]), atol=1e-15, rtol=0)


def test_ip_topk_contract():
    rng = np.random.default_rng(0)
    c = rng.standard_normal((200, 16)).astype(np.float32)
    c[17] = c[3]
    c[150] = c[3]            # exact ties -> ascending id
    q = rng.standard_normal((3, 16)).astype(np.float32)
    D, I = oracle.ip_topk(c, q, 250)
    s = (q.astype(np.float64) @ c.astype(np.float64).T)
    for qi in range(3):
        order = np.lexsort((np.arange(200), -s[qi]))
        assert I[qi, :200].tolist() == order.tolist()
        np.testing.assert_allclose(D[qi, :200], s[qi][order], rtol=1e-6)
        assert (I[qi, 200:] == -1).all() and (D[qi, 200:] < -3e38).all()
        pos = {int(i): p for p, i in enumerate(I[qi, :200])}
        assert pos[3] < pos[17] < pos[150] and pos[150] - pos[3] == 2
    # the timed BLAS baseline computes the same thing
    Db, Ib = oracle.ip_topk_blas(c, q, 20, chunk=64)
    D2, I2 = oracle.ip_topk(c, q, 20)
    assert np.array_equal(Ib, I2)
    np.testing.assert_allclose(Db, D2, atol=1e-5)


def test_quantize_roundtrip():
    x = np.array([1.0, 1.0 + 2 ** -9, 1.0 + 3 * 2 ** -9, -2.5, 65504.0, 1e-8], np.float32)
    b = oracle.quantize(x, "bf16")
    assert b[0] == 1.0 and b[1] == 1.0 and b[2] == np.float32(1.0 + 2 ** -7)  # ties-to-even
    h = oracle.quantize(x, "f16")
    assert h[4] == 65504.0 and h[3] == -2.5
    import torch
    t = torch.tensor(x)
    np.testing.assert_array_equal(b, t.to(torch.bfloat16).float().numpy())
    np.testing.assert_array_equal(h, t.to(torch.float16).float().numpy())


def test_merge_and_ndcg():
    s = np.array([[[0.9, 0.5, 0.1]], [[0.9, 0.8, -1.0]]], np.float32)   # [R=2,B=1,k=3]
    i = np.array([[[4, 7, 9]], [[12, 13, -1]]], np.int64)
    D, I = oracle.merge_topk(s, i, 4)
    assert I.tolist() == [[4, 12, 13, 7]]
    assert oracle.ndcg_at_k({"q": {"a": 1}}, {"q": {"a": 0.9, "b": 0.1}}) == 1.0
    assert oracle.ndcg_at_k({"q": {"a": 1}}, {"q": {"a": 0.1, "b": 0.9}}) == pytest.approx(1 / np.log2(3))
