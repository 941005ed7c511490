"""CPU tests of the product's host-side Python (the mirror of the reference API),
pinned to outputs of the reference's own code (tests/golden/reference_kat.json).
GPU entry points are replaced by the doubles in tests/doubles.py."""
import json
import os
from dataclasses import asdict

import numpy as np
import pytest
import torch

from doubles import OracleIndex, oracle_maxsim, oracle_maxsim_indexed, oracle_maxsim_indexed_batch
from oracle import oracle
from tristage_rag_amd.embedding_service import EmbeddingConfig, EmbeddingService
from tristage_rag_amd.encoders import CrossEncoderModel, HashTokenizer, SentenceEncoder
from tristage_rag_amd.retrieval_pipeline import PipelineConfig, RetrievalPipeline
from tristage_rag_amd.stage1_retriever import BM25Index, Stage1Config, Stage1Retriever
from tristage_rag_amd.stage2_rescorer import ColBERTScorer, Stage2Config
from tristage_rag_amd.stage3_reranker import AdaptiveCrossEncoderReranker, CrossEncoderReranker, Stage3Config
from tristage_rag_amd.tristage_mteb_model import ModelCard, TriStageMTEBModel, create_tristage_model

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kat.json")))
DOCS = KAT["bm25"]["documents"]


@pytest.fixture(scope="module")
def encoder():
    return SentenceEncoder("random:tiny", device="cpu")


def _stage1(encoder, tmp_path, **kw):
    cfg = Stage1Config(model_name="random:tiny", device="cpu", cache_dir=str(tmp_path / "m"),
                       index_dir=str(tmp_path / "i"), **kw)
    return Stage1Retriever(cfg, model=encoder, index_factory=lambda d: OracleIndex(d))


# ------------------------------------------------------------------ configs
def test_config_defaults_match_reference():
    want = KAT["config_defaults"]
    for cls, name in ((Stage1Config, "Stage1Config"), (Stage2Config, "Stage2Config"),
                      (Stage3Config, "Stage3Config"), (PipelineConfig, "PipelineConfig"),
                      (EmbeddingConfig, "EmbeddingConfig")):
        got = asdict(cls())
        for k, v in want[name].items():   # every reference field, same default; extra fields are additive
            assert got[k] == v, (name, k)
    mc = asdict(ModelCard())
    for k, v in want["ModelCard"].items():
        if k != "framework":
            assert mc[k] == v


# ------------------------------------------------------------------ BM25 + fusion
def test_bm25_matches_reference():
    b = KAT["bm25"]
    idx = BM25Index()
    idx.fit(DOCS)
    assert idx.doc_lens == b["doc_lens"] and idx.avg_doc_len == b["avg_doc_len"]
    assert {k: idx.idf[k] for k in sorted(idx.idf)} == pytest.approx(b["idf"], rel=1e-15)
    for q, toks, want in zip(b["queries"], b["tokenize"], b["search_top5"]):
        assert idx.tokenize(q) == toks
        got = idx.search(q, 5)
        assert [i for i, _ in got] == [i for i, _ in want]
        assert [s for _, s in got] == pytest.approx([s for _, s in want], rel=1e-15)
        for i, s in want:
            assert idx.score(q, i) == pytest.approx(s, rel=1e-15)
    idx.fit(DOCS + ["attention attention networks"])      # re-fit rebuilds (documented deviation)
    assert idx.corpus_size == 6 and len(idx.doc_freqs) == 6


def test_bm25_refit_compat_matches_reference_after_a_second_add():
    """`bm25_refit_compat` reproduces the reference's statistics after a second add_documents — its fit()
    appends to the previous fit's lists (src/stage1_retriever.py:56-80, re-fit at :316-322): df and the average length
    count the first batch twice and document i is scored with list entry i.  Pinned by the reference's own outputs;
    without the switch the index is rebuilt (= the reference after ONE add of the same corpus)."""
    b = KAT["bm25_refit"]
    first, allv = list(b["first"]), list(b["first"]) + list(b["second"])
    idx = BM25Index(refit_compat=True)
    idx.fit(first)
    idx.fit(allv)
    assert idx.corpus_size == b["corpus_size"] and idx.doc_lens == b["doc_lens"] and idx.avg_doc_len == b["avg_doc_len"]
    assert {k: idx.idf[k] for k in sorted(idx.idf)} == pytest.approx(b["idf"], rel=1e-15)
    for q, want in zip(b["queries"], b["search_top6"]):
        got = idx.search(q, 6)
        assert [i for i, _ in got] == [i for i, _ in want]
        assert [s for _, s in got] == pytest.approx([s for _, s in want], rel=1e-15)
        for i, s_ in want:
            assert idx.score(q, i) == pytest.approx(s_, rel=1e-15)
    plain, once = BM25Index(), BM25Index()
    plain.fit(first)
    plain.fit(allv)
    once.fit(allv)
    assert plain.idf == once.idf and plain.doc_lens == once.doc_lens       # the default re-fit rebuilds
    assert any(plain.search(q, 6) != [tuple(x) for x in w] for q, w in zip(b["queries"], b["search_top6"]))


def test_fusion_matches_reference(encoder, tmp_path):
    s1 = _stage1(encoder, tmp_path)
    for case in KAT["fusion"]:
        dense = [tuple(x) for x in case["dense"]]
        bm25 = [tuple(x) for x in case["bm25"]]
        assert [[i, s] for i, s in s1._reciprocal_rank_fusion(dense, bm25)] == case["rrf"]
        w = s1._weighted_fusion(dense, bm25)
        assert [i for i, _ in w] == [i for i, _ in case["weighted"]]
        assert [s for _, s in w] == pytest.approx([s for _, s in case["weighted"]], rel=1e-15)
    n = KAT["normalize"]
    y = s1._normalize_embeddings(np.array(n["x"], dtype=np.float32))
    assert str(y.dtype) == n["y_dtype"]
    np.testing.assert_array_equal(y.astype(np.float64), np.array(n["y"]))


# ------------------------------------------------------------------ stage 1 flow
def test_stage1_search_schema_and_batching(encoder, tmp_path):
    s1 = _stage1(encoder, tmp_path, enable_bm25=False)
    with pytest.raises(ValueError, match=r"No documents indexed\. Call add_documents\(\) first\."):
        s1.search("x")
    meta = [{"n": i} for i in range(len(DOCS))]
    s1.add_documents(DOCS[:2], meta[:2])
    s1.add_documents(DOCS[2:], meta[2:])          # incremental add appends to the same index
    assert s1.faiss_index.ntotal == 5 and s1.embedding_dim == 64
    res = s1.search("neural networks attention", top_k=3)
    assert len(res) == 3
    assert set(res[0]) == {"doc_id", "document", "score", "stage1_score", "metadata", "stage"}
    assert res[0]["stage"] == "stage1" and res[0]["score"] == res[0]["stage1_score"]
    assert res[0]["document"] == DOCS[res[0]["doc_id"]] and res[0]["metadata"] == meta[res[0]["doc_id"]]
    assert [r["score"] for r in res] == sorted((r["score"] for r in res), reverse=True)
    # scores are cosine similarities of normalised embeddings
    e = s1._normalize_embeddings(s1._encode_batch(DOCS))
    q = s1._normalize_embeddings(s1._encode_batch(["neural networks attention"]))
    assert res[0]["score"] == pytest.approx(float((e @ q[0]).max()), abs=1e-6)
    # k larger than the corpus: the -1 padding is dropped (reference :383)
    assert len(s1.search("x", top_k=50)) == 5
    many = s1.search_many(["neural networks attention", "language"], top_k=3)
    assert [r["doc_id"] for r in many[0]] == [r["doc_id"] for r in res]
    assert many[0][0]["score"] == pytest.approx(res[0]["score"], abs=1e-6)


def test_stage1_bm25_rrf_default_and_persistence(encoder, tmp_path):
    s1 = _stage1(encoder, tmp_path)               # BM25 + RRF on, like the reference default
    s1.add_documents(DOCS)
    res = s1.search("neural networks attention", top_k=5)
    dense = s1.faiss_index.search(s1._normalize_embeddings(s1._encode_batch(["neural networks attention"])), 5)
    dense = [(int(i), float(s)) for i, s in zip(dense[1][0], dense[0][0])]
    want = s1._reciprocal_rank_fusion(dense, s1.bm25_index.search("neural networks attention", 300))[:5]
    assert [(r["doc_id"], r["score"]) for r in res] == want
    st = s1.get_stats()
    assert st["total_documents"] == 5 and st["bm25_enabled"] and st["bm25_vocabulary_size"] > 10
    path = str(tmp_path / "i" / "stage1_index.pkl")
    s1.save_index(path)
    s2 = _stage1(encoder, tmp_path)
    s2.load_index(path)
    assert s2.documents == DOCS and s2.faiss_index.ntotal == 5
    assert [(r["doc_id"], r["score"]) for r in s2.search("neural networks attention", top_k=5)] == want
    with open(str(tmp_path / "evil.pkl"), "wb") as f:
        f.write(b"\x80\x04N.")
    with pytest.raises(ValueError, match="not a tristage-rag_amd index"):
        s2.load_index(str(tmp_path / "evil.pkl"))


# ------------------------------------------------------------------ stage 2
def test_stage2_rescoring_flow():
    cfg = Stage2Config(model_name="random:tiny", device="cpu", top_k_candidates=3, batch_size=2,
                       max_seq_length=16)
    sc = ColBERTScorer(cfg, maxsim_fn=oracle_maxsim)
    cands = [{"doc_id": i, "document": d, "score": 0.1 * i, "stage1_score": 0.1 * i, "metadata": {}, "stage": "stage1"}
             for i, d in enumerate(DOCS + ["", "   "])]
    out = sc.rescore_candidates("neural networks", cands)
    assert len(out) == 3 and all(o["stage"] == "stage2" for o in out)
    assert [o["stage2_score"] for o in out] == sorted((o["stage2_score"] for o in out), reverse=True)
    assert cands[0]["stage"] == "stage1" and "stage2_score" not in cands[0]   # inputs are copied
    # each score equals the reference formula on the model's own token embeddings
    q = sc.encode_query("neural networks")
    assert q.shape[0] == 1 and q.shape[1] == 4                               # [CLS] neural networks [SEP]
    docs = sc.encode_documents_batch([c["document"] for c in cands])
    assert docs[5].shape[0] == 3 and torch.allclose(docs[5], docs[6], atol=1e-5)   # "" and "   " -> "empty"
    long = sc.encode_documents_batch(["word " * 100])[0]
    assert long.shape[0] == 16                                               # truncated to max_seq_length
    from oracle import oracle
    by_id = {o["doc_id"]: o["stage2_score"] for o in sc.rescore_candidates("neural networks", cands[:5])}
    for i in by_id:
        want = oracle.maxsim_numpy(q[0].numpy(), docs[i].numpy())
        assert by_id[i] == pytest.approx(want, abs=1e-5)
    assert sc.rescore_candidates("q", []) == []
    info = sc.get_model_info()
    assert info["embedding_dim"] == 64 and info["scoring_method"] == "maxsim" and info["use_fp16"] is False
    # pooling helper parity with the reference
    p = KAT["pooling"]
    for method, want in p["out"].items():
        sc.config.pooling_method = method
        got = sc._pool_embeddings(torch.tensor(p["emb"]), torch.tensor(p["mask"]))
        np.testing.assert_allclose(got.numpy(), np.array(want), atol=1e-6)


def test_stage2_document_cache_gives_identical_scores():
    cfg = Stage2Config(model_name="random:tiny", device="cpu", top_k_candidates=10, batch_size=5,
                       cache_document_embeddings=True)
    sc = ColBERTScorer(cfg, maxsim_fn=oracle_maxsim)
    cands = [{"doc_id": i, "document": d} for i, d in enumerate(DOCS)]
    a = sc.rescore_candidates("attention", cands)
    assert len(sc._doc_cache) == 5
    b = sc.rescore_candidates("attention", cands)
    assert [(x["doc_id"], x["stage2_score"]) for x in a] == [(x["doc_id"], x["stage2_score"]) for x in b]


# ------------------------------------------------------------------ stage 3
def test_stage3_minmax_adaptive_and_rerank():
    cfg = Stage3Config(model_name="random:tiny", device="cpu", top_k_final=2)
    rr = AdaptiveCrossEncoderReranker(cfg)
    for case in KAT["minmax"]:
        assert rr._normalize_scores(list(case["in"])) == pytest.approx(case["out"], abs=0)
    for case in KAT["adaptive_batch"]:
        rr.config.batch_size = case["batch_size"]
        texts = [] if case["words"] is None else [" ".join(["w"] * case["words"])] * 3
        assert rr._adaptive_batch_size(texts) == case["out"]
    rr.config.batch_size = 32
    assert [list(p) for p in rr._prepare_input_pairs("q", ["a", "b"])] == KAT["prepare_pairs"]
    cands = [{"doc_id": i, "document": d, "stage2_score": 0.5} for i, d in enumerate(DOCS)]
    out = rr.rerank("what are transformers", cands)
    assert len(out) == 2 and out[0]["stage"] == "stage3" and rr.config.batch_size == 32
    assert out[0]["stage3_score"] == 1.0                                      # min-max: best is exactly 1
    scores = rr.predict("what are transformers", DOCS)
    assert min(scores) == 0.0 and max(scores) == 1.0 and len(scores) == 5
    raw = rr.model.predict([["what are transformers", d] for d in DOCS])
    assert int(np.argmax(raw)) == out[0]["doc_id"]
    assert rr.rerank("q", []) == [] and rr.predict("q", []) == []
    with pytest.raises(ValueError):
        rr.batch_rerank(["a"], [[], []])
    # the raw-HF path (tokenizer + seq-cls model) gives the same activation the reference applies
    hf = CrossEncoderReranker(Stage3Config(model_name="random:tiny", device="cpu", normalize_scores=False),
                              model=rr.model.model)
    hf.tokenizer = rr.model.tokenizer
    got = hf.predict("what are transformers", DOCS)
    np.testing.assert_allclose(got, 1 / (1 + np.exp(-rr.model.logits([["what are transformers", d] for d in DOCS])[:, 0].numpy())), atol=1e-6)


# ------------------------------------------------------------------ pipeline
def _pipeline(encoder, tmp_path, **cfg):
    pc = PipelineConfig(stage1_model="random:tiny", stage2_model="random:tiny", stage3_model="random:tiny",
                        device="cpu", cache_dir=str(tmp_path / "m"), index_dir=str(tmp_path / "i"),
                        log_file=str(tmp_path / "p.log"), stage1_top_k=4, stage2_top_k=3, stage3_top_k=2, **cfg)
    p = RetrievalPipeline(config=pc)
    p.stage1 = Stage1Retriever(Stage1Config(model_name="random:tiny", device="cpu", cache_dir=pc.cache_dir,
                                            index_dir=pc.index_dir, top_k_candidates=pc.stage1_top_k,
                                            enable_bm25=pc.stage1_enable_bm25),
                               model=encoder, index_factory=lambda d: OracleIndex(d))
    p.stage2 = ColBERTScorer(Stage2Config(model_name="random:tiny", device="cpu", top_k_candidates=pc.stage2_top_k),
                             maxsim_fn=oracle_maxsim)
    p.stage3 = AdaptiveCrossEncoderReranker(Stage3Config(model_name="random:tiny", device="cpu",
                                                         top_k_final=pc.stage3_top_k))
    return p


def test_pipeline_search_contract(encoder, tmp_path):
    p = _pipeline(encoder, tmp_path, save_intermediate_results=True)
    p.add_documents(DOCS)
    r = p.search("neural networks attention")
    assert set(r) == {"query", "results", "stage1_results", "stage2_results", "timing", "performance_stats"}
    assert len(r["stage1_results"]) == 4 and len(r["stage2_results"]) == 3 and len(r["results"]) == 2
    assert set(r["timing"]) == {"stage1_time", "stage2_time", "stage3_time", "total_time"}
    assert r["performance_stats"]["total_queries"] == 1
    assert all(k in r["results"][0] for k in ("stage1_score", "stage2_score", "stage3_score", "doc_id", "document"))
    assert len(p.search("neural networks attention", top_k=1)["results"]) == 1
    many = p.search_many(["neural networks attention", "language models"])
    seq = p.batch_search(["neural networks attention", "language models"])
    for a, b in zip(many, seq):
        assert [x["doc_id"] for x in a["results"]] == [x["doc_id"] for x in b["results"]]
        for x, y in zip(a["results"], b["results"]):     # batched forwards: padding noise only
            assert x["stage3_score"] == pytest.approx(y["stage3_score"], abs=1e-5)
            assert x["stage2_score"] == pytest.approx(y["stage2_score"], abs=1e-5)
        assert set(a) == set(b) and set(a["timing"]) == set(b["timing"])
    assert p.search_many([]) == []
    info = p.get_pipeline_info()
    assert info["stages_initialized"] == {"stage1": True, "stage2": True, "stage3": True}
    assert info["stage1_stats"]["total_documents"] == 5


def test_pipeline_stats_config_roundtrip(encoder, tmp_path):
    p = _pipeline(encoder, tmp_path)
    ps = KAT["perf_stats"]
    for u in ps["updates"]:
        p._update_performance_stats(*u)
    assert p.performance_stats == pytest.approx(ps["stats"])   # the reference's own running means
    empty = RetrievalPipeline(config=PipelineConfig(log_file=str(tmp_path / "e.log")))
    with pytest.raises(ValueError, match="Pipeline not initialized"):
        empty.save_index()
    y = str(tmp_path / "c.yaml")
    p.export_config(y)
    again = RetrievalPipeline(config_path=y)
    assert asdict(again.config) == asdict(p.config)
    # the reference's nested YAML layout
    open(y, "w").write("pipeline:\n  device: cpu\n  stage1: {model: a, top_k: 7, enable_bm25: false}\n"
                       "  stage2: {max_seq_length: 64}\n  stage3: {top_k: 3}\n  log_file: %s\n" % (tmp_path / "n.log"))
    c = RetrievalPipeline(config_path=y).config
    assert (c.stage1_model, c.stage1_top_k, c.stage1_enable_bm25, c.stage2_max_seq_length, c.stage3_top_k, c.device) == \
           ("a", 7, False, 64, 3, "cpu")
    assert asdict(RetrievalPipeline(config_path=str(tmp_path / "missing.yaml")).config)["stage1_top_k"] == 500


# ------------------------------------------------------------------ MTEB adapter + embedding service
def test_mteb_adapter(encoder, tmp_path):
    p = _pipeline(encoder, tmp_path, stage1_enable_bm25=False)
    m = TriStageMTEBModel(pipeline=p)
    for case in KAT["is_corpus_encoding"]:
        assert m._is_corpus_encoding(case["task_name"], case["kwargs"]) == case["is_corpus"]
    assert m.similarity_fn_name == "cosine" and m.max_seq_length == 512 and m.encode([]).size == 0
    assert m.search("anything") == []                       # nothing indexed: swallowed like the reference
    corpus = {f"d{i}": {"text": d, "title": ""} for i, d in enumerate(DOCS)}
    queries = {"q1": "neural networks attention", "q2": "human language"}
    res = m.search_cross_encoder(corpus, queries, top_k=2)
    assert set(res) == {"q1", "q2"} and all(len(v) == 2 for v in res.values())
    assert all(k.startswith("d") for v in res.values() for k in v)
    emb = m.encode(DOCS, task_name="LIMITSmallRetrieval")
    assert emb.shape == (5, 64) and np.allclose(np.linalg.norm(emb, axis=1), 1.0, atol=1e-5)
    qe = m.encode(["neural networks"], task_name="x", prompt_name="query")
    assert qe.shape == (1, 64)
    recs = m.search("neural networks attention", top_k=2)
    assert set(recs[0]) == {"id", "score", "text", "rank", "stage1_score", "stage2_score", "stage3_score"}
    assert recs[0]["rank"] == 1 and recs[0]["score"] == recs[0]["stage3_score"]
    qs = ["neural networks attention", "human language", "statistics"]
    batched = m.search_batch(qs, top_k=2, chunk=2)           # batched stage 1 == per-query search
    for q, b in zip(qs, batched):
        one = m.search(q, top_k=2)
        assert [r["id"] for r in one] == [r["id"] for r in b]
        assert np.allclose([r["score"] for r in one], [r["score"] for r in b], atol=1e-6)
    pairs = [("neural networks attention", DOCS[3]), ("neural networks attention", DOCS[0]), ("human language", DOCS[1])]
    scores = m.predict(pairs)
    assert len(scores) == 3 and all(isinstance(s, float) for s in scores)
    assert len(m.predict(["human language"], top_k=1)[0]) == 1
    assert isinstance(create_tristage_model(pipeline=p), TriStageMTEBModel)
    from tristage_rag_amd.evaluation import evaluate_retrieval, ndcg_at_k
    assert ndcg_at_k({"q": {"a": 1}}, {"q": {"b": 0.9, "a": 0.1}}) == pytest.approx(1 / np.log2(3))
    m2 = TriStageMTEBModel(pipeline=_pipeline(encoder, tmp_path, stage1_enable_bm25=False))
    out = evaluate_retrieval(m2, corpus, queries, {"q1": {"d3": 1}, "q2": {"d1": 1}}, top_k=2)
    assert 0.0 <= out["ndcg_at_2"] <= 1.0 and out["num_queries"] == 2


def test_embedding_service(encoder, tmp_path):
    EmbeddingService.reset_instance()
    es = EmbeddingService(str(tmp_path / "none.yaml"), model=encoder)
    assert EmbeddingService() is es                          # singleton
    c = KAT["cosine"]
    np.testing.assert_allclose(es.similarity(np.array(c["q"]), np.array(c["D"])), np.array(c["out"]), atol=1e-15)
    for text, ok in KAT["validate_text"]:
        assert es._validate_text(text if text is not None else 5) == ok
    e1 = es.encode_query("hello world")
    assert e1.shape == (64,) and es.encode_query("hello world") is e1    # cached object
    d = es.encode_document(["hello world", "another text"])
    assert d.shape == (2, 64) and np.array_equal(d[0], e1)
    with pytest.raises(ValueError):
        es.encode_document([])
    with pytest.raises(ValueError):
        es.encode_query("")
    es.config.cache_size = 2
    es.encode_query("third")                                # evicts the first-inserted entry
    assert es._get_cached_embedding("hello world") is None and es._get_cached_embedding("third") is not None
    EmbeddingService.reset_instance()


# ------------------------------------------------------------------ encoders
def test_hash_tokenizer_and_encoders():
    tok = HashTokenizer()
    one = tok("Hello, world!", return_tensors="pt", padding=False)
    assert one["input_ids"].shape == (1, 6) and one["input_ids"][0, 0] == 101 and one["input_ids"][0, -1] == 102
    b = tok(["a b c d e f", "a"], truncation=True, padding=True, max_length=5, return_tensors="pt")
    assert b["input_ids"].shape == (2, 5) and b["attention_mask"].tolist() == [[1] * 5, [1, 1, 1, 0, 0]]
    pr = tok(["q q"], ["d d d d d d d d"], truncation=True, padding=True, max_length=8, return_tensors="pt")
    assert pr["input_ids"].shape == (1, 8) and pr["token_type_ids"][0].tolist() == [0, 0, 0, 0, 1, 1, 1, 1]
    enc = SentenceEncoder("random:tiny", device="cpu")
    texts = ["short", "a much longer sentence with many more words in it", "mid length text"]
    e = enc.encode(texts, batch_size=2)
    assert e.shape == (3, 64) and e.dtype == np.float32
    for i, t in enumerate(texts):                            # length-sorted batching does not permute rows
        np.testing.assert_allclose(enc.encode(t), e[i], atol=1e-5)
    n = enc.encode(texts, normalize_embeddings=True)
    np.testing.assert_allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-5)
    ce = CrossEncoderModel("random:tiny", device="cpu")
    s = ce.predict([["q", "d1"], ["q", "a longer document"]], batch_size=1)
    assert s.shape == (2,) and ((s > 0) & (s < 1)).all()


# ------------------------------------------------------------------ config-0 caller
def test_chunk_text_and_three_stage_system(encoder, tmp_path):
    from tristage_rag_amd.three_stage_system import AppConfig, ThreeStageRetrievalSystem, chunk_text
    for case in KAT["chunk_text"]:
        assert chunk_text(case["text"], case["chunk_size"], case["overlap"]) == case["chunks"]
    s1 = _stage1(encoder, tmp_path)
    s2 = ColBERTScorer(Stage2Config(model_name="random:tiny", device="cpu", top_k_candidates=50),
                       maxsim_fn=oracle_maxsim)
    s3 = CrossEncoderReranker(Stage3Config(model_name="random:tiny", device="cpu", top_k_final=20))
    sys_ = ThreeStageRetrievalSystem(AppConfig(device="cpu"), stage1=s1, stage2=s2, stage3=s3)
    assert sys_.search("anything")["error"].startswith("No documents indexed")
    assert sys_.add_documents(DOCS + [DOCS[0], "  "]) == 5 and sys_.add_documents(DOCS) == 0
    out = sys_.search("neural networks attention", top_k=3)
    assert set(out) == {"query", "results", "stage1_time", "stage2_time", "stage3_time", "total_time",
                        "candidate_count", "final_count"}
    assert out["candidate_count"] == 5 and out["final_count"] == 3
    r = out["results"][0]
    assert set(r) == {"rank", "doc_id", "document", "final_score", "stage1_score", "stage2_score", "stage3_score"}
    assert r["rank"] == 1 and r["final_score"] == r["stage3_score"] == 1.0
    assert len(sys_.search_history) == 1


def test_stage2_token_store_matches_reencoding():
    cfg = Stage2Config(model_name="random:tiny", device="cpu", top_k_candidates=10, batch_size=2,
                       precompute_document_embeddings=True)
    sc = ColBERTScorer(cfg, maxsim_fn=oracle_maxsim, maxsim_indexed_fn=oracle_maxsim_indexed,
                       maxsim_indexed_batch_fn=oracle_maxsim_indexed_batch)
    cands = [{"doc_id": 100 + i, "document": d} for i, d in enumerate(DOCS)]
    plain = sc.rescore_candidates("neural attention", cands)          # store empty -> encode path
    sc.index_documents(DOCS[:3], 100)
    sc.index_documents(DOCS[3:], 103)                                  # appended in two calls
    assert len(sc.token_store) == 5 and sc.token_store.rows == sum(sc.token_store.lens)
    stored = sc.rescore_candidates("neural attention", cands[::-1])    # any candidate order
    a = {x["doc_id"]: x["stage2_score"] for x in plain}
    b = {x["doc_id"]: x["stage2_score"] for x in stored}
    assert a.keys() == b.keys()
    for k in a:
        assert a[k] == pytest.approx(b[k], abs=1e-5)                   # batch-padding noise only
    mixed = sc.rescore_candidates("neural attention", cands + [{"doc_id": 999, "document": "not stored"}])
    assert len(mixed) == 6                                             # unknown id -> encode path for all
    # several queries at once: one padded query forward + one (batched) MaxSim call
    qs = ["neural attention", "human language understanding and more words", "statistics"]
    lists = [cands[::-1], [], cands[:2]]
    many = sc.rescore_many(qs, lists)
    assert many[1] == [] and len(many[2]) == 2
    for q, cl, got in zip(qs, lists, many):
        want = sc.rescore_candidates(q, cl) if cl else []
        assert [x["doc_id"] for x in got] == [x["doc_id"] for x in want]
        for x, y in zip(got, want):
            assert x["stage2_score"] == pytest.approx(y["stage2_score"], abs=1e-5) and x["stage"] == "stage2"
    fallback = sc.rescore_many(qs[:1], [cands + [{"doc_id": 999, "document": "not stored"}]])
    assert len(fallback[0]) == 6


def test_stage3_rerank_many_equals_per_query_rerank():
    cfg = Stage3Config(model_name="random:tiny", device="cpu", top_k_final=3, batch_size=2, many_batch_size=4)
    rr = AdaptiveCrossEncoderReranker(cfg)
    cands = [{"doc_id": i, "document": d, "score": 1.0} for i, d in enumerate(DOCS)]
    qs = ["neural networks attention", "language", "x"]
    lists = [cands, [], cands[1:4]]
    many = rr.rerank_many(qs, lists)
    assert many[1] == []
    for q, cl, got in zip(qs, lists, many):
        want = rr.rerank(q, cl) if cl else []
        assert [x["doc_id"] for x in got] == [x["doc_id"] for x in want]
        for x, y in zip(got, want):
            assert x["stage3_score"] == pytest.approx(y["stage3_score"], abs=1e-5) and x["stage"] == "stage3"
    assert rr.config.batch_size == 2
    with pytest.raises(ValueError):
        rr.rerank_many(["a"], [])


def test_pipeline_persistence_restores_the_stage2_token_store(encoder, tmp_path):
    """save_index / load_index (reference src/retrieval_pipeline.py:450-493) also carry the resident
    stage-2 token store: from its safetensors file when it matches, re-encoded otherwise."""
    def build(sub):
        p = _pipeline(encoder, tmp_path / sub, stage1_enable_bm25=False, stage2_precompute_document_embeddings=True)
        p.stage2._maxsim_indexed_fn, p.stage2._maxsim_indexed_batch_fn = oracle_maxsim_indexed, oracle_maxsim_indexed_batch
        p.stage2.config.precompute_document_embeddings = True
        return p
    a = build("a")
    a.add_documents(DOCS)
    want = a.search("neural networks attention")
    path = str(tmp_path / "idx" / "pipeline_index.pkl")
    a.save_index(path)
    assert os.path.exists(str(tmp_path / "idx" / "pipeline_index.stage2_tokens.safetensors"))
    b = build("b")
    b.load_index(path)
    assert len(b.stage2.token_store) == len(DOCS) and b.stage2.token_store.lens == a.stage2.token_store.lens
    assert torch.equal(b.stage2.token_store.data[: b.stage2.token_store.rows].cpu(),
                       a.stage2.token_store.data[: a.stage2.token_store.rows].cpu())
    got = b.search("neural networks attention")
    assert [(r["doc_id"], r["stage2_score"]) for r in got["results"]] == [(r["doc_id"], r["stage2_score"]) for r in want["results"]]
    os.remove(str(tmp_path / "idx" / "pipeline_index.stage2_tokens.safetensors"))
    c = build("c")
    c.load_index(path)                                   # no file: the store is re-encoded
    assert len(c.stage2.token_store) == len(DOCS)
    got = c.search("neural networks attention")
    assert [r["doc_id"] for r in got["results"]] == [r["doc_id"] for r in want["results"]]


# ------------------------------------------------------------------ evaluation entry point (JSONL dir -> nDCG@10)
def _write_task(dirpath, docs, queries, qrels):
    import json as _json
    os.makedirs(dirpath, exist_ok=True)
    with open(os.path.join(dirpath, "corpus.jsonl"), "w") as f:
        for i, d in enumerate(docs):
            f.write(_json.dumps({"_id": f"d{i}", "title": "", "text": d}) + "\n")
    with open(os.path.join(dirpath, "queries.jsonl"), "w") as f:
        for qid, q in queries.items():
            f.write(_json.dumps({"_id": qid, "text": q}) + "\n")
    with open(os.path.join(dirpath, "qrels.jsonl"), "w") as f:
        for qid, rel in qrels.items():
            for did, sc in rel.items():
                f.write(_json.dumps({"query-id": qid, "corpus-id": did, "score": sc}) + "\n")


def test_evaluation_run_task_matches_oracle_ndcg(encoder, tmp_path):
    """JSONL directory (schema of reference benchmark/limit_mteb_tasks.py:129-158) -> TriStageMTEBModel ->
    the MTEB-style result entry; nDCG@10 equals the oracle's on the same run, in both modes."""
    from tristage_rag_amd import evaluation as ev
    queries = {"q1": "neural networks attention", "q2": "human language", "q3": "statistics"}
    qrels = {"q1": {"d3": 1, "d4": 2}, "q2": {"d1": 1}, "q3": {"d0": 1, "d2": 1}}
    _write_task(str(tmp_path / "task"), DOCS, queries, qrels)
    corpus, qs, qr = ev.load_jsonl_dataset(str(tmp_path / "task"))
    assert list(corpus) == [f"d{i}" for i in range(5)] and qs == queries and qr == qrels
    m = TriStageMTEBModel(pipeline=_pipeline(encoder, tmp_path, stage1_enable_bm25=False))
    entry = ev.run_task(m, str(tmp_path / "task"), "ToyRetrieval", mode="rerank", top_k=3)
    run = TriStageMTEBModel(pipeline=_pipeline(encoder, tmp_path, stage1_enable_bm25=False)).search_cross_encoder(
        corpus, qs, top_k=3)
    sc = entry["scores"]["test"][0]
    assert sc["ndcg_at_10"] == pytest.approx(oracle.ndcg_at_k(qrels, run, 10)) and entry["main_score"] == sc["ndcg_at_10"]
    assert sc["ndcg_at_3"] == pytest.approx(oracle.ndcg_at_k(qrels, run, 3))
    assert entry["num_queries"] == 3 and entry["num_documents"] == 5 and entry["task_name"] == "ToyRetrieval"
    # dense mode: encode() for corpus and queries, cosine top-k on the (stand-in) index
    m2 = TriStageMTEBModel(pipeline=_pipeline(encoder, tmp_path, stage1_enable_bm25=False))
    dense = ev.dense_results(m2, corpus, qs, top_k=4, index_factory=lambda d: OracleIndex(d))
    E = encoder.encode(DOCS, normalize_embeddings=True)
    Q = encoder.encode(list(queries.values()), normalize_embeddings=True)
    for r, qid in enumerate(queries):
        want = np.argsort(-(E @ Q[r]), kind="stable")[:4]
        assert list(dense[qid]) == [f"d{i}" for i in want]
    entry2 = ev.run_task(m2, str(tmp_path / "task"), "ToyRetrieval", mode="dense", index_factory=lambda d: OracleIndex(d))
    assert 0.0 <= entry2["main_score"] <= 1.0
    # metric helpers on a hand-made run
    run2 = {"q": {"a": 0.9, "b": 0.8, "c": 0.1}}
    assert ev.recall_at_k({"q": {"b": 1, "z": 1}}, run2, 2) == 0.5
    assert ev.mrr_at_k({"q": {"b": 1}}, run2, 10) == 0.5
    with pytest.raises(ValueError):
        ev.run_task(m2, str(tmp_path / "task"), mode="nope")


# ------------------------------------------------------------------ token-id pair assembly + array-level search_many
def _bert_tokenizer(tmp_path):
    """A REAL transformers tokenizer (WordPiece through the tokenizers crate) over a synthetic vocabulary."""
    from transformers import BertTokenizer
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "alpha", "bravo", "charlie", "delta", "echo", "foxtrot",
             "golf", "hotel"] + [f"w{i}" for i in range(200)]
    vf = tmp_path / "vocab.txt"
    vf.write_text("\n".join(words))
    return BertTokenizer(str(vf))


@pytest.mark.parametrize("kind", ["hash", "bert"])
@pytest.mark.parametrize("max_length", [12, 40, 41, 256])
def test_pair_assembler_reproduces_the_tokenizer(tmp_path, kind, max_length):
    """Cross-encoder inputs assembled from cached token ids == tokenizer(query, doc, truncation=True, padding=True)
    (reference src/stage3_reranker.py:139-160), including the longest-first truncation of BOTH known
    implementations (odd and even budgets, either side longer, ties), for our hashing tokenizer and for a real
    transformers tokenizer; the rule is probed from the tokenizer, not assumed."""
    from tristage_rag_amd.encoders import PairAssembler
    tok = HashTokenizer() if kind == "hash" else _bert_tokenizer(tmp_path)
    pa = PairAssembler(tok, max_length)
    assert pa.ok, pa.why
    rng = np.random.default_rng(max_length)
    vocab = [f"w{i}" for i in range(200)]
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(0, 70)))) for _ in range(120)] + ["", "w1"]
    qs = [" ".join(rng.choice(vocab, size=int(rng.integers(0, 60)))) for _ in range(12)] + ["", "w2 w3"]
    pa.add_documents(docs[:50])
    pa.add_documents(docs[50:])
    assert len(pa) == len(docs)
    pq = torch.from_numpy(rng.integers(0, len(qs), size=600))
    pd = torch.from_numpy(rng.integers(0, len(docs), size=600))
    plan = pa.plan([pa.ids_of(q) for q in qs], pq, pd, "cpu")
    got = pa.batch(plan, torch.arange(600))
    want = tok([qs[i] for i in pq.tolist()], [docs[i] for i in pd.tolist()], truncation=True, padding=True,
               max_length=max_length, return_tensors="pt")
    for k in got:
        if k != "lengths":
            assert torch.equal(got[k], want[k]), k
    assert got["lengths"].dtype == torch.int32 and torch.equal(got["lengths"].long(), want["attention_mask"].sum(1))
    assert int(plan["total"].max()) <= max_length
    sel = torch.tensor([5, 17, 3])                                 # a sub-batch, padded to a given width
    sub = pa.batch(plan, sel, width=int(plan["total"][sel].max()) + 2)
    assert sub["input_ids"].shape[1] == int(plan["total"][sel].max()) + 2
    assert torch.equal(sub["attention_mask"].sum(1), plan["total"][sel])
    assert torch.equal(sub["lengths"].long(), plan["total"][sel])


def test_pair_assembler_refuses_a_tokenizer_it_cannot_restate():
    from tristage_rag_amd.encoders import PairAssembler

    class Odd(HashTokenizer):                                      # truncates only the second text
        def __call__(self, text, text_pair=None, truncation=True, max_length=None, **kw):
            out = super().__call__(text, text_pair, truncation=False, max_length=max_length, **kw)
            if text_pair is not None:
                out["input_ids"] = [r[: max_length] for r in out["input_ids"]]
                out["token_type_ids"] = [r[: max_length] for r in out["token_type_ids"]]
            return out
    pa = PairAssembler(Odd(), 24)
    assert not pa.ok and pa.why


@pytest.mark.parametrize("bm25,fusion", [(False, "rrf"), (True, "rrf"), (True, "weighted")])
def test_search_many_on_arrays_equals_the_record_path(encoder, tmp_path, bm25, fusion):
    """RetrievalPipeline.search_many with every stage on arrays (ids / scores as matrices, token-store lookup,
    token-id pair assembly, device sorts) returns the SAME records as the per-record path: identical ids,
    fields and fused / stage-1 / stage-2 scores; stage-3 scores up to batch-padding noise."""
    from pipeline_pairs import cpu_pipeline, synth_docs
    docs = synth_docs(300, seed=5)
    qs = ["neural networks attention", "vector index search doc42", "token embedding", "gpu memory doc7 doc8"]
    kw = dict(stage1_enable_bm25=bm25, stage1_fusion_method=fusion, stage2_precompute_document_embeddings=True,
              stage1_top_k=60, stage2_top_k=20, stage3_top_k=7)
    a = cpu_pipeline(tmp_path, name="a", stage3_cache_document_tokens=True, **kw)
    b = cpu_pipeline(tmp_path, name="b", **kw)
    a.add_documents(docs[:200])
    a.add_documents(docs[200:])                                    # incremental adds extend the id cache
    b.add_documents(docs[:200])                                    # (same batches on both sides: same padding noise)
    b.add_documents(docs[200:])
    assert a.stage3._pairs_usable and a.stage3._pairs.rule == "second_on_ties"
    took = []
    orig = a._search_many_arrays
    a._search_many_arrays = lambda *x, **k: (took.append(1), orig(*x, **k))[1]
    ra, rb = a.search_many(qs, top_k=5), b.search_many(qs, top_k=5)
    assert took == [1] and all(r["results"] for r in ra)
    for x, y in zip(ra, rb):
        assert set(x) == set(y) and set(x["timing"]) == set(y["timing"])
        for st in ("stage1_results", "stage2_results", "results"):
            assert len(x[st]) == len(y[st])
            for u, v in zip(x[st], y[st]):
                assert set(u) == set(v)
                for k in u:
                    if k == "stage3_score":
                        assert u[k] == pytest.approx(v[k], abs=1e-5)
                    else:
                        assert u[k] == v[k], (st, k)
    # without the intermediate lists only the final records are built
    a.config.save_intermediate_results = False
    lean = a.search_many(qs, top_k=5)
    for x, y in zip(lean, ra):
        assert x["stage1_results"] == [] and x["stage2_results"] == []
        assert [r["doc_id"] for r in x["results"]] == [r["doc_id"] for r in y["results"]]
    # stage-3 batch widths padded to a multiple (fewer GEMM shapes for tune_gemms): masked padding, same ranking
    a.stage3.config.many_width_multiple = 16
    wide = a.search_many(qs, top_k=5)
    a.stage3.config.many_width_multiple = 1
    for x, y in zip(wide, ra):
        assert [r["doc_id"] for r in x["results"]] == [r["doc_id"] for r in y["results"]]
        for u, v in zip(x["results"], y["results"]):
            assert u["stage3_score"] == pytest.approx(v["stage3_score"], abs=1e-5)
    # search() takes the same path for one query; search_on_arrays=False restores the per-record path
    took.clear()
    one = a.search(qs[1], top_k=5)
    assert took == [1] and set(one) == set(ra[1])
    a.config.search_on_arrays = False
    rec = a.search(qs[1], top_k=5)
    assert took == [1]
    assert [r["doc_id"] for r in one["results"]] == [r["doc_id"] for r in rec["results"]] == [r["doc_id"] for r in ra[1]["results"]]
    for u, v in zip(one["results"], rec["results"]):
        assert u["stage3_score"] == pytest.approx(v["stage3_score"], abs=1e-5) and u["stage2_score"] == pytest.approx(v["stage2_score"], abs=1e-5)
    a.config.search_on_arrays = True
    # fewer documents than stage1_top_k: the record path (padded results) takes over
    small = cpu_pipeline(tmp_path, name="s", stage3_cache_document_tokens=True, **kw)
    small.add_documents(docs[:30])
    assert small._search_many_arrays(qs, 5) is None and len(small.search_many(qs, top_k=5)) == len(qs)
    if bm25 and fusion == "weighted":                               # the reference divides by max(bm25) == 0 here
        with pytest.raises(ZeroDivisionError):
            a.search_many(["zzz"], top_k=3)
        with pytest.raises(ZeroDivisionError):
            b.search_many(["zzz"], top_k=3)


@pytest.mark.parametrize("spec", ["random:tiny", "random:minilm", "random:xlmr-large:64:2:4"])
def test_lean_classifier_forward_equals_the_transformers_module(spec):
    """encoders.LeanBertClassifier (fused QKV, no autocast cache, no mask for unpadded batches) is the same
    arithmetic as the transformers module: identical logits in fp32, and in bf16 identical to the module under
    torch.autocast (BERT and the RoBERTa family with its padding-aware position ids)."""
    from tristage_rag_amd.encoders import LeanBertClassifier
    m = CrossEncoderModel(spec, device="cpu", use_amp=False)
    g = torch.Generator().manual_seed(0)
    B, L = 9, 31
    ids = torch.randint(1000, 20000, (B, L), generator=g)
    lens = torch.randint(4, L + 1, (B,), generator=g)
    lens[0] = L
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    ids = ids * mask
    types = ((torch.arange(L)[None, :] >= 5) & mask.bool()).long()
    for full in (False, True):                          # with padding / a batch without any (no mask tensor at all)
        mm = torch.ones_like(mask) if full else mask
        with torch.no_grad():
            ref = m.model(input_ids=ids, attention_mask=mm, token_type_ids=types).logits.float()
            with torch.autocast("cpu", dtype=torch.bfloat16):
                ref16 = m.model(input_ids=ids, attention_mask=mm, token_type_ids=types).logits.float()
        assert torch.allclose(LeanBertClassifier(m.model, None)(ids, mm, types), ref, atol=1e-6)
        assert torch.allclose(LeanBertClassifier(m.model, torch.bfloat16)(ids, mm, types), ref16, atol=2e-3)
    # CrossEncoderModel routes assembled id tensors through it; the switch restores the module's forward
    enc = {"input_ids": ids, "attention_mask": mask, "token_type_ids": types}
    a = m.logits_from_ids(enc)
    m.lean_forward = False
    assert torch.allclose(a, m.logits_from_ids(enc), atol=1e-6)


@pytest.mark.parametrize("spec", ["random:tiny", "random:xlmr-large:64:2:4"])
def test_lean_encoder_forward_equals_the_transformers_module(spec):
    """encoders.LeanBertEncoder on a bare AutoModel: the last hidden state of the transformers module (fp32 equal, bf16
    against the module under autocast), and SentenceEncoder.encode / ColBERTScorer._forward only route through it under
    16-bit autocast on a GPU (here, on the CPU, both keep the module's forward)."""
    from tristage_rag_amd.encoders import LeanBertEncoder, SentenceEncoder, lean_encoder_for, load_backbone
    tok, model, _ = load_backbone(spec, "/tmp/ts_models", "base")
    model.eval()
    g = torch.Generator().manual_seed(1)
    B, L = 7, 23
    ids = torch.randint(1000, 20000, (B, L), generator=g)
    lens = torch.randint(3, L + 1, (B,), generator=g)
    lens[0] = L
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    ids = ids * mask + (1 - mask) * int(getattr(model.config, "pad_token_id", 0) or 0)
    with torch.no_grad():
        ref = model(input_ids=ids, attention_mask=mask).last_hidden_state
        with torch.autocast("cpu", dtype=torch.bfloat16):
            ref16 = model(input_ids=ids, attention_mask=mask).last_hidden_state.float()
    valid = mask.bool()
    got = LeanBertEncoder(model, None)(ids, mask)
    assert got.dtype == torch.float32 and torch.allclose(got[valid], ref[valid], atol=2e-6)
    got16 = LeanBertEncoder(model, torch.bfloat16)(ids, mask)
    assert got16.dtype == torch.float32 and float((got16[valid] - ref16[valid]).abs().max()) < 0.05 * float(ref16.abs().max())
    assert lean_encoder_for(model, torch.bfloat16) is lean_encoder_for(model, torch.bfloat16)      # built once per dtype
    assert lean_encoder_for(torch.nn.Linear(2, 2), torch.bfloat16) is False                         # no such architecture
    with pytest.raises(ValueError):                                                                 # beyond the position table
        LeanBertEncoder(model, None)(torch.zeros((1, 600), dtype=torch.long), None)
    enc = SentenceEncoder(spec, device="cpu")
    with torch.autocast("cpu", dtype=torch.bfloat16):
        enc.encode(["a b c", "d"])
    assert "_ts_lean_encoders" not in enc.model.__dict__                                            # CPU: module forward


def test_lean_modernbert_forward_equals_the_transformers_module():
    """encoders.LeanModernBertEncoder (pre-LN blocks, rotary embedding, local window |q-k| <= 64 on the sliding
    layers, gated GELU) against transformers' ModernBertModel: fp32 equal on the valid tokens of a padded batch longer
    than the window, bf16 against the module under autocast; lean_encoder_for picks it by model type."""
    from tristage_rag_amd.encoders import LeanModernBertEncoder, lean_encoder_for, load_backbone
    tok, model, _ = load_backbone("random:modernbert:64:4:2", "/tmp/ts_models", "base")
    model.eval()
    assert {l.attention_type for l in model.layers} == {"full_attention", "sliding_attention"}
    g = torch.Generator().manual_seed(2)
    B, L = 5, 150
    ids = torch.randint(1000, 20000, (B, L), generator=g)
    lens = torch.tensor([L, 1, 64, 66, 131])
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    ids = ids * mask
    with torch.no_grad():
        ref = model(input_ids=ids, attention_mask=mask).last_hidden_state
        with torch.autocast("cpu", dtype=torch.bfloat16):
            ref16 = model(input_ids=ids, attention_mask=mask).last_hidden_state.float()
    valid = mask.bool()
    lean = LeanModernBertEncoder(model, None)
    assert lean.window == 64
    got = lean(ids, mask)
    assert got.dtype == torch.float32 and torch.allclose(got[valid], ref[valid], atol=5e-6)
    got16 = LeanModernBertEncoder(model, torch.bfloat16)(ids, mask)
    assert float((got16[valid] - ref16[valid]).abs().max()) < 0.05 * float(ref16[valid].abs().max())
    # an unpadded batch without a mask tensor
    full = LeanModernBertEncoder(model, None)(ids[:1], None)
    assert torch.allclose(full, ref[:1], atol=5e-6)
    assert isinstance(lean_encoder_for(model, None), LeanModernBertEncoder)
