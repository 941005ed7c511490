#!/usr/bin/env python3
"""Regenerates tests/golden/reference_kat.json from the reference's OWN code.

Runs only in the build container (needs /root/reference; never on the GPU box).
The reference's hot-path modules hard-import `faiss` and `sentence_transformers`
at top level (src/stage1_retriever.py:6,9); neither is installed here, and none
of the pure functions exercised below touches them, so two EMPTY placeholder
modules are registered under those names for the duration of the import.  No
model is loaded, nothing is downloaded, no FAISS call is made.

What is captured = inputs + the reference's outputs for its pure arithmetic
(SURVEY.md §8c): BM25, RRF / weighted fusion, embedding normalisation, MaxSim /
ColBERT score, min-max, adaptive batch size, cosine similarity, config defaults,
running-mean timing stats, the MTEB adapter's corpus/query heuristic.
"""
import json
import os
import sys
import types
from dataclasses import asdict

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kat.json")


def main() -> None:
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures are committed, nothing to do")
    for name, attrs in (("faiss", ()), ("sentence_transformers", ("SentenceTransformer", "CrossEncoder"))):
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, type(a, (), {}))
        sys.modules[name] = m
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir("/tmp")  # the reference writes log files relative to cwd
    import logging
    from src.stage1_retriever import BM25Index, Stage1Retriever, Stage1Config
    from src.stage2_rescorer import ColBERTScorer, Stage2Config
    from src.stage3_reranker import AdaptiveCrossEncoderReranker, CrossEncoderReranker, Stage3Config
    from src.retrieval_pipeline import PipelineConfig, RetrievalPipeline
    from src.embedding_service import EmbeddingService, EmbeddingConfig
    from benchmark.tristage_mteb_model import TriStageMTEBModel, ModelCard

    kat = {"_generated_by": "tests/golden/make_golden.py", "_reference": "NoliNobdon/TriStage-RAG"}

    # ---- config defaults (field names + values are API surface)
    kat["config_defaults"] = {
        "Stage1Config": asdict(Stage1Config()), "Stage2Config": asdict(Stage2Config()),
        "Stage3Config": asdict(Stage3Config()), "PipelineConfig": asdict(PipelineConfig()),
        "EmbeddingConfig": asdict(EmbeddingConfig()), "ModelCard": asdict(ModelCard()),
    }

    # ---- BM25 (src/stage1_retriever.py:35-112) on the reference's sample documents
    docs = json.load(open(os.path.join(REF, "non_mcp", "test_docs.json")))
    bm = BM25Index()
    bm.fit(docs)
    queries = ["neural networks attention", "machine learning algorithms",
               "information retrieval from large datasets", "Transformers!!! deep-learning", "zzz"]
    kat["bm25"] = {
        "documents": docs, "queries": queries,
        "tokenize": [bm.tokenize(q) for q in queries],
        "idf": {k: bm.idf[k] for k in sorted(bm.idf)},
        "avg_doc_len": bm.avg_doc_len, "doc_lens": bm.doc_lens,
        "search_top5": [[[int(i), float(s)] for i, s in bm.search(q, 5)] for q in queries],
    }

    # ---- BM25 after a SECOND add_documents: the reference re-fits on the whole corpus but never clears doc_freqs /
    # doc_lens (src/stage1_retriever.py:56-80, called at :316-322), so the statistics of the first batch are counted
    # twice and documents added later are scored with the term frequencies of earlier ones.  Captured as it is, for
    # the build's `bm25_refit_compat` switch.
    bm2 = BM25Index()
    first, second = docs[:3], docs[3:] + ["Attention is all you need: transformers for neural machine translation."]
    bm2.fit(list(first))
    bm2.fit(list(first) + list(second))
    kat["bm25_refit"] = {
        "first": first, "second": second, "queries": queries,
        "idf": {k: bm2.idf[k] for k in sorted(bm2.idf)},
        "avg_doc_len": bm2.avg_doc_len, "doc_lens": bm2.doc_lens, "corpus_size": bm2.corpus_size,
        "search_top6": [[[int(i), float(s)] for i, s in bm2.search(q, 6)] for q in queries],
    }

    # ---- fusion (src/stage1_retriever.py:326-366)
    s1 = Stage1Retriever.__new__(Stage1Retriever)
    s1.config = Stage1Config()
    dense = [(0, .9), (1, .8), (2, .7)]
    bm25r = [(2, 3.0), (0, 1.0), (4, .5)]
    rng = np.random.default_rng(7)
    dense2 = [(int(i), float(s)) for i, s in zip(rng.permutation(40)[:25], np.sort(rng.random(25))[::-1])]
    bm252 = [(int(i), float(s)) for i, s in zip(rng.permutation(40)[:20], np.sort(rng.random(20) * 9)[::-1])]
    kat["fusion"] = []
    for d_, b_ in ((dense, bm25r), (dense2, bm252)):
        kat["fusion"].append({
            "dense": [list(x) for x in d_], "bm25": [list(x) for x in b_],
            "rrf": [[int(i), float(s)] for i, s in s1._reciprocal_rank_fusion(d_, b_)],
            "weighted": [[int(i), float(s)] for i, s in s1._weighted_fusion(d_, b_)],
        })

    # ---- normalise (src/stage1_retriever.py:285-288)
    x = rng.standard_normal((6, 24)).astype(np.float32)
    x[3] = 0.0  # zero row: stays zero thanks to the +1e-8
    y = s1._normalize_embeddings(x)
    kat["normalize"] = {"x": x.tolist(), "y": y.astype(np.float64).tolist(), "y_dtype": str(y.dtype)}

    # ---- MaxSim / ColBERT score (src/stage2_rescorer.py:167-201)
    s2 = ColBERTScorer.__new__(ColBERTScorer)
    s2.config = Stage2Config()
    cases = []
    g = torch.Generator().manual_seed(11)
    for (lq, ld, h) in ((5, 9, 16), (1, 1, 8), (32, 47, 24), (7, 192, 16)):
        q = torch.randn(1, lq, h, generator=g)
        d = torch.randn(ld, h, generator=g)
        cases.append({"q": q[0].tolist(), "d": d.tolist(),
                      "maxsim": float(s2._maxsim_score(q, d)), "colbert": float(s2._colbert_score(q, d))})
    kat["maxsim"] = cases
    # pooling helper (src/stage2_rescorer.py:115-132), unused by the pipeline but public
    emb = torch.randn(2, 4, 6, generator=g)
    mask = torch.tensor([[1, 1, 1, 0], [1, 1, 0, 0]])
    pools = {}
    for method in ("cls", "mean", "max"):
        s2.config.pooling_method = method
        pools[method] = s2._pool_embeddings(emb.clone(), mask).tolist()
    kat["pooling"] = {"emb": emb.tolist(), "mask": mask.tolist(), "out": pools}

    # ---- min-max + adaptive batch size (src/stage3_reranker.py:212-228, 328-344)
    s3 = AdaptiveCrossEncoderReranker.__new__(AdaptiveCrossEncoderReranker)
    s3.config = Stage3Config()
    mm_in = [[2.0, -1.0, 0.5], [1.0, 1.0], [], [3.5], [-2.25, 7.0, 7.0, 0.0, -2.25]]
    kat["minmax"] = [{"in": a, "out": [float(v) for v in s3._normalize_scores(list(a))]} for a in mm_in]
    ab = []
    for words in (250, 201, 200, 120, 101, 100, 60, 51, 50, 1):
        for bs in (32, 64, 8):
            s3.config.batch_size = bs
            ab.append({"words": words, "batch_size": bs,
                       "out": int(s3._adaptive_batch_size([" ".join(["w"] * words)] * 3))})
    s3.config.batch_size = 32
    ab.append({"words": None, "batch_size": 32, "out": int(s3._adaptive_batch_size([]))})
    kat["adaptive_batch"] = ab
    kat["prepare_pairs"] = [list(p) for p in CrossEncoderReranker._prepare_input_pairs(s3, "q", ["a", "b"])]

    # ---- cosine similarity (src/embedding_service.py:228-237)
    es = object.__new__(EmbeddingService)
    es.logger = logging.getLogger("kat")
    qv = rng.standard_normal(12)
    dm = rng.standard_normal((5, 12))
    kat["cosine"] = {"q": qv.tolist(), "D": dm.tolist(), "out": es.similarity(qv, dm).tolist()}
    es.config = EmbeddingConfig()
    kat["validate_text"] = [[t if isinstance(t, str) else None, bool(es._validate_text(t))]
                            for t in ("", "a", "x" * 10000, "x" * 10001, 5)]

    # ---- running-mean timing stats (src/retrieval_pipeline.py:567-606)
    rp = RetrievalPipeline.__new__(RetrievalPipeline)
    rp.config = PipelineConfig()
    rp.performance_stats = {"total_queries": 0, "avg_stage1_time": 0.0, "avg_stage2_time": 0.0,
                            "avg_stage3_time": 0.0, "avg_total_time": 0.0, "stage_time_history": []}
    upd = [(0.1, 0.2, 0.3, 0.6), (0.3, 0.1, 0.2, 0.7), (0.05, 0.5, 0.25, 0.9)]
    for u in upd:
        rp._update_performance_stats(*u)
    kat["perf_stats"] = {"updates": [list(u) for u in upd], "stats": rp.performance_stats}

    # ---- MTEB adapter heuristic (benchmark/tristage_mteb_model.py:135-161)
    tm = TriStageMTEBModel.__new__(TriStageMTEBModel)
    hc = []
    for task, kw in (("LIMITSmallRetrieval", {}), ("corpus-encode", {}), ("my_query_task", {}),
                     ("STS12", {}), ("STS12", {"prompt_name": "passage"}), ("STS12", {"prompt_name": "Query"}),
                     ("", {}), ("DocumentRetrieval", {"prompt_name": "query"})):
        hc.append({"task_name": task, "kwargs": kw, "is_corpus": bool(tm._is_corpus_encoding(task, kw))})
    kat["is_corpus_encoding"] = hc

    # ---- chunking of the config-0 caller (non_mcp/embed_and_query.py:31-53)
    from non_mcp.embed_and_query import chunk_text
    base = ("Sentence number %d ends here. " * 1)
    long_text = "".join(base % i for i in range(120)) + "\n" + "tail without period " * 30
    texts = ["", "   ", "short text.", long_text, "x" * 2500, ("para one.\n" * 150)]
    kat["chunk_text"] = [{"text": t, "chunk_size": cs, "overlap": ov, "chunks": chunk_text(t, cs, ov)}
                         for t in texts for cs, ov in ((1000, 200), (300, 50))]

    os.chdir(cwd)
    with open(OUT, "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
