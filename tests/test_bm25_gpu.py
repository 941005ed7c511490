"""GPU BM25 (ts_bm25_*) vs the host implementation and the reference's own outputs:
float64 scores bit for bit, order = stable descending sort (ties by ascending doc id)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kat.json")))


def test_gpu_bm25_matches_reference_outputs():
    from tristage_rag_amd.stage1_retriever import BM25Index
    b = KAT["bm25"]
    idx = BM25Index(gpu_device=0)
    idx.fit(b["documents"])
    for q, want in zip(b["queries"], b["search_top5"]):
        got = idx.search(q, 5)
        assert [i for i, _ in got] == [i for i, _ in want]
        assert [s for _, s in got] == [s for _, s in want]          # float64, bit for bit
    idx.close()


def test_gpu_bm25_refit_compat_matches_reference_outputs():
    """The reference's statistics after a SECOND fit (appended lists), on the HIP scorer: bit for bit."""
    from tristage_rag_amd.stage1_retriever import BM25Index
    b = KAT["bm25_refit"]
    idx = BM25Index(gpu_device=0, refit_compat=True)
    idx.fit(list(b["first"]))
    idx.fit(list(b["first"]) + list(b["second"]))
    for q, want in zip(b["queries"], b["search_top6"]):
        got = idx.search(q, 6)
        assert [i for i, _ in got] == [i for i, _ in want]
        assert [s for _, s in got] == [s for _, s in want]
    idx.close()


def test_gpu_bm25_equals_host_bm25_on_a_larger_corpus():
    from tristage_rag_amd.stage1_retriever import BM25Index
    rng = np.random.default_rng(5)
    vocab = [f"w{i}" for i in range(400)]
    p = 1.0 / np.arange(1, 401)
    p /= p.sum()                                              # Zipf-like: a few very common terms
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(3, 60)), p=p)) for _ in range(20_000)]
    docs[100] = docs[7]
    docs[5000] = docs[7]                                      # identical documents: exact score ties
    host, gpu = BM25Index(), BM25Index(gpu_device=0)
    host.fit(docs)
    gpu.fit(docs)
    queries = ["w0 w1 w2", "w399", "w7 w7 w250 nosuchword", "zzz", " ".join(docs[7].split()[:6]), "w0"]
    for q in queries:
        for k in (1, 10, 300, 2048):
            a, b = host.search(q, k), gpu.search(q, k)
            assert [i for i, _ in a] == [i for i, _ in b], (q, k)
            assert [s for _, s in a] == [s for _, s in b], (q, k)
    # re-fit after more documents arrive (index is rebuilt and re-uploaded)
    docs2 = docs + ["w0 w1 brandnewterm"] * 3
    host.fit(docs2)
    gpu.fit(docs2)
    assert host.search("brandnewterm w1", 10) == gpu.search("brandnewterm w1", 10)
    gpu.close()


def test_gpu_bm25_batch_equals_single_queries():
    """BM25Index.search_many (ts_bm25_search_batch: one call, one synchronisation, the queries side by side in up to 64
    lanes — one launch per token position) == search() per query, bit for bit, including queries without a known term,
    repeated queries, lanes with long and short posting lists in the same launch, more queries than lanes, and a batch
    after a batch (state left clean)."""
    from tristage_rag_amd.stage1_retriever import BM25Index
    rng = np.random.default_rng(9)
    vocab = [f"w{i}" for i in range(300)]
    p = 1.0 / np.arange(1, 301)
    p /= p.sum()
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(3, 50)), p=p)) for _ in range(30_000)]
    host, gpu = BM25Index(), BM25Index(gpu_device=0)
    host.fit(docs)
    gpu.fit(docs)
    queries = ["w0 w1 w2", "zzz", "w299", "", "w0", "w5 w5 w17 nosuch", "w0 w1 w2"] + \
              [" ".join(rng.choice(vocab, size=int(rng.integers(1, 9)))) for _ in range(80)]   # 87 queries: two chunks of lanes
    for k in (10, 300):
        for _ in range(2):
            many = gpu.search_many(queries, k)
            assert len(many) == len(queries)
            for q, got in zip(queries, many):
                assert got == gpu.search(q, k) == host.search(q, k)
            for (ai, asc), got in zip(gpu.search_many_arrays(queries, k), many):      # the array form: same lists
                assert ai.dtype == np.int64 and asc.dtype == np.float64
                assert list(zip(ai.tolist(), asc.tolist())) == got
    for (ai, asc), q in zip(host.search_many_arrays(queries[:5], 7), queries[:5]):
        assert list(zip(ai.tolist(), asc.tolist())) == host.search(q, 7)
    assert gpu.search_many([], 5) == [] and host.search_many(queries[:3], 5) == [host.search(q, 5) for q in queries[:3]]
    gpu.close()


def test_gpu_bm25_prefilter_and_its_fallbacks():
    """Long touched lists go through the sampled threshold + chip-wide filter before the exact
    select; massive ties overflow the candidate list and must fall back to the exact select over
    everything — results stay bit-identical to the host implementation either way."""
    from tristage_rag_amd.stage1_retriever import BM25Index
    rng = np.random.default_rng(11)
    vocab = [f"w{i}" for i in range(300)]
    p = 1.0 / np.arange(1, 301)
    p /= p.sum()
    docs = [" ".join(rng.choice(vocab, size=int(rng.integers(5, 40)), p=p)) for _ in range(60_000)]
    host, gpu = BM25Index(), BM25Index(gpu_device=0)
    host.fit(docs)
    gpu.fit(docs)
    for q in ("w0 w1 w2 w3", "w0", "w5 w299", "w0 w0 w1 w200 w250"):      # w0 is in most documents
        for k in (1, 10, 300, 2048):
            a, b = host.search(q, k), gpu.search(q, k)
            assert a == b, (q, k)
    gpu.close()
    same = ["alpha beta gamma"] * 30_000 + ["alpha delta"] * 10     # 30 000 exact ties at the top score... and below
    host2, gpu2 = BM25Index(), BM25Index(gpu_device=0)
    host2.fit(same)
    gpu2.fit(same)
    for q in ("alpha", "beta alpha", "delta"):
        for k in (5, 300):
            assert host2.search(q, k) == gpu2.search(q, k), (q, k)
    gpu2.close()


def test_stage1_with_gpu_bm25_equals_host_bm25(tmp_path):
    from tristage_rag_amd.encoders import SentenceEncoder
    from tristage_rag_amd.stage1_retriever import Stage1Config, Stage1Retriever
    enc = SentenceEncoder("random:tiny", device="cuda")
    outs = []
    for on_gpu in (False, True):
        s1 = Stage1Retriever(Stage1Config(model_name="random:tiny", device="cuda", cache_dir=str(tmp_path / "m"),
                                          index_dir=str(tmp_path / "i"), bm25_on_gpu=on_gpu, use_fp16=False),
                             model=enc)
        s1.add_documents(KAT["bm25"]["documents"] * 3)
        outs.append([s1.search(q, top_k=8) for q in KAT["bm25"]["queries"]])
    for a, b in zip(*outs):
        assert [(r["doc_id"], r["score"]) for r in a] == [(r["doc_id"], r["score"]) for r in b]
