"""N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise the row-shard
planning, id offsets, the single packed all-gather and the merge of
tristage_rag_amd.sharded.ShardedFlatIPIndex.  The per-rank scan and the merge
kernel are GPU code, so here they are replaced by oracle-backed doubles; what is
under test is the distributed plumbing, which must reproduce the unsharded top-k
exactly, for every world size, including ties across shard boundaries."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tristage_rag_amd.sharded import ShardedFlatIPIndex, shard_bounds


def test_shard_bounds_cover_rows_once():
    for n in (0, 1, 7, 8, 9, 1000, 10_000_000):
        for r in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, r, i) for i in range(r)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d
    assert shard_bounds(10_000_000, 8, 3) == (3_750_000, 5_000_000)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d, k, B, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from doubles import OracleIndex, oracle_merge
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(42)                     # same data on every rank
        base = rng.standard_normal((n // 4 + 1, d)).astype(np.float32)
        corpus = base[rng.integers(0, base.shape[0], size=n)]   # duplicates -> exact ties across shards
        queries = rng.standard_normal((B, d)).astype(np.float32)
        idx = ShardedFlatIPIndex(d, n, local_index=OracleIndex(d), merge_fn=oracle_merge)
        assert (idx.lo, idx.hi) == shard_bounds(n, world, rank)
        idx.add_global(corpus)
        assert idx.local_index.ntotal == idx.hi - idx.lo and idx.ntotal == n
        D, I = idx.search(torch.from_numpy(queries), k)
        np.save(os.path.join(out_dir, f"D{rank}.npy"), D.numpy())
        np.save(os.path.join(out_dir, f"I{rank}.npy"), I.numpy())
        if rank == 0:
            np.save(os.path.join(out_dir, "corpus.npy"), corpus)
            np.save(os.path.join(out_dir, "queries.npy"), queries)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,k,B", [(2, 501, 40, 6), (3, 100, 60, 6), (2, 5, 8, 6),
                                          (2, 77, 25, 1), (3, 40, 7, 3)])     # odd B*k: ids start at an 8-byte pad
def test_sharded_search_equals_unsharded(tmp_path, world, n, k, B):
    from oracle import oracle
    d = 24
    mp.spawn(_worker, args=(world, _free_port(), n, d, k, B, str(tmp_path)), nprocs=world, join=True)
    corpus, queries = np.load(tmp_path / "corpus.npy"), np.load(tmp_path / "queries.npy")
    D0, I0 = oracle.ip_topk(corpus, queries, k)
    for r in range(world):                                    # identical on every rank, equal to unsharded
        assert np.array_equal(np.load(tmp_path / f"I{r}.npy"), I0)
        np.testing.assert_array_equal(np.load(tmp_path / f"D{r}.npy"), D0)


def _pipeline_worker(rank, world, port, out_dir):
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from doubles import OracleIndex, oracle_maxsim, oracle_merge
    from tristage_rag_amd.encoders import SentenceEncoder
    from tristage_rag_amd.parallel_pipeline import ShardedRetrievalPipeline
    from tristage_rag_amd.retrieval_pipeline import PipelineConfig, RetrievalPipeline
    from tristage_rag_amd.stage1_retriever import Stage1Config, Stage1Retriever
    from tristage_rag_amd.stage2_rescorer import ColBERTScorer, Stage2Config
    from tristage_rag_amd.stage3_reranker import AdaptiveCrossEncoderReranker, Stage3Config
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(1)
        words = "alpha beta gamma delta neural network retrieval index query vector gpu memory token".split()
        docs = [" ".join(rng.choice(words, size=int(rng.integers(3, 14)))) + f" d{i}" for i in range(57)]

        def build(cls, **kw):
            pc = PipelineConfig(stage1_model="random:tiny", stage2_model="random:tiny", stage3_model="random:tiny",
                                device="cpu", cache_dir=os.path.join(out_dir, "m"), index_dir=os.path.join(out_dir, "i"),
                                log_file=os.path.join(out_dir, f"r{rank}.log"), log_level="ERROR", stage1_top_k=20,
                                stage2_top_k=9, stage3_top_k=4, save_intermediate_results=True)
            p = cls(config=pc, **kw)
            p.stage1 = Stage1Retriever(Stage1Config(model_name="random:tiny", device="cpu", cache_dir=pc.cache_dir,
                                                    index_dir=pc.index_dir, top_k_candidates=20),
                                       model=SentenceEncoder("random:tiny", device="cpu"),
                                       index_factory=lambda d: OracleIndex(d))
            p.stage2 = ColBERTScorer(Stage2Config(model_name="random:tiny", device="cpu", top_k_candidates=9),
                                     maxsim_fn=oracle_maxsim)
            p.stage3 = AdaptiveCrossEncoderReranker(Stage3Config(model_name="random:tiny", device="cpu", top_k_final=4))
            return p

        par = build(ShardedRetrievalPipeline)
        par._merge_fn = oracle_merge
        par.add_documents(docs)
        assert par.stage1.faiss_index.local_index.ntotal == shard_bounds(57, world, rank)[1] - shard_bounds(57, world, rank)[0]
        single = build(RetrievalPipeline)
        single.add_documents(docs)
        res = []
        for q in ("neural network retrieval", "gpu memory d7", "zeta"):
            a, b = par.search(q), single.search(q)
            for stage, key in (("stage1_results", "stage1_score"), ("stage2_results", "stage2_score"), ("results", "stage3_score")):
                assert [r["doc_id"] for r in a[stage]] == [r["doc_id"] for r in b[stage]], (q, stage)
                np.testing.assert_allclose([r[key] for r in a[stage]], [r[key] for r in b[stage]], atol=2e-5)
            res.append([r["doc_id"] for r in a["results"]])
        # what a rank holds: its rows' text only; the records that come back are complete on every rank
        lo, hi = shard_bounds(57, world, rank)
        assert len(par.stage1.documents) == 57 and par.stage1.documents.items == docs[lo:hi]
        for stage in ("stage1_results", "stage2_results", "results"):
            assert all(r["document"] == docs[r["doc_id"]] for r in a[stage]), stage
        # batched: stage 1 collective, then every rank scores the candidates / pairs whose documents it owns
        qs = ["neural network retrieval", "gpu memory d7", "zeta", "alpha beta index", "token d3"]
        many, ref = par.search_many(qs), single.search_many(qs)
        assert par.search_many([]) == []
        for a, b in zip(many, ref):
            for stage, key in (("stage1_results", "stage1_score"), ("stage2_results", "stage2_score"), ("results", "stage3_score")):
                assert [r["doc_id"] for r in a[stage]] == [r["doc_id"] for r in b[stage]], (a["query"], stage)
                np.testing.assert_allclose([r[key] for r in a[stage]], [r[key] for r in b[stage]], atol=2e-5)
                assert [r["document"] for r in a[stage]] == [r["document"] for r in b[stage]]
            res.append([r["doc_id"] for r in a["results"]])
        again = par.search("gpu memory d7")
        assert [r["doc_id"] for r in again["results"]] == res[1]
        # the array path over the ranks: token store + cached stage-3 token ids ROW-SHARDED with the stage-1 rows
        from doubles import oracle_maxsim_indexed, oracle_maxsim_indexed_batch

        def build_arrays(cls, **kw):
            p = build(cls)
            p.config.stage2_precompute_document_embeddings = True
            p.config.stage3_cache_document_tokens = True
            for k, v in kw.items():
                setattr(p.config, k, v)
            p.stage2 = ColBERTScorer(Stage2Config(model_name="random:tiny", device="cpu", top_k_candidates=9,
                                                  precompute_document_embeddings=True),
                                     maxsim_fn=oracle_maxsim, maxsim_indexed_fn=oracle_maxsim_indexed,
                                     maxsim_indexed_batch_fn=oracle_maxsim_indexed_batch)
            return p
        par2, single2 = build_arrays(ShardedRetrievalPipeline), build_arrays(RetrievalPipeline)
        par2._merge_fn = oracle_merge
        par2.add_documents(docs)
        single2.add_documents(docs)
        # rank r holds token matrices / pair-token ids of ITS rows only
        assert par2.stage3._pairs_usable and len(par2.stage2.token_store) == hi - lo == len(par2.stage3._pairs)
        assert sorted(par2.stage2._store_slot) == list(range(lo, hi)) and par2.stage3._pairs_base == lo
        info = par2.get_pipeline_info()["sharding"]
        assert info["rows"] == [lo, hi] and info["documents_held"] == hi - lo and info["documents_total"] == 57
        assert info["stage2_token_rows"] == par2.stage2.token_store.rows < single2.stage2.token_store.rows
        assert info["stage3_id_cache_documents"] == hi - lo
        took = []
        orig = par2._search_many_arrays
        par2._search_many_arrays = lambda *a, **k: (lambda r: (took.append(r is not None), r)[1])(orig(*a, **k))
        many2, ref2 = par2.search_many(qs), single2.search_many(qs)
        assert took == [True]
        for a, b in zip(many2, ref2):
            for stage, key in (("stage1_results", "stage1_score"), ("stage2_results", "stage2_score"), ("results", "stage3_score")):
                assert [r["doc_id"] for r in a[stage]] == [r["doc_id"] for r in b[stage]], (a["query"], stage)
                np.testing.assert_allclose([r[key] for r in a[stage]], [r[key] for r in b[stage]], atol=2e-5)
                assert [r["document"] for r in a[stage]] == [r["document"] for r in b[stage]]
            res.append([r["doc_id"] for r in a["results"]])
        one = par2.search(qs[1])                                   # search() = the array path of one query
        assert took == [True, True] and [r["doc_id"] for r in one["results"]] == [r["doc_id"] for r in ref2[1]["results"]]
        # a query whose candidates ALL live on one rank: dense stage 1 with top_k no larger than the smallest shard and
        # a query that is (a copy of) a document of rank 0's rows -> check through the merged ids
        par3 = build_arrays(ShardedRetrievalPipeline, stage1_enable_bm25=False, stage1_top_k=20)
        single3 = build_arrays(RetrievalPipeline, stage1_enable_bm25=False, stage1_top_k=20)
        for p3 in (par3, single3):
            p3.stage1.config.enable_bm25 = False
        par3._merge_fn = oracle_merge
        # 40 near-copies of one text at the front (rank 0's rows when world == 2), unrelated text behind them
        skew = [f"omega omega omega sigma v{i}" for i in range(24)] + [f"lambda kappa d{i}" for i in range(40)]
        par3.add_documents(skew)
        single3.add_documents(skew)
        a3, b3 = par3.search_many(["omega omega omega sigma"])[0], single3.search_many(["omega omega omega sigma"])[0]
        lo3, hi3 = shard_bounds(len(skew), world, 0)
        assert all(lo3 <= r["doc_id"] < hi3 for r in a3["stage2_results"]), "the kept candidates were meant to sit on rank 0"
        for stage, key in (("stage1_results", "stage1_score"), ("stage2_results", "stage2_score"), ("results", "stage3_score")):
            assert [r["doc_id"] for r in a3[stage]] == [r["doc_id"] for r in b3[stage]], stage
            np.testing.assert_allclose([r[key] for r in a3[stage]], [r[key] for r in b3[stage]], atol=2e-5)
        res.append([r["doc_id"] for r in a3["results"]])
        # each rank hands over ONLY its rows (the ingestion call for corpora no host should hold): same results
        par4 = build_arrays(ShardedRetrievalPipeline, stage1_enable_bm25=False)
        par4.stage1.config.enable_bm25 = False
        par4._merge_fn = oracle_merge
        lo4, hi4 = shard_bounds(len(skew), world, rank)
        par4.add_documents_shard(skew[lo4:hi4], len(skew))
        a4 = par4.search_many(["omega omega omega sigma"])[0]
        assert [r["doc_id"] for r in a4["results"]] == [r["doc_id"] for r in a3["results"]]
        assert [r["document"] for r in a4["results"]] == [skew[r["doc_id"]] for r in a4["results"]]
        # ... also with the reference's default BM25 + RRF: the lexical index is sharded too (local postings, corpus-wide
        # statistics): BM25 lists bit-identical to one index over the whole corpus, and the pipeline's records equal
        from tristage_rag_amd.stage1_retriever import BM25Index
        par5 = build_arrays(ShardedRetrievalPipeline)
        par5._merge_fn = oracle_merge
        par5.add_documents_shard(docs[lo:hi], len(docs))
        whole = BM25Index()
        whole.fit(docs)
        for q in qs + ["d7 d30 gpu", "zzzz"]:
            assert par5.stage1.bm25_index.search(q, 12) == whole.search(q, 12), q        # float64 scores, bit for bit
        many5 = par5.search_many(qs)
        for a, b in zip(many5, ref2):
            for stage, key in (("stage1_results", "stage1_score"), ("stage2_results", "stage2_score"), ("results", "stage3_score")):
                assert [r["doc_id"] for r in a[stage]] == [r["doc_id"] for r in b[stage]], (a["query"], stage)
                np.testing.assert_allclose([r[key] for r in a[stage]], [r[key] for r in b[stage]], atol=2e-5)
        # per-rank persistence: every rank writes its shard (rows, token matrices, text), a fresh pipeline of the same
        # world size loads it back without re-encoding the corpus rows
        path = os.path.join(out_dir, "saved", "pipeline_index.pkl")
        par5.save_index(path)
        assert os.path.exists(os.path.join(out_dir, "saved", f"pipeline_index.shard{rank}of{world}.json"))
        par6 = build_arrays(ShardedRetrievalPipeline)
        par6._merge_fn = oracle_merge
        encodes = []
        orig_enc = par6.stage1._encode_batch
        par6.stage1._encode_batch = lambda texts: (encodes.append(len(texts)), orig_enc(texts))[1]
        par6.load_index(path)
        assert sum(encodes) == 0                      # no document went through the bi-encoder again
        assert len(par6.stage2.token_store) == hi - lo and par6.get_pipeline_info()["sharding"]["rows"] == [lo, hi]
        many6 = par6.search_many(qs)
        for a, b in zip(many6, many5):
            for stage, key in (("stage1_results", "stage1_score"), ("stage2_results", "stage2_score"), ("results", "stage3_score")):
                assert [r["doc_id"] for r in a[stage]] == [r["doc_id"] for r in b[stage]], (a["query"], stage)
                np.testing.assert_allclose([r[key] for r in a[stage]], [r[key] for r in b[stage]], atol=2e-5)
                assert [r["document"] for r in a[stage]] == [r["document"] for r in b[stage]]
        res.append([r["doc_id"] for r in many6[0]["results"]])
        json.dump(res, open(os.path.join(out_dir, f"res{rank}.json"), "w"))
    finally:
        dist.destroy_process_group()


def test_sharded_pipeline_equals_single_process_pipeline(tmp_path):
    """All three stages over 2 ranks — stage-1 rows, stage-2 token store, stage-3 token-id cache and document text all
    ROW-SHARDED (rank r holds its rows only), BM25+RRF replicated — == the single-process pipeline (ids exact, scores
    2e-5), identical on every rank; per-record path and array path, a query whose candidates all live on one rank, and
    the shard-only ingestion call."""
    import json
    mp.spawn(_pipeline_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert json.load(open(tmp_path / "res0.json")) == json.load(open(tmp_path / "res1.json"))


class _CountingAsyncIndex:
    """OracleIndex with the asynchronous protocol of FlatIPIndex as far as the wrapper sees it: room for 60 passes
    of <= 32 queries between finish() calls, counted only for searches that actually ran on this rank."""

    PENDING_PASSES = 60

    def __init__(self, d):
        from doubles import OracleIndex
        self._o = OracleIndex(d)
        self.auto_finish = True
        self.passes = 0
        self.finishes = 0

    def __getattr__(self, name):
        return getattr(self._o, name)

    def search(self, q, k, **kw):
        self.passes += (q.shape[0] + 31) // 32
        assert self.passes <= 60, "the wrapper let the local index run out of tickets"
        return self._o.search(q, k)

    def pending_room(self, n):
        return 60 - self.passes - (int(n) + 31) // 32

    def finish(self):
        self.passes = 0
        self.finishes += 1
        return []


def _empty_shard_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from doubles import oracle_merge
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        n, d, k, B = 2, 8, 2, 64                      # 2 rows on 3 ranks: ranks 1 and 2 hold NOTHING ... (ceil division: 1 row each on 0, 1)
        rng = np.random.default_rng(3)
        corpus = rng.standard_normal((n, d)).astype(np.float32)
        idx = ShardedFlatIPIndex(d, n, local_index=_CountingAsyncIndex(d), merge_fn=oracle_merge)
        idx.add_global(corpus)
        q = torch.from_numpy(rng.standard_normal((B, d)).astype(np.float32))
        D0, I0 = idx.search(q, k)
        idx.local_index.passes = 0                     # (a synchronous search holds no ticket in the real index)
        calls = []
        real_finish = idx.finish
        idx.finish = lambda: (calls.append(len(idx._pending)), real_finish())[1]
        outs = [idx.search(q, k, async_=True) for _ in range(70)]      # 2 passes each: > 60 passes twice
        idx.finish()
        for D, I in outs:
            assert torch.equal(I, I0) and torch.equal(D, D0)
        np.save(os.path.join(out_dir, f"calls{rank}.npy"), np.asarray(calls))
        np.save(os.path.join(out_dir, f"empty{rank}.npy"), np.asarray([idx.hi - idx.lo]))
    finally:
        dist.destroy_process_group()


def test_collective_finish_cadence_is_the_same_on_ranks_with_empty_shards(tmp_path):
    """the wrapper's finish() is collective (an all_reduce + re-exchange) but was triggered by the LOCAL
    index running out of tickets — which never happens on a rank whose shard is empty (fewer rows than ranks), so
    after ~30 asynchronous 64-query batches the non-empty ranks sat in the all_reduce and the empty ones in the
    next all_gather.  The cadence now follows the wrapper's own count: identical finish points on every rank."""
    world = 3
    mp.spawn(_empty_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    calls = [np.load(tmp_path / f"calls{r}.npy").tolist() for r in range(world)]
    sizes = [int(np.load(tmp_path / f"empty{r}.npy")[0]) for r in range(world)]
    assert 0 in sizes and max(sizes) > 0                       # at least one empty and one non-empty shard
    assert calls[0] == calls[1] == calls[2] and len(calls[0]) >= 3   # two forced finishes + the final one


def _tiny_pipeline_worker(rank, world, port, out_dir):
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import datetime
    from doubles import (OracleIndex, oracle_maxsim, oracle_maxsim_indexed, oracle_maxsim_indexed_batch, oracle_merge)
    from tristage_rag_amd.encoders import SentenceEncoder
    from tristage_rag_amd.parallel_pipeline import ShardedRetrievalPipeline
    from tristage_rag_amd.retrieval_pipeline import PipelineConfig, RetrievalPipeline
    from tristage_rag_amd.stage1_retriever import Stage1Config, Stage1Retriever
    from tristage_rag_amd.stage2_rescorer import ColBERTScorer, Stage2Config
    from tristage_rag_amd.stage3_reranker import AdaptiveCrossEncoderReranker, Stage3Config
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        docs = ["alpha beta gamma d0", "neural network retrieval d1", "gpu memory token index d2", "query vector alpha d3"]

        def build(cls):
            pc = PipelineConfig(stage1_model="random:tiny", stage2_model="random:tiny", stage3_model="random:tiny",
                                device="cpu", cache_dir=os.path.join(out_dir, "m"), index_dir=os.path.join(out_dir, "i"),
                                log_file=os.path.join(out_dir, f"t{rank}.log"), log_level="ERROR", stage1_top_k=4,
                                stage2_top_k=3, stage3_top_k=2, save_intermediate_results=True,
                                stage2_precompute_document_embeddings=True, stage3_cache_document_tokens=True)
            p = cls(config=pc)
            p.stage1 = Stage1Retriever(Stage1Config(model_name="random:tiny", device="cpu", cache_dir=pc.cache_dir,
                                                    index_dir=pc.index_dir, top_k_candidates=4),
                                       model=SentenceEncoder("random:tiny", device="cpu"),
                                       index_factory=lambda d: OracleIndex(d))
            p.stage2 = ColBERTScorer(Stage2Config(model_name="random:tiny", device="cpu", top_k_candidates=3,
                                                  precompute_document_embeddings=True),
                                     maxsim_fn=oracle_maxsim, maxsim_indexed_fn=oracle_maxsim_indexed,
                                     maxsim_indexed_batch_fn=oracle_maxsim_indexed_batch)
            p.stage3 = AdaptiveCrossEncoderReranker(Stage3Config(model_name="random:tiny", device="cpu", top_k_final=2))
            return p
        par, single = build(ShardedRetrievalPipeline), build(RetrievalPipeline)
        par._merge_fn = oracle_merge
        par.add_documents(docs)                       # 4 documents on 3 ranks: rows 0-1, 2-3 and NOTHING on rank 2
        single.add_documents(docs)
        assert par.hi - par.lo == (2, 2, 0)[rank] and par._arrays_agreed()
        qs = ["neural network", "alpha d3", "gpu"]
        for got, want in zip(par.search_many(qs) + [par.search(qs[1])], single.search_many(qs) + [single.search(qs[1])]):
            for stage, key in (("stage1_results", "stage1_score"), ("stage2_results", "stage2_score"), ("results", "stage3_score")):
                assert [r["doc_id"] for r in got[stage]] == [r["doc_id"] for r in want[stage]], stage
                np.testing.assert_allclose([r[key] for r in got[stage]], [r[key] for r in want[stage]], atol=2e-5)
                assert all(r["document"] == docs[r["doc_id"]] for r in got[stage])
        json.dump({"ok": True}, open(os.path.join(out_dir, f"tiny{rank}.json"), "w"))
    finally:
        dist.destroy_process_group()


def test_three_stage_pipeline_with_an_empty_shard(tmp_path):
    """Fewer documents than the ranks can share evenly (4 on 3 ranks: the ceil division leaves rank 2 with nothing): the
    rank without rows, token store, id cache or BM25 postings takes part in every collective (all-gathers, the two
    all-reduce(MAX), the text gather, the BM25 statistics exchange) and every rank returns the single-process records."""
    world = 3
    mp.spawn(_tiny_pipeline_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"tiny{r}.json") for r in range(world))
