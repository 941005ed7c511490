"""GPU: the shard exchange + merge on real kernels.  The box has one GPU, so
(a) R shards live on the same device and their real search outputs are merged by
the HIP merge kernel, and (b) ShardedFlatIPIndex runs its RCCL all-gather + merge
code path in a world of one rank."""
import os
import socket

import numpy as np
import pytest

from helpers import check_topk, make_corpus

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R", [2, 8])
def test_shards_merged_on_device_equal_unsharded(R):
    import torch
    from tristage_rag_amd.index import FlatIPIndex, merge_topk
    from tristage_rag_amd.sharded import shard_bounds
    n, d, k, B = 200_000, 128, 1000, 64
    rng = np.random.default_rng(8)
    corpus = make_corpus(n // 2, d, dtype="f16")
    corpus = np.concatenate([corpus, corpus[rng.permutation(n // 2)]])      # every row twice: ties across shards
    queries = make_corpus(B, d, seed=4321, dtype="f16")
    q = torch.from_numpy(queries).cuda().half()
    whole = FlatIPIndex(d, dtype="f16")
    whole.add(corpus)
    D0, I0 = whole.search(q, k)
    parts = []
    for r in range(R):
        lo, hi = shard_bounds(n, R, r)
        idx = FlatIPIndex(d, dtype="f16")
        idx.add(corpus[lo:hi])
        idx.set_id_offset(lo)
        parts.append(idx.search(q, k))
        idx.close()
    D, I = merge_topk(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]))
    assert torch.equal(I, I0) and torch.equal(D, D0)
    check_topk(D.cpu().numpy()[:4], I.cpu().numpy()[:4], corpus, queries[:4], k)
    whole.close()


def test_sharded_index_rccl_world_of_one():
    import torch
    import torch.distributed as dist
    from tristage_rag_amd.sharded import ShardedFlatIPIndex
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        n, d, k = 100_000, 128, 100
        corpus = make_corpus(n, d, dtype="f16")
        queries = make_corpus(64, d, seed=9, dtype="f16")
        idx = ShardedFlatIPIndex(d, n, dtype="f16", device=0)
        idx.always_exchange = True
        idx.add_global(torch.from_numpy(corpus).cuda().half())
        q = torch.from_numpy(queries).cuda().half()
        D, I = idx.search(q, k)
        check_topk(D.cpu().numpy(), I.cpu().numpy(), corpus, queries, k)
        outs = [idx.search(q, k, async_=True) for _ in range(3)]
        idx.finish()
        for Da, Ia in outs:
            assert torch.equal(Ia, I) and torch.equal(Da, D)
        # pipelined submission (internal streams) interleaved with the RCCL exchange + merge,
        # exactly what bench.py does for N > 1
        q2 = torch.from_numpy(make_corpus(64, d, seed=10, dtype="f16")).cuda().half()
        D2, I2 = idx.search(q2, k)
        torch.cuda.synchronize()
        outs = [idx.search(q if i % 2 == 0 else q2, k, async_=True, inputs_ready=True) for i in range(60)]
        idx.finish()
        for i, (Da, Ia) in enumerate(outs):
            assert torch.equal(Ia, I if i % 2 == 0 else I2) and torch.equal(Da, D if i % 2 == 0 else D2), i
    finally:
        dist.destroy_process_group()


def _two_rank_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from helpers import make_corpus as mk
    from tristage_rag_amd.sharded import ShardedFlatIPIndex, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        n, d, k = 150_001, 128, 200                      # odd row count: unequal shards
        corpus = mk(n, d, dtype="f16")
        corpus[n // 2 + 7] = corpus[5]                   # the same row on both shards: a tie across ranks
        lo, hi = shard_bounds(n, world, rank)
        idx = ShardedFlatIPIndex(d, n, dtype="f16", device=0)
        idx.add_local(torch.from_numpy(corpus[lo:hi]).cuda().half())
        assert idx.local_index.ntotal == hi - lo
        qs = [torch.from_numpy(mk(64, d, seed=40 + i, dtype="f16")).cuda().half() for i in range(3)]
        sync = [idx.search(q, k) for q in qs]
        torch.cuda.synchronize()
        outs = [idx.search(qs[i % 3], k, async_=True, inputs_ready=True) for i in range(130)]   # > 120 batches: a collective finish() mid-way
        idx.finish()
        for i, (Da, Ia) in enumerate(outs):
            assert torch.equal(Ia, sync[i % 3][1]) and torch.equal(Da, sync[i % 3][0]), i
        np.save(os.path.join(out_dir, f"D{rank}.npy"), torch.stack([s[0] for s in sync]).cpu().numpy())
        np.save(os.path.join(out_dir, f"I{rank}.npy"), torch.stack([s[1] for s in sync]).cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_full_protocol(tmp_path):
    """Two PROCESSES (gloo group, exchange staged through the host) each owning a row shard on
    the same GPU: the whole multi-rank protocol on the real kernels — global ids, pipelined
    asynchronous searches, the collective finish(), the packed merge — equals the oracle and is
    identical on both ranks."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    n, d, k = 150_001, 128, 200
    corpus = make_corpus(n, d, dtype="f16")
    corpus[n // 2 + 7] = corpus[5]
    D0, I0 = np.load(tmp_path / "D0.npy"), np.load(tmp_path / "I0.npy")
    assert np.array_equal(I0, np.load(tmp_path / "I1.npy")) and np.array_equal(D0, np.load(tmp_path / "D1.npy"))
    for i in range(3):
        q = make_corpus(64, d, seed=40 + i, dtype="f16")
        check_topk(D0[i][:8], I0[i][:8], corpus, q[:8], k)


def test_sharded_numpy_call_and_odd_k_on_the_real_index():
    """FAISS-style numpy queries through ShardedFlatIPIndex with the real HIP index underneath (the exchange
    buffer lives on the GPU: host queries are moved there first), and an odd B*k (the int64 id block of the
    packed buffer starts at an 8-byte pad)."""
    import torch
    from tristage_rag_amd.sharded import ShardedFlatIPIndex
    n, d = 60_000, 96
    corpus = make_corpus(n, d, dtype="f16")
    idx = ShardedFlatIPIndex(d, n, dtype="f16", device=0)
    idx.add_global(torch.from_numpy(corpus).cuda().half())
    for B, k in ((7, 50), (1, 25), (3, 33)):
        q = make_corpus(B, d, seed=70 + B, dtype="f16")
        D, I = idx.search(q, k)                                   # numpy in -> numpy out
        assert isinstance(D, np.ndarray) and D.shape == (B, k)
        check_topk(D, I, corpus, q, k)
        Dt, It = idx.search(torch.from_numpy(q), k)               # host tensor in
        assert np.array_equal(It.cpu().numpy(), I) and np.array_equal(Dt.cpu().numpy(), D)


def test_sharded_async_repairs_survive_many_pending_batches():
    """with more asynchronous batches in flight than the local index tracks, a batch whose fused
    filter failed (20 000 exact ties overflow the candidate list) must still be re-exchanged after its local
    repair.  RCCL world of one with the exchange forced on; 260 batches (the wrapper finishes collectively every 120),
    every other one failing; odd B*k too."""
    import torch
    import torch.distributed as dist
    from tristage_rag_amd.sharded import ShardedFlatIPIndex
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        n, d, k = 120_000, 128, 301
        corpus = make_corpus(n, d, seed=31, dtype="f16")
        corpus[40_000:60_000] = corpus[11]
        ok_q = make_corpus(9, d, seed=32, dtype="f16")
        bad_q = ok_q.copy()
        bad_q[4] = corpus[11]
        idx = ShardedFlatIPIndex(d, n, dtype="f16", device=0)
        idx.always_exchange = True
        idx.add_global(torch.from_numpy(corpus).cuda().half())
        qa, qb = torch.from_numpy(ok_q).cuda().half(), torch.from_numpy(bad_q).cuda().half()
        Da, Ia = idx.search(qa, k)
        Db, Ib = idx.search(qb, k)
        assert idx.local_index.last_search_info()["path"] == "filter+dense-fallback"
        check_topk(Db.cpu().numpy(), Ib.cpu().numpy(), corpus, bad_q, k)
        torch.cuda.synchronize()
        for ready in (False, True):
            outs = [idx.search(qb if i % 2 else qa, k, async_=True, inputs_ready=ready) for i in range(260)]
            idx.finish()
            for i, (D, I) in enumerate(outs):
                wd, wi = (Db, Ib) if i % 2 else (Da, Ia)
                assert torch.equal(I, wi) and torch.equal(D, wd), (ready, i)
        # the plain index: its own automatic finish() must hand the repaired tickets to the next finish()
        li = FlatLocal = idx.local_index
        li.auto_finish = True
        outs = [li.search(qb if i % 2 else qa, k, async_=True) for i in range(250)]   # > PENDING_PASSES: one internal finish()
        redone = li.finish()
        assert len(redone) == 125, redone
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("R,k,B", [(20, 1000, 5), (9, 2048, 3), (33, 700, 2)])
def test_merge_of_more_lists_than_fit_one_workgroup(R, k, B):
    """nlists * k above the 16384 keys one workgroup selects among in LDS (e.g. 9 ranks x k = 2048): the lists are
    merged in groups, then the groups' results — same canonical result as the oracle's merge, duplicated scores
    across lists and padding entries included."""
    import torch
    from oracle import oracle
    from tristage_rag_amd.index import merge_topk
    rng = np.random.default_rng(R * 1000 + k)
    pool = np.round(rng.standard_normal(4000), 2).astype(np.float32)           # many exact score ties
    scores = np.empty((R, B, k), dtype=np.float32)
    ids = np.empty((R, B, k), dtype=np.int64)
    for r in range(R):
        for b in range(B):
            n_valid = k if (r + b) % 5 else k - 37                               # some lists end in padding
            sc = np.sort(rng.choice(pool, size=n_valid))[::-1]
            idv = rng.choice(10_000_000, size=n_valid, replace=False).astype(np.int64) + r * 10_000_000
            order = np.lexsort((idv, -sc))                                      # canonical order inside a list
            scores[r, b, :n_valid], ids[r, b, :n_valid] = sc[order], idv[order]
            scores[r, b, n_valid:], ids[r, b, n_valid:] = -3.4028234663852886e38, -1
    D0, I0 = oracle.merge_topk(scores, ids, k)
    D, I = merge_topk(torch.from_numpy(scores).cuda(), torch.from_numpy(ids).cuda())
    assert np.array_equal(I.cpu().numpy(), I0) and np.array_equal(D.cpu().numpy(), D0)


def _sharded_pipeline_worker(rank, world, port, out_dir):
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from pipeline_pairs import synth_docs
    from tristage_rag_amd.parallel_pipeline import ShardedRetrievalPipeline
    from tristage_rag_amd.retrieval_pipeline import PipelineConfig, RetrievalPipeline
    from tristage_rag_amd.sharded import shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        docs = synth_docs(601, seed=5)
        queries = ["neural networks attention", "language retrieval system", "gpu memory index", docs[11], "vector search rank",
                   docs[590]]

        def cfg(name, bm25):
            return PipelineConfig(stage1_model="random:tiny", stage2_model="random:tiny", stage3_model="random:tiny",
                                  device="cuda", cache_dir=os.path.join(out_dir, "m"), index_dir=os.path.join(out_dir, "i"),
                                  log_file=os.path.join(out_dir, f"{name}{rank}.log"), log_level="WARNING", stage1_top_k=200,
                                  stage2_top_k=40, stage3_top_k=10, stage1_enable_bm25=bm25, stage1_use_fp16=False,
                                  stage2_use_fp16=False, stage3_use_fp16=False, save_intermediate_results=True,
                                  stage2_precompute_document_embeddings=True, stage3_cache_document_tokens=True)
        res = {}
        for bm25 in (False, True):
            par = ShardedRetrievalPipeline(config=cfg("par", bm25))
            par.add_documents(docs)
            lo, hi = shard_bounds(len(docs), world, rank)
            info = par.get_pipeline_info()["sharding"]
            assert info["rows"] == [lo, hi] and len(par.stage2.token_store) == hi - lo == len(par.stage3._pairs)
            assert par._arrays_agreed()
            many = par.search_many(queries)
            one = par.search(queries[3])
            single = RetrievalPipeline(config=cfg("one", bm25))
            single.add_documents(docs)
            assert info["stage2_token_rows"] < single.stage2.token_store.rows
            ref = single.search_many(queries)
            for a, b in zip(many + [one], ref + [ref[3]]):
                for stage, key in (("stage1_results", "stage1_score"), ("stage2_results", "stage2_score"), ("results", "stage3_score")):
                    ia, ib = [x["doc_id"] for x in a[stage]], [x["doc_id"] for x in b[stage]]
                    sa, sb = np.array([x[key] for x in a[stage]]), np.array([x[key] for x in b[stage]])
                    assert len(ia) == len(ib), (stage, len(ia), len(ib))
                    np.testing.assert_allclose(sa, sb, atol=2e-5)     # fp32 models: the batches differ in shape only
                    for x, y, u, v in zip(ia, ib, sa, sb):
                        assert x == y or abs(u - v) < 2e-5, (stage, x, y, u, v)
                    assert all(x["document"] == docs[x["doc_id"]] for x in a[stage])
            res[str(bm25)] = [[x["doc_id"] for x in a["results"]] for a in many]
        json.dump(res, open(os.path.join(out_dir, f"ids{rank}.json"), "w"))
    finally:
        dist.destroy_process_group()


def test_row_sharded_three_stage_pipeline_on_the_real_kernels(tmp_path):
    """Two PROCESSES on the one GPU of the test box (gloo group, collectives staged through the host), each holding
    its half of the rows in the HIP index, its half of the stage-2 token store and its half of the stage-3 token-id
    cache: the array path of search_many / search — owner-scored MaxSim (ts_maxsim_indexed_batch on ragged owned
    candidates) and cross-encoder pairs, one all-reduce(MAX) per stage — equals the single-process pipeline (ids
    exact up to 2e-5 near-ties, scores 2e-5), dense and with BM25 + RRF, and is identical on both ranks."""
    import json
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sharded_pipeline_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert json.load(open(tmp_path / "ids0.json")) == json.load(open(tmp_path / "ids1.json"))
