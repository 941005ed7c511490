"""GPU: BASELINE.json configs[2] and the pipeline half of configs[4] AT THEIR SHAPES, with parity.

configs[2]  NFCorpus-shaped: 3 633 documents, stage 1 top-1000 -> stage 2 keep 100 -> stage 3 top-10, bf16
            (bf16 index, bf16 token store, bf16 autocast forwards), models of the reference's architectures
            (BERT-base-shaped bi-encoder, ModernBERT-base backbone, MiniLM-L6 cross-encoder; randomly
            initialised — no weights exist offline, so the forwards are "parity unpinned" for real
            checkpoints; what is pinned is everything the HIP kernels and the host logic compute).
configs[4]  pipeline half: a 2^20-row x 1024 bf16 shard (fused-filter path of the scan) feeding stage 2 and a
            bge-reranker-large-shaped (XLM-R-large: 24 layers, hidden 1024) stage 3 with max_length 256 over
            100 pairs per query.  Stages 1 / 2 use reduced-DEPTH encoders of the right width so that the
            one-million-document index build stays inside a test's time; depth does not change any shape
            the HIP kernels see.

Checks, per query, for both ``search`` and ``search_many`` (semantics of reference
src/retrieval_pipeline.py:323-424):
  stage 1  ids + scores == oracle.ip_topk on the STORED (bf16-rounded) embeddings and the query embedding the
           pipeline actually used (captured), by the near-tie rule of oracle.check_topk;
  stage 2  the kept 100 == the stable top-100 of oracle.maxsim_scores on the STORED token matrices and the
           captured query tokens, scores to 1e-5;
  stage 3  raw scores of the bf16 forward within bf16 tolerance of a CPU fp32 forward of the same model, the
           pipeline's final records == min-max + stable sort of its own raw scores, and — with the forward in
           fp32 on the GPU — the ORDER of the CPU fp32 forward (query subset)."""
import numpy as np
import pytest

from oracle import oracle
from pipeline_pairs import assert_same_ranking

pytestmark = pytest.mark.gpu


class Capture:
    """Records what a bound method returns (the embeddings a stage really used)."""

    def __init__(self, obj, name):
        self.out = []
        orig = getattr(obj, name)

        def wrapped(*a, **k):
            r = orig(*a, **k)
            self.out.append(r)
            return r
        setattr(obj, name, wrapped)


def _texts(n_docs, n_queries, lo, hi, seed=0):
    rng = np.random.default_rng(seed)
    vocab = [f"w{i}" for i in range(5000)]

    def make(n, a, b):   # (one draw of all word indices: a million rng.choice calls on a list take minutes)
        lens = rng.integers(a, b, size=n)
        words = rng.integers(0, len(vocab), size=int(lens.sum())).tolist()
        out, at = [], 0
        for m in lens.tolist():
            out.append(" ".join(vocab[i] for i in words[at: at + m]))
            at += m
        return out
    return make(n_docs, lo, hi), make(n_queries, 4, 16)


def _check_stage1(records, stored, q_used, k, dtype="bf16"):
    ids = np.array([[r["doc_id"] for r in recs] for recs in records], dtype=np.int64)
    sc = np.array([[r["stage1_score"] for r in recs] for recs in records], dtype=np.float32)
    assert ids.shape == (len(records), k)
    return oracle.check_topk(sc, ids, stored, oracle.quantize(q_used, dtype), k)


def _check_stage2(p, s1_records, s2_records, q_tokens, keep):
    st = p.stage2.token_store
    store = st.data
    qv = oracle.quantize(q_tokens.float().cpu().numpy(), "bf16")           # the kernel reads the query in the store's dtype
    docs = []
    for r in s1_records:
        slot = p.stage2._store_slot[r["doc_id"]]
        docs.append(store[st.starts[slot]: st.starts[slot] + st.lens[slot]].float().cpu().numpy())
    want = oracle.maxsim_scores(qv, docs)
    order = np.argsort(-want, kind="stable")[:keep]                         # stable sort desc, keep top_k (reference :293-297)
    assert_same_ranking([r["doc_id"] for r in s2_records], [r["stage2_score"] for r in s2_records],
                        [s1_records[i]["doc_id"] for i in order], want[order], atol=1e-5, tie=2e-6, what="stage 2")


def _check_stage3_host_logic(p, query, s2_records, final, top_k):
    """final == min-max + stable sort + cut of the reranker's own raw scores (reference :212-264)."""
    again = p.stage3.rerank(query, s2_records)[:top_k]
    assert [r["doc_id"] for r in again] == [r["doc_id"] for r in final]
    np.testing.assert_allclose([r["stage3_score"] for r in again], [r["stage3_score"] for r in final], atol=1e-6)
    s = np.array([r["stage3_score"] for r in final])
    assert (np.diff(s) <= 0).all() and 0.0 <= s.min() and s.max() <= 1.0


def _stage3_against_cpu(p, spec, query, s2_records, max_length, bf16_tol):
    import torch
    from tristage_rag_amd.encoders import CrossEncoderModel
    pairs = [[query, r["document"]] for r in s2_records]
    raw_gpu = p.stage3.model.predict(pairs, batch_size=64)                 # bf16 autocast, as the pipeline runs it
    cpu = CrossEncoderModel(spec, device="cpu", max_length=max_length, use_amp=False)
    raw_cpu = cpu.predict(pairs, batch_size=32)
    np.testing.assert_allclose(raw_gpu, raw_cpu, atol=bf16_tol)            # bf16 forward vs fp32 forward
    # the ORDER: same weights, fp32 on the GPU, against the CPU fp32 forward; only near-ties may swap
    g32 = CrossEncoderModel(spec, device="cuda", max_length=max_length, use_amp=False)
    raw_g32 = g32.predict(pairs, batch_size=64)
    ids = [r["doc_id"] for r in s2_records]
    og, oc = np.argsort(-raw_g32, kind="stable"), np.argsort(-raw_cpu, kind="stable")
    assert_same_ranking([ids[i] for i in og], raw_g32[og], [ids[i] for i in oc], raw_cpu[oc], atol=1e-4, tie=2e-5,
                        what="stage 3 fp32 order")
    del g32, cpu
    torch.cuda.empty_cache()


def _run_config(tmp_path, models, n_docs, doc_words, n_single, n_many, s1k, s2k, topk, s3_max_length, s3_cpu_queries,
                expect_path, bf16_tol, build_batch=64):
    import torch
    from tristage_rag_amd.retrieval_pipeline import PipelineConfig, RetrievalPipeline
    docs, queries = _texts(n_docs, n_single + n_many, *doc_words)
    pc = PipelineConfig(stage1_model=models[0], stage2_model=models[1], stage3_model=models[2], device="cuda",
                        cache_dir=str(tmp_path / "m"), index_dir=str(tmp_path / "i"), log_file=str(tmp_path / "p.log"),
                        log_level="WARNING", stage1_top_k=s1k, stage2_top_k=s2k, stage3_top_k=topk,
                        stage1_enable_bm25=False, stage1_index_dtype="bf16", stage1_batch_size=max(256, build_batch),
                        stage2_batch_size=build_batch, stage3_batch_size=64, stage3_max_length=s3_max_length,
                        stage2_precompute_document_embeddings=True, save_intermediate_results=True)
    p = RetrievalPipeline(config=pc)
    p.add_documents(docs)
    idx = p.stage1.faiss_index
    assert type(idx).__name__ == "FlatIPIndex" and idx.ntotal == n_docs and idx.storage_dtype == "bf16"
    assert len(p.stage2.token_store) == n_docs and p.stage2.token_store.data.dtype == torch.bfloat16
    stored = idx.reconstruct_n(0, n_docs)
    cap_q1 = Capture(p.stage1, "_normalized_query_tensor")
    cap_q2 = Capture(p.stage2, "encode_query")
    cap_q2m = Capture(p.stage2, "encode_queries_batch")
    swaps = 0
    # ---- search(): one query at a time
    for j in range(n_single):
        q = queries[j]
        r = p.search(q, top_k=topk)
        assert idx.last_search_info()["path"] == expect_path
        assert len(r["stage1_results"]) == s1k and len(r["stage2_results"]) == s2k and len(r["results"]) == topk
        swaps += _check_stage1([r["stage1_results"]], stored, cap_q1.out[-1].float().cpu().numpy(), s1k)
        _check_stage2(p, r["stage1_results"], r["stage2_results"], cap_q2.out[-1][0], s2k)
        n1, n2 = len(cap_q1.out), len(cap_q2.out)
        _check_stage3_host_logic(p, q, r["stage2_results"], r["results"], topk)
        del cap_q1.out[n1:], cap_q2.out[n2:]
        if j < s3_cpu_queries:
            _stage3_against_cpu(p, models[2], q, r["stage2_results"], s3_max_length, bf16_tol)
        assert set(r["timing"]) == {"stage1_time", "stage2_time", "stage3_time", "total_time"}
    # ---- search_many(): every stage batched over the queries
    qs = queries[n_single:]
    many = p.search_many(qs, top_k=topk)
    assert len(many) == len(qs)
    q1 = cap_q1.out[-1].float().cpu().numpy()
    assert q1.shape[0] == len(qs)
    swaps += _check_stage1([m["stage1_results"] for m in many], stored, q1, s1k)
    q2 = cap_q2m.out[-1]
    for j, m in enumerate(many):
        assert m["query"] == qs[j] and len(m["results"]) == topk
        _check_stage2(p, m["stage1_results"], m["stage2_results"], q2[j], s2k)
        s = np.array([x["stage3_score"] for x in m["results"]])
        assert (np.diff(s) <= 0).all() and s.max() <= 1.0 and s.min() >= 0.0
        assert {x["doc_id"] for x in m["results"]} <= {x["doc_id"] for x in m["stage2_results"]}
    # the batched stage 3 sees the same pairs in other batches: its raw scores stay within bf16 noise of the
    # per-query forward (the order of near-ties may differ — the score spread of a random cross-encoder is ~1e-2)
    m0 = many[0]
    pairs = [[qs[0], x["document"]] for x in m0["stage2_results"]]
    raw = np.asarray(p.stage3.model.predict(pairs, batch_size=64), dtype=np.float64)
    norm = np.array(p.stage3._normalize_scores(list(raw)))
    by_id = {x["doc_id"]: v for x, v in zip(m0["stage2_results"], norm)}
    spread = raw.max() - raw.min()
    for x in m0["results"]:
        assert abs(x["stage3_score"] - by_id[x["doc_id"]]) <= 2 * bf16_tol / max(spread, 1e-9) + 1e-6
    return p, swaps


def test_config2_nfcorpus_shape_three_stage_bf16(tmp_path):
    p, _ = _run_config(tmp_path, ("random:bert", "random:modernbert", "random:minilm"), n_docs=3633, doc_words=(40, 160),
                       n_single=3, n_many=8, s1k=1000, s2k=100, topk=10, s3_max_length=256, s3_cpu_queries=2,
                       expect_path="dense", bf16_tol=2e-3)
    info = p.get_pipeline_info()
    assert info["stage1_stats"]["total_documents"] == 3633 and info["performance_stats"]["total_queries"] >= 11


def test_config4_pipeline_half_xlmr_large_stage3(tmp_path):
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 60e9:
        pytest.skip("needs ~40 GB of free HBM")
    n = 1 << 20
    p, _ = _run_config(tmp_path, ("random:bert:1024:2:16", "random:modernbert:768:2:12", "random:xlmr-large"),
                       n_docs=n, doc_words=(5, 11), n_single=2, n_many=4, s1k=1000, s2k=100, topk=10,
                       s3_max_length=256, s3_cpu_queries=1, expect_path="filter", bf16_tol=4e-3, build_batch=1024)
    assert p.stage1.embedding_dim == 1024 and p.stage3.model.model.config.num_hidden_layers == 24
    assert p.stage3.model.model.config.hidden_size == 1024 and p.stage3.config.max_length == 256
