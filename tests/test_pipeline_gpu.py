"""GPU: the whole three-stage pipeline on the real HIP entry points (FlatIPIndex,
ts_maxsim) against the same pipeline on CPU with the oracle-backed doubles, same
randomly initialised models (fp32 on both sides so the comparison is tight)."""
import json
import os

import numpy as np
import pytest

from doubles import OracleIndex, oracle_maxsim

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kat.json")))


def _corpus(n=300):
    rng = np.random.default_rng(3)
    words = ("neural network attention transformer language retrieval index vector query document "
             "learning model data system search rank score token embedding gpu memory").split()
    docs = [" ".join(rng.choice(words, size=int(rng.integers(4, 30)))) for _ in range(n)]
    return docs + list(KAT["bm25"]["documents"])


def _build(device, tmp_path, doubles, **extra):
    from tristage_rag_amd.retrieval_pipeline import PipelineConfig, RetrievalPipeline
    from tristage_rag_amd.stage1_retriever import Stage1Config, Stage1Retriever
    from tristage_rag_amd.stage2_rescorer import ColBERTScorer, Stage2Config
    pc = PipelineConfig(stage1_model="random:tiny", stage2_model="random:tiny", stage3_model="random:tiny",
                        device=device, cache_dir=str(tmp_path / "m"), index_dir=str(tmp_path / "i"),
                        log_file=str(tmp_path / f"{device}.log"), stage1_top_k=40, stage2_top_k=15,
                        stage3_top_k=5, stage1_use_fp16=False, stage2_use_fp16=False, stage3_use_fp16=False,
                        save_intermediate_results=True, **extra)
    p = RetrievalPipeline(config=pc)
    p.initialize_stages() if not doubles else None
    if doubles:
        from tristage_rag_amd.encoders import SentenceEncoder
        from tristage_rag_amd.stage3_reranker import AdaptiveCrossEncoderReranker, Stage3Config
        p.stage1 = Stage1Retriever(Stage1Config(model_name="random:tiny", device="cpu", cache_dir=pc.cache_dir,
                                                index_dir=pc.index_dir, top_k_candidates=40,
                                                enable_bm25=pc.stage1_enable_bm25, use_fp16=False),
                                   model=SentenceEncoder("random:tiny", device="cpu"),
                                   index_factory=lambda d: OracleIndex(d))
        p.stage2 = ColBERTScorer(Stage2Config(model_name="random:tiny", device="cpu", top_k_candidates=15,
                                              use_fp16=False), maxsim_fn=oracle_maxsim)
        p.stage3 = AdaptiveCrossEncoderReranker(Stage3Config(model_name="random:tiny", device="cpu",
                                                             top_k_final=5, use_fp16=False))
    return p


@pytest.mark.parametrize("bm25", [False, True])
def test_pipeline_gpu_matches_cpu_doubles(tmp_path, bm25):
    docs = _corpus()
    gpu = _build("cuda", tmp_path, doubles=False, stage1_enable_bm25=bm25)
    cpu = _build("cpu", tmp_path, doubles=True, stage1_enable_bm25=bm25)
    gpu.add_documents(docs)
    cpu.add_documents(docs)
    assert type(gpu.stage1.faiss_index).__name__ == "FlatIPIndex" and gpu.stage1.faiss_index.ntotal == len(docs)
    queries = ["neural networks attention", "language retrieval system", "gpu memory index", "zzz unknown words"]
    for q in queries:
        a, b = gpu.search(q), cpu.search(q)
        for stage, key in (("stage1_results", "stage1_score"), ("stage2_results", "stage2_score"), ("results", "stage3_score")):
            ia = [r["doc_id"] for r in a[stage]]
            ib = [r["doc_id"] for r in b[stage]]
            sa = np.array([r[key] for r in a[stage]])
            sb = np.array([r[key] for r in b[stage]])
            assert len(ia) == len(ib)
            np.testing.assert_allclose(sa, sb, atol=1e-3)          # north_star: scores within 1e-3
            if ia != ib:                                            # only near-ties may swap
                assert sorted(ia) == sorted(ib) or np.abs(sa - sb).max() < 1e-4
                for x, y, u, v in zip(ia, ib, sa, sb):
                    assert x == y or abs(u - v) < 1e-4
    many = gpu.search_many(queries)            # every stage batched over the queries
    for q, r in zip(queries, many):
        one = gpu.search(q)
        for stage, key in (("stage1_results", "stage1_score"), ("stage2_results", "stage2_score"), ("results", "stage3_score")):
            ia, ib = [x["doc_id"] for x in r[stage]], [x["doc_id"] for x in one[stage]]
            sa, sb = np.array([x[key] for x in r[stage]]), np.array([x[key] for x in one[stage]])
            assert len(ia) == len(ib)
            np.testing.assert_allclose(sa, sb, atol=1e-4)           # batch-padding noise only
            for x, y, u, v in zip(ia, ib, sa, sb):
                assert x == y or abs(u - v) < 1e-4                  # only near-ties may swap


def test_bf16_pipeline_runs_and_ranks(tmp_path):
    """The production setting (bf16 autocast, fp16 index): sanity, not bit parity."""
    from tristage_rag_amd.retrieval_pipeline import PipelineConfig, RetrievalPipeline
    pc = PipelineConfig(stage1_model="random:tiny", stage2_model="random:tiny", stage3_model="random:tiny",
                        device="cuda", cache_dir=str(tmp_path / "m"), index_dir=str(tmp_path / "i"),
                        log_file=str(tmp_path / "b.log"), stage1_top_k=50, stage2_top_k=20, stage3_top_k=10,
                        stage1_enable_bm25=False, stage1_index_dtype="f16",
                        stage2_cache_document_embeddings=True)
    p = RetrievalPipeline(config=pc)
    docs = _corpus(500)
    p.add_documents(docs)
    r = p.search(docs[7])                      # a document as its own query
    assert len(r["results"]) == 10 and r["timing"]["total_time"] > 0
    s1 = p.stage1.search(docs[7], 5)
    assert s1[0]["doc_id"] == 7 and s1[0]["score"] > 0.99
    r2 = p.search(docs[7])                     # second time: stage-2 token matrices come from the cache
    assert [x["doc_id"] for x in r2["results"]] == [x["doc_id"] for x in r["results"]]


def test_token_store_pipeline_equals_reencoding_pipeline(tmp_path):
    """Stage 2 from the resident token store (filled at add_documents) vs re-encoding the
    candidates per query (the reference's behaviour): same ranking, scores to 1e-3."""
    from tristage_rag_amd.retrieval_pipeline import PipelineConfig, RetrievalPipeline
    docs = _corpus(400)
    outs = []
    for store in (False, True):
        pc = PipelineConfig(stage1_model="random:tiny", stage2_model="random:tiny", stage3_model="random:tiny",
                            device="cuda", cache_dir=str(tmp_path / "m"), index_dir=str(tmp_path / "i"),
                            log_file=str(tmp_path / f"s{store}.log"), stage1_top_k=60, stage2_top_k=20,
                            stage3_top_k=5, stage1_enable_bm25=False, stage2_use_fp16=False,
                            save_intermediate_results=True, stage2_precompute_document_embeddings=store)
        p = RetrievalPipeline(config=pc)
        p.initialize_stages()
        p.add_documents(docs[:250])
        p.add_documents(docs[250:])
        if store:
            assert len(p.stage2.token_store) == len(docs)
        qs = ("neural network attention", "gpu memory", docs[5])
        outs.append([p.search(q) for q in qs])
        if store:   # batched: ONE ts_maxsim_indexed_batch launch for the three queries
            for one, many in zip(outs[-1], p.search_many(list(qs))):
                sa = {r["doc_id"]: r["stage2_score"] for r in one["stage2_results"]}
                sb = {r["doc_id"]: r["stage2_score"] for r in many["stage2_results"]}
                assert len(set(sa) ^ set(sb)) <= 2
                for k in set(sa) & set(sb):
                    assert abs(sa[k] - sb[k]) < 1e-4
    for a, b in zip(*outs):
        sa = {r["doc_id"]: r["stage2_score"] for r in a["stage2_results"]}
        sb = {r["doc_id"]: r["stage2_score"] for r in b["stage2_results"]}
        assert set(sa) == set(sb) or len(set(sa) ^ set(sb)) <= 2
        for k in set(sa) & set(sb):
            assert abs(sa[k] - sb[k]) < 1e-3


def test_hip_graph_query_forwards_match_eager(tmp_path):
    """Batch-1 encoder forwards replayed from captured HIP graphs (padded to a length bucket)
    give the eager results: stage-1 embeddings, stage-2 token matrices, final ranking."""
    import torch
    from tristage_rag_amd.encoders import SentenceEncoder
    from tristage_rag_amd.stage2_rescorer import ColBERTScorer, Stage2Config
    eager = SentenceEncoder("random:tiny", device="cuda")
    graph = SentenceEncoder("random:tiny", device="cuda", use_hip_graph=True)
    texts = ["a", "neural network attention", " ".join(["word"] * 7), " ".join(["tok"] * 30), " ".join(["z"] * 200)]
    for t in texts * 2:                                   # second round replays the captured graphs
        np.testing.assert_allclose(graph.encode(t), eager.encode(t), atol=1e-4)
    assert graph._graphed is not None and not graph._graphed._broken and len(graph._graphed._graphs) >= 3
    # a BATCH of queries: padded to a (rows, length) bucket, replayed; twice (capture, then replay)
    for _ in range(2):
        np.testing.assert_allclose(graph.encode(texts[:4], batch_size=64), eager.encode(texts[:4], batch_size=64), atol=1e-4)
        many = [f"query number {i} " + "w " * (i % 9) for i in range(37)]
        np.testing.assert_allclose(graph.encode(many, batch_size=64), eager.encode(many, batch_size=64), atol=1e-4)
    assert any(r > 1 for r, _ in graph._graphed._graphs)
    s_e = ColBERTScorer(Stage2Config(model_name="random:tiny", device="cuda", use_fp16=False))
    s_g = ColBERTScorer(Stage2Config(model_name="random:tiny", device="cuda", use_fp16=False, use_hip_graph=True))
    for t in texts:
        a, b = s_e.encode_query(t), s_g.encode_query(t)
        assert a.shape == b.shape
        assert torch.allclose(a, b, atol=1e-4)
    for _ in range(2):
        for a, b in zip(s_e.encode_queries_batch(texts[:4] * 5), s_g.encode_queries_batch(texts[:4] * 5)):
            assert a.shape == b.shape and torch.allclose(a, b, atol=1e-4)
    assert any(r > 1 for r, _ in s_g._graphed._graphs)
    # stage 3: one query's pairs as a single graph replay (rows and columns padded to a bucket)
    from tristage_rag_amd.encoders import CrossEncoderModel
    ce_e = CrossEncoderModel("random:tiny", device="cuda", use_amp=False)
    ce_g = CrossEncoderModel("random:tiny", device="cuda", use_amp=False, use_hip_graph=True)
    cdocs = _corpus(120)
    for npairs in (1, 17, 100, 100, 120):
        pairs = [["neural network attention", d] for d in cdocs[:npairs]]
        np.testing.assert_allclose(ce_g.predict(pairs), ce_e.predict(pairs, batch_size=32), atol=1e-5)
    assert ce_g._graphed is not None and not ce_g._graphed._broken and len(ce_g._graphed._graphs) >= 3
    big = [["q", d] for d in (cdocs * 3)[:300]]              # more rows than the largest bucket: eager path
    np.testing.assert_allclose(ce_g.predict(big), ce_e.predict(big), atol=1e-5)
    from tristage_rag_amd.retrieval_pipeline import PipelineConfig, RetrievalPipeline
    docs = _corpus(200)
    outs = []
    for graphs in (False, True):
        pc = PipelineConfig(stage1_model="random:tiny", stage2_model="random:tiny", stage3_model="random:tiny",
                            device="cuda", cache_dir=str(tmp_path / "m"), index_dir=str(tmp_path / "i"),
                            log_file=str(tmp_path / f"g{graphs}.log"), stage1_top_k=30, stage2_top_k=10,
                            stage3_top_k=5, stage1_enable_bm25=False, stage1_use_fp16=False,
                            stage2_use_fp16=False, stage3_use_fp16=False, use_hip_graphs=graphs)
        p = RetrievalPipeline(config=pc)
        p.add_documents(docs)
        outs.append([[r["doc_id"] for r in p.search(q)["results"]] for q in ("neural network", "gpu memory index")])
    assert outs[0] == outs[1]


def test_array_path_search_many_on_gpu(tmp_path):
    """search_many with every stage on arrays (HIP index -> token-store MaxSim + device sort -> cross-encoder inputs
    assembled on the GPU from cached token ids -> LeanBertClassifier) against the per-record path of the same
    pipeline and against search(): same records; with BM25 + RRF too; fp32 so that the comparison is tight, then
    bf16 (lean forward vs the transformers module under autocast) at bf16 tolerance."""
    import torch
    from tristage_rag_amd.retrieval_pipeline import PipelineConfig, RetrievalPipeline
    docs = _corpus(600)
    queries = ["neural networks attention", "language retrieval system", "gpu memory index", docs[11], "vector search rank"]
    for bm25 in (False, True):
        pc = PipelineConfig(stage1_model="random:tiny", stage2_model="random:tiny", stage3_model="random:tiny",
                            device="cuda", cache_dir=str(tmp_path / "m"), index_dir=str(tmp_path / "i"),
                            log_file=str(tmp_path / f"a{bm25}.log"), log_level="WARNING", stage1_top_k=200, stage2_top_k=40,
                            stage3_top_k=10, stage1_enable_bm25=bm25, stage1_use_fp16=False, stage2_use_fp16=False,
                            stage3_use_fp16=False, save_intermediate_results=True,
                            stage2_precompute_document_embeddings=True, stage3_cache_document_tokens=True)
        p = RetrievalPipeline(config=pc)
        p.add_documents(docs[:350])
        p.add_documents(docs[350:])
        assert p.stage3._pairs_usable and len(p.stage3._pairs) == len(docs)
        fast = p.search_many(queries)
        assert p._search_many_arrays(queries, 10) is not None
        p.stage3._pairs_usable = False                       # the same pipeline, per-record path
        slow = p.search_many(queries)
        p.stage3._pairs_usable = True
        for a, b, q in zip(fast, slow, queries):
            one = p.search(q)
            for other in (b, one):
                for stage, key in (("stage1_results", "stage1_score"), ("stage2_results", "stage2_score"), ("results", "stage3_score")):
                    ia, ib = [x["doc_id"] for x in a[stage]], [x["doc_id"] for x in other[stage]]
                    sa, sb = np.array([x[key] for x in a[stage]]), np.array([x[key] for x in other[stage]])
                    assert len(ia) == len(ib)
                    np.testing.assert_allclose(sa, sb, atol=1e-4)
                    for x, y, u, v in zip(ia, ib, sa, sb):
                        assert x == y or abs(u - v) < 1e-4
            assert set(a["results"][0]) == set(one["results"][0])
    # bf16: the written-out forward vs the transformers module under autocast, on real assembled batches
    from tristage_rag_amd.encoders import CrossEncoderModel, PairAssembler
    ce = CrossEncoderModel("random:minilm", device="cuda", use_amp=True)
    pa = PairAssembler(ce.tokenizer, 256)
    pa.add_documents(docs)
    pq = torch.arange(3, device="cuda").repeat_interleave(100)
    pd = torch.randint(0, len(docs), (300,), device="cuda")
    plan = pa.plan([pa.ids_of(q) for q in queries[:3]], pq, pd, "cuda")
    enc = pa.batch(plan, torch.arange(300, device="cuda"))
    lean = ce.logits_from_ids(enc)
    assert ce._lean
    ce.lean_forward = False
    ref = ce.logits_from_ids(enc)
    assert float((lean - ref).abs().max()) < 4e-3            # bf16 GEMMs of different shapes (fused QKV)
    text = ce.logits([[queries[int(a)], docs[int(b)]] for a, b in zip(pq.tolist(), pd.tolist())], batch_size=300)
    assert float((text - ref).abs().max()) < 4e-3            # assembled ids == tokenised text pairs
    # the attention kernel's three entries: promised lengths, a prefix mask it verifies itself, and a mask with a hole
    # (not a right-padded batch: torch's masked kernel takes over) — each against the transformers module
    ce.lean_forward = True
    no_len = {k: v for k, v in enc.items() if k != "lengths"}
    assert float((ce.logits_from_ids(no_len) - ref).abs().max()) < 4e-3
    ce._lean.fused_attention = False
    assert float((ce.logits_from_ids(no_len) - ref).abs().max()) < 4e-3
    ce._lean.fused_attention = True
    holed = {k: v.clone() for k, v in no_len.items()}
    holed["attention_mask"][:, 3] = 0
    ce.lean_forward = False
    ref_h = ce.logits_from_ids(holed)
    ce.lean_forward = True
    assert float((ce.logits_from_ids(holed) - ref_h).abs().max()) < 4e-3


@pytest.mark.parametrize("xdt,H,rows", [("bf16", 384, 5000), ("f16", 1024, 777), ("f32", 64, 33), ("bf16", 2048, 65), ("bf16", 772, 100)])
def test_fused_add_layernorm_matches_torch(xdt, H, rows):
    """ts_add_layernorm (residual add + LayerNorm + cast in one pass) against torch: (x + residual) in fp32,
    torch.layer_norm in fp32, .to(16-bit) — fp32 output to 2e-6, the 16-bit output equal up to one rounding step."""
    import torch
    import torch.nn.functional as F
    from tristage_rag_amd.index import add_layernorm
    g = torch.Generator(device="cuda").manual_seed(H + rows)
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[xdt]
    x = (torch.randn((rows, H), generator=g, device="cuda") * 0.7).to(tdt)
    res = torch.randn((rows, H), generator=g, device="cuda") * 2.0 + 0.3
    gamma = torch.rand((H,), generator=g, device="cuda") + 0.5
    beta = torch.randn((H,), generator=g, device="cuda") * 0.1
    for lp in (torch.bfloat16, torch.float16):
        for r in (res, None):
            want = F.layer_norm(x.float() + r if r is not None else x.float(), (H,), gamma, beta, 1e-12)
            y32, ylp = add_layernorm(x, r, gamma, beta, 1e-12, lp_dtype=lp)
            assert y32.dtype == torch.float32 and ylp.dtype == lp
            assert float((y32 - want).abs().max()) < 2e-6 * max(1.0, float(want.abs().max()))
            step = 2.0 ** (-8 if lp == torch.bfloat16 else -11)
            assert float((ylp.float() - want).abs().max()) <= step * float(want.abs().max()) + 1e-6
            assert torch.equal(ylp, y32.to(lp))                   # the 16-bit copy is the rounded fp32 result
    only32, none_lp = add_layernorm(x, res, gamma, beta, 1e-5, lp_dtype=None)
    assert none_lp is None and only32 is not None
    with pytest.raises(ValueError):                                # parameter vectors of another width
        add_layernorm(x, res, gamma[:-4], beta, 1e-5, lp_dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        add_layernorm(x, res.half(), gamma, beta, 1e-5, lp_dtype=torch.bfloat16)
    # 3-D input, as the encoder passes it
    y32, ylp = add_layernorm(x.view(1, rows, H), res.view(1, rows, H), gamma, beta, 1e-12, lp_dtype=torch.bfloat16)
    assert y32.shape == (1, rows, H)


@pytest.mark.parametrize("dt,B,L,nh,dh", [("bf16", 37, 168, 12, 32), ("f16", 9, 256, 6, 64), ("bf16", 5, 33, 2, 32),
                                          ("bf16", 3, 512, 4, 64), ("f16", 4, 1000, 2, 32)])
def test_attention_varlen_matches_torch(dt, B, L, nh, dh):
    """ts_attention_varlen against an fp32 softmax(QK^T/sqrt(dh))V over each sequence's valid tokens (float64-free
    torch reference computed from the same 16-bit q, k, v), and against torch's masked SDPA in the same dtype.
    Tolerance: probabilities and the output are rounded to the 16-bit type (2^-8 bf16 / 2^-11 fp16 relative steps) —
    3 steps of the output scale."""
    import torch
    import torch.nn.functional as F
    from tristage_rag_amd.index import attention_varlen
    g = torch.Generator(device="cuda").manual_seed(B * L + dh)
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    H = nh * dh
    qkv = (torch.randn((B, L, 3 * H), generator=g, device="cuda") * 1.5).to(tdt)
    lens = torch.randint(1, L + 1, (B,), generator=g, device="cuda").to(torch.int32)
    lens[0] = L
    if B > 2:
        lens[1], lens[2] = 1, 32
    sentinel = 7.0
    out = torch.full((B, L, H), sentinel, dtype=tdt, device="cuda")
    got = attention_varlen(qkv, lens, nh, out=out)
    assert got.data_ptr() == out.data_ptr()
    q, k, v = (qkv.view(B, L, 3, nh, dh)[:, :, i].transpose(1, 2) for i in range(3))
    valid = torch.arange(L, device="cuda")[None, :] < lens[:, None]
    s = (q.float() @ k.float().transpose(-1, -2)) * dh ** -0.5
    s = s.masked_fill(~valid[:, None, None, :], float("-inf"))
    want = (torch.softmax(s, -1) @ v.float()).transpose(1, 2).reshape(B, L, H)
    step = 2.0 ** (-8 if dt == "bf16" else -11)
    scale = float(want[valid].abs().max())
    assert float((got.float() - want)[valid].abs().max()) <= 3 * step * scale
    assert bool((got[~valid] == sentinel).all())                   # padded rows untouched
    sd = F.scaled_dot_product_attention(q, k, v, attn_mask=valid[:, None, None, :]).transpose(1, 2).reshape(B, L, H)
    assert float((got.float() - sd.float())[valid].abs().max()) <= 4 * step * scale
    # default output buffer: zeros at the padding
    z = attention_varlen(qkv, lens, nh)
    assert torch.equal(z[valid], got[valid]) and bool((z[~valid] == 0).all())


def test_attention_varlen_argument_errors():
    import torch
    from tristage_rag_amd.index import attention_varlen
    from tristage_rag_amd import _lib
    qkv = torch.zeros((2, 64, 3 * 2 * 16), dtype=torch.bfloat16, device="cuda")
    lens = torch.full((2,), 64, dtype=torch.int32, device="cuda")
    with pytest.raises(_lib.TriStageNativeError):
        attention_varlen(qkv, lens, 2)                              # head dimension 16
    with pytest.raises(ValueError):
        attention_varlen(qkv, lens.long(), 2)
    big = torch.zeros((1, 2048, 3 * 64), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(_lib.TriStageNativeError):
        attention_varlen(big, torch.full((1,), 2048, dtype=torch.int32, device="cuda"), 1)   # K and V^T do not fit LDS


@pytest.mark.parametrize("H,V,shape,types", [(384, 30522, (64, 168), True), (1024, 5000, (7, 33), False), (772, 100, (3, 5), True)])
def test_fused_embed_layernorm_matches_torch(H, V, shape, types):
    """ts_embed_layernorm against the torch ops it replaces: (word[ids] + type[tt]) + pos[p] in fp32, layer_norm, cast."""
    import torch
    import torch.nn.functional as F
    from tristage_rag_amd.index import embed_layernorm
    g = torch.Generator(device="cuda").manual_seed(H + V)
    word = torch.randn((V, H), generator=g, device="cuda")
    pos = torch.randn((514, H), generator=g, device="cuda") * 0.5
    typ = torch.randn((2, H), generator=g, device="cuda") * 0.1
    gamma = torch.rand((H,), generator=g, device="cuda") + 0.5
    beta = torch.randn((H,), generator=g, device="cuda") * 0.1
    ids = torch.randint(0, V, shape, generator=g, device="cuda")
    ids[0, 0], ids[-1, -1] = V - 1, 0
    pid = torch.arange(shape[1], device="cuda")[None, :].expand(*shape) + 2          # a non-contiguous view, like the forward's
    tt = torch.randint(0, 2, shape, generator=g, device="cuda") if types else None
    want = F.layer_norm((word[ids] + typ[tt if tt is not None else torch.zeros_like(ids)]) + pos[pid], (H,), gamma, beta, 1e-12)
    for lp in (torch.bfloat16, torch.float16):
        y32, ylp = embed_layernorm(ids, pid, tt, word, pos, typ, gamma, beta, 1e-12, lp_dtype=lp)
        assert y32.shape == shape + (H,) and ylp.dtype == lp
        assert float((y32 - want).abs().max()) < 2e-6 * max(1.0, float(want.abs().max()))
        assert torch.equal(ylp, y32.to(lp))
    with pytest.raises(ValueError):
        embed_layernorm(ids, pid, tt, word.half(), pos, typ, gamma, beta, 1e-12, lp_dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        embed_layernorm(ids, pid[:, :-1], tt, word, pos, typ, gamma, beta, 1e-12, lp_dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        embed_layernorm(ids, pid, tt, word, pos[:, :-4].contiguous(), typ, gamma, beta, 1e-12, lp_dtype=torch.bfloat16)


def test_stage1_and_stage2_encoders_use_the_written_out_forward_under_amp():
    """SentenceEncoder.encode and ColBERTScorer._forward under bf16 autocast on the GPU run LeanBertEncoder (HIP
    embedding / LayerNorm / attention kernels between the GEMMs) for BERT-family models; against the transformers
    module under the same autocast: embeddings to 2e-2 absolute on unit vectors' raw hidden scale / cosine > 0.999."""
    import torch
    from tristage_rag_amd.encoders import SentenceEncoder
    from tristage_rag_amd.stage2_rescorer import ColBERTScorer, Stage2Config
    rng = np.random.default_rng(3)
    vocab = [f"w{i}" for i in range(500)]
    texts = [" ".join(rng.choice(vocab, size=int(n))) for n in rng.integers(1, 120, size=150)] + ["", "w1"]
    enc = SentenceEncoder("random:bert", device="cuda")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lean = enc.encode(texts, batch_size=64, convert_to_tensor=True, normalize_embeddings=True)
        assert enc.model.__dict__.get("_ts_lean_encoders", {}).get(torch.bfloat16)
        enc.lean_forward = False
        ref = enc.encode(texts, batch_size=64, convert_to_tensor=True, normalize_embeddings=True)
    assert float((lean * ref).sum(1).min()) > 0.999
    plain = enc.encode(texts[:5], convert_to_tensor=True)               # no autocast: the module's fp32 forward
    assert plain.dtype == torch.float32
    s2 = ColBERTScorer(Stage2Config(model_name="random:minilm", device="cuda", use_fp16=True, max_seq_length=64))
    batch = s2._tokenize_batch(texts[:40])
    a = s2._forward(batch)
    assert s2.model.__dict__.get("_ts_lean_encoders", {}).get(torch.bfloat16)
    s2.lean_forward = False
    b = s2._forward(batch)
    valid = batch["attention_mask"].bool()
    assert a.dtype == b.dtype == torch.float32
    assert float((a[valid] - b[valid]).abs().max()) < 0.03 * float(b[valid].abs().max())


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_rope_geglu_prenorm_and_window_kernels_match_torch(dt):
    """The four kernels of the ModernBERT forward against the torch ops they replace."""
    import torch
    import torch.nn.functional as F
    from tristage_rag_amd.index import add_layernorm, attention_varlen, geglu, rope_inplace
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    step = 2.0 ** (-8 if dt == "bf16" else -11)
    g = torch.Generator(device="cuda").manual_seed(11)
    B, L, nh, dh = 6, 200, 3, 64
    H = nh * dh
    # rotary embedding, transformers' formula in fp32, one rounding: bit-equal
    qkv = torch.randn((B, L, 3 * H), generator=g, device="cuda").to(tdt)
    ang = torch.rand((L, dh // 2), generator=g, device="cuda") * 6.28
    cos, sin = torch.cat((ang.cos(), ang.cos()), -1).contiguous(), torch.cat((ang.sin(), ang.sin()), -1).contiguous()
    v5 = qkv.view(B, L, 3, nh, dh)
    half = lambda z: torch.cat((-z[..., dh // 2:], z[..., : dh // 2]), dim=-1)
    want = qkv.clone().view(B, L, 3, nh, dh)
    for j in (0, 1):
        z = v5[:, :, j].float()
        want[:, :, j] = ((z * cos[None, :, None]) + (half(z) * sin[None, :, None])).to(tdt)
    got = rope_inplace(qkv.clone(), cos, sin, nh).view(B, L, 3, nh, dh)
    assert torch.equal(got, want)
    # gated GELU: two roundings like the two torch ops
    u = (torch.randn((B, L, 2 * 1152), generator=g, device="cuda") * 2).to(tdt)
    a, b = u.chunk(2, dim=-1)
    ref = F.gelu(a) * b
    out = geglu(u)
    assert out.shape == ref.shape and float((out.float() - ref.float()).abs().max()) <= 2 * step * float(ref.abs().max())
    assert float((out != ref).float().mean()) < 0.01               # (erf of the two libraries: rare last-bit differences)
    # pre-LN: residual stream out, normalised 16-bit copy, no bias
    x = torch.randn((B * L, 768), generator=g, device="cuda").to(tdt)
    res = torch.randn((B * L, 768), generator=g, device="cuda")
    gamma = torch.rand((768,), generator=g, device="cuda") + 0.5
    s32, ylp = add_layernorm(x, res, gamma, None, 1e-5, lp_dtype=tdt, prenorm=True)
    assert torch.equal(s32, x.float() + res)
    assert torch.equal(ylp, F.layer_norm(x.float() + res, (768,), gamma, None, 1e-5).to(tdt)) or \
        float((ylp.float() - F.layer_norm(x.float() + res, (768,), gamma, None, 1e-5)).abs().max()) <= step * 8
    y32, _ = add_layernorm(x, res, gamma, None, 1e-5, lp_dtype=tdt)
    assert float((y32 - F.layer_norm(x.float() + res, (768,), gamma, None, 1e-5)).abs().max()) < 1e-5
    # local attention: |q - k| <= window, with padding
    lens = torch.tensor([L, 1, 64, 66, 131, 199], dtype=torch.int32, device="cuda")
    q, k, v = (v5[:, :, j].transpose(1, 2) for j in range(3))
    valid = torch.arange(L, device="cuda")[None, :] < lens[:, None]
    t = torch.arange(L, device="cuda")
    for window in (64, 10, 500):
        near = (t[:, None] - t[None, :]).abs() <= window
        s = (q.float() @ k.float().transpose(-1, -2)) * dh ** -0.5
        s = s.masked_fill(~(valid[:, None, None, :] & near[None, None]), float("-inf"))
        want_a = (torch.softmax(s, -1) @ v.float()).transpose(1, 2).reshape(B, L, H)
        got_a = attention_varlen(qkv, lens, nh, window=window)
        assert float((got_a.float() - want_a)[valid].abs().max()) <= 3 * step * float(want_a[valid].abs().max())
        # rotary embedding applied inside the attention kernel == rope_inplace first, then the plain kernel: bit for bit
        rotated = rope_inplace(qkv.clone(), cos, sin, nh)
        two_pass = attention_varlen(rotated, lens, nh, window=window)
        before = qkv.clone()
        fused = attention_varlen(qkv, lens, nh, window=window, rope=(cos, sin))
        assert torch.equal(fused, two_pass) and torch.equal(qkv, before)       # (qkv itself is left alone)
    for nh2, dh2 in ((4, 32), (1, 64)):                                        # the other head dimension, short rows
        L2 = 70
        qkv2 = torch.randn((3, L2, 3 * nh2 * dh2), generator=g, device="cuda").to(tdt)
        ang2 = torch.rand((L2, dh2 // 2), generator=g, device="cuda") * 6.28
        c2_, s2_ = torch.cat((ang2.cos(), ang2.cos()), -1).contiguous(), torch.cat((ang2.sin(), ang2.sin()), -1).contiguous()
        lens2 = torch.tensor([L2, 33, 1], dtype=torch.int32, device="cuda")
        a1 = attention_varlen(rope_inplace(qkv2.clone(), c2_, s2_, nh2), lens2, nh2)
        a2 = attention_varlen(qkv2, lens2, nh2, rope=(c2_, s2_))
        assert torch.equal(a1, a2)
    with pytest.raises(ValueError):
        attention_varlen(qkv, lens, nh, rope=(cos[:, :-8].contiguous(), sin))


def test_lean_modernbert_on_gpu_matches_module_under_autocast():
    """ColBERTScorer._forward with the reference's default stage-2 architecture (ModernBERT-base shape) under bf16
    AMP runs LeanModernBertEncoder with the HIP kernels; against the transformers module under the same autocast."""
    import torch
    from tristage_rag_amd.stage2_rescorer import ColBERTScorer, Stage2Config
    from tristage_rag_amd.encoders import LeanModernBertEncoder
    rng = np.random.default_rng(5)
    vocab = [f"w{i}" for i in range(500)]
    texts = [" ".join(rng.choice(vocab, size=int(n))) for n in rng.integers(1, 180, size=40)] + ["", "w1"]
    s2 = ColBERTScorer(Stage2Config(model_name="random:modernbert", device="cuda", use_fp16=True, max_seq_length=192))
    batch = s2._tokenize_batch(texts)
    assert batch["input_ids"].shape[1] > 130                        # beyond the local window
    a = s2._forward(batch)
    lean = s2.model.__dict__.get("_ts_lean_encoders", {}).get(torch.bfloat16)
    assert isinstance(lean, LeanModernBertEncoder)
    s2.lean_forward = False
    b = s2._forward(batch)
    valid = batch["attention_mask"].bool()
    assert a.dtype == b.dtype == torch.float32
    err = float((a[valid] - b[valid]).abs().max()) / float(b[valid].abs().max())
    assert err < 0.04, err                                          # 22 layers of bf16 GEMMs in different shapes
    cosine = torch.nn.functional.cosine_similarity(a[valid], b[valid], dim=-1)
    assert float(cosine.min()) > 0.998
    # the torch fallback of the same class (holes in the mask) agrees too
    holed = {k: v.clone() for k, v in batch.items() if k not in ("lengths", "lengths_host")}   # (no promise of a prefix mask any more)
    holed["attention_mask"][:, 2] = 0
    s2.lean_forward = True
    c = s2._forward(holed)
    s2.lean_forward = False
    d = s2._forward(holed)
    hv = holed["attention_mask"].bool()
    assert float((c[hv] - d[hv]).abs().max()) / float(d[hv].abs().max()) < 0.04


def test_attention_varlen_random_shapes_sweep():
    """Seeded sweep over small / ragged / edge shapes of ts_attention_varlen (L = 1, 31..33, lengths 0 and L, both head
    dimensions, windows smaller and larger than L) against the fp32 reference: no launch fails, padded rows untouched,
    results within three 16-bit steps."""
    import torch
    from tristage_rag_amd.index import attention_varlen
    rng = np.random.default_rng(int(os.environ.get("TS_TEST_SEED", "20240611")))
    g = torch.Generator(device="cuda").manual_seed(17)
    shapes = [(1, 1, 1, 32), (2, 31, 2, 64), (3, 32, 1, 32), (3, 33, 3, 64), (5, 65, 2, 32), (2, 575, 1, 64), (1, 1119, 1, 32)]
    for _ in range(12):
        shapes.append((int(rng.integers(1, 40)), int(rng.integers(1, 300)), int(rng.integers(1, 5)), int(rng.choice([32, 64]))))
    for B, L, nh, dh in shapes:
        for tdt, step in ((torch.bfloat16, 2.0 ** -8), (torch.float16, 2.0 ** -11)):
            H = nh * dh
            qkv = torch.randn((B, L, 3 * H), generator=g, device="cuda").to(tdt)
            lens_h = rng.integers(0, L + 1, size=B)
            lens_h[0] = L
            if B > 1:
                lens_h[1] = 0                                         # an empty sequence: nothing written
            lens = torch.as_tensor(lens_h, dtype=torch.int32, device="cuda")
            window = int(rng.choice([0, 0, 1, 7, 64, 5000]))
            out = torch.full((B, L, H), 3.0, dtype=tdt, device="cuda")
            attention_varlen(qkv, lens, nh, out=out, window=window)
            q, k, v = (qkv.view(B, L, 3, nh, dh)[:, :, i].transpose(1, 2).float() for i in range(3))
            t = torch.arange(L, device="cuda")
            valid = t[None, :] < lens[:, None]
            ok = valid[:, None, None, :] & (((t[:, None] - t[None, :]).abs() <= window) if window else torch.ones((L, L), dtype=torch.bool, device="cuda"))[None, None]
            s = (q @ k.transpose(-1, -2)) * dh ** -0.5
            s = s.masked_fill(~ok, float("-inf"))
            want = torch.nan_to_num(torch.softmax(s, -1), nan=0.0) @ v
            want = want.transpose(1, 2).reshape(B, L, H)
            if bool(valid.any()):
                err = float((out.float() - want)[valid].abs().max())
                assert err <= 3 * step * max(1.0, float(want[valid].abs().max())), (B, L, nh, dh, window, err)
            assert bool((out[~valid] == 3.0).all()), (B, L, nh, dh)


@pytest.mark.parametrize("amp", ["bf16", "fp16"])
def test_amp_dtype_option_runs_the_forwards_and_the_token_store_in_that_type(tmp_path, amp):
    """PipelineConfig.amp_dtype: "fp16" is the reference's own GPU precision (torch.cuda.amp.autocast), "bf16" the
    BASELINE configs[2] setting.  Either way the written-out forwards and the token store use that 16-bit type, the
    array path and the record path agree, and the top results overlap with the fp32 (parity-setting) pipeline's."""
    import torch
    from tristage_rag_amd.retrieval_pipeline import PipelineConfig, RetrievalPipeline
    docs = _corpus(700)
    queries = ["neural network attention", "gpu memory index", docs[5], "language model retrieval"]

    def build(name, **kw):
        pc = PipelineConfig(stage1_model="random:minilm", stage2_model="random:modernbert:64:4:2", stage3_model="random:minilm",
                            device="cuda", cache_dir=str(tmp_path / "m"), index_dir=str(tmp_path / name),
                            log_file=str(tmp_path / f"{name}.log"), log_level="WARNING", stage1_top_k=200, stage2_top_k=40,
                            stage3_top_k=10, stage1_enable_bm25=False, stage2_precompute_document_embeddings=True,
                            stage3_cache_document_tokens=True, **kw)
        p = RetrievalPipeline(config=pc)
        p.add_documents(docs)
        return p
    want = torch.bfloat16 if amp == "bf16" else torch.float16
    p = build("amp", amp_dtype=amp)
    assert p.stage2.token_store.data.dtype == want
    res = p.search_many(queries)
    assert all(len(r["results"]) == 10 for r in res)
    for m in (p.stage1.model.model, p.stage2.model):
        assert want in m.__dict__.get("_ts_lean_encoders", {})            # the written-out forward in that type
    assert p.stage3.model._lean and p.stage3.model._lean.cd == want
    one = [p.search(q) for q in queries]                                 # array path for one query
    p.config.search_on_arrays = False
    rec = [p.search(q) for q in queries]                                 # per-record path
    for a, b, c in zip(res, one, rec):
        ia, ib, ic = ([x["doc_id"] for x in r["results"]] for r in (a, b, c))
        assert len(set(ia) & set(ib)) >= 9 and len(set(ia) & set(ic)) >= 8   # batch-padding / tokenised-vs-assembled noise
    ref = build("f32", stage1_use_fp16=False, stage2_use_fp16=False, stage3_use_fp16=False)
    for a, b in zip(res, ref.search_many(queries)):
        top_ref = [x["doc_id"] for x in b["stage1_results"][:5]] if b["stage1_results"] else None
        ia, ib = ([x["doc_id"] for x in r["results"]] for r in (a, b))
        assert len(set(ia) & set(ib)) >= 6, (amp, ia, ib)                 # 16-bit forwards of random-init models vs fp32
    with pytest.raises(ValueError):
        build("bad", amp_dtype="fp8")


def test_packed_batches_equal_padded_batches():
    """PACKED batches (the sequences' tokens concatenated; ts_attention_varlen with offsets, LayerNorm / GEMMs / GELU on
    sum(lengths) rows): the attention kernel gives bit for bit what it gives on the padded layout, PairAssembler's
    packed ids are the padded ids without their padding, and the classifier's logits agree with the padded forward's
    (other GEMM row counts: bf16 accumulation-order noise) and with the transformers module under autocast."""
    import torch
    from tristage_rag_amd.encoders import CrossEncoderModel, PairAssembler
    from tristage_rag_amd.index import attention_varlen
    g = torch.Generator(device="cuda").manual_seed(23)
    B, L, nh, dh = 9, 150, 3, 32
    H = nh * dh
    lens = torch.tensor([L, 1, 33, 64, 150, 17, 96, 2, 128], dtype=torch.int32, device="cuda")
    qkv = torch.randn((B, L, 3 * H), generator=g, device="cuda").to(torch.bfloat16)
    valid = torch.arange(L, device="cuda")[None, :] < lens[:, None]
    offs = (torch.cumsum(lens, 0) - lens).to(torch.int32)
    padded = attention_varlen(qkv, lens, nh)
    packed = attention_varlen(qkv[valid].contiguous(), lens, nh, offs=offs, max_len=L)
    assert packed.shape == (int(lens.sum()), H) and torch.equal(packed, padded[valid])
    for window in (0, 20):
        ang = torch.rand((L, dh // 2), generator=g, device="cuda") * 6.28
        cos, sin = torch.cat((ang.cos(), ang.cos()), -1).contiguous(), torch.cat((ang.sin(), ang.sin()), -1).contiguous()
        a = attention_varlen(qkv, lens, nh, window=window, rope=(cos, sin))
        b = attention_varlen(qkv[valid].contiguous(), lens, nh, window=window, rope=(cos, sin), offs=offs, max_len=L)
        assert torch.equal(b, a[valid])
    with pytest.raises(ValueError):
        attention_varlen(qkv[valid].contiguous(), lens, nh, offs=offs)                 # max_len missing
    # the assembled pairs
    docs = _corpus(400)
    queries = ["neural network attention", "gpu memory index retrieval system", "token"]
    # (the third: the reference's own GPU AMP type, fp16 — the feed-forward kernel computes its GELU there, no table)
    for spec, amp in (("random:minilm", torch.bfloat16), ("random:xlmr-large:64:2:2", torch.bfloat16), ("random:minilm", torch.float16)):
        ce = CrossEncoderModel(spec, device="cuda", use_amp=True, amp_dtype=amp)
        pa = PairAssembler(ce.tokenizer, 64)
        pa.add_documents(docs)
        n = 500
        pq = torch.randint(0, 3, (n,), generator=g, device="cuda")
        pd = torch.randint(0, len(docs), (n,), generator=g, device="cuda")
        plan = pa.plan([pa.ids_of(q) for q in queries], pq, pd, "cuda")
        sel = torch.arange(n, device="cuda")
        width = int(plan["total"].max())
        enc = pa.batch(plan, sel, width=width)
        pk = pa.batch_packed(plan, sel, int(plan["total"].sum()), width)
        m = enc["attention_mask"].bool()
        assert torch.equal(pk["input_ids"], enc["input_ids"][m]) and torch.equal(pk["lengths"], enc["lengths"])
        if pk["token_type_ids"] is not None:
            assert torch.equal(pk["token_type_ids"], enc["token_type_ids"][m])
        assert torch.equal(pk["positions"], torch.arange(width, device="cuda")[None, :].expand(n, width)[m])
        assert ce.packed_ok(width, n)
        a = ce.logits_from_ids(enc)
        b = ce.logits_from_ids(pk)
        assert a.shape == b.shape and float((a - b).abs().max()) < 4e-3
        # projection + residual + LayerNorm as one kernel, and the GELU folded into the down kernel's row staging, are
        # re-arrangements of the same arithmetic: the SAME logits as the kernel-per-step forward, bit for bit
        lean = ce._lean_model()
        if lean.layers[0]["t2_ln"] is not None and int(pk["input_ids"].shape[0]) >= lean.min_linear_rows:
            for fo, gd, fm in ((False, False, False), (True, False, False), (True, True, False)):   # (b, a: everything on, the one-kernel feed-forward block included)
                lean.fused_output_layernorm, lean.gelu_in_down, lean.fused_mlp = fo, gd, fm
                assert torch.equal(ce.logits_from_ids(pk), b) and torch.equal(ce.logits_from_ids(enc), a)
            lean.fused_output_layernorm, lean.gelu_in_down, lean.fused_mlp = True, True, True
        ce.lean_forward = False
        ref = ce.logits_from_ids(enc)
        assert float((b - ref).abs().max()) < 4e-3


def test_projection_kernels_are_repeatable_bit_for_bit():
    """The streamed-weight kernels hand data between waves through LDS with counters (ffn_stream_kernel's slots), bare
    barriers with rings in flight (proj_ln_kernel, mlp_ln_kernel) and an LDS table built at run time: 200 launches of each
    on the same input must give the same bits every time (a lost ordering shows up as a rare different value), at a size
    with several tiles per compute unit."""
    import torch
    from tristage_rag_amd.index import TiledLinear, mlp_add_layernorm
    g = torch.Generator(device="cuda").manual_seed(5)
    M, H, I = 70001, 384, 1536
    x = (torch.randn((M, H), generator=g, device="cuda") * 0.8).to(torch.bfloat16)
    w1 = (torch.randn((I, H), generator=g, device="cuda") * 0.06).to(torch.bfloat16)
    b1 = (torch.randn((I,), generator=g, device="cuda") * 0.1).to(torch.bfloat16)
    w2 = (torch.randn((H, I), generator=g, device="cuda") * 0.03).to(torch.bfloat16)
    b2 = (torch.randn((H,), generator=g, device="cuda") * 0.1).to(torch.bfloat16)
    res = torch.randn((M, H), generator=g, device="cuda")
    gamma = 1.0 + 0.1 * torch.randn((H,), generator=g, device="cuda")
    beta = 0.1 * torch.randn((H,), generator=g, device="cuda")
    up, down = TiledLinear(w1, b1), TiledLinear(w2, b2, with_layernorm=True)
    first = None
    for _ in range(200):
        u = up(x, gelu=True)
        d32, dlp = down.add_layernorm(u, res, gamma, beta, 1e-12)
        m32, mlp = mlp_add_layernorm(up, down, x, res, gamma, beta, 1e-12)
        now = (u, d32, dlp, m32, mlp)
        if first is None:
            first = now
            assert torch.equal(d32, m32) and torch.equal(dlp, mlp)
        else:
            assert all(torch.equal(a, b) for a, b in zip(first, now))


def test_rrf_fusion_on_the_gpu_is_bit_identical_to_the_host_code():
    """Stage1Retriever._fuse_rrf_device (one stable float64 sort for a whole query batch) against _fuse_arrays per query
    (itself pinned to the reference's dictionary code on the CPU): same ids, same float64 scores, same tie order —
    overlapping, disjoint and identical dense / BM25 lists."""
    import torch
    from tristage_rag_amd.stage1_retriever import Stage1Config, Stage1Retriever
    s1 = object.__new__(Stage1Retriever)
    s1.config = Stage1Config()
    rng = np.random.default_rng(8)
    for k1, k2, top_k, universe in ((1000, 300, 1000, 3633), (50, 50, 50, 60), (200, 300, 120, 100000), (64, 10, 64, 64)):
        B = 17
        dense = np.stack([rng.permutation(universe)[:k1] for _ in range(B)]).astype(np.int64)
        bm = np.stack([rng.permutation(universe)[:k2] for _ in range(B)]).astype(np.int64)
        bm[0, : min(k1, k2)] = dense[0, : min(k1, k2)]                        # identical prefixes
        dscores = np.sort(rng.random((B, k1)).astype(np.float32), axis=1)[:, ::-1].copy()
        gi, gs = s1._fuse_rrf_device(torch.from_numpy(dense).cuda(), bm, top_k)
        assert gi.dtype == torch.int64 and gs.dtype == torch.float64 and gi.shape == (B, top_k)
        gi, gs = gi.cpu().numpy(), gs.cpu().numpy()
        for q in range(B):
            hi, hs = s1._fuse_arrays(dense[q], dscores[q], (bm[q], np.ones(k2)))
            assert np.array_equal(gi[q], hi[:top_k]) and np.array_equal(gs[q], hs[:top_k])


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_tiled_linear_matches_torch(dt):
    """index.TiledLinear (ts_linear_tile_weight + ts_linear_act: the weight streamed from L2, the rows of x in LDS) against
    F.linear and F.gelu(F.linear): within one 16-bit step of the result (accumulation order, and the erf of the GELU
    epilogue is a 1.5e-7 approximation), ragged row counts, with and without bias, the shapes of the encoders."""
    import torch
    import torch.nn.functional as F
    from tristage_rag_amd.index import TiledLinear
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    step = 2.0 ** (-8 if dt == "bf16" else -11)
    g = torch.Generator(device="cuda").manual_seed(31)
    # (the last four: more tiles than compute units — persistent workgroups take several —, an odd number of 32-feature
    # blocks above 8 (the two compute waves a storer serves own different numbers of blocks), and reduction dimensions whose
    # next image does not fit the storers' registers (64- and 32-row tiles))
    for M, K, N in ((5000, 384, 1152), (4097, 384, 384), (1, 384, 1536), (777, 256, 1024), (100, 256, 256), (300, 128, 32),
                    (60000, 384, 288), (50011, 128, 352), (40000, 768, 64), (20000, 1536, 96)):
        x = (torch.randn((M, K), generator=g, device="cuda") * 0.8).to(tdt)
        w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(tdt)
        b = (torch.randn((N,), generator=g, device="cuda") * 0.1).to(tdt)
        for bias in (b, None):
            lin = TiledLinear(w, bias, with_layernorm=not TiledLinear.usable(N, K))   # (K > 384: admitted as the LayerNorm kernel's weight)
            ref = F.linear(x, w, bias)
            got = lin(x)
            assert got.shape == ref.shape and got.dtype == tdt
            assert float((got.float() - ref.float()).abs().max()) <= 2 * step * max(1.0, float(ref.abs().max()))
            refg = F.gelu(ref)
            gotg = lin(x, gelu=True)
            assert float((gotg.float() - refg.float()).abs().max()) <= 2 * step * max(1.0, float(refg.abs().max()))
        x3 = x.view(1, M, K)                                                  # leading dimensions are kept
        assert lin(x3).shape == (1, M, N)
    assert not TiledLinear.usable(384, 1536) and not TiledLinear.usable(100, 384) and not TiledLinear.usable(3072, 768)
    with pytest.raises(ValueError):
        TiledLinear(torch.zeros((384, 1536), dtype=tdt, device="cuda"))       # K too long: the library GEMM's case
    with pytest.raises(ValueError):
        lin(x.float())


@pytest.mark.gpu
@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_tiled_linear_add_layernorm_equals_the_two_kernels(dt):
    """TiledLinear.add_layernorm (ts_linear_add_layernorm: BertSelfOutput / BertOutput as one kernel — the rows' reduction
    in 384-wide chunks through a double-buffered LDS image, the projection's output staged in LDS, LayerNorm rows by the
    same routine as ts_add_layernorm) against ts_linear_act followed by ts_add_layernorm: the SAME bits (fp32 stream and
    16-bit copy), and against torch in fp32 within the 16-bit rounding of the projection.  Ragged row counts (the last
    workgroup's tile is partial), one and four chunks, N below 384, no residual / no beta / no fp32 output."""
    import torch
    import torch.nn.functional as F
    from tristage_rag_amd.index import TiledLinear, add_layernorm
    tdt = torch.bfloat16 if dt == "bf16" else torch.float16
    step = 2.0 ** (-8 if dt == "bf16" else -11)
    g = torch.Generator(device="cuda").manual_seed(47)
    for M, K, N in ((5000, 384, 384), (4097, 1536, 384), (1, 384, 384), (95, 768, 256), (97, 384, 32), (20000, 1536, 384)):
        x = (torch.randn((M, K), generator=g, device="cuda") * 0.8).to(tdt)
        w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(tdt)
        b = (torch.randn((N,), generator=g, device="cuda") * 0.1).to(tdt)
        res = torch.randn((M, N), generator=g, device="cuda")
        gamma = 1.0 + 0.1 * torch.randn((N,), generator=g, device="cuda")
        beta = 0.1 * torch.randn((N,), generator=g, device="cuda")
        lin = TiledLinear(w, b, with_layernorm=True)
        for r, bt in ((res, beta), (None, beta), (res, None)):
            y32, ylp = lin.add_layernorm(x, r, gamma, bt, 1e-12)
            assert y32.shape == (M, N) and y32.dtype == torch.float32 and ylp.dtype == tdt
            # the two-kernel path on the same tiled weight (ts_linear_act takes any K that is a multiple of 128)
            o = lin(x)
            e32, elp = add_layernorm(o, r, gamma, bt, 1e-12, lp_dtype=tdt)
            if N > 128:
                assert torch.equal(y32, e32), (M, K, N, float((y32 - e32).abs().max()))
                assert torch.equal(ylp, elp)
            else:   # (ts_add_layernorm keeps ONE chunk per lane at N <= 128 and the compiler contracts that instance differently: 1e-6)
                assert float((y32 - e32).abs().max()) <= 2e-6
            # torch, fp32 end to end: the projection's output is rounded to 16 bits once on our side
            ref = F.layer_norm(F.linear(x.float(), w.float(), b.float()) + (r if r is not None else 0.0), (N,), gamma, bt, 1e-12)
            assert float((y32 - ref).abs().max()) <= 12 * step * max(1.0, float(ref.abs().max()))
        only_lp = lin.add_layernorm(x, res, gamma, beta, 1e-12, want_f32=False)
        assert only_lp[0] is None and torch.equal(only_lp[1], lin.add_layernorm(x, res, gamma, beta, 1e-12)[1])
        x3 = x.view(1, M, K)
        assert lin.add_layernorm(x3, res.view(1, M, N), gamma, beta, 1e-12)[0].shape == (1, M, N)
    # BertIntermediate's GELU folded into BertOutput: the up projection WITHOUT its activation, the down kernel applies the
    # erf GELU (and its rounding) while it stages its rows == the up projection with the GELU in its epilogue, bit for bit
    for M, H in ((5000, 384), (97, 384), (1000, 256)):
        x0 = (torch.randn((M, H), generator=g, device="cuda") * 0.8).to(tdt)
        w1 = (torch.randn((1536, H), generator=g, device="cuda") * 0.06).to(tdt)
        b1 = (torch.randn((1536,), generator=g, device="cuda") * 0.1).to(tdt)
        w2 = (torch.randn((H, 1536), generator=g, device="cuda") * 0.03).to(tdt)
        b2 = (torch.randn((H,), generator=g, device="cuda") * 0.1).to(tdt)
        res = torch.randn((M, H), generator=g, device="cuda")
        gamma = 1.0 + 0.1 * torch.randn((H,), generator=g, device="cuda")
        beta = 0.1 * torch.randn((H,), generator=g, device="cuda")
        up, down = TiledLinear(w1, b1), TiledLinear(w2, b2, with_layernorm=True)
        a32, alp = down.add_layernorm(up(x0), res, gamma, beta, 1e-12, gelu_input=True)
        e32, elp = down.add_layernorm(up(x0, gelu=True), res, gamma, beta, 1e-12)
        assert torch.equal(a32, e32) and torch.equal(alp, elp)
        # ... and the whole block as ONE kernel (ts_mlp_add_layernorm: the intermediate never leaves the CU), hidden size 384
        from tristage_rag_amd.index import mlp_add_layernorm, mlp_usable
        assert mlp_usable(up, down) == (H == 384)
        if H == 384:
            for r, bt in ((res, beta), (None, None)):
                m32, mlp = mlp_add_layernorm(up, down, x0, r, gamma, bt, 1e-12)
                w32, wlp = down.add_layernorm(up(x0, gelu=True), r, gamma, bt, 1e-12)
                assert torch.equal(m32, w32) and torch.equal(mlp, wlp)
            nb_up, nb_down = TiledLinear(w1, None), TiledLinear(w2, None, with_layernorm=True)       # no biases
            assert torch.equal(mlp_add_layernorm(nb_up, nb_down, x0, res, gamma, beta, 1e-12)[0],
                               nb_down.add_layernorm(nb_up(x0, gelu=True), res, gamma, beta, 1e-12)[0])
            half_up, half_down = TiledLinear(w1[:768].contiguous(), b1[:768].contiguous()), TiledLinear(w2[:, :768].contiguous(), b2, with_layernorm=True)
            assert torch.equal(mlp_add_layernorm(half_up, half_down, x0, res, gamma, beta, 1e-12, want_f32=False)[1],    # I = 768: two chunks
                               half_down.add_layernorm(half_up(x0, gelu=True), res, gamma, beta, 1e-12)[1])
        else:
            with pytest.raises(ValueError):
                mlp_add_layernorm(up, down, x0, res, gamma, beta, 1e-12)
        ref = F.layer_norm(F.linear(F.gelu(F.linear(x0.float(), w1.float(), b1.float())), w2.float(), b2.float()) + res, (H,), gamma, beta, 1e-12)
        assert float((a32 - ref).abs().max()) <= 16 * step * max(1.0, float(ref.abs().max()))
    assert TiledLinear.usable_with_layernorm(384, 1536) and not TiledLinear.usable_with_layernorm(768, 768)
    assert not TiledLinear.usable_with_layernorm(384, 512)
    with pytest.raises(ValueError):
        TiledLinear(torch.zeros((768, 768), dtype=tdt, device="cuda"), with_layernorm=True)
    with pytest.raises(ValueError):
        lin.add_layernorm(x, res.to(tdt), gamma, beta, 1e-12)                  # the residual stream is fp32
