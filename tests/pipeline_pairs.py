"""Test infrastructure shared by the GPU pipeline / adapter tests: the SAME randomly initialised models
built twice — once on the real HIP entry points (FlatIPIndex, ts_maxsim*, GPU BM25) and once on the CPU
with the oracle-backed doubles — plus the comparison rule for their result records."""
import numpy as np

from doubles import OracleIndex, oracle_maxsim, oracle_maxsim_indexed, oracle_maxsim_indexed_batch

WORDS = ("neural network attention transformer language retrieval index vector query document "
         "learning model data system search rank score token embedding gpu memory").split()


def synth_docs(n, seed=3, lo=4, hi=30, vocab=None, tag=True):
    rng = np.random.default_rng(seed)
    vocab = vocab or WORDS
    return [" ".join(rng.choice(vocab, size=int(rng.integers(lo, hi)))) + (f" doc{i}" if tag else "")
            for i in range(n)]


def pipeline_config(device, tmp_path, name, models=("random:tiny",) * 3, **cfg):
    from tristage_rag_amd.retrieval_pipeline import PipelineConfig
    base = dict(stage1_model=models[0], stage2_model=models[1], stage3_model=models[2], device=device,
                cache_dir=str(tmp_path / "m"), index_dir=str(tmp_path / "i"),
                log_file=str(tmp_path / f"{name}.log"), log_level="WARNING",
                stage1_top_k=40, stage2_top_k=15, stage3_top_k=5,
                stage1_use_fp16=False, stage2_use_fp16=False, stage3_use_fp16=False,
                save_intermediate_results=True)
    base.update(cfg)
    return PipelineConfig(**base)


def gpu_pipeline(tmp_path, name="gpu", models=("random:tiny",) * 3, **cfg):
    from tristage_rag_amd.retrieval_pipeline import RetrievalPipeline
    p = RetrievalPipeline(config=pipeline_config("cuda", tmp_path, name, models, **cfg))
    p.initialize_stages()
    return p


def cpu_pipeline(tmp_path, name="cpu", models=("random:tiny",) * 3, index_dtype="f32", **cfg):
    """The same pipeline on the CPU: encoders in fp32, index / MaxSim replaced by the oracle-backed doubles."""
    from tristage_rag_amd.encoders import SentenceEncoder
    from tristage_rag_amd.retrieval_pipeline import RetrievalPipeline
    from tristage_rag_amd.stage1_retriever import Stage1Config, Stage1Retriever
    from tristage_rag_amd.stage2_rescorer import ColBERTScorer, Stage2Config
    from tristage_rag_amd.stage3_reranker import AdaptiveCrossEncoderReranker, Stage3Config
    pc = pipeline_config("cpu", tmp_path, name, models, **cfg)
    p = RetrievalPipeline(config=pc)
    p.stage1 = Stage1Retriever(Stage1Config(model_name=models[0], device="cpu", cache_dir=pc.cache_dir,
                                            index_dir=pc.index_dir, top_k_candidates=pc.stage1_top_k,
                                            batch_size=pc.stage1_batch_size, enable_bm25=pc.stage1_enable_bm25,
                                            bm25_top_k=pc.stage1_bm25_top_k, fusion_method=pc.stage1_fusion_method,
                                            use_fp16=False, bm25_on_gpu=False),
                               model=SentenceEncoder(models[0], device="cpu"),
                               index_factory=lambda d: OracleIndex(d, dtype=index_dtype))
    p.stage2 = ColBERTScorer(Stage2Config(model_name=models[1], device="cpu", top_k_candidates=pc.stage2_top_k,
                                          max_seq_length=pc.stage2_max_seq_length, batch_size=pc.stage2_batch_size,
                                          use_fp16=False, scoring_method=pc.stage2_scoring_method,
                                          precompute_document_embeddings=pc.stage2_precompute_document_embeddings),
                             maxsim_fn=oracle_maxsim, maxsim_indexed_fn=oracle_maxsim_indexed,
                             maxsim_indexed_batch_fn=oracle_maxsim_indexed_batch)
    p.stage3 = AdaptiveCrossEncoderReranker(Stage3Config(model_name=models[2], device="cpu",
                                                         max_length=pc.stage3_max_length, batch_size=pc.stage3_batch_size,
                                                         top_k_final=pc.stage3_top_k, use_fp16=False))
    return p


def assert_same_ranking(ids_a, scores_a, ids_b, scores_b, atol=1e-3, tie=1e-4, what=""):
    """north_star bar: same ids in the same order, scores within `atol`; two entries may trade places only
    when their scores are within `tie` of each other (GPU vs CPU forward noise on near-ties)."""
    ids_a, ids_b = list(ids_a), list(ids_b)
    sa, sb = np.asarray(scores_a, dtype=np.float64), np.asarray(scores_b, dtype=np.float64)
    assert len(ids_a) == len(ids_b), (what, len(ids_a), len(ids_b))
    if not len(ids_a):
        return
    np.testing.assert_allclose(sa, sb, atol=atol, rtol=0, err_msg=what)
    for x, y, u, v in zip(ids_a, ids_b, sa, sb):
        assert x == y or abs(u - v) < tie, (what, x, y, u, v)
    if ids_a != ids_b:   # whatever entered / left at the cut-off must be a near-tie with the boundary
        for x in set(ids_a) ^ set(ids_b):
            s = sa[ids_a.index(x)] if x in ids_a else sb[ids_b.index(x)]
            assert abs(s - min(sa.min(), sb.min())) < tie, (what, x, s)


def assert_same_records(a, b, stages=(("stage1_results", "stage1_score"), ("stage2_results", "stage2_score"),
                                      ("results", "stage3_score")), atol=1e-3, tie=1e-4, what=""):
    for stage, key in stages:
        assert_same_ranking([r["doc_id"] for r in a[stage]], [r[key] for r in a[stage]],
                            [r["doc_id"] for r in b[stage]], [r[key] for r in b[stage]], atol, tie,
                            what=f"{what} {stage}")
