"""tristage-rag_amd — MI355X-native retrieval hot path of TriStage-RAG.

Import it as ``tristage_rag_amd`` (the hyphenated directory name required by
the repository layout is not a Python identifier; ``tristage_rag_amd/`` at the
repository root is a two-line alias whose ``__path__`` points here).

Layout: ``csrc/`` HIP kernels + the C ABI (include/tristage.h);
``_lib`` ctypes binding; ``index`` FAISS-shaped index object, shard merge and
MaxSim entry points; the ``stage*`` / ``retrieval_pipeline`` /
``tristage_mteb_model`` modules mirror the reference's Python API surface.
"""
__version__ = "0.1.0"

_LAZY = {
    "FlatIPIndex": ("index", "FlatIPIndex"),
    "merge_topk": ("index", "merge_topk"),
    "maxsim": ("index", "maxsim"),
    "ShardedFlatIPIndex": ("sharded", "ShardedFlatIPIndex"),
    "RetrievalPipeline": ("retrieval_pipeline", "RetrievalPipeline"),
    "PipelineConfig": ("retrieval_pipeline", "PipelineConfig"),
    "Stage1Retriever": ("stage1_retriever", "Stage1Retriever"),
    "Stage1Config": ("stage1_retriever", "Stage1Config"),
    "BM25Index": ("stage1_retriever", "BM25Index"),
    "ColBERTScorer": ("stage2_rescorer", "ColBERTScorer"),
    "Stage2Config": ("stage2_rescorer", "Stage2Config"),
    "CrossEncoderReranker": ("stage3_reranker", "CrossEncoderReranker"),
    "AdaptiveCrossEncoderReranker": ("stage3_reranker", "AdaptiveCrossEncoderReranker"),
    "Stage3Config": ("stage3_reranker", "Stage3Config"),
    "EmbeddingService": ("embedding_service", "EmbeddingService"),
    "TriStageMTEBModel": ("tristage_mteb_model", "TriStageMTEBModel"),
    "create_tristage_model": ("tristage_mteb_model", "create_tristage_model"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f"{__name__}.{mod}"), attr)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
