#!/usr/bin/env python3
"""The reference's workflow (index documents, search) on the MI355X-native pipeline.

    python examples/search_pipeline.py [--model-dir ./models]

With model directories under --model-dir (looked up as <cache_dir>/<basename>, exactly like the
reference, src/stage1_retriever.py:148-151) the configured encoders are loaded; without any —
this environment has no network — randomly initialised models of the same architectures stand
in ("random:<arch>"), which exercises every kernel but ranks arbitrarily."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-dir", default="./models")
    args = ap.parse_args()
    from tristage_rag_amd.retrieval_pipeline import PipelineConfig, RetrievalPipeline

    def model(name, arch):
        return name if os.path.isdir(os.path.join(args.model_dir, os.path.basename(name))) else f"random:{arch}"

    cfg = PipelineConfig(
        stage1_model=model("google/embeddinggemma-300m", "bert"),
        stage2_model=model("lightonai/GTE-ModernColBERT-v1", "modernbert"),
        stage3_model=model("cross-encoder/ms-marco-MiniLM-L6-v2", "minilm"),
        device="cuda", cache_dir=args.model_dir, index_dir="/tmp/ts_example_index", log_file="/tmp/ts_example.log",
        stage1_top_k=100, stage2_top_k=20, stage3_top_k=5,
        # additive knobs of this build:
        stage1_index_dtype="f16",                       # corpus kept as fp16: half the bytes scanned
        stage2_precompute_document_embeddings=True,     # token matrices resident in HBM, scored in place
        stage3_cache_document_tokens=True,              # documents tokenised once; with the line above search() and
                                                        # search_many() run every stage on arrays
        use_hip_graphs=True)                            # per-query forwards replayed from HIP graphs
    pipe = RetrievalPipeline(config=cfg)
    docs = [f"Document {i}: " + " ".join(w for w in ("retrieval", "ranking", "gpu", "memory", "attention", "index",
                                                    "query", "token")[i % 5:i % 5 + 3]) for i in range(500)]
    pipe.add_documents(docs)
    one = pipe.search("gpu memory attention", top_k=3)                         # the reference's call
    for r in one["results"]:
        print(f"search      doc {r['doc_id']:4d}  s1 {r['stage1_score']:.4f}  s2 {r['stage2_score']:.4f}  "
              f"s3 {r['stage3_score']:.4f}")
    many = pipe.search_many(["gpu memory attention", "token index", "ranking"], top_k=3)   # every stage batched
    for q in many:
        print("search_many", repr(q["query"]), [r["doc_id"] for r in q["results"]])
    pipe.save_index()                                                           # + the stage-2 token store
    print("timing of the last search (s):", {k: round(v, 5) for k, v in one["timing"].items()})


if __name__ == "__main__":
    main()
