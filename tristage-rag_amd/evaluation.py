"""Retrieval evaluation without mteb / pytrec_eval (both absent offline).

Counterpart of the parts of the reference harness that touch the hot path:
the local LIMIT / BEIR JSONL schema (``corpus.jsonl`` / ``queries.jsonl`` rows
``{_id, text[, title]}``, ``qrels.jsonl`` rows ``{query-id, corpus-id, score}``;
reference benchmark/limit_mteb_tasks.py:129-158, benchmark/run_mteb_evaluation.py:41-80)
and the metric MTEB reports for retrieval tasks, nDCG@10 with trec_eval
semantics (gain = relevance, discount 1/log2(rank+1), ties broken by document id
descending), which run_mteb_evaluation.py:343-392 prints as the main score.

Entry point (counterpart of ``python benchmark/run_mteb_evaluation.py``, reference
:116-392, minus the model / dataset downloads, which need a network):

    python -m tristage_rag_amd.evaluation --limit-path DIR [--tasks NAME] [--mode rerank|dense]
        [--output benchmark/mteb_results] [--device auto] [--cache-dir ../models]
        [--index-dir ./faiss_index] [--sample-size N] [--low-mem] [--stage1-model M] ...

``DIR`` holds corpus.jsonl / queries.jsonl / qrels.jsonl.  ``--mode rerank`` drives
``TriStageMTEBModel.search_cross_encoder`` (the full three stages, what MTEB v2 calls for a
reranking-style model, reference benchmark/tristage_mteb_model.py:402-481); ``--mode dense``
drives ``TriStageMTEBModel.encode`` for corpus and queries and ranks by cosine on the HIP
index (what MTEB does with an encoder-style model, reference :106-133).  Prints the
reference's "Summary of results" block and writes ``<output>/<task>.json``.
"""
from __future__ import annotations

import argparse
import json
import logging
import math
import os
import sys
import time
from typing import Any, Dict, Iterable, List, Optional, Tuple


def _jsonl(path: str) -> Iterable[dict]:
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if line:
                yield json.loads(line)


def load_jsonl_dataset(data_dir: str) -> Tuple[Dict[str, dict], Dict[str, str], Dict[str, Dict[str, int]]]:
    """-> (corpus {id: {text, title}}, queries {id: text}, qrels {qid: {doc_id: rel}})."""
    corpus = {str(r["_id"]): {"text": r.get("text", ""), "title": r.get("title", "")}
              for r in _jsonl(os.path.join(data_dir, "corpus.jsonl"))}
    queries = {str(r["_id"]): r.get("text", "") for r in _jsonl(os.path.join(data_dir, "queries.jsonl"))}
    qrels: Dict[str, Dict[str, int]] = {}
    for r in _jsonl(os.path.join(data_dir, "qrels.jsonl")):
        qrels.setdefault(str(r["query-id"]), {})[str(r["corpus-id"])] = int(r["score"])
    return corpus, queries, qrels


def ndcg_at_k(qrels: Dict[str, Dict[str, int]], results: Dict[str, Dict[str, float]], k: int = 10) -> float:
    vals = []
    for qid, rels in qrels.items():
        run = results.get(qid, {})
        ranked = sorted(run.items(), key=lambda kv: (kv[1], kv[0]), reverse=True)[:k]
        dcg = sum(rels.get(doc, 0) / math.log2(r + 2) for r, (doc, _) in enumerate(ranked))
        ideal = sorted((v for v in rels.values() if v > 0), reverse=True)[:k]
        idcg = sum(g / math.log2(r + 2) for r, g in enumerate(ideal))
        vals.append(dcg / idcg if idcg > 0 else 0.0)
    return sum(vals) / len(vals) if vals else 0.0


def evaluate_retrieval(model, corpus: Dict[str, dict], queries: Dict[str, str],
                       qrels: Dict[str, Dict[str, int]], top_k: int = 10) -> Dict[str, float]:
    """Runs ``model.search_cross_encoder`` (the TriStageMTEBModel entry point MTEB v2 uses for
    reranking-style retrieval) and scores it."""
    results = model.search_cross_encoder(corpus, queries, top_k=top_k)
    return {f"ndcg_at_{top_k}": ndcg_at_k(qrels, results, top_k), "num_queries": float(len(queries))}


def _ranked(run: Dict[str, float]) -> List[Tuple[str, float]]:
    return sorted(run.items(), key=lambda kv: (kv[1], kv[0]), reverse=True)


def recall_at_k(qrels: Dict[str, Dict[str, int]], results: Dict[str, Dict[str, float]], k: int) -> float:
    vals = []
    for qid, rels in qrels.items():
        pos = {d for d, v in rels.items() if v > 0}
        if not pos:
            continue
        top = {d for d, _ in _ranked(results.get(qid, {}))[:k]}
        vals.append(len(pos & top) / len(pos))
    return sum(vals) / len(vals) if vals else 0.0


def mrr_at_k(qrels: Dict[str, Dict[str, int]], results: Dict[str, Dict[str, float]], k: int) -> float:
    vals = []
    for qid, rels in qrels.items():
        rr = 0.0
        for r, (d, _) in enumerate(_ranked(results.get(qid, {}))[:k]):
            if rels.get(d, 0) > 0:
                rr = 1.0 / (r + 1)
                break
        vals.append(rr)
    return sum(vals) / len(vals) if vals else 0.0


def score_run(qrels, results, ks=(1, 3, 5, 10, 100)) -> Dict[str, float]:
    out: Dict[str, float] = {}
    for k in ks:
        out[f"ndcg_at_{k}"] = ndcg_at_k(qrels, results, k)
        out[f"recall_at_{k}"] = recall_at_k(qrels, results, k)
    out["mrr_at_10"] = mrr_at_k(qrels, results, 10)
    out["main_score"] = out["ndcg_at_10"]          # MTEB's main score for retrieval tasks
    return out


def dense_results(model, corpus: Dict[str, dict], queries: Dict[str, str], top_k: int = 100,
                  task_name: str = "", batch_size: int = 32, index_dtype: str = "f32",
                  index_factory=None) -> Dict[str, Dict[str, float]]:
    """What MTEB does with an encoder-style model: ``model.encode`` for the corpus
    (``title + " " + text``) and for the queries, cosine top-k.  The embeddings come back
    L2-normalised (reference benchmark/tristage_mteb_model.py:187-193, 223-229), so cosine is the
    inner product and the top-k is one exact search on the HIP index."""
    import numpy as np
    ids = list(corpus)
    docs = [((corpus[i].get("title", "") or "") + " " + (corpus[i].get("text", "") or "")).strip() for i in ids]
    E = np.asarray(model.encode(docs, task_name=task_name or "corpus", prompt_name="passage", batch_size=batch_size),
                   dtype=np.float32)
    qids = list(queries)
    Q = np.asarray(model.encode([queries[q] for q in qids], task_name="query", prompt_name="query",
                                batch_size=batch_size), dtype=np.float32)
    if index_factory is None:
        from .index import FlatIPIndex   # the HIP index; raises without the library or a GPU
        index = FlatIPIndex(int(E.shape[1]), dtype=index_dtype)
    else:
        index = index_factory(int(E.shape[1]))
    index.add(E)
    k = min(int(top_k), len(ids))
    D, I = index.search(Q, k)
    if hasattr(index, "close"):
        index.close()
    return {qid: {ids[int(j)]: float(s) for s, j in zip(D[r], I[r]) if j >= 0} for r, qid in enumerate(qids)}


def run_task(model, data_dir: str, task_name: str = "LIMITSmallRetrieval", mode: str = "rerank",
             top_k: int = 10, sample_size: Optional[int] = None, index_factory=None) -> Dict[str, Any]:
    """One retrieval task from a JSONL directory -> the result entry MTEB would write
    (``scores: {test: [{ndcg_at_10, main_score, ...}]}``)."""
    corpus, queries, qrels = load_jsonl_dataset(data_dir)
    if not corpus or not queries:
        raise ValueError("No queries or corpus found in dataset")
    if sample_size:
        keep = list(corpus)[: int(sample_size)]
        corpus = {k: corpus[k] for k in keep}
    t0 = time.time()
    if mode == "dense":
        results = dense_results(model, corpus, queries, top_k=max(top_k, 100), task_name=task_name,
                                index_factory=index_factory)
    elif mode == "rerank":
        results = model.search_cross_encoder(corpus, queries, top_k=top_k)
    else:
        raise ValueError(f"unknown mode {mode!r}")
    dt = time.time() - t0
    scores = score_run(qrels, results)
    return {"task_name": task_name, "mteb_dataset_name": task_name, "mode": mode,
            "scores": {"test": [dict(scores, hf_subset="default", languages=["eng-Latn"])]},
            "main_score": scores["main_score"], "evaluation_time": dt,
            "num_queries": len(queries), "num_documents": len(corpus)}


LOW_MEM_OVERRIDES = {   # reference benchmark/run_mteb_evaluation.py:272-287
    "stage1_model": "sentence-transformers/all-MiniLM-L6-v2", "stage1_batch_size": 32, "stage1_top_k": 200,
    "stage1_use_fp16": False,
    "stage2_model": "sentence-transformers/all-MiniLM-L6-v2", "stage2_batch_size": 8, "stage2_top_k": 50,
    "stage2_use_fp16": False,
    "stage3_model": "cross-encoder/ms-marco-MiniLM-L-6-v2", "stage3_batch_size": 16, "stage3_top_k": 10,
    "stage3_use_fp16": False,
}


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(description="Evaluate TriStage-RAG (MI355X build) on a local JSONL retrieval task")
    ap.add_argument("--tasks", nargs="+", default=["LIMITSmallRetrieval"],
                    help="task names (labels of the result files; every task reads --limit-path)")
    ap.add_argument("--output", type=str, default="benchmark/mteb_results")
    ap.add_argument("--limit-path", type=str, required=True, help="directory with corpus/queries/qrels .jsonl")
    ap.add_argument("--log-level", choices=["DEBUG", "INFO", "WARNING", "ERROR"], default="INFO")
    ap.add_argument("--device", type=str, default="auto")
    ap.add_argument("--cache-dir", type=str, default="../models")
    ap.add_argument("--index-dir", type=str, default="./faiss_index")
    ap.add_argument("--sample-size", type=int, default=None)
    ap.add_argument("--low-mem", action="store_true")
    ap.add_argument("--stage1-model", type=str, default=None)
    # additive
    ap.add_argument("--stage2-model", type=str, default=None)
    ap.add_argument("--stage3-model", type=str, default=None)
    ap.add_argument("--mode", choices=["rerank", "dense"], default="rerank")
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--no-bm25", action="store_true", help="stage1_enable_bm25=False (pure dense stage 1)")
    ap.add_argument("--token-store", action="store_true", help="stage-2 token matrices and stage-3 token ids resident (search_many on arrays)")
    ap.add_argument("--index-dtype", default="f32", choices=["f32", "f16", "bf16"])
    args = ap.parse_args(argv)
    logging.basicConfig(level=getattr(logging, args.log_level),
                        format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    from .tristage_mteb_model import TriStageMTEBModel
    overrides: Dict[str, Any] = dict(LOW_MEM_OVERRIDES) if args.low_mem else {}
    for key, val in (("stage1_model", args.stage1_model), ("stage2_model", args.stage2_model),
                     ("stage3_model", args.stage3_model)):
        if val:
            overrides[key] = val
    if args.no_bm25:
        overrides["stage1_enable_bm25"] = False
    if args.token_store:
        overrides["stage2_precompute_document_embeddings"] = True
        overrides["stage3_cache_document_tokens"] = True      # with the token store: every stage of search_many on arrays
    if args.index_dtype != "f32":
        overrides["stage1_index_dtype"] = args.index_dtype
    if args.log_level != "INFO":
        overrides["log_level"] = args.log_level
    print(f"Using LIMIT dataset from: {args.limit_path}")
    print("Initializing TriStage-RAG model for MTEB evaluation...")
    model = TriStageMTEBModel(device=args.device, cache_dir=args.cache_dir, index_dir=args.index_dir,
                              pipeline_config=overrides if overrides else None)
    print(f"Model created: {model}")
    print(f"Pipeline info: {model.get_pipeline_info()}")
    os.makedirs(args.output, exist_ok=True)
    results = []
    try:
        for i, task in enumerate(args.tasks):
            if i:   # a task starts from an empty index, as a fresh MTEB task would
                model = TriStageMTEBModel(device=args.device, cache_dir=args.cache_dir, index_dir=args.index_dir,
                                          pipeline_config=overrides if overrides else None)
            entry = run_task(model, args.limit_path, task, mode=args.mode, top_k=args.top_k,
                             sample_size=args.sample_size)
            results.append(entry)
            with open(os.path.join(args.output, f"{task}.json"), "w") as f:
                json.dump(entry, f, indent=2)
            print(f"Evaluation completed in {entry['evaluation_time']:.2f} seconds")
    except Exception as e:  # the reference prints and returns (:389-392); the exit code tells a script
        print(f"Evaluation failed: {e}")
        import traceback
        traceback.print_exc()
        return 1
    print("\nEvaluation completed successfully!")
    print(f"Results saved to: {args.output}")
    print("\nSummary of results:")
    for entry in results:
        print(f"  {entry['task_name']}: {float(entry['main_score']):.4f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
