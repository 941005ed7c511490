"""Retrieval evaluation without mteb / pytrec_eval (both absent offline).

Counterpart of the parts of the reference harness that touch the hot path:
the local LIMIT / BEIR JSONL schema (``corpus.jsonl`` / ``queries.jsonl`` rows
``{_id, text[, title]}``, ``qrels.jsonl`` rows ``{query-id, corpus-id, score}``;
reference benchmark/limit_mteb_tasks.py:129-158, benchmark/run_mteb_evaluation.py:41-80)
and the metric MTEB reports for retrieval tasks, nDCG@10 with trec_eval
semantics (gain = relevance, discount 1/log2(rank+1), ties broken by document id
descending), which run_mteb_evaluation.py:343-392 prints as the main score.
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, Iterable, Tuple


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
