"""Stand-alone caller of the three stages (BASELINE.json config 0).

Counterpart of the reference's non-MCP application core: ``AppConfig`` and
``ThreeStageRetrievalSystem`` (reference non_mcp/main.py:41-50, 132-339 — fixed
funnel stage 1 top-100 -> stage 2 on the first 50 -> stage 3 on the first 20, and
the flat result schema ``rank / doc_id / document / final_score / stage{1,2,3}_score``
plus per-stage timings) and ``chunk_text`` (non_mcp/embed_and_query.py:31-53:
1000-character windows, 200 overlap, snapped back to the last '.' or newline when
that is less than 200 characters from the window end).  The CLI, pickle document
store, web UI and file-format extraction around them are out of scope.
"""
from __future__ import annotations

import logging
import time
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

from .stage1_retriever import Stage1Config, Stage1Retriever
from .stage2_rescorer import ColBERTScorer, Stage2Config
from .stage3_reranker import CrossEncoderReranker, Stage3Config


@dataclass
class AppConfig:
    models_dir: str = "../models"
    data_dir: str = "../data"
    index_dir: str = "../faiss_index"
    max_results: int = 20
    enable_bm25: bool = True
    device: str = "auto"
    log_level: str = "INFO"
    # additive: model names (the reference hard-codes them, non_mcp/main.py:167-203)
    stage1_model: str = "google/embeddinggemma-300m"
    stage2_model: str = "lightonai/GTE-ModernColBERT-v1"
    stage3_model: str = "cross-encoder/ms-marco-MiniLM-L6-v2"


def chunk_text(text: str, chunk_size: int = 1000, overlap: int = 200) -> List[str]:
    text = (text or "").strip()
    out: List[str] = []
    n, start = len(text), 0
    while start < n:
        end = min(start + chunk_size, n)
        if end < n:
            window = text[start:end]
            cut = max(window.rfind("."), window.rfind("\n"))
            if cut > 0 and end - (start + cut) < 200:
                end = start + cut + 1
        piece = text[start:end].strip()
        if piece:
            out.append(piece)
        if end >= n:
            break
        start = max(end - overlap, 0)
    return out


class ThreeStageRetrievalSystem:
    def __init__(self, config: AppConfig, stage1: Optional[Stage1Retriever] = None,
                 stage2: Optional[ColBERTScorer] = None, stage3: Optional[CrossEncoderReranker] = None):
        self.config = config
        self.logger = logging.getLogger(__name__)
        self.documents: List[str] = []
        self.search_history: List[Dict[str, Any]] = []
        self.stage1, self.stage2, self.stage3 = stage1, stage2, stage3
        if self.stage1 is None or self.stage2 is None or self.stage3 is None:
            self._initialize_stages()

    def _initialize_stages(self) -> None:
        c = self.config
        if self.stage1 is None:
            self.stage1 = Stage1Retriever(Stage1Config(
                model_name=c.stage1_model, device=c.device, cache_dir=c.models_dir, index_dir=c.index_dir,
                top_k_candidates=100, batch_size=16, enable_bm25=c.enable_bm25, use_fp16=False))
        if self.stage2 is None:
            self.stage2 = ColBERTScorer(Stage2Config(
                model_name=c.stage2_model, device=c.device, cache_dir=c.models_dir, top_k_candidates=50,
                batch_size=8, max_seq_length=192, use_fp16=False))
        if self.stage3 is None:
            self.stage3 = CrossEncoderReranker(Stage3Config(
                model_name=c.stage3_model, device=c.device, cache_dir=c.models_dir,
                top_k_final=c.max_results, batch_size=16, max_length=256, use_fp16=False))

    def add_documents(self, documents: List[str], source: str = "manual") -> int:
        """Adds the documents not seen before (exact text match); returns how many."""
        seen = set(self.documents)
        new = []
        for d in documents:
            if d and d.strip() and d not in seen:
                seen.add(d)
                new.append(d)
        if new:
            self.documents.extend(new)
            self.stage1.add_documents(new)
        return len(new)

    def search(self, query: str, top_k: Optional[int] = None) -> Dict[str, Any]:
        if top_k is None:
            top_k = self.config.max_results
        t0 = time.time()
        try:
            t = time.time()
            candidates = self.stage1.search(query, top_k=100)
            t1 = time.time() - t
            if not candidates:
                return {"query": query, "results": [], "stage1_time": t1, "stage2_time": 0, "stage3_time": 0,
                        "total_time": time.time() - t0, "candidate_count": 0, "final_count": 0}
            t = time.time()
            rescored = self.stage2.rescore_candidates(query, candidates[:50])
            t2 = time.time() - t
            t = time.time()
            final = self.stage3.rerank(query, rescored[:20])
            t3 = time.time() - t
            results = []
            for i, r in enumerate(final[:top_k]):
                s1 = r.get("stage1_score")
                if s1 is None:
                    s1 = r.get("score", 0)
                s2, s3 = r.get("stage2_score", 0), r.get("stage3_score", 0)
                fs = s3 if s3 is not None else (s2 if s2 is not None else (s1 if s1 is not None else 0))
                results.append({"rank": i + 1, "doc_id": r.get("doc_id", f"doc_{i}"),
                                "document": r.get("document", ""), "final_score": fs,
                                "stage1_score": s1 if s1 is not None else 0,
                                "stage2_score": s2 if s2 is not None else 0,
                                "stage3_score": s3 if s3 is not None else 0})
            total = time.time() - t0
            self.search_history.append({"query": query, "timestamp": time.time(), "total_time": total,
                                        "result_count": len(results), "stage1_time": t1, "stage2_time": t2,
                                        "stage3_time": t3})
            self.search_history = self.search_history[-100:]
            return {"query": query, "results": results, "stage1_time": t1, "stage2_time": t2,
                    "stage3_time": t3, "total_time": total, "candidate_count": len(candidates),
                    "final_count": len(results)}
        except Exception as e:  # the reference reports the error in the result instead of raising
            self.logger.error(f"Error during search: {e}")
            return {"query": query, "results": [], "stage1_time": 0, "stage2_time": 0, "stage3_time": 0,
                    "total_time": time.time() - t0, "candidate_count": 0, "final_count": 0, "error": str(e)}
