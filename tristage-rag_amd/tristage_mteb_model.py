"""TriStageMTEBModel — MTEB-facing adapter, API-compatible with the reference's
benchmark/tristage_mteb_model.py so it drops in under
benchmark/run_mteb_evaluation.py (:294-299 builds it, :98-104 hands it to MTEB).

Kept: constructor arguments, ``encode`` (normalised
stage-1 embeddings; corpus/query decided by the task/prompt-name heuristic
:135-161; per-call result caches :168,:207), ``search`` (record format :291-306,
final score = stage 3, else stage 2, else stage 1), ``predict`` in its three
calling patterns (:310-400), ``search_cross_encoder`` (-> {qid: {doc_id: score}},
:402-481), ``__call__``, ``get_pipeline_info``, attributes
``similarity_fn_name = "cosine"``, ``max_seq_length = 512``, ``model_card_data``,
and ``create_tristage_model``.  The optional registration with mteb by ``exec``
(:529-555) is not reproduced: mteb resolves a model object passed to it directly.
"""
from __future__ import annotations

import logging
from collections import OrderedDict, defaultdict
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, List, Optional, Union

import numpy as np

from .retrieval_pipeline import PipelineConfig, RetrievalPipeline


@dataclass
class ModelCard:
    model_name: str = "TriStage-RAG"
    name: str = "TriStage-RAG"
    description: str = "3-stage retrieval pipeline with embedding, ColBERT, and cross-encoder models"
    architecture: str = "TriStage-RAG"
    framework: list = None
    model_type: str = "retrieval"
    languages: list = None
    language: list = None
    base_model_revision: str = "main"
    release_date: str = "2025-09-08"
    version: str = "1.0.0"

    def __post_init__(self):
        if self.framework is None:
            self.framework = ["PyTorch-ROCm", "HIP (gfx950)"]
        if self.languages is None:
            self.languages = ["eng-Latn"]
        if self.language is None:
            self.language = ["eng-Latn"]


def _final_score(result: Dict[str, Any]) -> float:
    v = result.get("stage3_score", result.get("stage2_score", result.get("score", 0.0)))
    if isinstance(v, list):
        v = v[0] if v else 0.0
    try:
        return float(v)
    except Exception:
        return 0.0


class TriStageMTEBModel:
    def __init__(self, pipeline_config: Optional[Dict[str, Any]] = None, device: str = "auto",
                 cache_dir: str = "./models", index_dir: str = "./faiss_index",
                 pipeline: Optional[RetrievalPipeline] = None):
        self.logger = logging.getLogger(__name__)
        repo_root = Path(__file__).resolve().parent.parent
        if not Path(cache_dir).is_absolute():
            cache_dir = str((repo_root / cache_dir).resolve())
        if not Path(index_dir).is_absolute():
            index_dir = str((repo_root / index_dir).resolve())
        if pipeline is not None:
            self.pipeline = pipeline
        elif pipeline_config:
            pipeline_config = dict(pipeline_config)
            pipeline_config["cache_dir"] = cache_dir
            pipeline_config.setdefault("index_dir", index_dir)
            self.pipeline = RetrievalPipeline(config=PipelineConfig(**pipeline_config))
        else:
            self.pipeline = RetrievalPipeline(config=PipelineConfig(cache_dir=cache_dir, index_dir=index_dir))
        if device != "auto":
            self.pipeline.config.device = device
        self.similarity_metric_name = "cosine"
        self.similarity_fn_name = "cosine"
        self.max_seq_length = 512
        self._document_cache: Dict[str, np.ndarray] = {}
        self._query_cache: Dict[str, np.ndarray] = {}
        self._doc_id_map: Dict[int, str] = {}
        self.model_card_data = ModelCard()

    # -- encode ----------------------------------------------------------------
    def _is_corpus_encoding(self, task_name: str, kwargs: Dict[str, Any]) -> bool:
        """reference :135-161"""
        corpus_kw, query_kw = ("corpus", "document", "passage"), ("query", "question")
        t = task_name.lower()
        if any(k in t for k in corpus_kw):
            return True
        if any(k in t for k in query_kw):
            return False
        p = kwargs.get("prompt_name", "").lower()
        if p:
            if any(k in p for k in corpus_kw):
                return True
            if any(k in p for k in query_kw):
                return False
        return "retrieval" in t

    def _ensure_stage1(self):
        if not getattr(self.pipeline, "stage1", None):
            self.pipeline.initialize_stages()
        return self.pipeline.stage1

    def _ensure_documents_indexed(self, documents: List[str]) -> None:
        stage1 = self._ensure_stage1()
        if len(getattr(stage1, "documents", [])) == 0:
            self.pipeline.add_documents(documents)

    def _encode(self, texts: List[str], **kwargs) -> np.ndarray:
        stage1 = self._ensure_stage1()
        bs = kwargs.get("batch_size", self.pipeline.config.stage1_batch_size)
        return stage1.model.encode(texts, batch_size=bs, convert_to_numpy=True, show_progress_bar=False,
                                   normalize_embeddings=True)

    def _encode_corpus(self, documents: List[str], task_name: str, **kwargs) -> np.ndarray:
        key = f"corpus_{task_name}_{hash(str(documents[:10]))}"
        if key in self._document_cache:
            return self._document_cache[key]
        self._ensure_documents_indexed(documents)
        emb = self._encode(documents, **kwargs)
        self._document_cache[key] = emb
        return emb

    def _encode_queries(self, queries: List[str], task_name: str, **kwargs) -> np.ndarray:
        key = f"query_{task_name}_{hash(str(queries[:10]))}"
        if key in self._query_cache:
            return self._query_cache[key]
        emb = self._encode(queries, **kwargs)
        self._query_cache[key] = emb
        return emb

    def encode(self, sentences: List[str], task_name: str = "", **kwargs) -> np.ndarray:
        if not sentences:
            return np.array([])
        if self._is_corpus_encoding(task_name, kwargs):
            return self._encode_corpus(sentences, task_name, **kwargs)
        return self._encode_queries(sentences, task_name, **kwargs)

    # -- search ----------------------------------------------------------------
    def _format(self, results: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
        formatted = []
        for i, r in enumerate(results):
            iid = r.get("doc_id", i)
            ext = (self._doc_id_map.get(int(iid), str(iid)) if isinstance(iid, (int, np.integer)) else str(iid))
            formatted.append({"id": ext, "score": _final_score(r), "text": r.get("document", ""),
                              "rank": i + 1, "stage1_score": r.get("stage1_score", 0.0),
                              "stage2_score": r.get("stage2_score", 0.0),
                              "stage3_score": r.get("stage3_score", 0.0)})
        return formatted

    def search(self, query: str, top_k: int = 10, task_name: str = "") -> List[Dict[str, Any]]:
        try:
            out = self.pipeline.search(query, top_k=top_k)
        except ValueError as e:
            if "No documents indexed" in str(e):
                self.logger.warning("Pipeline has no indexed documents. Returning empty results.")
                return []
            raise
        return self._format(out.get("results", []) if isinstance(out, dict) else out)

    def search_batch(self, queries: List[str], top_k: int = 10, chunk: int = 64) -> List[List[Dict[str, Any]]]:
        """Same records as calling :meth:`search` per query, but stage 1 sweeps the corpus once
        per `chunk` queries (RetrievalPipeline.search_many) instead of once per query."""
        out: List[List[Dict[str, Any]]] = []
        for s in range(0, len(queries), chunk):
            try:
                res = self.pipeline.search_many(list(queries[s:s + chunk]), top_k=top_k)
            except ValueError as e:
                if "No documents indexed" in str(e):
                    out.extend([[] for _ in queries[s:s + chunk]])
                    continue
                raise
            out.extend(self._format(r.get("results", [])) for r in res)
        return out

    def predict(self, queries, corpus: List[str] = None, top_k: int = 10, task_name: str = "", **kwargs):
        # pattern 1: list of (query, doc[, instruction]) pairs, no corpus -> one score per pair
        if queries and isinstance(queries[0], (tuple, list)) and len(queries[0]) in (2, 3) and corpus is None:
            pairs = queries
            unique_docs: "OrderedDict[str, None]" = OrderedDict()
            for p in pairs:
                unique_docs.setdefault(str(p[1]), None)
            key = hash(tuple(unique_docs.keys()))
            if getattr(self, "_last_pair_doc_key", None) != key:
                self.pipeline.add_documents(list(unique_docs.keys()))
                self._last_pair_doc_key = key
            groups: Dict[str, List] = defaultdict(list)
            for idx, p in enumerate(pairs):
                groups[str(p[0])].append((idx, str(p[1])))
            scores: List[float] = [0.0] * len(pairs)
            for q, items in groups.items():
                try:
                    res = self.pipeline.search(q, top_k=max(1, len(items)))
                except Exception as e:
                    self.logger.warning(f"Pipeline search failed for a group ({e}); using zeros")
                    res = {"results": []}
                by_text = {r.get("document", ""): _final_score(r) for r in res.get("results", [])}
                for idx, d in items:
                    scores[idx] = by_text.get(d, 0.0)
            return scores
        # patterns 2 and 3: queries (+ optional corpus) -> list of result lists
        if corpus:
            self._ensure_documents_indexed(corpus)
        return self.search_batch([str(q) for q in queries], top_k=top_k)

    @staticmethod
    def _extract_corpus(c):
        ids: List[str] = []
        texts: List[str] = []
        if isinstance(c, dict):
            for cid, v in c.items():
                ids.append(str(cid))
                texts.append((v.get("text", "") or "") if isinstance(v, dict) else str(v))
            return ids, texts
        for i, row in enumerate(c):
            if isinstance(row, dict):
                ids.append(str(row.get("_id", row.get("id", i))))
                texts.append(row.get("text", "") or "")
            else:
                ids.append(str(i))
                texts.append(str(row))
        return ids, texts

    def search_cross_encoder(self, corpus, queries, top_k: int = 10, **kwargs):
        if corpus is not None and len(corpus) > 0:
            ids, texts = self._extract_corpus(corpus)
            stage1 = getattr(self.pipeline, "stage1", None)
            start = len(getattr(stage1, "documents", [])) if stage1 else 0
            self.pipeline.add_documents(texts)
            for off, cid in enumerate(ids):
                self._doc_id_map[start + off] = cid
        out: Dict[str, Dict[str, float]] = {}
        qids: List[str] = []
        texts: List[str] = []
        items = queries.items() if isinstance(queries, dict) else enumerate(queries)
        for i, q in items:
            if isinstance(queries, dict):
                qid, text = str(i), str(q)
            elif isinstance(q, dict):
                qid, text = str(q.get("_id", i)), q.get("text", "")
            else:
                qid, text = str(i), str(q)
            qids.append(qid)
            texts.append(text)
        for qid, recs in zip(qids, self.search_batch(texts, top_k=top_k)):
            out[qid] = {str(r.get("id", "")): float(r.get("score", 0.0)) for r in recs}
        return out

    def __call__(self, *args, **kwargs):
        if len(args) == 2 and isinstance(args[0], list) and isinstance(args[1], list):
            return self.predict(args[0], args[1], **kwargs)
        return self.encode(*args, **kwargs)

    def get_pipeline_info(self) -> Dict[str, Any]:
        c = self.pipeline.config
        return {"stage1_model": c.stage1_model, "stage2_model": c.stage2_model, "stage3_model": c.stage3_model,
                "device": c.device, "stage1_top_k": c.stage1_top_k, "stage2_top_k": c.stage2_top_k,
                "stage3_top_k": c.stage3_top_k, "similarity_metric": self.similarity_metric_name,
                "max_seq_length": self.max_seq_length}

    def __repr__(self):
        c = self.pipeline.config
        return f"TriStageMTEBModel(stage1={c.stage1_model}, stage2={c.stage2_model}, stage3={c.stage3_model})"


def create_tristage_model(model_name: str = "tristage-rag", **kwargs) -> TriStageMTEBModel:
    """Factory kept for MTEB's loading pattern (reference :514-526); model_name is ignored."""
    return TriStageMTEBModel(**kwargs)
