"""RetrievalPipeline — the three-stage orchestrator, API-compatible with the
reference's src/retrieval_pipeline.py.

Kept verbatim: ``PipelineConfig`` field names and defaults
(reference :52-87), ``RetrievalPipeline(config_path=None, config=None)``, lazy
``initialize_stages`` (:238-290), ``add_documents`` (:292-321), ``search`` with its
result keys ``query / results / stage1_results / stage2_results / timing /
performance_stats`` and the early returns when a stage yields nothing (:323-424),
``batch_search``, ``save_index`` / ``load_index``, ``get_pipeline_info``,
``export_config``, the running-mean ``performance_stats`` (:567-606).

MI355X-side differences: stage 1 is the exact HIP index, stage 2 one MaxSim
kernel launch per query, no ``torch.cuda.empty_cache()`` per query, and
``search_many`` pushes a whole query batch through one stage-1 sweep.
Additive config fields carry a default so reference YAML files still load.
"""
from __future__ import annotations

import logging
import os
import time
from dataclasses import asdict, dataclass, fields
from typing import Any, Dict, List, Optional

import yaml

from .stage1_retriever import Stage1Config, Stage1Retriever
from .stage2_rescorer import ColBERTScorer, Stage2Config
from .stage3_reranker import AdaptiveCrossEncoderReranker, Stage3Config


@dataclass
class PipelineConfig:
    """All knobs of the three-stage pipeline (reference src/retrieval_pipeline.py:15-87)."""
    # Stage 1
    stage1_model: str = "google/embeddinggemma-300m"
    stage1_top_k: int = 500
    stage1_batch_size: int = 32
    stage1_enable_bm25: bool = True
    stage1_bm25_top_k: int = 300
    stage1_fusion_method: str = "rrf"
    stage1_use_fp16: bool = True
    # Stage 2
    stage2_model: str = "lightonai/GTE-ModernColBERT-v1"
    stage2_top_k: int = 100
    stage2_batch_size: int = 16
    stage2_max_seq_length: int = 192
    stage2_use_fp16: bool = True
    stage2_scoring_method: str = "maxsim"
    # Stage 3
    stage3_model: str = "cross-encoder/ms-marco-MiniLM-L6-v2"
    stage3_top_k: int = 20
    stage3_batch_size: int = 32
    stage3_max_length: int = 256
    stage3_use_fp16: bool = True
    # General
    device: str = "auto"
    cache_dir: str = "./models"
    index_dir: str = "./faiss_index"
    log_level: str = "INFO"
    log_file: str = "retrieval_pipeline.log"
    enable_timing: bool = True
    save_intermediate_results: bool = False
    auto_cleanup: bool = True
    max_memory_usage_gb: float = 4.0
    # ---- additive (MI355X) ----
    stage1_index_dtype: str = "f32"          # corpus storage on the GPU: f32 | f16 | bf16
    stage1_bm25_on_gpu: Optional[bool] = None  # BM25 postings in HBM, HIP scoring; None = when a GPU is used
    stage1_bm25_refit_compat: bool = False     # BM25 after a second add_documents as the reference computes it (appended statistics)
    stage2_cache_document_embeddings: bool = False
    stage2_precompute_document_embeddings: bool = False  # token store filled by add_documents
    use_hip_graphs: bool = False             # query forwards of stages 1/2 and a query's stage-3 pairs replayed from HIP graphs
    stage2_token_store_dtype: str = "auto"   # "auto": bf16 under AMP, else the encoder's output type; or bf16 | f16 | f32
    amp_dtype: str = "bf16"                     # what stage*_use_fp16 means on the GPU: "bf16" (BASELINE configs[2]) or "fp16"
                                                # (what torch.cuda.amp.autocast() gives the reference on a GPU)
    stage1_index_batch_size: int = 256          # documents per encoder forward at add_documents time (stage 1, device path)
    stage2_index_batch_size: int = 256          # ... and for the stage-2 token store
    tune_gemms: bool = False                    # PyTorch TunableOp: the fastest hipBLASLt / rocBLAS solution per GEMM shape,
                                                # found by timing at the shape's first occurrence (~1 s each, kept in the
                                                # process and in tune_gemms_file): the 1024-pair stage-3 forward 9.6 -> 8.7 ms
    tune_gemms_file: str = ""                   # results file (read at start, written at exit); "" = TunableOp's default name
    stage3_width_multiple: int = 1              # pad the token width of search_many's stage-3 batches to a multiple of this
                                                # (16 with tune_gemms: bounds the number of GEMM shapes to tune)
    search_on_arrays: bool = True               # search() takes the array path of search_many when its preconditions hold
    stage3_cache_document_tokens: bool = False  # tokenise every document once at add time; search_many then assembles
                                                # the cross-encoder inputs from token ids on the GPU
    stage3_many_batch_size: int = 1024       # pairs per cross-encoder forward in search_many
    stage3_many_packed_batch_size: int = 4096  # ... when the batch is packed (written-out forward on the GPU): no padding to pay


# (section, key) in the reference's YAML layout -> PipelineConfig field (reference :182-217)
_YAML_MAP = {
    ("stage1", "model"): "stage1_model", ("stage1", "top_k"): "stage1_top_k",
    ("stage1", "batch_size"): "stage1_batch_size", ("stage1", "enable_bm25"): "stage1_enable_bm25",
    ("stage1", "bm25_top_k"): "stage1_bm25_top_k", ("stage1", "fusion_method"): "stage1_fusion_method",
    ("stage1", "use_fp16"): "stage1_use_fp16", ("stage1", "index_dtype"): "stage1_index_dtype",
    ("stage2", "model"): "stage2_model", ("stage2", "top_k"): "stage2_top_k",
    ("stage2", "batch_size"): "stage2_batch_size", ("stage2", "max_seq_length"): "stage2_max_seq_length",
    ("stage2", "use_fp16"): "stage2_use_fp16", ("stage2", "scoring_method"): "stage2_scoring_method",
    ("stage3", "model"): "stage3_model", ("stage3", "top_k"): "stage3_top_k",
    ("stage3", "batch_size"): "stage3_batch_size", ("stage3", "max_length"): "stage3_max_length",
    ("stage3", "use_fp16"): "stage3_use_fp16",
}
_GENERAL_KEYS = ("device", "cache_dir", "index_dir", "log_level", "log_file", "enable_timing",
                 "save_intermediate_results", "auto_cleanup", "max_memory_usage_gb")


class RetrievalPipeline:
    """Stage 1 candidate generation -> stage 2 MaxSim rescoring -> stage 3 cross-encoder."""

    def __init__(self, config_path: Optional[str] = None, config: Optional[PipelineConfig] = None):
        self.logger = logging.getLogger(__name__)
        if config_path:
            self.config = self._load_config(config_path)
        elif config:
            self.config = config
        else:
            self.config = PipelineConfig()
        self._setup_logging()
        self.stage1: Optional[Stage1Retriever] = None
        self.stage2: Optional[ColBERTScorer] = None
        self.stage3: Optional[AdaptiveCrossEncoderReranker] = None
        self.performance_stats: Dict[str, Any] = {
            "total_queries": 0, "avg_stage1_time": 0.0, "avg_stage2_time": 0.0,
            "avg_stage3_time": 0.0, "avg_total_time": 0.0, "stage_time_history": []}
        self.logger.info("RetrievalPipeline initialized")

    # -- configuration -------------------------------------------------------
    def _load_config(self, config_path: str) -> PipelineConfig:
        """YAML with a top-level ``pipeline:`` mapping; nested stage sections as in the
        reference, or the flat field names that ``export_config`` writes."""
        try:
            with open(config_path, "r") as f:
                data = yaml.safe_load(f) or {}
            pd = data.get("pipeline", {}) or {}
            kw: Dict[str, Any] = {}
            names = {f.name for f in fields(PipelineConfig)}
            for k, v in pd.items():
                if k in names and not isinstance(v, dict):
                    kw[k] = v
            for (section, key), field_name in _YAML_MAP.items():
                sec = pd.get(section)
                if isinstance(sec, dict) and key in sec:
                    kw[field_name] = sec[key]
            for k in _GENERAL_KEYS:
                if k in pd:
                    kw[k] = pd[k]
            return PipelineConfig(**kw)
        except Exception as e:  # reference :219-221: fall back to defaults
            self.logger.error(f"Error loading config: {e}")
            return PipelineConfig()

    def _setup_logging(self) -> None:
        level = getattr(logging, str(self.config.log_level).upper(), logging.INFO)
        handlers: List[logging.Handler] = [logging.StreamHandler()]
        try:
            handlers.insert(0, logging.FileHandler(self.config.log_file))
        except OSError:
            pass
        logging.basicConfig(level=level, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s",
                            handlers=handlers)

    def initialize_stages(self) -> None:
        self.logger.info("Initializing pipeline stages...")
        c = self.config
        if c.tune_gemms:
            import torch
            if torch.cuda.is_available():
                if c.tune_gemms_file:
                    torch.cuda.tunable.set_filename(c.tune_gemms_file)
                torch.cuda.tunable.enable(True)
                torch.cuda.tunable.tuning_enable(True)
        try:
            self.stage1 = Stage1Retriever(Stage1Config(
                model_name=c.stage1_model, device=c.device, cache_dir=c.cache_dir, index_dir=c.index_dir,
                top_k_candidates=c.stage1_top_k, batch_size=c.stage1_batch_size,
                enable_bm25=c.stage1_enable_bm25, bm25_top_k=c.stage1_bm25_top_k,
                fusion_method=c.stage1_fusion_method, use_fp16=c.stage1_use_fp16,
                index_dtype=c.stage1_index_dtype, bm25_on_gpu=c.stage1_bm25_on_gpu,
                bm25_refit_compat=c.stage1_bm25_refit_compat,
                use_hip_graph=c.use_hip_graphs, index_batch_size=c.stage1_index_batch_size, amp_dtype=c.amp_dtype))
            self.logger.info("Stage 1 initialized")
            self.stage2 = ColBERTScorer(Stage2Config(
                model_name=c.stage2_model, device=c.device, cache_dir=c.cache_dir,
                max_seq_length=c.stage2_max_seq_length, batch_size=c.stage2_batch_size,
                top_k_candidates=c.stage2_top_k, use_fp16=c.stage2_use_fp16,
                scoring_method=c.stage2_scoring_method,
                cache_document_embeddings=c.stage2_cache_document_embeddings,
                precompute_document_embeddings=c.stage2_precompute_document_embeddings,
                token_store_dtype=c.stage2_token_store_dtype,
                use_hip_graph=c.use_hip_graphs, index_batch_size=c.stage2_index_batch_size, amp_dtype=c.amp_dtype))
            self.logger.info("Stage 2 initialized")
            self.stage3 = AdaptiveCrossEncoderReranker(Stage3Config(
                model_name=c.stage3_model, device=c.device, cache_dir=c.cache_dir,
                max_length=c.stage3_max_length, batch_size=c.stage3_batch_size,
                top_k_final=c.stage3_top_k, use_fp16=c.stage3_use_fp16, use_hip_graph=c.use_hip_graphs,
                many_batch_size=c.stage3_many_batch_size, amp_dtype=c.amp_dtype,
                many_width_multiple=c.stage3_width_multiple,
                many_packed_batch_size=max(c.stage3_many_batch_size, c.stage3_many_packed_batch_size)))
            self.logger.info("Stage 3 initialized")
        except Exception as e:
            self.logger.error(f"Error initializing pipeline stages: {e}")
            raise

    # -- indexing --------------------------------------------------------------
    def add_documents(self, documents: List[str], metadata: Optional[List[Dict[str, Any]]] = None):
        if not self.stage1:
            self.initialize_stages()
        self.logger.info(f"Adding {len(documents)} documents to pipeline")
        try:
            first_id = len(self.stage1.documents)
            self.stage1.add_documents(documents, metadata)
            if self.stage2 is not None and self.stage2.config.precompute_document_embeddings:
                self.stage2.index_documents(list(documents), first_id)
            if self.stage3 is not None and self.config.stage3_cache_document_tokens:
                self.stage3.index_documents(list(documents), first_id)
        except Exception as e:
            self.logger.error(f"Error adding documents: {e}")
            raise

    # -- search ----------------------------------------------------------------
    def _now(self) -> Optional[float]:
        return time.time() if self.config.enable_timing else None

    def _get_timing(self, enable_timing: bool) -> Optional[float]:
        return time.time() if enable_timing else None

    def _calculate_timing(self, total_start, stage1_time, stage2_time, stage3_time) -> Dict[str, float]:
        if not self.config.enable_timing:
            return {}
        total = time.time() - total_start if total_start else None
        return {"stage1_time": stage1_time or 0.0, "stage2_time": stage2_time or 0.0,
                "stage3_time": stage3_time or 0.0, "total_time": total or 0.0}

    def _run_later_stages(self, query: str, top_k: int, stage1_results, total_start, stage1_time):
        if not stage1_results:
            return {"query": query, "results": [], "stage1_results": [], "stage2_results": [],
                    "timing": self._calculate_timing(total_start, stage1_time, None, None),
                    "performance_stats": self.performance_stats}
        t = self._now()
        stage2_results = self.stage2.rescore_candidates(query, stage1_results)
        stage2_time = time.time() - t if t else None
        if not stage2_results:
            return {"query": query, "results": [], "stage1_results": stage1_results, "stage2_results": [],
                    "timing": self._calculate_timing(total_start, stage1_time, stage2_time, None),
                    "performance_stats": self.performance_stats}
        t = self._now()
        final = self.stage3.rerank(query, stage2_results)
        stage3_time = time.time() - t if t else None
        final = final[:top_k]
        total_time = time.time() - total_start if total_start else None
        if self.config.enable_timing:
            self._update_performance_stats(stage1_time, stage2_time, stage3_time, total_time)
        keep = self.config.save_intermediate_results
        result = {"query": query, "results": final,
                  "stage1_results": stage1_results if keep else [],
                  "stage2_results": stage2_results if keep else [],
                  "timing": self._calculate_timing(total_start, stage1_time, stage2_time, stage3_time),
                  "performance_stats": self.performance_stats.copy()}
        if self.config.auto_cleanup:
            self._cleanup_memory()
        return result

    def search(self, query: str, top_k: Optional[int] = None) -> Dict[str, Any]:
        if not self.stage1 or not self.stage2 or not self.stage3:
            self.initialize_stages()
        top_k = top_k or self.config.stage3_top_k
        if self.config.search_on_arrays:
            # one query through the array path of search_many (resident token store + stage-3 id cache): no candidate
            # text is tokenised or re-encoded, one cross-encoder forward for the query's pairs; same records (tests)
            fast = self._search_many_arrays([query], top_k)
            if fast is not None:
                return fast[0]
        total_start = self._now()
        try:
            t = self._now()
            stage1_results = self.stage1.search(query, self.config.stage1_top_k)
            stage1_time = time.time() - t if t else None
            return self._run_later_stages(query, top_k, stage1_results, total_start, stage1_time)
        except Exception as e:
            self.logger.error(f"Error during search: {e}")
            raise

    def batch_search(self, queries: List[str], top_k: Optional[int] = None) -> List[Dict[str, Any]]:
        """Sequential, like the reference (:444-448); see search_many for the batched form."""
        return [self.search(q, top_k) for q in queries]

    def search_many(self, queries: List[str], top_k: Optional[int] = None) -> List[Dict[str, Any]]:
        """Same result records as batch_search, every stage batched over the queries: one
        bi-encoder pass and one sweep of the corpus per 64 queries (stage 1), one query forward
        and one MaxSim launch (stage 2, with the resident token store), one length-sorted pass
        of the cross-encoder over all (query, candidate) pairs (stage 3).  Stage times are
        reported as equal shares of the batch's stage times."""
        if not self.stage1 or not self.stage2 or not self.stage3:
            self.initialize_stages()
        top_k = top_k or self.config.stage3_top_k
        if not queries:
            return []
        queries = list(queries)
        n = len(queries)
        # a batch builds ~n x stage1_top_k candidate records; the cyclic collector would rescan
        # them on every allocation burst (measured: a list comprehension over 6400 pairs took 69 ms)
        import gc
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            fast = self._search_many_arrays(queries, top_k)
            if fast is not None:
                return fast
            total_start = self._now()
            t = self._now()
            s1 = self.stage1.search_many(queries, self.config.stage1_top_k)
            t1 = (time.time() - t) / n if t else None
            s2, s3, t2, t3 = self._later_stages_many(queries, s1)
            total = (time.time() - total_start) / n if total_start else None
            return self._assemble_many(queries, top_k, s1, s2, s3, t1, t2, t3, total)
        finally:
            if gc_was_on:
                gc.enable()

    def _search_many_arrays(self, queries: List[str], top_k: int) -> Optional[List[Dict[str, Any]]]:
        """search_many with every stage on ARRAYS (the candidate sets never become Python records): stage 1
        returns id / score matrices, stage 2 looks the candidates up in the resident token store and keeps the
        best by one stable device sort, stage 3 assembles its (query, document) inputs from cached token ids on
        the GPU; dictionaries are built only for what is returned (all three lists with
        ``save_intermediate_results``, else the final top_k per query).  Same records as the per-record path
        (tests).  None when a precondition is missing (HIP index on the device path, complete token store,
        complete stage-3 id cache, at least stage1_top_k documents) — the caller then takes the record path."""
        if not self._arrays_ready():
            return None
        n = len(queries)
        m0 = self._tick()
        # what the later stages need from the QUERIES alone is started first: the stage-2 query forward runs beside
        # stage 1's sweep of the corpus, stage 3's query tokens are ready when its pairs are assembled
        for st in (self.stage2, self.stage3):
            if hasattr(st, "prefetch_queries"):
                st.prefetch_queries(queries)
        mp = self._tick()
        got = self._arrays_stage1(queries)
        if got is None:
            return None
        ids1_dev, sc1 = got
        m1 = self._tick()
        later = self._arrays_stage23(queries, ids1_dev)
        if later is None:
            return None
        pos2_h, sc2_h, pos3_h, sc3_h, t2s, t3s = later       # (the host has waited for the GPU in there: marks are complete)
        t1 = self._span(mp, m1)
        t1 = t1 / n if t1 is not None else None
        pre = self._span(m0, mp)                              # the prefetched query work belongs to stages 2 / 3
        t2 = (t2s + (pre or 0.0)) / n if t2s is not None else None
        t3 = t3s / n if t3s is not None else None
        ids1_h = ids1_dev.cpu().numpy()
        sc1_h = sc1.cpu().numpy() if hasattr(sc1, "cpu") else sc1
        total = (time.time() - m0[0]) / n if m0 is not None else None
        return self._records_from_arrays(queries, top_k, ids1_h, sc1_h, pos2_h, sc2_h, pos3_h, sc3_h, t1, t2, t3, total)

    def _tick(self):
        """A stage boundary of the array path: (host time, HIP event recorded on the current stream) — NOT a
        synchronisation.  Round 2 synchronised the device at every boundary (six times per search): correct stage
        times, but the GPU drained six times per query and `search()` paid 1.5 ms of its 5.5 ms for it.  The stage
        times are now the GPU-side intervals between the events, read after the call's results have been copied
        out (`_span`); the total stays wall-clock."""
        if not self.config.enable_timing:
            return None
        import torch
        ev = None
        if torch.cuda.is_available():
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        return (time.time(), ev)

    @staticmethod
    def _span(a, b) -> Optional[float]:
        """Seconds between two _tick() marks (device time when both carry an event, else host time)."""
        if a is None or b is None:
            return None
        if a[1] is not None and b[1] is not None:
            b[1].synchronize()
            return a[1].elapsed_time(b[1]) / 1e3
        return b[0] - a[0]

    def _arrays_ready(self) -> bool:
        s1, s2, s3 = self.stage1, self.stage2, self.stage3
        return bool(getattr(s2, "token_store", None) is not None and len(s2.token_store)
                    and getattr(s3, "_pairs_usable", False) and hasattr(s1, "search_many_arrays"))

    def _arrays_stage1(self, queries: List[str]):
        """-> (ids int64 [B, k] on the token store's device, scores [B, k] tensor or numpy) or None."""
        import torch
        got = self.stage1.search_many_arrays(queries, self.config.stage1_top_k)
        if got is None:
            return None
        ids1, sc1 = got
        dev = self.stage2.token_store.data.device
        ids1_dev = ids1.to(dev) if torch.is_tensor(ids1) else torch.from_numpy(ids1).to(dev)
        return ids1_dev, sc1

    def _arrays_stage23(self, queries: List[str], ids1_dev):
        """Stages 2 and 3 of `queries` (row i of ids1_dev = the stage-1 ids of queries[i]) ->
        (pos2, sc2, pos3, sc3 as host arrays, stage-2 seconds, stage-3 seconds) or None."""
        import torch
        ma = self._tick()
        r2 = self.stage2.rescore_arrays(queries, ids1_dev, lazy=True)
        if r2 is None:
            return None
        pos2, sc2, bad2 = r2
        ids2_dev = torch.gather(ids1_dev.to(pos2.device), 1, pos2)
        mb = self._tick()
        r3 = self.stage3.rerank_arrays(queries, ids2_dev, lazy=True)
        if r3 is None:
            return None
        pos3, sc3, bad3 = r3
        mc = self._tick()
        out = (pos2.cpu().numpy(), sc2.cpu().numpy(), pos3.cpu().numpy(), sc3.cpu().numpy())
        if bool(bad2) or bool(bad3):     # a candidate outside the token store / id cache (looked at only now: the copies
            return None                  # above were the first time the host waited for the GPU since stage 3's batch plan)
        return out + (self._span(ma, mb), self._span(mb, mc))

    def _records_from_arrays(self, queries, top_k, ids1_h, sc1_h, pos2_h, sc2_h, pos3_h, sc3_h, t1, t2, t3, total):
        s1 = self.stage1
        n = len(queries)
        docs, meta = s1.documents, s1.doc_metadata
        keep = self.config.save_intermediate_results

        def rec1(q, j):
            i = int(ids1_h[q, j])
            s = float(sc1_h[q, j])
            return {"doc_id": i, "document": docs[i], "score": s, "stage1_score": s, "metadata": meta[i],
                    "stage": "stage1"}

        def rec2(q, j):
            r = rec1(q, int(pos2_h[q, j]))
            r["stage2_score"] = float(sc2_h[q, j])
            r["stage"] = "stage2"
            return r

        def rec3(q, j):
            r = rec2(q, int(pos3_h[q, j]))
            r["stage3_score"] = float(sc3_h[q, j])
            r["stage"] = "stage3"
            return r
        out1 = [[rec1(q, j) for j in range(ids1_h.shape[1])] if keep else None for q in range(n)]
        out2 = [[rec2(q, j) for j in range(pos2_h.shape[1])] if keep else None for q in range(n)]
        out3 = [[rec3(q, j) for j in range(pos3_h.shape[1])] for q in range(n)]
        if not keep:   # _assemble_many only tests the lists of a non-empty stage for truth
            out1 = [[True]] * n
            out2 = [[True]] * n
        return self._assemble_many(queries, top_k, out1, out2, out3, t1, t2, t3, total)

    def _later_stages_many(self, queries, s1):
        """Stages 2 and 3 for several queries -> (stage-2 lists, stage-3 lists, per-query time shares)."""
        n = max(len(queries), 1)
        t = self._now()
        s2 = self.stage2.rescore_many(queries, s1)
        t2 = (time.time() - t) / n if t else None
        t = self._now()
        s3 = self.stage3.rerank_many(queries, s2)
        t3 = (time.time() - t) / n if t else None
        return s2, s3, t2, t3

    def _assemble_many(self, queries, top_k, s1, s2, s3, t1, t2, t3, total):
        keep = self.config.save_intermediate_results
        out = []
        for q, r1, r2, r3 in zip(queries, s1, s2, s3):
            if not r1 or not r2:   # the reference's early returns (:363-371, :380-388)
                out.append({"query": q, "results": [], "stage1_results": r1 if r1 else [], "stage2_results": [],
                            "timing": ({"stage1_time": t1 or 0.0, "stage2_time": (t2 or 0.0) if r1 else 0.0,
                                        "stage3_time": 0.0, "total_time": total or 0.0}
                                       if self.config.enable_timing else {}),
                            "performance_stats": self.performance_stats})
                continue
            if self.config.enable_timing:
                self._update_performance_stats(t1, t2, t3, total)
            out.append({"query": q, "results": r3[:top_k],
                        "stage1_results": r1 if keep else [], "stage2_results": r2 if keep else [],
                        "timing": ({"stage1_time": t1 or 0.0, "stage2_time": t2 or 0.0, "stage3_time": t3 or 0.0,
                                    "total_time": total or 0.0} if self.config.enable_timing else {}),
                        "performance_stats": self.performance_stats.copy()})
        if self.config.auto_cleanup:
            self._cleanup_memory()
        return out

    # -- persistence -------------------------------------------------------------
    def save_index(self, index_path: Optional[str] = None):
        if not self.stage1:
            raise ValueError("Pipeline not initialized")
        if index_path is None:
            index_path = os.path.join(self.config.index_dir, "pipeline_index.pkl")
        self.stage1.save_index(index_path)
        if self.stage2 is not None and self.stage2.config.precompute_document_embeddings:
            self.stage2.save_token_store(self._token_store_path(index_path))
        self.logger.info(f"Pipeline index saved to {index_path}")

    @staticmethod
    def _token_store_path(index_path: str) -> str:
        return os.path.splitext(index_path)[0] + ".stage2_tokens.safetensors"

    def load_index(self, index_path: Optional[str] = None):
        if not self.stage1:
            self.initialize_stages()
        if index_path is None:
            index_path = os.path.join(self.config.index_dir, "pipeline_index.pkl")
        self.stage1.load_index(index_path)
        if self.stage2 is not None and self.stage2.config.precompute_document_embeddings and self.stage1.documents:
            # the resident stage-2 token store comes back from its file, or is re-encoded
            if not self.stage2.load_token_store(self._token_store_path(index_path), len(self.stage1.documents)):
                self.stage2.reset_token_store()
                self.stage2.index_documents(list(self.stage1.documents), 0)
        if self.stage3 is not None and self.config.stage3_cache_document_tokens and self.stage1.documents:
            self.stage3._pairs = None     # token ids are cheap to recompute: no file of their own
            self.stage3.index_documents(list(self.stage1.documents), 0)
        self.logger.info(f"Pipeline index loaded from {index_path}")

    # -- introspection -----------------------------------------------------------
    def get_pipeline_info(self) -> Dict[str, Any]:
        info = {"config": asdict(self.config),
                "stages_initialized": {"stage1": self.stage1 is not None, "stage2": self.stage2 is not None,
                                       "stage3": self.stage3 is not None},
                "performance_stats": self.performance_stats}
        if self.stage1:
            info["stage1_stats"] = self.stage1.get_stats()
        if self.stage2:
            info["stage2_info"] = self.stage2.get_model_info()
        if self.stage3:
            info["stage3_info"] = self.stage3.get_model_info()
        return info

    def _update_performance_stats(self, stage1_time, stage2_time, stage3_time, total_time):
        """Cumulative mean with alpha = 1/n and a 100-entry history (reference :567-606)."""
        ps = self.performance_stats
        ps["total_queries"] += 1
        alpha = 1.0 / ps["total_queries"]
        for key, val in (("avg_stage1_time", stage1_time), ("avg_stage2_time", stage2_time),
                         ("avg_stage3_time", stage3_time), ("avg_total_time", total_time)):
            ps[key] = (1 - alpha) * ps[key] + alpha * val
        ps["stage_time_history"].append({"stage1": stage1_time, "stage2": stage2_time,
                                         "stage3": stage3_time, "total": total_time})
        if len(ps["stage_time_history"]) > 100:
            ps["stage_time_history"] = ps["stage_time_history"][-100:]

    def _cleanup_memory(self):
        try:
            if self.stage2:
                self.stage2.clear_gpu_memory()
            if self.stage3:
                self.stage3.clear_gpu_memory()
        except Exception as e:
            self.logger.warning(f"Error during memory cleanup: {e}")

    def export_config(self, config_path: str):
        with open(config_path, "w") as f:
            yaml.dump({"pipeline": asdict(self.config)}, f, default_flow_style=False)
        self.logger.info(f"Configuration exported to {config_path}")
