"""Stage 3: cross-encoder reranking.

Mirror of reference src/stage3_reranker.py (``Stage3Config``,
``CrossEncoderReranker``, ``AdaptiveCrossEncoderReranker``; same method names,
result fields and arithmetic: pair preparation :113-118, activation :173-176,
min-max normalisation :212-228, stable descending sort and top_k_final :256-260,
adaptive batch size :328-344).  The forward pass is a batched bf16 run of the
sequence-classification model on PyTorch-ROCm (tristage_rag_amd.encoders.
CrossEncoderModel, the restatement of sentence_transformers.CrossEncoder the
reference prefers at :65-70), with one device->host copy per rerank call.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class Stage3Config:
    model_name: str = "cross-encoder/ms-marco-MiniLM-L6-v2"
    device: str = "auto"
    cache_dir: str = "./models"
    max_length: int = 256
    batch_size: int = 32
    top_k_final: int = 20
    use_fp16: bool = True
    use_gpu_if_available: bool = True
    activation_fxn: str = "sigmoid"  # "sigmoid" or "softmax" (HF path)
    normalize_scores: bool = True
    # additive
    use_hip_graph: bool = False  # one query's pairs as a single forward replayed from a HIP graph
    many_batch_size: int = 1024  # pairs per forward when several queries are reranked together (rerank_many)
    amp_dtype: str = "bf16"      # what use_fp16 means on the GPU: "bf16" or "fp16" (the reference's autocast)
    many_width_multiple: int = 1  # rerank_arrays pads the token width of a batch to a multiple of this (fewer GEMM shapes)
    many_packed_batch_size: int = 4096   # ... pairs per forward of a PACKED batch (no padding to pay for a wide length range;
                                         # 1024 -> 4096: 1132 -> 1200 queries/s on the configs[2] shape)
    many_packed: bool = True      # rerank_arrays runs PACKED batches (the pairs' tokens concatenated, no padded position
                                  # computed) when the written-out forward with its HIP kernels is in use


class CrossEncoderReranker:
    """Stage 3: cross-encoder reranker for the final ranking."""

    def __init__(self, config: Stage3Config, model: Any = None):
        self.config = config
        self.logger = logging.getLogger(__name__)
        self.model = model
        self.tokenizer = None
        self.device = self._get_device()
        self._load_model()

    def _get_device(self) -> str:
        if self.config.device == "auto":
            return "cuda" if (torch.cuda.is_available() and self.config.use_gpu_if_available) else "cpu"
        return self.config.device

    def _load_model(self) -> None:
        if self.model is None:
            from .encoders import CrossEncoderModel
            self.logger.info(f"Loading Stage 3 model: {self.config.model_name}")
            self.model = CrossEncoderModel(self.config.model_name, device=self.device,
                                           max_length=self.config.max_length,
                                           cache_folder=self.config.cache_dir,
                                           use_amp=self.config.use_fp16, amp_dtype=self._amp_dtype(),
                                           use_hip_graph=self.config.use_hip_graph)
        # CrossEncoder-style object (predict on sentence pairs), like the reference's preferred path
        self.use_sentence_transformers = hasattr(self.model, "predict")
        if not self.use_sentence_transformers:
            self.tokenizer = getattr(self.model, "tokenizer", None)
        self.use_amp = self.config.use_fp16 and str(self.device).startswith("cuda")

    def _amp_dtype(self):
        from .stage1_retriever import amp_torch_dtype
        return amp_torch_dtype(getattr(self.config, "amp_dtype", "bf16"))

    def _prepare_input_pairs(self, query: str, documents: List[str]) -> List[Tuple[str, str]]:
        return [(query, doc) for doc in documents]

    def _predict_with_sentence_transformers(self, pairs: List[Tuple[str, str]]) -> List[float]:
        """reference :120-137"""
        sentence_pairs = [[q, d] for q, d in pairs]
        scores = self.model.predict(sentence_pairs, batch_size=self.config.batch_size, show_progress_bar=False)
        return np.asarray(scores).tolist()

    def _predict_with_huggingface(self, pairs: List[Tuple[str, str]]) -> List[float]:
        """reference :139-190 — raw HF sequence-classification model + tokenizer."""
        all_scores: List[float] = []
        for i in range(0, len(pairs), self.config.batch_size):
            batch = pairs[i:i + self.config.batch_size]
            enc = self.tokenizer([p[0] for p in batch], [p[1] for p in batch], truncation=True, padding=True,
                                 max_length=self.config.max_length, return_tensors="pt")
            enc = {k: v.to(self.device) for k, v in enc.items()}
            with torch.no_grad():
                if self.use_amp:
                    with torch.autocast("cuda", dtype=self._amp_dtype()):
                        logits = self.model(**enc).logits
                else:
                    logits = self.model(**enc).logits
                if self.config.activation_fxn == "sigmoid":
                    scores = torch.sigmoid(logits.float()).squeeze(-1)
                else:
                    scores = F.softmax(logits.float(), dim=-1)[:, 1]
                all_scores.extend(scores.cpu().tolist())
        return all_scores

    def raw_scores(self, query: str, documents: List[str]) -> List[float]:
        """Activated model scores of (query, doc) pairs, before the min-max step."""
        pairs = self._prepare_input_pairs(query, documents)
        return (self._predict_with_sentence_transformers(pairs) if self.use_sentence_transformers
                else self._predict_with_huggingface(pairs))

    def predict(self, query: str, documents: List[str]) -> List[float]:
        """reference :192-210"""
        if not documents:
            return []
        scores = self.raw_scores(query, documents)
        if self.config.normalize_scores:
            scores = self._normalize_scores(scores)
        return scores

    def _normalize_scores(self, scores: List[float]) -> List[float]:
        """reference :212-228"""
        if not scores:
            return scores
        a = np.array(scores)
        mn, mx = a.min(), a.max()
        normalized = (a - mn) / (mx - mn) if mx > mn else np.zeros_like(a)
        return normalized.tolist()

    def rerank(self, query: str, candidates: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
        """reference :230-264"""
        if not candidates:
            return []
        self.logger.info(f"Reranking {len(candidates)} candidates with Stage 3")
        scores = self.predict(query, [c["document"] for c in candidates])
        reranked = []
        for cand, s in zip(candidates, scores):
            u = cand.copy()
            u["stage3_score"] = s
            u["stage"] = "stage3"
            reranked.append(u)
        reranked.sort(key=lambda x: x["stage3_score"], reverse=True)
        final = reranked[: self.config.top_k_final]
        self.logger.info(f"Stage 3 reranking completed. Top score: {final[0]['stage3_score'] if final else 0:.4f}")
        return final

    def rerank_many(self, queries: List[str], candidates_list: List[List[Dict[str, Any]]]) -> List[List[Dict[str, Any]]]:
        """rerank() for several queries with ONE pass of the cross-encoder over all their pairs
        (length-sorted, large batches); min-max, stable sort and top_k_final stay per query."""
        if len(queries) != len(candidates_list):
            raise ValueError("Number of queries must match number of candidate lists")
        pairs: List[Tuple[str, str]] = []
        bounds = [0]
        for q, cands in zip(queries, candidates_list):
            pairs.extend(self._prepare_input_pairs(q, [c["document"] for c in cands]))
            bounds.append(len(pairs))
        if not pairs:
            return [[] for _ in queries]
        original = self.config.batch_size
        self.config.batch_size = max(original, self.config.many_batch_size)
        try:
            raw = (self._predict_with_sentence_transformers(pairs) if self.use_sentence_transformers
                   else self._predict_with_huggingface(pairs))
        finally:
            self.config.batch_size = original
        out = []
        for cands, a, b in zip(candidates_list, bounds[:-1], bounds[1:]):
            if not cands:
                out.append([])
                continue
            scores = raw[a:b]
            if self.config.normalize_scores:
                scores = self._normalize_scores(scores)
            reranked = []
            for cand, s in zip(cands, scores):
                u = cand.copy()
                u["stage3_score"] = s
                u["stage"] = "stage3"
                reranked.append(u)
            reranked.sort(key=lambda x: x["stage3_score"], reverse=True)
            out.append(reranked[: self.config.top_k_final])
        return out

    # -- token-id level path (additive): documents tokenised once, pairs assembled on the GPU ------------
    def index_documents(self, documents: List[str], first_doc_id: int) -> bool:
        """Tokenise `documents` once and keep their ids for rerank_arrays(); document j gets pipeline id
        first_doc_id + j.  The FIRST call fixes the id of slot 0 (0 for a whole corpus, the first row of a row shard:
        parallel_pipeline.ShardedRetrievalPipeline); later calls must continue where the previous one ended.  False
        (and the id cache is unusable) when the tokenizer cannot be restated on the device (encoders.PairAssembler
        probes it) or the ids are not consecutive."""
        from .encoders import CrossEncoderModel, PairAssembler
        if not isinstance(self.model, CrossEncoderModel):
            self._pairs = None
            return False
        pa = getattr(self, "_pairs", None)
        if pa is None:
            pa = PairAssembler(self.model.tokenizer, self.config.max_length)
            self._pairs = pa
            self._pairs_base = int(first_doc_id)
        if not pa.ok or first_doc_id != self._pairs_base + len(pa):
            self._pairs_usable = False
            return False
        pa.add_documents(documents)
        self._pairs_usable = True
        return True

    def prefetch_queries(self, queries: List[str]) -> None:
        """Tokenise the queries of a coming rerank_arrays / raw_arrays_partial call now (host work that needs nothing
        from stages 1 and 2: the pipeline does it while the GPU is busy with them)."""
        pa = getattr(self, "_pairs", None)
        if pa is not None and getattr(self, "_pairs_usable", False):
            qs = list(queries)
            self._q_prefetch = (qs, [pa.ids_of(q) for q in qs])

    def raw_arrays_partial(self, queries: List[str], doc_ids: torch.Tensor, compact: Optional[bool] = None) -> torch.Tensor:
        """Activated cross-encoder scores float32 [B, C] of the pairs (queries[q], document doc_ids[q, j]) whose
        document lives in THIS process's token-id cache; -inf everywhere else.  The (query, document) inputs are
        assembled from cached token ids on the GPU with the tokenizer's own truncation and run through the
        cross-encoder in length-sorted batches of ``many_batch_size`` pairs (packed when the written-out forward
        takes them); no collective.  With a row-sharded cache every pair is owned by exactly one rank.
        ``compact`` (default ``self.owner_compact``): True = only the owned pairs run (their count is a host sync);
        False = all B x C pairs run, a pair whose document is not cached reads slot 0 and is masked afterwards."""
        pa = getattr(self, "_pairs", None)
        dev = torch.device(self.model.device) if hasattr(self.model, "device") else doc_ids.device
        doc_ids = doc_ids.to(dev)
        B, C = doc_ids.shape
        raw_full = torch.full((B * C,), float("-inf"), dtype=torch.float32, device=dev)
        if pa is None or not getattr(self, "_pairs_usable", False) or not len(pa) or B == 0 or C == 0:
            return raw_full.view(B, C)
        base = int(getattr(self, "_pairs_base", 0))
        flat_ids = doc_ids.reshape(-1)
        owned = (flat_ids >= base) & (flat_ids < base + len(pa))
        whole = B * C
        if compact is None:
            compact = bool(getattr(self, "owner_compact", False))
        if compact:
            idx = torch.nonzero(owned).flatten()      # positions (row-major) of the owned pairs
            P = int(idx.numel())                      # host sync
            if P == 0:
                self._q_prefetch = None
                return raw_full.view(B, C)
            pair_q = torch.div(idx, C, rounding_mode="floor")
            pair_slot = flat_ids[idx] - base
        else:
            idx, P = None, whole
            pair_q = torch.arange(B, device=dev).repeat_interleave(C)
            pair_slot = (flat_ids - base).clamp(0, len(pa) - 1)
        pre = getattr(self, "_q_prefetch", None)
        self._q_prefetch = None
        q_ids = pre[1] if pre is not None and pre[0] == list(queries) else [pa.ids_of(q) for q in queries]
        plan = pa.plan(q_ids, pair_q, pair_slot, dev)
        order = torch.argsort(plan["total"], descending=True, stable=True)
        raw = torch.empty((P,), dtype=torch.float32, device=dev)
        wm = max(int(getattr(self.config, "many_width_multiple", 1) or 1), 1)
        bs = max(self.config.batch_size, self.config.many_batch_size)
        # packed batches (no padded position in the GEMMs, LayerNorms or GELU) whenever the written-out forward takes them
        # one query's pairs: the replayed graph wins — if they fit its largest bucket (a search_many batch of a few dozen queries
        # is "all pairs in one batch" too, but thousands of rows: those go through the packed forward)
        from .encoders import GraphedClassifier
        graph_rows = GraphedClassifier.ROWS[-1]
        graph_one = P <= min(bs, graph_rows) and P == whole and getattr(self.model, "use_hip_graph", False)
        packed = (wm == 1 and not graph_one and getattr(self.config, "many_packed", True) and hasattr(self.model, "packed_ok")
                  and self.model.packed_ok(int(self.config.max_length), min(max(bs, int(getattr(self.config, "many_packed_batch_size", bs))), P)))
        if packed:
            bs = max(bs, int(getattr(self.config, "many_packed_batch_size", bs)))
        tot = plan["total"][order]
        csum = torch.cumsum(tot, 0)
        ends = torch.arange(bs, P + bs, bs, device=dev).clamp(max=P) - 1
        host = torch.stack([tot[::bs], csum[ends]]).tolist()      # one host sync: every batch's padded width and token count
        widths = host[0]
        tokens = [int(e - s) for e, s in zip(host[1], [0] + host[1][:-1])]
        if wm > 1:                                            # (padding is masked: same scores up to batch-padding noise)
            widths = [min(-(-int(w) // wm) * wm, max(int(self.config.max_length), int(w))) for w in widths]
        for j, s in enumerate(range(0, P, bs)):
            sel = order[s: s + bs]
            if packed:
                enc = pa.batch_packed(plan, sel, tokens[j], int(widths[j]))
                raw[sel] = self.model.activate(self.model.logits_from_ids(enc)).reshape(-1)
                continue
            enc = pa.batch(plan, sel, width=int(widths[j]))
            lg = self.model.logits_graphed(enc) if (P <= bs and P == whole) else None   # one query's pairs, all of them here: graph replay if enabled
            raw[sel] = self.model.activate(lg if lg is not None else self.model.logits_from_ids(enc)).reshape(-1)
        if idx is None:
            return torch.where(owned, raw, raw_full).view(B, C)
        raw_full[idx] = raw
        return raw_full.view(B, C)

    def finish_arrays(self, raw: torch.Tensor):
        """raw activated scores [B, C] -> (pos int64 [B, keep], scores float64 [B, keep]): min-max per query in
        float64 like reference :212-228 (all-equal -> zeros), stable descending sort (:256-260), top_k_final."""
        a = raw.to(torch.float64)
        if self.config.normalize_scores:
            mn, mx = a.min(dim=1, keepdim=True).values, a.max(dim=1, keepdim=True).values
            a = torch.where(mx > mn, (a - mn) / (mx - mn), torch.zeros_like(a))
        srt, pos = torch.sort(a, dim=1, descending=True, stable=True)
        keep = min(self.config.top_k_final, raw.shape[1])
        return pos[:, :keep].contiguous(), srt[:, :keep].contiguous()

    def rerank_arrays(self, queries: List[str], doc_ids: torch.Tensor, lazy: bool = False):
        """rerank_many on arrays: doc_ids int64 [B, C] (CUDA; row q = the stage-2 list of query q, in stage-2
        order) -> (pos int64 [B, keep], scores float64 [B, keep]): positions in stage-2 order of the
        ``top_k_final`` best by stage-3 score (min-max normalised per query like reference :212-228, stable
        descending sort :256-260).  None when the id cache does not cover the documents."""
        pa = getattr(self, "_pairs", None)
        if pa is None or not getattr(self, "_pairs_usable", False) or doc_ids.dim() != 2:
            return None
        B, C = doc_ids.shape
        if B == 0 or C == 0:
            return None
        raw = self.raw_arrays_partial(queries, doc_ids)
        bad = torch.isinf(raw).any()                  # a document outside the id cache
        if lazy:                                      # third result: 0-dim bool tensor, looked at by the caller (no sync here)
            return self.finish_arrays(raw) + (bad,)
        if bool(bad):
            return None
        return self.finish_arrays(raw)

    def batch_rerank(self, queries: List[str], candidates_list: List[List[Dict[str, Any]]]):
        if not queries or not candidates_list:
            return []
        if len(queries) != len(candidates_list):
            raise ValueError("Number of queries must match number of candidate lists")
        return [self.rerank(q, c) for q, c in zip(queries, candidates_list)]

    def get_model_info(self) -> Dict[str, Any]:
        info = {"model_name": self.config.model_name, "device": self.device,
                "max_length": self.config.max_length, "batch_size": self.config.batch_size,
                "use_fp16": self.use_amp, "activation_function": self.config.activation_fxn,
                "normalize_scores": self.config.normalize_scores, "top_k_final": self.config.top_k_final}
        if self.use_sentence_transformers:
            info["model_type"] = "CrossEncoder (tristage_rag_amd.encoders.CrossEncoderModel)"
        else:
            info["model_type"] = "HuggingFace AutoModel"
            info["num_labels"] = getattr(self.model, "num_labels", None)
        return info

    def clear_gpu_memory(self):
        return None  # see ColBERTScorer.clear_gpu_memory


class AdaptiveCrossEncoderReranker(CrossEncoderReranker):
    """Reranker that adapts its batch size to the input length (reference :321-367)."""

    def __init__(self, config: Stage3Config, model: Any = None):
        super().__init__(config, model=model)
        self.max_text_length = config.max_length // 2

    def _adaptive_batch_size(self, texts: List[str]) -> int:
        texts = [t for t in texts if t is not None]   # (a row-sharded pipeline: text lives with its owner rank only)
        if not texts:
            return self.config.batch_size
        avg = sum(len(t.split()) for t in texts) / len(texts)
        if avg > 200:
            return max(4, self.config.batch_size // 4)
        elif avg > 100:
            return max(8, self.config.batch_size // 2)
        elif avg > 50:
            return max(16, self.config.batch_size // 1)
        return self.config.batch_size

    def rerank(self, query: str, candidates: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
        if not candidates:
            return []
        documents = [c["document"] for c in candidates]
        adaptive = self._adaptive_batch_size([query] + documents)
        original = self.config.batch_size
        self.config.batch_size = adaptive  # not re-entrant, exactly like the reference (:358-365)
        try:
            return super().rerank(query, candidates)
        finally:
            self.config.batch_size = original
