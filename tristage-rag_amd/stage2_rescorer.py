"""Stage 2: ColBERT-style MaxSim rescoring, batched on the GPU.

Mirror of reference src/stage2_rescorer.py (class ``ColBERTScorer``, config
``Stage2Config``, same method names and result fields).  The arithmetic is the
reference's — token embeddings = ``last_hidden_state`` of a plain ``AutoModel``
truncated to each text's true length (:134-165, :207-242), score =
mean/softmax-weighted max cosine (:167-201), stable descending sort, keep
``top_k_candidates`` (:293-297) — but instead of one tiny matmul plus one
``.item()`` host sync per candidate (:268-276) all candidates of a query are
scored by ONE launch of the HIP MaxSim kernel (ts_maxsim) on their packed token
matrices, and there is one device->host copy per query.

Optional (``cache_document_embeddings``): token matrices are kept per document
text, so a document is encoded once instead of once per query that retrieves it
(SURVEY.md §8f-2); scores are unchanged up to batch-padding noise.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional

import numpy as np
import torch


@dataclass
class Stage2Config:
    model_name: str = "lightonai/GTE-ModernColBERT-v1"
    device: str = "auto"
    cache_dir: str = "./models"
    max_seq_length: int = 192
    batch_size: int = 16
    top_k_candidates: int = 100
    use_fp16: bool = True
    pooling_method: str = "cls"  # "cls", "mean", or "max"
    normalize_embeddings: bool = True
    scoring_method: str = "maxsim"  # "maxsim" or "colbert"
    use_gpu_if_available: bool = True
    # additive
    cache_document_embeddings: bool = False      # memoise token matrices per document text
    precompute_document_embeddings: bool = False  # resident token store filled at add time
    token_store_dtype: str = "auto"               # storage type of the resident token store: "auto" = bf16 when the
                                                  # encoder runs under AMP (its last LayerNorm hands back fp32 even
                                                  # then), else the encoder's own output type; or "bf16" | "f16" | "f32"
    use_hip_graph: bool = False                   # replay the batch-1 query forward from a HIP graph
    amp_dtype: str = "bf16"                       # what use_fp16 means on the GPU: "bf16" or "fp16" (the reference's autocast)
    index_batch_size: int = 256                   # documents per forward wherever MANY documents are encoded: filling the
                                                  # token store at add time, re-encoding a query's candidates without one
                                                  # (batch_size, the reference's 16, keeps both launch-bound)


class TokenStore:
    """Token matrices of indexed documents, resident on the GPU as one [rows, H]
    tensor plus per-document (start, length); grown geometrically.  Stage 2 then
    reads candidates in place (ts_maxsim_indexed) instead of re-encoding them for
    every query (reference src/stage2_rescorer.py:254-259)."""

    def __init__(self):
        self.data: Optional[torch.Tensor] = None
        self.rows = 0
        self.starts: List[int] = []
        self.lens: List[int] = []
        self._starts_dev: Optional[torch.Tensor] = None
        self._lens_dev: Optional[torch.Tensor] = None

    def __len__(self) -> int:
        return len(self.starts)

    def append(self, mats: List[torch.Tensor]) -> None:
        if not mats:
            return
        add = sum(int(m.shape[0]) for m in mats)
        H = int(mats[0].shape[1])
        if self.data is None or self.rows + add > self.data.shape[0]:
            cap = max(self.rows + add, int(1.5 * (self.data.shape[0] if self.data is not None else 0)), 1024)
            new = torch.empty((cap, H), dtype=mats[0].dtype, device=mats[0].device)
            if self.data is not None and self.rows:
                new[: self.rows].copy_(self.data[: self.rows])
            self.data = new
        for m in mats:
            n = int(m.shape[0])
            self.data[self.rows: self.rows + n].copy_(m)
            self.starts.append(self.rows)
            self.lens.append(n)
            self.rows += n
        self._starts_dev = None

    def append_packed(self, rows: torch.Tensor, lens: List[int]) -> None:
        """`rows` [sum(lens), H]: the token rows of len(lens) documents, back to back."""
        add = int(rows.shape[0])
        if not lens:
            return
        if self.data is None or self.rows + add > self.data.shape[0]:
            cap = max(self.rows + add, int(1.5 * (self.data.shape[0] if self.data is not None else 0)), 1024)
            new = torch.empty((cap, int(rows.shape[1])), dtype=rows.dtype, device=rows.device)
            if self.data is not None and self.rows:
                new[: self.rows].copy_(self.data[: self.rows])
            self.data = new
        self.data[self.rows: self.rows + add].copy_(rows)
        at = self.rows
        for n in lens:
            self.starts.append(at)
            self.lens.append(int(n))
            at += int(n)
        self.rows += add
        self._starts_dev = None

    def device_tables(self):
        if self._starts_dev is None:
            dev = self.data.device
            self._starts_dev = torch.tensor(self.starts, dtype=torch.int64, device=dev)
            self._lens_dev = torch.tensor(self.lens, dtype=torch.int32, device=dev)
        return self._starts_dev, self._lens_dev


class ColBERTScorer:
    """ColBERT-style MaxSim scoring for multi-vector retrieval."""

    def __init__(self, config: Stage2Config, model: Any = None, tokenizer: Any = None,
                 maxsim_fn: Optional[Callable] = None, maxsim_indexed_fn: Optional[Callable] = None,
                 maxsim_indexed_batch_fn: Optional[Callable] = None):
        self.config = config
        self.logger = logging.getLogger(__name__)
        self.model = model
        self.tokenizer = tokenizer
        self.device = self._get_device()
        self._maxsim_fn = maxsim_fn
        self._maxsim_indexed_fn = maxsim_indexed_fn
        self._maxsim_indexed_batch_fn = maxsim_indexed_batch_fn
        self._doc_cache: Dict[str, torch.Tensor] = {}
        self.token_store = TokenStore()        # filled by index_documents()
        self._store_slot: Dict[int, int] = {}  # pipeline doc_id -> slot in the store
        self._slot_version = 0                 # bumped whenever _store_slot is replaced or extended (see _slot_table)
        self._graphed = None
        self._load_model()

    def _get_device(self) -> str:
        if self.config.device == "auto":
            return "cuda" if (torch.cuda.is_available() and self.config.use_gpu_if_available) else "cpu"
        return self.config.device

    def _load_model(self) -> None:
        if self.model is None or self.tokenizer is None:
            from .encoders import load_backbone
            self.logger.info(f"Loading Stage 2 model: {self.config.model_name}")
            self.tokenizer, self.model, _ = load_backbone(self.config.model_name, self.config.cache_dir, "base")
        self.model.to(self.device)
        self.model.eval()
        self.use_amp = self.config.use_fp16 and str(self.device).startswith("cuda")

    # -- encoding ------------------------------------------------------------
    def _model_inputs(self, enc: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        keep = ("input_ids", "attention_mask", "token_type_ids")
        from .encoders import to_device_async
        out = {k: to_device_async(v, self.device) for k, v in enc.items() if k in keep}
        if "token_type_ids" in out and not hasattr(self.model.config, "type_vocab_size"):
            out.pop("token_type_ids")
        return out

    def _amp_dtype(self):
        from .stage1_retriever import amp_torch_dtype
        return amp_torch_dtype(getattr(self.config, "amp_dtype", "bf16"))

    def _forward(self, enc: Dict[str, torch.Tensor]) -> torch.Tensor:
        lengths = enc.get("lengths")
        enc = {k: v for k, v in enc.items() if k not in ("lengths", "lengths_host")}
        with torch.no_grad():
            if self.use_amp:
                if getattr(self, "lean_forward", True):
                    from .encoders import lean_encoder_for   # BERT-family / ModernBERT token encoders: the written-out forward
                    lean = lean_encoder_for(self.model, self._amp_dtype())
                    if lean:
                        return lean(enc["input_ids"], enc["attention_mask"], enc.get("token_type_ids"), lengths=lengths)
                with torch.autocast("cuda", dtype=self._amp_dtype()):
                    return self.model(**enc).last_hidden_state
            return self.model(**enc).last_hidden_state

    def _tokenize_batch(self, texts: List[str]) -> Dict[str, torch.Tensor]:
        """reference :100-113"""
        texts = [t if t and t.strip() else "empty" for t in texts]
        enc = self.tokenizer(texts, truncation=True, padding=True, max_length=self.config.max_seq_length,
                             return_tensors="pt")
        from .encoders import _prefix_lengths, to_device_async
        lengths = _prefix_lengths(enc, self.tokenizer)      # on the host: the written-out forward then needs no mask check
        out = self._model_inputs(enc)
        if lengths is not None:
            out["lengths"] = to_device_async(lengths, self.device)
            out["lengths_host"] = [int(x) for x in lengths.tolist()]   # the same counts for the host: no .tolist() of a device tensor later
        return out

    def _pool_embeddings(self, embeddings: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        """reference :115-132 (not used by the scoring path; kept for API parity)"""
        if self.config.pooling_method == "cls":
            return embeddings[:, 0, :]
        if self.config.pooling_method == "mean":
            m = attention_mask.unsqueeze(-1).expand(embeddings.size()).float()
            return torch.sum(embeddings * m, 1) / torch.clamp(m.sum(1), min=1e-9)
        if self.config.pooling_method == "max":
            m = attention_mask.unsqueeze(-1).expand(embeddings.size()).float()
            embeddings = embeddings.masked_fill(m == 0, -1e9)
            return torch.max(embeddings, 1)[0]
        raise ValueError(f"Unknown pooling method: {self.config.pooling_method}")

    def _encode_single_text(self, text: str) -> torch.Tensor:
        """reference :134-165 -> [1, L, H]"""
        if not text or not text.strip():
            text = "empty"
        enc = self.tokenizer(text, truncation=True, max_length=self.config.max_seq_length,
                             return_tensors="pt", padding=False)
        enc = self._model_inputs(enc)
        if self.config.use_hip_graph and str(self.device).startswith("cuda"):
            if self._graphed is None:
                from .encoders import GraphedForward
                self._graphed = GraphedForward(self.model, getattr(self.tokenizer, "pad_token_id", 0),
                                               self._amp_dtype() if self.use_amp else None)
            return self._graphed(enc["input_ids"], enc["attention_mask"])  # unpadded input: all tokens valid
        hidden = self._forward(enc)
        n = int(enc["attention_mask"].sum().item())
        return hidden[:, :n, :]

    def encode_query(self, query: str) -> torch.Tensor:
        return self._encode_single_text(query)

    def encode_single_document(self, document: str) -> torch.Tensor:
        return self._encode_single_text(document)

    def encode_documents_batch(self, documents: List[str]) -> List[torch.Tensor]:
        """reference :207-242 -> list of [L_i, H] (true lengths, padding cut off)."""
        out: List[Optional[torch.Tensor]] = [None] * len(documents)
        todo = list(range(len(documents)))
        if self.config.cache_document_embeddings:
            todo = []
            for i, d in enumerate(documents):
                hit = self._doc_cache.get(d)
                if hit is not None:
                    out[i] = hit
                else:
                    todo.append(i)
        # (the reference encodes candidates 16 at a time, :207-242; on this GPU that is 63 launch-bound forwards for a
        # query's 1000 candidates: length-sorted batches of index_batch_size instead, same values up to padding noise)
        bs = max(self.config.batch_size, getattr(self.config, "index_batch_size", 0) or 0, 1)
        if bs > self.config.batch_size:
            todo = sorted(todo, key=lambda i: -len(documents[i]))
        for s in range(0, len(todo), bs):
            idx = todo[s:s + bs]
            enc = self._tokenize_batch([documents[i] for i in idx])
            hidden = self._forward(enc)
            lens = enc["attention_mask"].sum(dim=1).tolist()  # one sync per batch, not per document
            for j, i in enumerate(idx):
                e = hidden[j, :int(lens[j]), :]
                out[i] = e
                if self.config.cache_document_embeddings:
                    self._doc_cache[documents[i]] = e
        return out  # type: ignore[return-value]

    # -- scoring -------------------------------------------------------------
    def _maxsim_score(self, query_embeddings: torch.Tensor, doc_embeddings: torch.Tensor) -> torch.Tensor:
        """reference :167-183 for ONE pair — same HIP kernel as the batched path."""
        return torch.tensor(self.score_all(query_embeddings, [doc_embeddings.reshape(-1, doc_embeddings.shape[-1])],
                                           mode="maxsim")[0])

    def _colbert_score(self, query_embeddings: torch.Tensor, doc_embeddings: torch.Tensor) -> torch.Tensor:
        """reference :185-201 for ONE pair — same HIP kernel as the batched path."""
        return torch.tensor(self.score_all(query_embeddings, [doc_embeddings.reshape(-1, doc_embeddings.shape[-1])],
                                           mode="colbert")[0])

    def score_all(self, query_embeddings: torch.Tensor, doc_embeddings_list: List[torch.Tensor],
                  mode: Optional[str] = None) -> List[float]:
        """Scores of every candidate with one kernel launch and one host copy."""
        if not doc_embeddings_list:
            return []
        # 16-bit token matrices (bf16 autocast) stay 16-bit: half the bytes and the streaming kernel;
        # the products are exact either way, so the scores are the ones the fp32 up-cast would give
        dts = {d.dtype for d in doc_embeddings_list} | {query_embeddings.dtype}
        dt = next(iter(dts)) if len(dts) == 1 and next(iter(dts)) in (torch.float16, torch.bfloat16) else torch.float32
        q = query_embeddings.squeeze(0).to(dt).contiguous()
        lens = [int(d.shape[0]) for d in doc_embeddings_list]
        off = torch.zeros(len(lens) + 1, dtype=torch.int32)
        off[1:] = torch.cumsum(torch.tensor(lens, dtype=torch.int32), 0)
        packed = torch.cat([d.reshape(-1, q.shape[1]) for d in doc_embeddings_list], 0).to(dt).contiguous()
        fn = self._maxsim_fn
        if fn is None:
            from .index import maxsim  # HIP kernel; raises without the library or a GPU
            fn = maxsim
        scores = fn(q, packed, off.to(q.device), mode or self.config.scoring_method)
        return [float(x) for x in scores.detach().cpu().tolist()]

    def store_dtype(self) -> Optional[torch.dtype]:
        name = self.config.token_store_dtype
        if name == "auto":
            return self._amp_dtype() if self.use_amp else None
        return {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[name]

    def index_documents(self, documents: List[str], first_doc_id: int) -> None:
        """Encode `documents` once and keep their token matrices on the GPU; document j
        gets pipeline id first_doc_id + j (the doc_id stage 1 reports).  Same tokenisation, padding
        and forward as encode_documents_batch (reference :207-242); the valid rows of a batch go
        into the store with ONE masked copy (row-major mask order = document order)."""
        bs = max(self.config.batch_size, getattr(self.config, "index_batch_size", 0) or 0, 1)
        dt = self.store_dtype()
        # length-sorted batches (little padding in the GEMMs); the slot table remembers where each document went
        order = sorted(range(len(documents)), key=lambda i: -len(documents[i])) if bs > self.config.batch_size \
            else list(range(len(documents)))
        for s in range(0, len(order), bs):
            idx = order[s: s + bs]
            enc = self._tokenize_batch([documents[i] for i in idx])
            hidden = self._forward(enc)
            mask = enc["attention_mask"].bool()
            rows = hidden[mask]                                  # [sum(lens), H]
            if dt is not None and rows.dtype != dt:
                rows = rows.to(dt)
            base = len(self.token_store)
            self.token_store.append_packed(rows, mask.sum(dim=1).tolist())
            for j, i in enumerate(idx):
                self._store_slot[first_doc_id + i] = base + j
        self._slot_version += 1

    # -- persistence of the token store (additive; the reference has nothing to persist for stage 2)
    def save_token_store(self, path: str) -> bool:
        """Token matrices + (start, length, doc id) tables as one safetensors file (data only)."""
        if not len(self.token_store):
            return False
        from safetensors.torch import save_file
        st = self.token_store
        ids = sorted(self._store_slot, key=self._store_slot.get)
        save_file({"tokens": st.data[: st.rows].contiguous().cpu(),
                   "starts": torch.tensor(st.starts, dtype=torch.int64),
                   "lens": torch.tensor(st.lens, dtype=torch.int32),
                   "doc_ids": torch.tensor(ids, dtype=torch.int64)}, path,
                  metadata={"format": "tristage-rag_amd/token-store/1", "model": str(self.config.model_name),
                            "max_seq_length": str(self.config.max_seq_length)})
        return True

    def load_token_store(self, path: str, expected_docs: Optional[int] = None) -> bool:
        """Restore what save_token_store wrote.  False (store left empty) when the file is absent,
        was produced by another model / sequence length, or does not cover `expected_docs`."""
        import os
        if not os.path.exists(path):
            return False
        from safetensors import safe_open
        with safe_open(path, framework="pt", device="cpu") as f:
            meta = f.metadata() or {}
            if (meta.get("format") != "tristage-rag_amd/token-store/1" or meta.get("model") != str(self.config.model_name)
                    or meta.get("max_seq_length") != str(self.config.max_seq_length)):
                return False
            lens = f.get_tensor("lens")
            if expected_docs is not None and int(lens.numel()) != int(expected_docs):
                return False
            tokens, starts, ids = f.get_tensor("tokens"), f.get_tensor("starts"), f.get_tensor("doc_ids")
        st = TokenStore()
        st.data = tokens.to(self.device)
        st.rows = int(tokens.shape[0])
        st.starts, st.lens = [int(x) for x in starts.tolist()], [int(x) for x in lens.tolist()]
        self.token_store = st
        self._store_slot = {int(d): j for j, d in enumerate(ids.tolist())}
        self._slot_version += 1
        return True

    def _score_from_store(self, query_embeddings: torch.Tensor, candidates: List[Dict[str, Any]]):
        """Scores straight from the resident token store, or None if a candidate is not in it."""
        if not len(self.token_store):
            return None
        slots = [self._store_slot.get(c.get("doc_id"), -1) for c in candidates]
        if min(slots) < 0:
            return None
        starts_all, lens_all = self.token_store.device_tables()
        sel = torch.tensor(slots, dtype=torch.int64, device=starts_all.device)
        q = query_embeddings.squeeze(0).to(self.token_store.data.dtype).contiguous()
        fn = self._maxsim_indexed_fn
        if fn is None:
            from .index import maxsim_indexed  # HIP kernel; raises without the library or a GPU
            fn = maxsim_indexed
        scores = fn(q, self.token_store.data, starts_all[sel], lens_all[sel], self.config.scoring_method)
        return [float(x) for x in scores.detach().cpu().tolist()]

    def score_candidates(self, query: str, candidates: List[Dict[str, Any]]) -> List[float]:
        """stage-2 score of every candidate, in order (one kernel launch, one host copy)."""
        query_embeddings = self.encode_query(query)
        scores = self._score_from_store(query_embeddings, candidates)
        if scores is None:
            documents = [c["document"] for c in candidates]
            doc_embeddings_list = self.encode_documents_batch(documents)
            scores = self.score_all(query_embeddings, doc_embeddings_list)
        return scores

    def encode_queries_batch(self, queries: List[str]) -> List[torch.Tensor]:
        """Token matrices [Lq_j, H] of several queries from ONE padded forward (the reference
        encodes one query per call, :203-205; same values up to batch-padding noise)."""
        out: List[torch.Tensor] = []
        graphs = self.config.use_hip_graph and str(self.device).startswith("cuda")
        if len(queries) == 1 and graphs:
            return [self._encode_single_text(queries[0])[0]]       # one query: the graph-replayed batch-1 forward
        bs = max(self.config.batch_size, 1)
        if graphs:
            from .encoders import GraphedForward
            bs = max(bs, GraphedForward.ROWS[-1])                  # up to 128 queries per replayed forward
        for s in range(0, len(queries), bs):
            enc = self._tokenize_batch(list(queries[s:s + bs]))
            if graphs and "lengths" in enc:                        # (a right-padding mask, checked on the host)
                if self._graphed is None:
                    self._graphed = GraphedForward(self.model, getattr(self.tokenizer, "pad_token_id", 0),
                                                   self._amp_dtype() if self.use_amp else None)
                hidden = self._graphed(enc["input_ids"], enc["attention_mask"])
                lens = enc["lengths_host"]
                out.extend(hidden[j, :int(n), :] for j, n in enumerate(lens))
                continue
            hidden = self._forward(enc)
            lens = enc["lengths_host"] if "lengths_host" in enc else enc["attention_mask"].sum(dim=1).tolist()
            out.extend(hidden[j, :int(n), :] for j, n in enumerate(lens))
        return out

    def prefetch_queries(self, queries: List[str]) -> None:
        """Start the query forward of a coming score_arrays_partial / rescore_arrays call NOW: it does not depend on
        stage 1, so the pipeline enqueues it before stage 1's search — its kernels run beside the corpus sweep and its
        host side (tokenising, ~200 launches or a graph replay) no longer sits between stage 1's result and the MaxSim
        launch.  Same forward, same values; used once, by the next call with the same queries."""
        qs = list(queries)
        self._q_prefetch = (qs, self.encode_queries_batch(qs))

    def _query_embeddings(self, queries: List[str]) -> List[torch.Tensor]:
        pre = getattr(self, "_q_prefetch", None)
        self._q_prefetch = None
        if pre is not None and pre[0] == list(queries):
            return pre[1]
        return self.encode_queries_batch(list(queries))

    def score_candidates_many(self, queries: List[str], candidates_list: List[List[Dict[str, Any]]]) -> List[List[float]]:
        """stage-2 scores for several queries: one query forward for all of them and, when every
        candidate lives in the token store, ONE MaxSim launch (ts_maxsim_indexed_batch)."""
        if not queries:
            return []
        slots_list = None
        if len(self.token_store):
            slots_list = [[self._store_slot.get(c.get("doc_id"), -1) for c in cands] for cands in candidates_list]
            if any(sl and min(sl) < 0 for sl in slots_list):
                slots_list = None
        if slots_list is None:   # no resident store (or a stranger among the candidates): query by query
            return [self.score_candidates(q, c) if c else [] for q, c in zip(queries, candidates_list)]
        q_embs = self.encode_queries_batch(list(queries))
        store = self.token_store
        starts_all, lens_all = store.device_tables()
        dt = store.data.dtype
        q_off, c_off = [0], [0]
        for e, sl in zip(q_embs, slots_list):
            q_off.append(q_off[-1] + int(e.shape[0]))
            c_off.append(c_off[-1] + len(sl))
        if c_off[-1] == 0:
            return [[] for _ in queries]
        q_packed = torch.cat([e.to(dt) for e in q_embs], 0).contiguous()
        sel = torch.tensor([x for sl in slots_list for x in sl], dtype=torch.int64, device=starts_all.device)
        fn = self._maxsim_indexed_batch_fn
        if fn is None:
            from .index import maxsim_indexed_batch  # HIP kernel; raises without the library or a GPU
            fn = maxsim_indexed_batch
        flat = fn(q_packed, q_off, store.data, starts_all[sel], lens_all[sel], c_off,
                  self.config.scoring_method).detach().cpu().tolist()
        return [[float(x) for x in flat[a:b]] for a, b in zip(c_off[:-1], c_off[1:])]

    def reset_token_store(self) -> None:
        """Drop the resident token store (and every table derived from it)."""
        self.token_store, self._store_slot = TokenStore(), {}
        self._slot_version += 1

    def _slot_table(self, device):
        """(base, int64 [max id - base + 1]): pipeline doc id -> slot of the token store (-1 = not in the store), for
        the id range this store covers — a row shard's store starts at its first row, not at 0.  None for an empty
        store.  The cache is keyed on a version counter that every writer of ``_store_slot`` bumps (round 2 keyed it on
        the document COUNT: a reloaded store of the same size kept the old table and scored other documents' tokens)."""
        if not self._store_slot:
            return None
        cached = getattr(self, "_slot_tab", None)
        key = (self._slot_version, id(self._store_slot), str(torch.device(device)))
        if cached is not None and cached[0] == key:
            return cached[1]
        keys = torch.tensor(list(self._store_slot.keys()), dtype=torch.int64)
        base = int(keys.min())
        tab = torch.full((int(keys.max()) - base + 1,), -1, dtype=torch.int64)
        tab[keys - base] = torch.tensor(list(self._store_slot.values()), dtype=torch.int64)
        out = (base, tab.to(device))
        self._slot_tab = (key, out)
        return out

    def score_arrays_partial(self, queries: List[str], cand_ids: torch.Tensor, compact: Optional[bool] = None) -> torch.Tensor:
        """MaxSim scores float32 [B, C] of the candidates ``cand_ids`` int64 [B, C] that live in THIS process's token
        store; -inf everywhere else (ids outside the store's range, padding ids < 0).  One padded query forward, one
        ts_maxsim_indexed_batch launch, no collective.  With a row-sharded store
        (parallel_pipeline.ShardedRetrievalPipeline) every candidate is owned by exactly one rank, so an element-wise
        MAX over the ranks yields the complete score matrix.

        ``compact`` (default ``self.owner_compact``): True = score only the owned candidates (a ragged launch; the
        per-query counts cost ONE host sync) — a rank of R owns ~1/R of them; False = score all B x C positions (a
        position that is not owned reads document slot 0 and is masked afterwards): nothing here waits for the GPU,
        which is what the single-process pipeline wants, where everything is owned anyway."""
        store = self.token_store
        dev = store.data.device if store.data is not None else cand_ids.device
        cand_ids = cand_ids.to(dev)
        B, C = cand_ids.shape
        neg = float("-inf")
        got = self._slot_table(dev) if len(store) else None
        if got is None or B == 0 or C == 0:
            self._q_prefetch = None
            return torch.full((B, C), neg, dtype=torch.float32, device=dev)
        if compact is None:
            compact = bool(getattr(self, "owner_compact", False))
        base, tab = got
        rel = cand_ids - base
        inside = (rel >= 0) & (rel < tab.numel())
        slots = torch.where(inside, tab[rel.clamp(0, tab.numel() - 1)], torch.full_like(rel, -1))
        owned = slots >= 0
        if compact:
            counts = owned.sum(dim=1).tolist()        # the one host sync of the step: ragged candidate offsets
            if sum(counts) == 0:
                self._q_prefetch = None
                return torch.full((B, C), neg, dtype=torch.float32, device=dev)
            sel = slots[owned]                        # row-major: grouped by query, stage-1 order inside a query
        else:
            counts = [C] * B
            sel = slots.clamp(min=0).reshape(-1)
        q_embs = self._query_embeddings(queries)
        starts_all, lens_all = store.device_tables()
        dt = store.data.dtype
        q_off, c_off = [0], [0]
        for e, c in zip(q_embs, counts):
            q_off.append(q_off[-1] + int(e.shape[0]))
            c_off.append(c_off[-1] + int(c))
        q_packed = (q_embs[0].to(dt) if len(q_embs) == 1 else torch.cat([e.to(dt) for e in q_embs], 0)).contiguous()
        fn = self._maxsim_indexed_batch_fn
        if fn is None:
            from .index import maxsim_indexed_batch  # HIP kernel; raises without the library or a GPU
            fn = maxsim_indexed_batch
        flat = fn(q_packed, q_off, store.data, starts_all[sel], lens_all[sel], c_off, self.config.scoring_method)
        flat = flat.to(dev).to(torch.float32)
        if compact:
            out = torch.full((B, C), neg, dtype=torch.float32, device=dev)
            out[owned] = flat
            return out
        return torch.where(owned, flat.view(B, C), torch.full((), neg, dtype=torch.float32, device=dev))

    def keep_top_arrays(self, sc: torch.Tensor):
        """Stable descending sort of the score matrix [B, C] (= the reference's stable ``sort``, :293-297) ->
        (pos int64 [B, keep], scores float32 [B, keep]), keep = top_k_candidates."""
        srt, pos = torch.sort(sc, dim=1, descending=True, stable=True)
        keep = min(self.config.top_k_candidates, sc.shape[1])
        return pos[:, :keep].contiguous(), srt[:, :keep].contiguous()

    def rescore_arrays(self, queries: List[str], cand_ids: torch.Tensor, lazy: bool = False):
        """rescore_many on arrays: ``cand_ids`` int64 [B, C] (CUDA; every row the stage-1 candidates of that
        query, in stage-1 order) -> (pos int64 [B, keep], scores float32 [B, keep]): the positions, in
        stage-1 order, of the ``top_k_candidates`` best candidates by stage-2 score — the stable descending
        sort of the reference (:293-297) — and their scores.  Everything stays on the GPU: one padded query
        forward, one MaxSim launch over the resident token store, one sort.  None if a candidate is not in
        the store — or, with ``lazy``, a third result: a 0-dim bool tensor that is true in that case (no host sync
        here)."""
        if not len(self.token_store) or cand_ids.dim() != 2:
            return None
        sc = self.score_arrays_partial(queries, cand_ids)
        bad = torch.isinf(sc).any()                   # (a MaxSim score is a mean of cosines: never infinite)
        if lazy:                                      # the caller looks at `bad` when it copies the results out anyway
            return self.keep_top_arrays(sc) + (bad,)
        if bool(bad):
            return None
        return self.keep_top_arrays(sc)

    def _keep_top(self, candidates: List[Dict[str, Any]], scores: List[float]) -> List[Dict[str, Any]]:
        scored = []
        for cand, s in zip(candidates, scores):
            u = cand.copy()
            u["stage2_score"] = s
            u["stage"] = "stage2"
            scored.append(u)
        scored.sort(key=lambda x: x["stage2_score"], reverse=True)  # stable, like the reference
        return scored[: self.config.top_k_candidates]

    def rescore_many(self, queries: List[str], candidates_list: List[List[Dict[str, Any]]]) -> List[List[Dict[str, Any]]]:
        """rescore_candidates for several queries at once (same records per query)."""
        scores = self.score_candidates_many(queries, candidates_list)
        return [self._keep_top(c, s) if c else [] for c, s in zip(candidates_list, scores)]

    def rescore_candidates(self, query: str, candidates: List[Dict[str, Any]]) -> List[Dict[str, Any]]:
        if not candidates:
            return []
        self.logger.info(f"Rescoring {len(candidates)} candidates with Stage 2")
        scores = self.score_candidates(query, candidates)
        scored = []
        for cand, s in zip(candidates, scores):
            u = cand.copy()
            u["stage2_score"] = s
            u["stage"] = "stage2"
            scored.append(u)
        scored.sort(key=lambda x: x["stage2_score"], reverse=True)  # stable, like the reference
        top = scored[: self.config.top_k_candidates]
        self.logger.info(f"Stage 2 rescoring completed. Top score: {top[0]['stage2_score'] if top else 0:.4f}")
        return top

    def compute_similarity_matrix(self, query: str, documents: List[str]) -> np.ndarray:
        q = self.encode_query(query)
        return np.array(self.score_all(q, self.encode_documents_batch(documents)))

    def get_model_info(self) -> Dict[str, Any]:
        return {"model_name": self.config.model_name, "device": self.device,
                "max_seq_length": self.config.max_seq_length, "use_fp16": self.use_amp,
                "pooling_method": self.config.pooling_method, "scoring_method": self.config.scoring_method,
                "batch_size": self.config.batch_size,
                "embedding_dim": self.model.config.hidden_size if self.model else None}

    def clear_gpu_memory(self):
        """The reference calls torch.cuda.empty_cache() after every query
        (src/retrieval_pipeline.py:417-418); with 288 GB of HBM that only costs
        allocator round trips, so this is a no-op.  See clear_document_cache()."""
        return None

    def clear_document_cache(self):
        self._doc_cache.clear()
