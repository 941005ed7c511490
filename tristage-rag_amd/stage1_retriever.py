"""Stage 1: dense candidate generation on the MI355X index (+ optional BM25 fusion).

Mirror of reference src/stage1_retriever.py: same class names, config fields,
attributes callers reach into (``documents``, ``doc_metadata``, ``faiss_index``,
``bm25_index``, ``model``) and result-dict schema (:403-416).
What differs underneath:

* ``faiss_index`` is a tristage_rag_amd.index.FlatIPIndex (HIP, exact inner
  product).  The reference switches to IndexIVFFlat(nlist=100, nprobe=10) when the
  first add has >1000 rows (:262-273), an approximation of the exact result this
  index returns; exact search is kept for every size.
* row normalisation ``x / (|x| + 1e-8)`` (:285-288) runs on the GPU inside
  ``add`` when the embeddings are already on the device.
* ``search_many`` batches queries through one index call (the reference loops one
  query at a time, src/retrieval_pipeline.py:444-448).
* BM25 keeps an inverted index instead of per-document dict scans; scores, tie
  order and the fusion arithmetic are the reference's (:35-112, :326-366).  One
  deliberate difference: ``fit`` rebuilds its statistics from scratch, where the
  reference appends to ``doc_freqs`` on every re-fit (:73-74) and misaligns
  document ids after a second ``add_documents``.
"""
from __future__ import annotations

import json
import logging
import math
import os
import re
from collections import defaultdict
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class Stage1Config:
    model_name: str = "google/embeddinggemma-300m"
    device: str = "auto"
    cache_dir: str = "./models"
    index_dir: str = "./faiss_index"
    top_k_candidates: int = 500
    batch_size: int = 32
    index_batch_size: int = 256     # additive: documents per encoder forward in add_documents on the device path
    fuse_on_gpu: bool = True        # additive: search_many's RRF fusion for the whole query batch on the GPU (same float64
                                    # arithmetic and tie order as the per-query host code)
    amp_dtype: str = "bf16"         # additive: what use_fp16 means on the GPU — "bf16" (BASELINE configs[2]) or "fp16"
                                    # (what torch.cuda.amp.autocast() gives the reference, :235)
    max_text_length: int = 512
    enable_bm25: bool = True
    bm25_top_k: int = 300
    fusion_method: str = "rrf"  # "rrf" (Reciprocal Rank Fusion) or "weighted"
    rrf_k: int = 60
    dense_weight: float = 0.7
    bm25_weight: float = 0.3
    use_fp16: bool = True
    nlist: int = 100  # kept for config compatibility; the index is exact
    nprobe: int = 10
    # additive knobs (not in the reference)
    index_dtype: str = "f32"   # storage dtype of the corpus matrix: f32 | f16 | bf16
    gpu_index_device: int = 0
    bm25_on_gpu: Optional[bool] = None  # BM25 postings in HBM + HIP scoring kernels; None = whenever
                                        # the HIP index is in use (a GPU is present)
    use_hip_graph: bool = False  # replay single-query encoder forwards from HIP graphs
    bm25_refit_compat: bool = False  # BM25 statistics after a SECOND add_documents exactly as the reference computes them
                                     # (its fit() appends to the previous fit's lists; see BM25Index)


def amp_torch_dtype(name: str):
    """"bf16" / "fp16" (also "bfloat16", "f16", "float16", "half") -> the torch dtype of the AMP forwards."""
    import torch
    key = str(name).lower()
    if key in ("bf16", "bfloat16"):
        return torch.bfloat16
    if key in ("fp16", "f16", "float16", "half"):
        return torch.float16
    raise ValueError(f"amp_dtype must be 'bf16' or 'fp16', not {name!r}")


class BM25Index:
    """BM25 (k1=1.2, b=0.75) with the reference's tokenizer and idf
    (reference src/stage1_retriever.py:35-112), over an inverted index."""

    def __init__(self, k1: float = 1.2, b: float = 0.75, gpu_device: Optional[int] = None, refit_compat: bool = False):
        self.k1 = k1
        self.b = b
        self.gpu_device = gpu_device   # None: score on the host; int: HIP kernels on that GPU
        # refit_compat: reproduce what the reference's BM25Index does when fit() is called AGAIN (every add_documents
        # after the first, src/stage1_retriever.py:316-322): its fit() appends to doc_freqs / doc_lens instead of
        # rebuilding them (:56-80), so the statistics of earlier fits are counted again in df and the average length,
        # and document i is scored with list entry i — for documents added later that is an EARLIER document's term
        # frequencies.  Off (default): fit() rebuilds, i.e. the scores the reference gives after ONE add_documents.
        self.refit_compat = bool(refit_compat)
        self._gpu = None
        import threading
        self._gpu_lock = threading.Lock()   # the GPU handle serves one caller at a time
        self._term_id: Dict[str, int] = {}
        self.doc_freqs: List[Dict[str, int]] = []
        self.idf: Dict[str, float] = {}
        self.doc_lens: List[int] = []
        self.avg_doc_len = 0
        self.corpus_size = 0
        self.vocabulary = set()
        self.documents: List[str] = []
        self._postings: Dict[str, Tuple[np.ndarray, np.ndarray]] = {}
        self._len_norm = np.zeros(0)

    def tokenize(self, text: str) -> List[str]:
        text = text.lower()
        text = re.sub(r"[^a-z0-9\s]", " ", text)
        return text.split()

    def fit(self, documents: Sequence[str], stats_exchange=None) -> None:
        """``stats_exchange(df, total_len, n_docs) -> (df, total_len, n_docs)`` (optional): the corpus-wide document
        frequencies, token count and document count when `documents` is only one row shard of the corpus
        (parallel_pipeline.ShardedBM25 all-gathers them): idf and the average length are then the GLOBAL ones, so a
        document's score is exactly what an index over the whole corpus gives it; ids stay local."""
        self.documents = list(documents)
        self.corpus_size = len(self.documents)
        keep = self.refit_compat and bool(self.doc_freqs)
        if not keep:
            self.doc_freqs, self.doc_lens = [], []
        self.idf = {}
        for doc in self.documents:
            tf: Dict[str, int] = defaultdict(int)
            toks = self.tokenize(doc)
            for t in toks:
                tf[t] += 1
            self.doc_freqs.append(tf)         # (refit_compat: BEHIND the entries of the earlier fits, like the reference)
            self.doc_lens.append(len(toks))
        n = self.corpus_size
        # document i is scored with list entry i (reference score(), :83-101); df and the average length run over
        # ALL entries (:76-80) — the same thing unless refit_compat kept entries of earlier fits
        post_d: Dict[str, List[int]] = defaultdict(list)
        post_tf: Dict[str, List[int]] = defaultdict(list)
        df: Dict[str, int] = defaultdict(int)
        for i, tf in enumerate(self.doc_freqs):
            for t, c in tf.items():
                df[t] += 1
                if i < n:
                    post_d[t].append(i)
                    post_tf[t].append(c)
        total_len, n_idf = sum(self.doc_lens), n
        if stats_exchange is not None:
            df, total_len, n_idf = stats_exchange(dict(df), total_len, n)
        self.vocabulary = set(df)
        self.avg_doc_len = total_len / n_idf if n_idf > 0 else 0
        for t, d_ in df.items():
            self.idf[t] = math.log((n_idf - d_ + 0.5) / (d_ + 0.5) + 1.0)
        self._postings = {t: (np.asarray(post_d[t], dtype=np.int64), np.asarray(post_tf[t], dtype=np.float64))
                          for t in post_d}
        lens = np.asarray(self.doc_lens[:n], dtype=np.float64)
        self._len_norm = (self.k1 * (1 - self.b + self.b * lens / self.avg_doc_len)
                          if self.avg_doc_len else np.zeros_like(lens))
        if self.gpu_device is not None:
            self._upload()

    # -- GPU mode ------------------------------------------------------------
    def _upload(self) -> None:
        """CSR postings + statistics -> HBM (ts_bm25_set_index)."""
        import ctypes
        from . import _lib
        lib = _lib.load()
        if self._gpu is None:
            self._gpu = ctypes.c_void_p()
            _lib.check(lib.ts_bm25_create(int(self.gpu_device), ctypes.byref(self._gpu)))
        terms = sorted(self._postings)
        self._term_id = {t: i for i, t in enumerate(terms)}
        off = np.zeros(len(terms) + 1, dtype=np.int64)
        for i, t in enumerate(terms):
            off[i + 1] = off[i] + len(self._postings[t][0])
        nnz = int(off[-1])
        docs = (np.concatenate([self._postings[t][0] for t in terms]).astype(np.int32)
                if nnz else np.zeros(1, np.int32))
        tfs = (np.concatenate([self._postings[t][1] for t in terms]).astype(np.float32)
               if nnz else np.zeros(1, np.float32))
        idf = np.array([self.idf[t] for t in terms], dtype=np.float64) if terms else np.zeros(1)
        ln = np.ascontiguousarray(self._len_norm, dtype=np.float64) if self.corpus_size else np.zeros(1)
        _lib.check(lib.ts_bm25_set_index(self._gpu, self.corpus_size, len(terms), nnz,
                                         off.ctypes.data, docs.ctypes.data, tfs.ctypes.data,
                                         idf.ctypes.data, ln.ctypes.data, float(self.k1 + 1)))

    def _search_gpu(self, query: str, top_k: int) -> List[Tuple[int, float]]:
        return self._search_gpu_many([query], top_k)[0]

    def _search_gpu_many(self, queries: Sequence[str], top_k: int, arrays: bool = False):
        """All queries through ts_bm25_search_batch: one call and one synchronisation for the batch."""
        from . import _lib
        lib = _lib.load()
        terms = [np.array([self._term_id[t] for t in self.tokenize(q) if t in self._term_id], dtype=np.int32) for q in queries]
        off = np.zeros(len(queries) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(t) for t in terms])
        flat = np.concatenate(terms) if len(terms) and off[-1] else np.zeros(1, dtype=np.int32)
        k = min(int(top_k), max(self.corpus_size, 1))
        nq = len(queries)
        out_s = np.zeros((nq, max(k, 1)), dtype=np.float64)
        out_i = np.zeros((nq, max(k, 1)), dtype=np.int64)
        n_out = np.zeros(max(nq, 1), dtype=np.int32)
        if k > 0 and nq and off[-1]:
            with self._gpu_lock:      # one accumulator per handle, and ctypes drops the GIL for the call
                _lib.check(lib.ts_bm25_search_batch(self._gpu, flat.ctypes.data, off.ctypes.data, nq, k, out_s.ctypes.data,
                                                    out_i.ctypes.data, n_out.ctypes.data, None))
        if arrays:      # (ids, scores) per query, no tuples: the array path of Stage1Retriever fuses them as they are
            want = min(int(top_k), self.corpus_size)
            out = []
            for q in range(nq):
                n = int(n_out[q])
                if n < want:    # documents without a query term score exactly 0.0 and follow in ascending id order
                    pad = self._pad_with_zero_scores(list(zip(out_i[q, :n].tolist(), out_s[q, :n].tolist())), top_k)
                    out.append((np.array([i for i, _ in pad], dtype=np.int64), np.array([v for _, v in pad], dtype=np.float64)))
                else:
                    out.append((out_i[q, :n], out_s[q, :n]))
            return out
        return [self._pad_with_zero_scores(list(zip(out_i[q, : n_out[q]].tolist(), out_s[q, : n_out[q]].tolist())), top_k)
                for q in range(nq)]

    def _pad_with_zero_scores(self, res: List[Tuple[int, float]], top_k: int) -> List[Tuple[int, float]]:
        if len(res) < min(top_k, self.corpus_size):
            # every document with a non-zero score is listed; the rest score exactly 0.0 and
            # follow in ascending id order (the reference's stable sort)
            seen = {i for i, _ in res}
            d = 0
            while len(res) < min(top_k, self.corpus_size):
                if d not in seen:
                    res.append((d, 0.0))
                d += 1
        return res

    def close(self) -> None:
        if self._gpu is not None:
            from . import _lib
            _lib.load().ts_bm25_destroy(self._gpu)
            self._gpu = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def score(self, query: str, doc_idx: int) -> float:
        if doc_idx >= len(self.doc_freqs):
            return 0.0
        f, dl, s = self.doc_freqs[doc_idx], self.doc_lens[doc_idx], 0.0
        for tok in self.tokenize(query):
            if tok in f and tok in self.idf:
                tf = f[tok]
                s += self.idf[tok] * ((tf * (self.k1 + 1)) /
                                      (tf + self.k1 * (1 - self.b + self.b * dl / self.avg_doc_len)))
        return s

    def scores(self, query: str) -> np.ndarray:
        """Scores of every document, accumulated term by term in query order
        (the same order of float additions as the per-document loop)."""
        acc = np.zeros(self.corpus_size, dtype=np.float64)
        for tok in self.tokenize(query):
            p = self._postings.get(tok)
            if p is None:
                continue
            ds, tf = p
            acc[ds] += self.idf[tok] * ((tf * (self.k1 + 1)) / (tf + self._len_norm[ds]))
        return acc

    def search_many(self, queries: Sequence[str], top_k: int = 10) -> List[List[Tuple[int, float]]]:
        """search() for several queries (the GPU index takes them in one call)."""
        if self._gpu is not None:
            return self._search_gpu_many(list(queries), top_k)
        return [self.search(q, top_k) for q in queries]

    def search_many_arrays(self, queries: Sequence[str], top_k: int = 10):
        """search_many() as (ids int64 [n], scores float64 [n]) per query."""
        if self._gpu is not None:
            return self._search_gpu_many(list(queries), top_k, arrays=True)
        out = []
        for q in queries:
            r = self.search(q, top_k)
            out.append((np.fromiter((i for i, _ in r), dtype=np.int64, count=len(r)),
                        np.fromiter((v for _, v in r), dtype=np.float64, count=len(r))))
        return out

    def search(self, query: str, top_k: int = 10) -> List[Tuple[int, float]]:
        if self._gpu is not None:
            return self._search_gpu(query, top_k)
        s = self.scores(query)
        order = np.argsort(-s, kind="stable")[:top_k]  # ties keep ascending doc order
        return [(int(i), float(s[i])) for i in order]


class Stage1Retriever:
    """Stage 1: dense embeddings + exact MI355X index + optional BM25 fusion."""

    def __init__(self, config: Stage1Config, model: Any = None,
                 index_factory: Optional[Callable[[int], Any]] = None):
        self.config = config
        self.logger = logging.getLogger(__name__)
        self.model = model
        self.embedding_dim: Optional[int] = None
        self.faiss_index = None
        self.bm25_index: Optional[BM25Index] = None
        self.documents: List[str] = []
        self.doc_metadata: List[Dict[str, Any]] = []
        self._index_factory = index_factory
        os.makedirs(self.config.cache_dir, exist_ok=True)
        os.makedirs(self.config.index_dir, exist_ok=True)
        self._load_model()

    # -- model -------------------------------------------------------------
    def _load_model(self) -> None:
        if self.model is None:
            from .encoders import SentenceEncoder
            self.logger.info(f"Loading Stage 1 model: {self.config.model_name}")
            self.model = SentenceEncoder(self.config.model_name, device=self.config.device,
                                         cache_folder=self.config.cache_dir,
                                         use_hip_graph=self.config.use_hip_graph)
        if hasattr(self.model, "get_sentence_embedding_dimension"):
            self.embedding_dim = self.model.get_sentence_embedding_dimension()
        else:
            self.embedding_dim = int(np.asarray(self.model.encode("sample text", convert_to_numpy=True)).shape[0])
        self.logger.info(f"Model loaded successfully. Embedding dimension: {self.embedding_dim}")

    def _amp_dtype(self):
        import torch
        return amp_torch_dtype(getattr(self.config, "amp_dtype", "bf16"))

    def _encode_batch(self, texts: List[str]) -> np.ndarray:
        """reference :230-254 — float32 [n, d] embeddings (AMP on the GPU when use_fp16)."""
        import torch
        dev = str(getattr(self.model, "device", "cpu"))
        if self.config.use_fp16 and dev.startswith("cuda"):
            with torch.autocast("cuda", dtype=self._amp_dtype()):
                emb = self.model.encode(texts, batch_size=self.config.batch_size, convert_to_numpy=True,
                                        show_progress_bar=False)
        else:
            emb = self.model.encode(texts, batch_size=self.config.batch_size, convert_to_numpy=True,
                                    show_progress_bar=False)
        return np.asarray(emb).astype(np.float32)

    def _normalize_embeddings(self, embeddings: np.ndarray) -> np.ndarray:
        """reference :285-288"""
        norms = np.linalg.norm(embeddings, axis=1, keepdims=True)
        return embeddings / (norms + 1e-8)

    # Device-resident variants: when the encoder runs on the GPU and the index is the HIP
    # index, embeddings never visit the host — rows are normalised inside ts_index_add
    # (TS_FLAG_NORMALIZE, the same x/(|x|+1e-8) in fp32) and queries stay tensors.
    def _bm25_device(self) -> Optional[int]:
        on = self.config.bm25_on_gpu
        if on is None:
            import torch
            on = self._index_factory is None and torch.cuda.is_available()
        return self.config.gpu_index_device if on else None

    def _device_path(self) -> bool:
        dev = str(getattr(self.model, "device", "cpu"))
        return (dev.startswith("cuda") and self._index_factory is None and
                getattr(self.model, "encode", None) is not None)

    def _encode_batch_tensor(self, texts: List[str], bulk: bool = False):
        import torch
        ctx = (torch.autocast("cuda", dtype=self._amp_dtype()) if self.config.use_fp16
               else torch.autocast("cuda", enabled=False))
        bs = max(self.config.batch_size, getattr(self.config, "index_batch_size", 0) or 0) if bulk else self.config.batch_size
        with ctx:
            emb = self.model.encode(texts, batch_size=bs, convert_to_numpy=False,
                                    convert_to_tensor=True, show_progress_bar=False)
        return emb.float()

    def _normalized_query_tensor(self, texts: List[str]):
        q = self._encode_batch_tensor(texts)
        return q / (q.norm(dim=1, keepdim=True) + 1e-8)

    # -- index -------------------------------------------------------------
    def _create_faiss_index(self, embeddings: np.ndarray) -> None:
        d = int(embeddings.shape[1])
        if self._index_factory is not None:
            self.faiss_index = self._index_factory(d)
        else:
            from .index import FlatIPIndex  # raises without libtristage.so / a GPU: no CPU fallback
            self.faiss_index = FlatIPIndex(d, dtype=self.config.index_dtype,
                                           device=self.config.gpu_index_device)
        self.faiss_index.add(embeddings)
        self.logger.info(f"Index created with {len(embeddings)} vectors (exact inner product)")

    def add_documents(self, documents: List[str], metadata: Optional[List[Dict[str, Any]]] = None):
        if not documents:
            return
        self.logger.info(f"Adding {len(documents)} documents to Stage 1 index")
        self.documents.extend(documents)
        if metadata is None:
            metadata = [{}] * len(documents)  # one shared dict, as in the reference (:302)
        self.doc_metadata.extend(metadata)
        if self._device_path():
            emb = self._encode_batch_tensor(list(documents), bulk=True)
            if self.faiss_index is None:
                from .index import FlatIPIndex
                self.faiss_index = FlatIPIndex(int(emb.shape[1]), dtype=self.config.index_dtype,
                                               device=self.config.gpu_index_device)
            self.faiss_index.add(emb, normalize=True)
        else:
            embeddings = self._normalize_embeddings(self._encode_batch(list(documents)))
            if self.faiss_index is None:
                self._create_faiss_index(embeddings)
            else:
                self.faiss_index.add(embeddings)
        if self.config.enable_bm25:
            if self.bm25_index is None:
                self.bm25_index = BM25Index(gpu_device=self._bm25_device(),
                                            refit_compat=getattr(self.config, "bm25_refit_compat", False))
            self.bm25_index.fit(self.documents)
        self.logger.info(f"Documents added successfully. Total documents: {len(self.documents)}")

    # -- fusion ------------------------------------------------------------
    def _reciprocal_rank_fusion(self, dense_results, bm25_results):
        """reference :326-343"""
        scores: Dict[int, float] = defaultdict(float)
        for rank, (doc_idx, _) in enumerate(dense_results):
            scores[doc_idx] += 1.0 / (self.config.rrf_k + rank + 1)
        for rank, (doc_idx, _) in enumerate(bm25_results):
            scores[doc_idx] += 1.0 / (self.config.rrf_k + rank + 1)
        fused = [(i, s) for i, s in scores.items()]
        fused.sort(key=lambda x: x[1], reverse=True)
        return fused

    def _weighted_fusion(self, dense_results, bm25_results):
        """reference :345-366"""
        scores: Dict[int, float] = defaultdict(float)
        if dense_results:
            mx = max(s for _, s in dense_results)
            for i, s in dense_results:
                scores[i] += self.config.dense_weight * (s / mx)
        if bm25_results:
            mx = max(s for _, s in bm25_results)
            for i, s in bm25_results:
                scores[i] += self.config.bm25_weight * (s / mx)
        fused = [(i, s) for i, s in scores.items()]
        fused.sort(key=lambda x: x[1], reverse=True)
        return fused

    # -- search ------------------------------------------------------------
    def _finish(self, query: str, dense_results: List[Tuple[int, float]], top_k: int) -> List[Dict[str, Any]]:
        bm25_results: List[Tuple[int, float]] = []
        if self.config.enable_bm25 and self.bm25_index is not None:
            bm25_results = self.bm25_index.search(query, self.config.bm25_top_k)
        if self.config.enable_bm25 and bm25_results:
            if self.config.fusion_method == "rrf":
                fused = self._reciprocal_rank_fusion(dense_results, bm25_results)
            else:
                fused = self._weighted_fusion(dense_results, bm25_results)
            final = fused[:top_k]
        else:
            final = dense_results[:top_k]
        results = []
        for doc_idx, score in final:
            if doc_idx < len(self.documents):
                results.append({"doc_id": doc_idx, "document": self.documents[doc_idx], "score": score,
                                "stage1_score": score, "metadata": self.doc_metadata[doc_idx],
                                "stage": "stage1"})
        return results

    def search(self, query: str, top_k: Optional[int] = None) -> List[Dict[str, Any]]:
        if self.faiss_index is None:
            raise ValueError("No documents indexed. Call add_documents() first.")
        top_k = top_k or self.config.top_k_candidates
        if self._device_path():
            D, I = self.faiss_index.search(self._normalized_query_tensor([query]), top_k)
            scores, ids = D.cpu().numpy(), I.cpu().numpy()
        else:
            q = self._normalize_embeddings(self._encode_batch([query]))
            scores, ids = self.faiss_index.search(q, top_k)
        dense = [(int(i), float(s)) for i, s in zip(ids[0], scores[0]) if i >= 0]
        results = self._finish(query, dense, top_k)
        self.logger.info(f"Stage 1 search completed. Found {len(results)} candidates")
        return results

    def search_many(self, queries: Sequence[str], top_k: Optional[int] = None) -> List[List[Dict[str, Any]]]:
        """All queries through ONE encoder pass and ONE index call (64 queries share a
        single sweep over the corpus on the GPU)."""
        if self.faiss_index is None:
            raise ValueError("No documents indexed. Call add_documents() first.")
        top_k = top_k or self.config.top_k_candidates
        if not queries:
            return []
        if self._device_path():
            D, I = self.faiss_index.search(self._normalized_query_tensor(list(queries)), top_k)
            scores, ids = D.cpu().numpy(), I.cpu().numpy()
        else:
            q = self._normalize_embeddings(self._encode_batch(list(queries)))
            scores, ids = self.faiss_index.search(q, top_k)
        out = []
        for qi, query in enumerate(queries):
            dense = [(int(i), float(s)) for i, s in zip(ids[qi], scores[qi]) if i >= 0]
            out.append(self._finish(query, dense, top_k))
        return out

    def search_many_arrays(self, queries: Sequence[str], top_k: Optional[int] = None):
        """Stage 1 for a query batch WITHOUT building result records: (ids int64 [B, k'], scores [B, k'])
        as torch tensors on the index's device for the pure dense search, numpy arrays (float64 fused
        scores) when BM25 fusion is on.  Row q, in order, is exactly what ``search_many`` would list for
        query q.  None when the corpus holds fewer than top_k rows (the caller then uses ``search_many``,
        which deals with padded results)."""
        import torch
        if self.faiss_index is None:
            raise ValueError("No documents indexed. Call add_documents() first.")
        top_k = top_k or self.config.top_k_candidates
        n = int(self.faiss_index.ntotal)
        if top_k > n or n != len(self.documents):
            return None
        fuse = self.config.enable_bm25 and self.bm25_index is not None
        if self._device_path():
            qt = self._normalized_query_tensor(list(queries))
            if not fuse and hasattr(self.faiss_index, "search_in_stream_order"):
                D, I = self.faiss_index.search_in_stream_order(qt, top_k)   # (ids stay on the GPU for stage 2: no wait here)
            else:
                D, I = self.faiss_index.search(qt, top_k)
        else:
            D, I = self.faiss_index.search(self._normalize_embeddings(self._encode_batch(list(queries))), top_k)
            D, I = torch.as_tensor(D), torch.as_tensor(I)
        if not fuse:
            return I, D
        bm25 = self.bm25_index
        if (self.config.fusion_method == "rrf" and I.is_cuda and getattr(self.config, "fuse_on_gpu", True)
                and hasattr(bm25, "search_many_arrays")):
            bms = bm25.search_many_arrays(list(queries), self.config.bm25_top_k)
            k2 = len(bms[0][0]) if bms else 0
            if k2 > 0 and all(len(b[0]) == k2 for b in bms):      # (ragged BM25 lists: the per-query host code below)
                return self._fuse_rrf_device(I, np.stack([b[0] for b in bms]), top_k)
        ids, scores = I.cpu().numpy(), D.cpu().numpy()
        out_i = np.empty((len(queries), top_k), dtype=np.int64)
        out_s = np.empty((len(queries), top_k), dtype=np.float64)
        bms = (bm25.search_many_arrays(list(queries), self.config.bm25_top_k) if hasattr(bm25, "search_many_arrays")
               else [bm25.search(q, self.config.bm25_top_k) for q in queries])
        for qi, bm in enumerate(bms):
            fi, fs = self._fuse_arrays(ids[qi], scores[qi], bm)
            if len(fi) < top_k:
                return None
            out_i[qi], out_s[qi] = fi[:top_k], fs[:top_k]
        return out_i, out_s

    def _fuse_rrf_device(self, dense_ids, bm25_ids: np.ndarray, top_k: int):
        """Reciprocal rank fusion of a whole query batch on the GPU: dense_ids int64 [B, k1] (device, rank order), bm25_ids
        int64 [B, k2] (host, rank order) -> (ids int64 [B, top_k], fused scores float64 [B, top_k]) on the device.  The
        arithmetic of _fuse_arrays / _reciprocal_rank_fusion (reference :326-340) in float64 — 1 / (k + rank + 1), the
        dense term first, then the BM25 term — and its order: descending fused score, ties in first-seen order (dense
        list, then the BM25-only documents in BM25 order), by ONE stable sort per batch; bit-identical to the host code
        (tested)."""
        import torch
        dev = dense_ids.device
        B, k1 = dense_ids.shape
        b_ids = torch.from_numpy(np.ascontiguousarray(bm25_ids)).to(dev)
        k2 = int(b_ids.shape[1])
        rk = float(self.config.rrf_k)
        d_part = 1.0 / (rk + torch.arange(k1, dtype=torch.float64, device=dev) + 1)
        b_part = 1.0 / (rk + torch.arange(k2, dtype=torch.float64, device=dev) + 1)
        sorted_ids, order = torch.sort(dense_ids, dim=1, stable=True)
        pos = torch.searchsorted(sorted_ids, b_ids).clamp(max=k1 - 1)
        hit = torch.gather(sorted_ids, 1, pos) == b_ids                       # the BM25 document is in the dense list
        where = torch.gather(order, 1, pos)                                   # ... at this rank
        fused = d_part.expand(B, k1).clone()
        fused.scatter_add_(1, where, torch.where(hit, b_part.expand(B, k2), torch.zeros((), dtype=torch.float64, device=dev)))
        all_ids = torch.cat([dense_ids, b_ids], dim=1)
        minus_inf = torch.full((), float("-inf"), dtype=torch.float64, device=dev)
        all_sc = torch.cat([fused, torch.where(hit, minus_inf, 0.0 + b_part.expand(B, k2))], dim=1)
        srt, rank = torch.sort(all_sc, dim=1, descending=True, stable=True)
        return torch.gather(all_ids, 1, rank[:, :top_k]).contiguous(), srt[:, :top_k].contiguous()

    def _fuse_arrays(self, dense_ids: np.ndarray, dense_scores: np.ndarray, bm25_results):
        """The fusion of _finish() on arrays: same float64 arithmetic, same order (descending fused score,
        ties in first-seen order: dense list first, then the BM25-only documents) as the dictionary code of
        _reciprocal_rank_fusion / _weighted_fusion (reference :326-366)."""
        if isinstance(bm25_results, tuple):          # (ids, scores) arrays
            b_ids, b_sc = bm25_results
            if not len(b_ids):
                return dense_ids, dense_scores.astype(np.float64)
        else:                                         # [(id, score), ...]
            if not bm25_results:
                return dense_ids, dense_scores.astype(np.float64)
            b_ids = np.fromiter((i for i, _ in bm25_results), dtype=np.int64, count=len(bm25_results))
            b_sc = np.fromiter((s for _, s in bm25_results), dtype=np.float64, count=len(bm25_results))
        if self.config.fusion_method == "rrf":
            d_part = 1.0 / (self.config.rrf_k + np.arange(len(dense_ids), dtype=np.float64) + 1)
            b_part = 1.0 / (self.config.rrf_k + np.arange(len(b_ids), dtype=np.float64) + 1)
        else:
            d_sc = dense_scores.astype(np.float64)
            if (len(d_sc) and d_sc.max() == 0.0) or b_sc.max() == 0.0:
                raise ZeroDivisionError("float division by zero")   # what _weighted_fusion (and the reference) does
            d_part = self.config.dense_weight * (d_sc / d_sc.max()) if len(d_sc) else d_sc
            b_part = self.config.bm25_weight * (b_sc / b_sc.max())
        # position of every BM25 document in the dense list (or -1)
        order = np.argsort(dense_ids, kind="stable")
        pos = np.searchsorted(dense_ids[order], b_ids)
        pos = np.where(pos < len(order), pos, 0)
        hit = dense_ids[order][pos] == b_ids if len(order) else np.zeros(len(b_ids), dtype=bool)
        fused = d_part.copy()
        fused[order[pos[hit]]] = fused[order[pos[hit]]] + b_part[hit]          # dense term first, then the BM25 term
        all_ids = np.concatenate([dense_ids, b_ids[~hit]])
        all_sc = np.concatenate([fused, (0.0 + b_part[~hit])])
        rank = np.argsort(-all_sc, kind="stable")
        return all_ids[rank], all_sc[rank]

    # -- persistence (reference :421-465; raw matrix + JSON instead of pickle + faiss file)
    def save_index(self, index_path: Optional[str] = None):
        if index_path is None:
            index_path = os.path.join(self.config.index_dir, "stage1_index.pkl")
        base = os.path.splitext(index_path)[0]
        os.makedirs(os.path.dirname(os.path.abspath(index_path)), exist_ok=True)
        manifest = {"format": "tristage-rag_amd/1", "documents": self.documents,
                    "doc_metadata": self.doc_metadata, "config": dict(self.config.__dict__),
                    "ntotal": 0, "dim": self.embedding_dim, "matrix": None}
        if self.faiss_index is not None:
            mat = self.faiss_index.reconstruct_n(0, self.faiss_index.ntotal)
            np.save(base + ".matrix.npy", mat)
            manifest.update(ntotal=int(mat.shape[0]), dim=int(mat.shape[1]),
                            matrix=os.path.basename(base + ".matrix.npy"))
        with open(index_path, "w") as f:  # JSON under the reference's file name
            json.dump(manifest, f)
        self.logger.info(f"Stage 1 index saved to {index_path}")

    def load_index(self, index_path: Optional[str] = None):
        if index_path is None:
            index_path = os.path.join(self.config.index_dir, "stage1_index.pkl")
        if not os.path.exists(index_path):
            self.logger.warning(f"Index file not found: {index_path}")
            return
        with open(index_path, "rb") as f:
            head = f.read(1)
        if head != b"{":
            raise ValueError(f"{index_path} is not a tristage-rag_amd index manifest (a pickle written by the "
                             "reference is not loaded: unpickling executes code)")
        manifest = json.load(open(index_path))
        self.documents = manifest["documents"]
        self.doc_metadata = manifest["doc_metadata"]
        self.faiss_index = None
        if manifest.get("matrix"):
            mat = np.load(os.path.join(os.path.dirname(os.path.abspath(index_path)), manifest["matrix"]),
                          allow_pickle=False)
            self._create_faiss_index(mat.astype(np.float32))
        if self.config.enable_bm25 and self.documents:
            self.bm25_index = BM25Index(gpu_device=self._bm25_device())
            self.bm25_index.fit(self.documents)
        self.logger.info(f"Stage 1 index loaded from {index_path}")

    def get_stats(self) -> Dict[str, Any]:
        return {"total_documents": len(self.documents), "embedding_dimension": self.embedding_dim,
                "faiss_index_type": type(self.faiss_index).__name__ if self.faiss_index else None,
                "bm25_enabled": self.config.enable_bm25,
                "bm25_vocabulary_size": len(self.bm25_index.vocabulary) if self.bm25_index else 0,
                "config": self.config.__dict__}
