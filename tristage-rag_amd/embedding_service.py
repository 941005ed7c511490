"""EmbeddingService — façade named by BASELINE.json's north_star.

Mirror of reference src/embedding_service.py (a singleton wrapper with an
md5-keyed FIFO cache of 1000 embeddings, text validation 1..10000 chars,
``encode_query`` :152-170, ``encode_document`` :172-226, cosine ``similarity``
:228-237).  Nothing in the reference imports it; it is kept as a thin layer over
the same SentenceEncoder the stage-1 retriever uses.  ``similarity`` computes the
same cosine; for a large document matrix on a machine with a GPU the products are
taken by the HIP dense scan (``FlatIPIndex.scores`` -> ``ts_index_scores``), else by
numpy exactly as the reference does.
"""
from __future__ import annotations

import hashlib
import logging
import threading
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import numpy as np
import yaml


@dataclass
class EmbeddingConfig:
    model_name: str = "google/embeddinggemma-300m"
    device: str = "auto"
    max_length: int = 512
    batch_size: int = 32
    cache_dir: str = "./models"
    enable_caching: bool = True
    cache_size: int = 1000
    log_level: str = "INFO"
    log_format: str = "%(asctime)s - %(name)s - %(levelname)s - %(message)s"
    log_file: str = "embedding_service.log"
    max_text_length: int = 10000
    min_text_length: int = 1


class EmbeddingService:
    _instance = None
    _lock = threading.Lock()

    def __new__(cls, config_path: str = "config.yaml", model: Any = None):
        if cls._instance is None:
            with cls._lock:
                if cls._instance is None:
                    cls._instance = super().__new__(cls)
        return cls._instance

    def __init__(self, config_path: str = "config.yaml", model: Any = None):
        if getattr(self, "_initialized", False):
            return
        self._initialized = True
        self.config = self._load_config(config_path)
        self.logger = logging.getLogger(__name__)
        self._model = model
        self._model_lock = threading.Lock()
        self._cache: Dict[str, np.ndarray] = {}

    @classmethod
    def reset_instance(cls) -> None:
        with cls._lock:
            cls._instance = None

    def _load_config(self, config_path: str) -> EmbeddingConfig:
        """reference :46-78: read the pipeline YAML's stage-1 section; defaults on any error."""
        try:
            with open(config_path, "r") as f:
                data = yaml.safe_load(f) or {}
            pd = data.get("pipeline", {}) or {}
            s1 = pd.get("stage1", {}) or {}
            device = pd.get("device", "cpu")
            if device == "auto":
                device = "cpu"
            return EmbeddingConfig(model_name=s1.get("model", "google/embeddinggemma-300m"), device=device,
                                   max_length=s1.get("max_text_length", 512),
                                   batch_size=s1.get("batch_size", 32),
                                   cache_dir=pd.get("cache_dir", "./models"),
                                   log_level=pd.get("log_level", "INFO"),
                                   log_file=pd.get("log_file", "embedding_service.log"))
        except Exception:
            return EmbeddingConfig()

    def _get_model(self):
        if self._model is None:
            with self._model_lock:
                if self._model is None:
                    from .encoders import SentenceEncoder
                    self._model = SentenceEncoder(self.config.model_name, device=self.config.device,
                                                  cache_folder=self.config.cache_dir)
        return self._model

    def get_model_info(self) -> Dict[str, Any]:
        m = self._get_model()
        return {"model_name": self.config.model_name, "device": m.device,
                "max_seq_length": m.max_seq_length,
                "embedding_dimension": m.get_sentence_embedding_dimension(),
                "cache_size": len(self._cache) if self.config.enable_caching else 0,
                "enable_caching": self.config.enable_caching}

    def _validate_text(self, text) -> bool:
        return (isinstance(text, str) and
                self.config.min_text_length <= len(text) <= self.config.max_text_length)

    def _get_text_hash(self, text: str) -> str:
        return hashlib.md5(text.encode()).hexdigest()

    def _cache_embedding(self, text: str, embedding: np.ndarray) -> None:
        if not self.config.enable_caching:
            return
        if len(self._cache) >= self.config.cache_size:
            del self._cache[next(iter(self._cache))]  # evicts the first-inserted entry (:141-143)
        self._cache[self._get_text_hash(text)] = embedding

    def _get_cached_embedding(self, text: str) -> Optional[np.ndarray]:
        return self._cache.get(self._get_text_hash(text)) if self.config.enable_caching else None

    def _bad_text(self, what: str) -> ValueError:
        return ValueError(f"Invalid {what} text: must be between {self.config.min_text_length} and "
                          f"{self.config.max_text_length} characters")

    def encode_query(self, query: str) -> np.ndarray:
        if not self._validate_text(query):
            raise self._bad_text("query")
        hit = self._get_cached_embedding(query)
        if hit is not None:
            return hit
        emb = self._get_model().encode(query, convert_to_numpy=True)
        self._cache_embedding(query, emb)
        return emb

    def encode_document(self, documents: List[str]) -> np.ndarray:
        if not documents:
            raise ValueError("Documents list cannot be empty")
        for d in documents:
            if not self._validate_text(d):
                raise self._bad_text("document")
        got: Dict[int, np.ndarray] = {}
        todo = []
        for i, d in enumerate(documents):
            hit = self._get_cached_embedding(d)
            if hit is not None:
                got[i] = hit
            else:
                todo.append(i)
        if todo:
            new = self._get_model().encode([documents[i] for i in todo], batch_size=self.config.batch_size,
                                           convert_to_numpy=True)
            for i, e in zip(todo, new):
                self._cache_embedding(documents[i], e)
                got[i] = e
        return np.array([got[i] for i in range(len(documents))])

    GPU_SIMILARITY_MIN_ELEMS = 1 << 22   # below this the upload costs more than numpy's product

    def similarity(self, query_embedding: np.ndarray, document_embeddings: np.ndarray,
                   use_gpu: Optional[bool] = None) -> np.ndarray:
        """Cosine similarity, shape (1, N) (reference :228-237).  ``use_gpu``: None = when a GPU is
        present and the matrix is large; True/False force the HIP scan / the numpy product."""
        q = np.asarray(query_embedding)
        D = np.asarray(document_embeddings)
        if use_gpu is None:
            use_gpu = D.ndim == 2 and D.size >= self.GPU_SIMILARITY_MIN_ELEMS and self._gpu_present()
        if use_gpu:
            return self._similarity_gpu(q, D)
        qn = q / np.linalg.norm(q)
        Dn = D / np.linalg.norm(D, axis=1, keepdims=True)
        return np.dot(Dn, qn).reshape(1, -1)

    @staticmethod
    def _gpu_present() -> bool:
        try:
            import torch
            return torch.cuda.is_available()
        except Exception:
            return False

    def _similarity_gpu(self, q: np.ndarray, D: np.ndarray) -> np.ndarray:
        """The same cosine with the N products taken on the MI355X: rows are normalised while they
        are laid out in HBM (x/(|x|+1e-8), fp32 storage) and one dense scan writes <row, q/|q|> for
        every row, in row order.  Differs from the numpy form by the 1e-8 in the row norm
        (relative 1e-8 for unit-scale rows) and by fp32 accumulation order; an all-zero row gives
        0 where numpy gives NaN."""
        from .index import FlatIPIndex   # raises without libtristage.so / a GPU: no silent fallback
        idx = FlatIPIndex(int(D.shape[1]), dtype="f32")
        try:
            idx.add(np.ascontiguousarray(D, dtype=np.float32), normalize=True)
            qn = (q / np.linalg.norm(q)).astype(np.float32).reshape(1, -1)
            return idx.scores(qn).astype(np.float64 if D.dtype == np.float64 else np.float32).reshape(1, -1)
        finally:
            idx.close()

    def clear_cache(self) -> None:
        self._cache.clear()
