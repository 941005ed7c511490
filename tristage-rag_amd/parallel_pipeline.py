"""Three-stage pipeline over R GPUs of one node (one process per GPU, RCCL).

SURVEY.md §8e.  The reference has no multi-device code (its batch_search is a
sequential loop, src/retrieval_pipeline.py:444-448); this is the scale-out of the
same ``RetrievalPipeline.search``.  EVERYTHING that grows with the corpus is
row-sharded with the stage-1 rows — rank r of R owns documents
``shard_bounds(N, R, r)`` and nothing of the others':

  stage 1  its rows of the corpus matrix (encoded and indexed by rank r only: the
           index build is data parallel); a query batch is searched on every shard,
           the partial top-k lists are all-gathered and merged
           (tristage_rag_amd.sharded).  Ids are global.
  stage 2  its documents' token matrices (the resident token store,
           reference src/stage2_rescorer.py:244-301 re-encodes instead): after the
           merge every rank scores, with ONE ts_maxsim_indexed_batch launch, the
           candidates it OWNS (lo <= id < hi) into a [B, k] matrix pre-filled with
           -inf; ONE all-reduce(MAX) of that matrix puts every score on every rank
           (each candidate has exactly one owner); the stable sort / keep-100 is
           replicated.
  stage 3  its documents' cross-encoder token ids (reference
           src/stage3_reranker.py:230-264 tokenises text pairs per query): the
           owner assembles and scores the (query, document) pairs of the kept
           candidates it owns, ONE all-reduce(MAX) of the [B, 100] matrix, then the
           replicated float64 min-max and stable sort.
  text     its documents' text and metadata; the records a call returns are
           completed by their owners through one all-gather of the few texts
           involved (top-k per query).

Round 2 kept the token store, the stage-3 id cache and all text WHOLE on every
rank: at ~100 tokens x 768 x 2 B per document one 288 GB GPU holds < 1.9 M
documents, so the three-stage pipeline did not exist at the 10 M / 50 M-document
sizes of BASELINE.json's configs[3] / [4].  Sharded with the rows, 8 GPUs hold
~15 M documents of that size and every stage's work splits 1/R as well (for a
corpus whose relevant documents are spread over the shards).

BM25 statistics need the whole corpus: with ``add_documents`` (every rank is given
the full list) the BM25 index is replicated and fitted from that list; with
``add_documents_shard`` (each rank is given only its rows) BM25 must be off.
Every rank issues the same sequence of calls (the collectives are matched by
construction: their cadence never depends on rank-local state).
"""
from __future__ import annotations

import time
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch

from .retrieval_pipeline import PipelineConfig, RetrievalPipeline
from .sharded import ShardedFlatIPIndex, shard_bounds
from .stage1_retriever import BM25Index


def _world(group):
    import torch.distributed as dist
    if not dist.is_initialized():
        return dist, 1, 0
    return dist, dist.get_world_size(group), dist.get_rank(group)


class ShardedList:
    """List-shaped view of a row-sharded sequence: ``len()`` is the GLOBAL length, item i is the real item on the
    rank that owns row i and ``missing`` elsewhere.  Stands where the reference keeps ``Stage1Retriever.documents``
    / ``doc_metadata`` (src/stage1_retriever.py:127-128), which callers index by global doc id."""

    def __init__(self, n_total: int, lo: int, items: Sequence[Any], missing: Any = None):
        self.n_total, self.lo, self.hi = int(n_total), int(lo), int(lo) + len(items)
        self.items = list(items)
        self.missing = missing

    def __len__(self) -> int:
        return self.n_total

    def owns(self, i: int) -> bool:
        return self.lo <= i < self.hi

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self.n_total))]
        i = int(i)
        if i < 0:
            i += self.n_total
        if not 0 <= i < self.n_total:
            raise IndexError(i)
        return self.items[i - self.lo] if self.lo <= i < self.hi else self.missing

    def __iter__(self):
        return (self[i] for i in range(self.n_total))

    def extend(self, more) -> None:
        raise ValueError("a row-sharded corpus is given in ONE add_documents call")


class ShardedRetrievalPipeline(RetrievalPipeline):
    def __init__(self, config_path: Optional[str] = None, config: Optional[PipelineConfig] = None,
                 group=None):
        super().__init__(config_path=config_path, config=config)
        self.group = group
        self._dist, self.world_size, self.rank = _world(group)
        self._indexed = False
        self.lo = self.hi = self.n_total = 0

    # -- collectives -------------------------------------------------------------
    def _host_staged(self) -> bool:
        return self.world_size > 1 and self._dist.get_backend(self.group) == "gloo"

    def _all_reduce_max(self, t: torch.Tensor) -> torch.Tensor:
        """Element-wise MAX over the ranks, in place where the backend can (RCCL on device memory); a gloo group —
        the CPU tests, or ranks sharing one GPU — is staged through the host."""
        if self.world_size == 1:
            return t
        if t.is_cuda and self._host_staged():
            h = t.cpu()
            self._dist.all_reduce(h, op=self._dist.ReduceOp.MAX, group=self.group)
            t.copy_(h)
            return t
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self.group)
        return t

    def _all_agree(self, flag: bool) -> bool:
        if self.world_size == 1:
            return bool(flag)
        flags: List[Any] = [None] * self.world_size
        self._dist.all_gather_object(flags, bool(flag), group=self.group)
        return all(flags)

    # -- indexing: each rank encodes and stores only its row shard ---------------
    def add_documents(self, documents: List[str], metadata: Optional[List[Dict[str, Any]]] = None):
        """Every rank passes the SAME full list; each keeps (encodes, tokenises, stores) only its rows."""
        n = len(documents)
        lo, hi = shard_bounds(n, self.world_size, self.rank)
        self._add(list(documents[lo:hi]), None if metadata is None else list(metadata[lo:hi]), n, lo,
                  all_documents=documents)

    def add_documents_shard(self, my_documents: List[str], n_total: int,
                            metadata: Optional[List[Dict[str, Any]]] = None):
        """Each rank passes ONLY the documents of its own rows, shard_bounds(n_total, R, rank), in row order — the
        ingestion call for corpora no single host should hold (configs[3]: 10 M documents).  BM25 needs corpus-wide
        statistics and has to be off here."""
        lo, hi = shard_bounds(int(n_total), self.world_size, self.rank)
        if len(my_documents) != hi - lo:
            raise ValueError(f"rank {self.rank} of {self.world_size} owns rows [{lo}, {hi}) of {n_total}: expected "
                             f"{hi - lo} documents, got {len(my_documents)}")
        self._add(list(my_documents), metadata, int(n_total), lo, all_documents=None)

    def _add(self, mine: List[str], my_meta, n: int, lo: int, all_documents) -> None:
        if self._indexed:
            raise ValueError("ShardedRetrievalPipeline takes the corpus in one add_documents call")
        if not self.stage1:
            self.initialize_stages()
        s1 = self.stage1
        if s1.config.enable_bm25 and all_documents is None:
            raise ValueError("BM25 needs corpus-wide statistics: use add_documents (full list on every rank) or "
                             "set stage1_enable_bm25=False")
        hi = lo + len(mine)
        self.lo, self.hi, self.n_total = lo, hi, n
        s1.documents = ShardedList(n, lo, mine, missing=None)
        s1.doc_metadata = ShardedList(n, lo, my_meta if my_meta is not None else [{}] * len(mine), missing={})
        if s1._device_path():
            emb = s1._encode_batch_tensor(mine, bulk=True) if mine else None
            d = int(emb.shape[1]) if emb is not None else int(s1.embedding_dim)
            local = None
            normalize = True
        else:
            emb = s1._normalize_embeddings(s1._encode_batch(mine)) if mine else None
            d = int(emb.shape[1]) if emb is not None else int(s1.embedding_dim)
            local = s1._index_factory(d) if s1._index_factory is not None else None
            normalize = False
        index = ShardedFlatIPIndex(d, n, dtype=s1.config.index_dtype, device=s1.config.gpu_index_device,
                                   group=self.group, local_index=local,
                                   merge_fn=getattr(self, "_merge_fn", None))
        if emb is not None:
            if normalize:
                index.local_index.add(emb, normalize=True)
            else:
                index.local_index.add(emb)
        s1.faiss_index = index
        if s1.config.enable_bm25:
            s1.bm25_index = BM25Index(gpu_device=s1._bm25_device())
            s1.bm25_index.fit(all_documents)
            s1.bm25_index.documents = []      # (fit keeps a copy of the texts it was given; nothing reads it afterwards)
        if self.stage2 is not None and self.stage2.config.precompute_document_embeddings and mine:
            self.stage2.index_documents(mine, lo)          # token matrices of MY rows only
        if self.stage3 is not None and self.config.stage3_cache_document_tokens and mine:
            self.stage3.index_documents(mine, lo)          # cross-encoder token ids of MY rows only
        if self.world_size > 1:      # a rank owns ~1/R of any candidate list: score the owned ones only (ragged launches)
            for st in (self.stage2, self.stage3):
                if st is not None:
                    st.owner_compact = True
        self._install_owner_scoring()
        self._indexed = True

    # -- the per-record path (search(): one query at a time, result dictionaries) ---------------------------
    def _install_owner_scoring(self) -> None:
        """stage2.score_candidates / stage3.raw_scores see a query's whole candidate list on every rank; each rank
        computes the entries whose document it owns and one all-reduce(MAX) completes the vector."""
        if self.world_size == 1:
            return
        s2, s3 = self.stage2, self.stage3
        base2, base3 = s2.score_candidates, s3.raw_scores
        neg = float("-inf")

        def score_candidates(query, candidates):
            own = [i for i, c in enumerate(candidates) if self.lo <= int(c.get("doc_id", -1)) < self.hi]
            vec = torch.full((len(candidates),), neg, dtype=torch.float32)
            if own:
                vec[own] = torch.tensor(base2(query, [candidates[i] for i in own]), dtype=torch.float32)
            return [float(x) for x in self._all_reduce_max(vec).tolist()]

        def raw_scores(query, documents):
            own = [i for i, d in enumerate(documents) if d is not None]      # text lives with its owner only
            vec = torch.full((len(documents),), neg, dtype=torch.float32)
            if own:
                vec[own] = torch.tensor([float(x) for x in base3(query, [documents[i] for i in own])], dtype=torch.float32)
            return [float(x) for x in self._all_reduce_max(vec).tolist()]

        s2.score_candidates = score_candidates
        s3.raw_scores = raw_scores

    def _complete_records(self, results: List[Dict[str, Any]]) -> None:
        """Fill ``document`` / ``metadata`` of the records the calls return: every rank knows the same doc ids, the
        owners contribute text and metadata, ONE all-gather of those few items."""
        if self.world_size == 1:
            return
        docs, meta = self.stage1.documents, self.stage1.doc_metadata
        ids = set()           # the SAME set on every rank (what a rank lacks is exactly what it does not own)
        for res in results:
            for key in ("results", "stage1_results", "stage2_results"):
                for r in res.get(key) or []:
                    if isinstance(r, dict):
                        ids.add(int(r["doc_id"]))
        mine = {i: (docs[i], meta[i]) for i in ids if docs.owns(i)}
        gathered: List[Any] = [None] * self.world_size
        self._dist.all_gather_object(gathered, mine, group=self.group)
        have: Dict[int, Any] = {}
        for g in gathered:
            have.update(g)
        for res in results:
            for key in ("results", "stage1_results", "stage2_results"):
                for r in res.get(key) or []:
                    if isinstance(r, dict) and r.get("document") is None and int(r["doc_id"]) in have:
                        r["document"], r["metadata"] = have[int(r["doc_id"])]

    def search(self, query: str, top_k: Optional[int] = None) -> Dict[str, Any]:
        if self.world_size == 1:
            return super().search(query, top_k)
        if self.config.search_on_arrays and self._arrays_agreed():
            return self.search_many([query], top_k)[0]
        return self._search_records(query, top_k)

    def _search_records(self, query: str, top_k: Optional[int]) -> Dict[str, Any]:
        if not self.stage1 or not self.stage2 or not self.stage3:
            self.initialize_stages()
        top_k = top_k or self.config.stage3_top_k
        total_start = self._now()
        t = self._now()
        stage1_results = self.stage1.search(query, self.config.stage1_top_k)      # collective (sharded index)
        stage1_time = time.time() - t if t else None
        out = self._run_later_stages(query, top_k, stage1_results, total_start, stage1_time)
        self._complete_records([out])
        return out

    # -- the array path ------------------------------------------------------------------------------------
    def _arrays_ready(self) -> bool:
        """Local preconditions of the array path.  A rank whose shard is EMPTY (fewer documents than ranks) has no
        token store and no id cache and is ready by definition: it contributes -inf everywhere."""
        s1, s2, s3 = self.stage1, self.stage2, self.stage3
        if not hasattr(s1, "search_many_arrays") or not self._indexed:
            return False
        if self.hi == self.lo:
            return True
        return bool(getattr(s2, "token_store", None) is not None and len(s2.token_store) == self.hi - self.lo
                    and getattr(s3, "_pairs_usable", False))

    def _arrays_agreed(self) -> bool:
        cached = getattr(self, "_arrays_ok", None)
        if cached is None:        # a property of the stores, which do not change after add_documents: agreed on once
            cached = self._arrays_ok = self._all_agree(self._arrays_ready())
        return cached

    def _work_device(self):
        s2 = self.stage2
        if getattr(s2, "token_store", None) is not None and s2.token_store.data is not None:
            return s2.token_store.data.device
        return torch.device(str(s2.device)) if s2 is not None else torch.device("cpu")

    def _arrays_stage1(self, queries: List[str]):
        got = self.stage1.search_many_arrays(queries, self.config.stage1_top_k)   # collective inside
        if got is None:
            return None
        ids1, sc1 = got
        dev = self._work_device()
        ids1_dev = ids1.to(dev) if torch.is_tensor(ids1) else torch.from_numpy(ids1).to(dev)
        return ids1_dev, sc1

    def _arrays_stage23(self, queries: List[str], ids1_dev):
        """Stages 2 and 3 of ALL queries on every rank, each rank scoring what it owns; two all-reduces."""
        if self.world_size == 1:
            return super()._arrays_stage23(queries, ids1_dev)
        ma = self._tick()
        sc2_all = self._all_reduce_max(self.stage2.score_arrays_partial(queries, ids1_dev))
        bad = bool(torch.isinf(sc2_all).any())       # a candidate nobody owns: identical on every rank after the reduce
        if bad:
            return None
        pos2, sc2 = self.stage2.keep_top_arrays(sc2_all)
        ids2_dev = torch.gather(ids1_dev.to(pos2.device), 1, pos2)
        mb = self._tick()
        raw3 = self._all_reduce_max(self.stage3.raw_arrays_partial(queries, ids2_dev))
        if bool(torch.isinf(raw3).any()):
            return None
        pos3, sc3 = self.stage3.finish_arrays(raw3)
        mc = self._tick()
        out = (pos2.cpu().numpy(), sc2.cpu().numpy(), pos3.cpu().numpy(), sc3.cpu().numpy())
        return out + (self._span(ma, mb), self._span(mb, mc))

    def search_many(self, queries: List[str], top_k: Optional[int] = None) -> List[Dict[str, Any]]:
        """Batched search over R ranks; every rank returns every query's records.  Array path (token store + id
        cache on the shards): three collectives per call besides stage 1's — two all-reduces of small score matrices
        and one gather of the returned texts.  Otherwise the per-record path, query by query."""
        if self.world_size == 1:
            return super().search_many(queries, top_k)
        if not self.stage1 or not self.stage2 or not self.stage3:
            self.initialize_stages()
        top_k = top_k or self.config.stage3_top_k
        queries = list(queries)
        if not queries:
            return []
        if self._arrays_agreed():
            import gc
            gc_was_on = gc.isenabled()
            gc.disable()          # (see RetrievalPipeline.search_many)
            try:
                fast = self._search_many_arrays(queries, top_k)
            finally:
                if gc_was_on:
                    gc.enable()
            # a None here is a property of the merged ids / reduced scores, i.e. the same on every rank
            if fast is not None:
                self._complete_records(fast)
                return fast
        return [self._search_records(q, top_k) for q in queries]

    # -- introspection -------------------------------------------------------------------------------------
    def shard_info(self) -> Dict[str, Any]:
        """What THIS rank holds (bytes of HBM / host memory that grow with the corpus)."""
        s1, s2, s3 = self.stage1, self.stage2, self.stage3
        info: Dict[str, Any] = {"rank": self.rank, "world_size": self.world_size, "rows": [self.lo, self.hi],
                                "documents_total": self.n_total, "documents_held": self.hi - self.lo}
        idx = getattr(getattr(s1, "faiss_index", None), "local_index", None)
        if idx is not None and hasattr(idx, "storage_dtype"):
            esz = 4 if idx.storage_dtype == "f32" else 2
            gran = 64 if esz == 4 else 128
            info["stage1_index_bytes"] = int(-(-idx.ntotal // 32) * 32 * (-(-idx.d // gran) * gran) * esz)
        st = getattr(s2, "token_store", None)
        if st is not None and st.data is not None:
            info["stage2_token_rows"] = int(st.rows)
            info["stage2_token_store_bytes"] = int(st.rows) * int(st.data.shape[1]) * st.data.element_size()
            info["stage2_token_store_allocated_bytes"] = int(st.data.numel()) * st.data.element_size()
        pa = getattr(s3, "_pairs", None)
        if pa is not None:
            info["stage3_id_cache_documents"] = len(pa)
            info["stage3_id_cache_bytes"] = int(sum(len(x) for x in pa._doc_ids)) * 4
        docs = getattr(s1, "documents", None)
        if isinstance(docs, ShardedList):
            info["text_bytes_held"] = int(sum(len(x) for x in docs.items))
        return info

    def get_pipeline_info(self) -> Dict[str, Any]:
        info = super().get_pipeline_info()
        info["sharding"] = self.shard_info()
        return info

    # -- persistence: one file set per rank ----------------------------------------------------------------
    def save_index(self, index_path: Optional[str] = None):
        raise ValueError("a row-sharded pipeline is rebuilt from its documents (add_documents / add_documents_shard): "
                         "per-rank persistence is not part of this build")

    def load_index(self, index_path: Optional[str] = None):
        raise ValueError("a row-sharded pipeline is rebuilt from its documents (add_documents / add_documents_shard): "
                         "per-rank persistence is not part of this build")
