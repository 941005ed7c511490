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

  BM25     its documents' postings, scored with the CORPUS-WIDE statistics (document
           frequencies, count and average length all-gathered once at fit time):
           bit-identical scores; per-shard top-k lists all-gathered and merged in the
           reference's order (ShardedBM25).

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


class ShardedBM25:
    """BM25 over a row-sharded corpus: every rank indexes ITS documents with the corpus-wide statistics (document
    frequencies, document count and average length all-gathered once at fit time), so a document's score is bit for bit
    the one reference src/stage1_retriever.py:83-101 gives it over the whole corpus; a query is scored on every shard
    (HIP scorer when the local index lives on the GPU), the per-shard top-k lists (global ids) are all-gathered and
    merged in the reference's order — score descending, ties by ascending id (its stable sort over ascending ids,
    :103-112).  Same interface as BM25Index as far as Stage1Retriever uses it."""

    def __init__(self, dist, group, world_size: int, rank: int, lo: int, gpu_device=None):
        self._dist, self.group, self.world_size, self.rank, self.lo = dist, group, world_size, rank, int(lo)
        self.local = BM25Index(gpu_device=gpu_device)
        self.n_total = 0

    def _exchange(self, df, total_len, n_docs):
        parts: List[Any] = [None] * self.world_size
        self._dist.all_gather_object(parts, (df, int(total_len), int(n_docs)), group=self.group)
        g: Dict[str, int] = {}
        tl = nd = 0
        for d_, t_, n_ in parts:
            for k, v in d_.items():
                g[k] = g.get(k, 0) + int(v)
            tl += t_
            nd += n_
        self.n_total = nd
        return g, tl, nd

    def fit(self, my_documents: Sequence[str]) -> None:
        self.local.fit(my_documents, stats_exchange=self._exchange if self.world_size > 1 else None)
        self.local.documents = []
        if self.world_size == 1:
            self.n_total = self.local.corpus_size

    @property
    def vocabulary(self):
        return self.local.vocabulary

    @property
    def corpus_size(self) -> int:
        return self.n_total

    def _merge(self, ids: np.ndarray, scores: np.ndarray, k: int):
        """ids int64 [nq, k_local] (global, -1 = nothing), scores float64 -> the global top-k of every query."""
        R = self.world_size
        nq, kl = ids.shape
        if R > 1:
            t = torch.empty((R, nq, kl, 2), dtype=torch.float64)
            mine = torch.stack([torch.from_numpy(ids.astype(np.float64)), torch.from_numpy(scores)], dim=-1)   # ids < 2^53: exact
            self._dist.all_gather_into_tensor(t.view(R, -1), mine.reshape(1, -1).contiguous(), group=self.group)
            ids = t[..., 0].permute(1, 0, 2).reshape(nq, R * kl).numpy().astype(np.int64)
            scores = t[..., 1].permute(1, 0, 2).reshape(nq, R * kl).numpy()
        out = []
        want = min(int(k), self.n_total)
        for q in range(nq):
            keep = ids[q] >= 0
            i, s_ = ids[q][keep], scores[q][keep]
            order = np.lexsort((i, -s_))[:want]            # score descending, then id ascending
            out.append((i[order], s_[order]))
        return out

    def search_many_arrays(self, queries: Sequence[str], top_k: int = 10):
        k = int(top_k)
        kl = min(k, max(self.local.corpus_size, 0))
        got = self.local.search_many_arrays(list(queries), kl) if kl > 0 else [(np.zeros(0, np.int64), np.zeros(0))] * len(queries)
        ids = np.full((len(queries), k), -1, dtype=np.int64)
        sc = np.zeros((len(queries), k), dtype=np.float64)
        for q, (i, s_) in enumerate(got):
            ids[q, : len(i)] = np.asarray(i, dtype=np.int64) + self.lo
            sc[q, : len(i)] = s_
        return self._merge(ids, sc, k)

    def search_many(self, queries: Sequence[str], top_k: int = 10):
        return [list(zip(i.tolist(), s_.tolist())) for i, s_ in self.search_many_arrays(queries, top_k)]

    def search(self, query: str, top_k: int = 10):
        return self.search_many([query], top_k)[0]


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
        self._add(list(documents[lo:hi]), None if metadata is None else list(metadata[lo:hi]), n, lo)

    def add_documents_shard(self, my_documents: List[str], n_total: int,
                            metadata: Optional[List[Dict[str, Any]]] = None):
        """Each rank passes ONLY the documents of its own rows, shard_bounds(n_total, R, rank), in row order — the
        ingestion call for corpora no single host should hold (configs[3]: 10 M documents)."""
        lo, hi = shard_bounds(int(n_total), self.world_size, self.rank)
        if len(my_documents) != hi - lo:
            raise ValueError(f"rank {self.rank} of {self.world_size} owns rows [{lo}, {hi}) of {n_total}: expected "
                             f"{hi - lo} documents, got {len(my_documents)}")
        self._add(list(my_documents), metadata, int(n_total), lo)

    def _place_on_this_ranks_gpu(self) -> None:
        """One process per GPU: the HIP index and the BM25 postings of this rank live on the process's CURRENT device
        (the launcher's LOCAL_RANK via torch.cuda.set_device), not on Stage1Config's default device 0."""
        s1 = self.stage1
        if torch.cuda.is_available() and s1._device_path():
            s1.config.gpu_index_device = int(torch.cuda.current_device())

    def _add(self, mine: List[str], my_meta, n: int, lo: int) -> None:
        if self._indexed:
            raise ValueError("ShardedRetrievalPipeline takes the corpus in one add_documents call")
        if not self.stage1:
            self.initialize_stages()
        s1 = self.stage1
        self._place_on_this_ranks_gpu()
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
            # the lexical index is sharded like everything else: local postings, corpus-wide statistics
            s1.bm25_index = ShardedBM25(self._dist, self.group, self.world_size, self.rank, lo, gpu_device=s1._bm25_device())
            s1.bm25_index.fit(mine)
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
    def _shard_files(self, index_path: Optional[str]):
        import os
        if index_path is None:
            index_path = os.path.join(self.config.index_dir, "pipeline_index.pkl")
        base = os.path.splitext(index_path)[0] + f".shard{self.rank}of{self.world_size}"
        return base + ".json", base + ".matrix.npy", base + ".stage2_tokens.safetensors"

    def save_index(self, index_path: Optional[str] = None):
        """Every rank writes what IT holds next to `index_path`: `<base>.shard<r>of<R>.json` (rows, documents and
        metadata of the shard; JSON like the single-process manifest, never a pickle), `.matrix.npy` (its rows of
        the corpus matrix) and `.stage2_tokens.safetensors` (its token matrices).  The reference's counterpart is
        RetrievalPipeline.save_index (src/retrieval_pipeline.py:450-470) over one process."""
        import json
        import os
        if not self.stage1 or not self._indexed:
            raise ValueError("Pipeline not initialized")
        manifest_path, matrix_path, tokens_path = self._shard_files(index_path)
        os.makedirs(os.path.dirname(os.path.abspath(manifest_path)), exist_ok=True)
        s1 = self.stage1
        local = s1.faiss_index.local_index
        mat = local.reconstruct_n(0, local.ntotal) if local.ntotal else np.zeros((0, int(s1.faiss_index.d)), np.float32)
        np.save(matrix_path, np.asarray(mat, dtype=np.float32))
        have_tokens = bool(self.stage2 is not None and self.stage2.config.precompute_document_embeddings
                           and self.stage2.save_token_store(tokens_path))
        manifest = {"format": "tristage-rag_amd/shard/1", "world_size": self.world_size, "rank": self.rank,
                    "n_total": self.n_total, "rows": [self.lo, self.hi], "dim": int(s1.faiss_index.d),
                    "documents": list(s1.documents.items), "doc_metadata": list(s1.doc_metadata.items),
                    "matrix": os.path.basename(matrix_path),
                    "stage2_tokens": os.path.basename(tokens_path) if have_tokens else None}
        with open(manifest_path, "w") as f:
            json.dump(manifest, f)
        if self.world_size > 1:
            self._dist.barrier(group=self.group)
        self.logger.info(f"Pipeline shard {self.rank}/{self.world_size} saved to {manifest_path}")

    def load_index(self, index_path: Optional[str] = None):
        """Restore what save_index wrote for THIS rank of the SAME world size: the corpus rows go back into the HIP
        index as they were stored (no re-encoding), the token store comes from its file (re-encoded if the file is
        missing or was written for another model), the stage-3 token ids and the BM25 postings are recomputed from
        the shard's text (the BM25 statistics are all-gathered again)."""
        import json
        import os
        if self._indexed:
            raise ValueError("ShardedRetrievalPipeline takes the corpus in one add_documents / load_index call")
        if not self.stage1:
            self.initialize_stages()
        manifest_path, matrix_path, tokens_path = self._shard_files(index_path)
        if not os.path.exists(manifest_path):
            raise ValueError(f"{manifest_path} not found: a sharded index is loaded by the same number of ranks that saved it")
        manifest = json.load(open(manifest_path))
        if (manifest.get("format") != "tristage-rag_amd/shard/1" or manifest["world_size"] != self.world_size
                or manifest["rank"] != self.rank):
            raise ValueError(f"{manifest_path} was not written by rank {self.rank} of {self.world_size}")
        s1 = self.stage1
        self._place_on_this_ranks_gpu()
        n, (lo, hi) = int(manifest["n_total"]), manifest["rows"]
        if (lo, hi) != tuple(shard_bounds(n, self.world_size, self.rank)):
            raise ValueError("shard bounds of the file do not match this world size")
        mine = list(manifest["documents"])
        self.lo, self.hi, self.n_total = lo, hi, n
        s1.documents = ShardedList(n, lo, mine, missing=None)
        s1.doc_metadata = ShardedList(n, lo, list(manifest["doc_metadata"]), missing={})
        mat = np.load(os.path.join(os.path.dirname(os.path.abspath(manifest_path)), manifest["matrix"]), allow_pickle=False)
        d = int(manifest["dim"])
        local = s1._index_factory(d) if s1._index_factory is not None else None
        index = ShardedFlatIPIndex(d, n, dtype=s1.config.index_dtype, device=s1.config.gpu_index_device,
                                   group=self.group, local_index=local, merge_fn=getattr(self, "_merge_fn", None))
        if mat.shape[0]:
            index.local_index.add(np.ascontiguousarray(mat, dtype=np.float32))   # stored rows are already normalised
        s1.faiss_index = index
        if s1.config.enable_bm25:
            s1.bm25_index = ShardedBM25(self._dist, self.group, self.world_size, self.rank, lo, gpu_device=s1._bm25_device())
            s1.bm25_index.fit(mine)
        if self.stage2 is not None and self.stage2.config.precompute_document_embeddings and mine:
            ok = bool(manifest.get("stage2_tokens")) and self.stage2.load_token_store(tokens_path, len(mine))
            if not ok:
                self.stage2.reset_token_store()
                self.stage2.index_documents(mine, lo)
        if self.stage3 is not None and self.config.stage3_cache_document_tokens and mine:
            self.stage3._pairs = None
            self.stage3.index_documents(mine, lo)
        if self.world_size > 1:
            for st in (self.stage2, self.stage3):
                if st is not None:
                    st.owner_compact = True
        self._install_owner_scoring()
        self._indexed = True
        self.logger.info(f"Pipeline shard {self.rank}/{self.world_size} loaded from {manifest_path}")
