"""Three-stage pipeline over R GPUs of one node (one process per GPU, RCCL).

SURVEY.md §8e.  The reference has no multi-device code (its batch_search is a
sequential loop, src/retrieval_pipeline.py:444-448); this is the scale-out of the
same ``RetrievalPipeline.search``:

  stage 1  the corpus matrix is row-sharded: rank r encodes and indexes only rows
           shard_bounds(N, R, r) (index build is data parallel too); a query is
           searched on every shard, partial top-k lists are all-gathered and merged
           (tristage_rag_amd.sharded).  BM25 statistics need the whole corpus and
           are replicated.
  stage 2  replicas: every rank holds the encoder; rank r scores candidates r, r+R, …
           and the float32 scores are all-gathered.
  stage 3  same for the cross-encoder pairs; min-max normalisation and sorting happen
           after the gather, so every rank returns the identical result — the
           single-GPU one up to batch-padding noise.

Documents are passed identically to every rank, in ONE ``add_documents`` call.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

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


def gather_interleaved(local_scores: List[float], n_total: int, group=None, device="cpu") -> List[float]:
    """Rank r computed the scores of items r, r+R, r+2R, …; returns all n_total scores in
    item order on every rank (one all-gather of float32)."""
    dist, R, rank = _world(group)
    if R == 1:
        return list(local_scores)
    per = -(-n_total // R)
    buf = torch.full((per,), float("nan"), dtype=torch.float32, device=device)
    if local_scores:
        buf[: len(local_scores)] = torch.tensor(local_scores, dtype=torch.float32, device=device)
    out = torch.empty((R * per,), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(out, buf, group=group)
    table = out.view(R, per).cpu().numpy()
    return [float(table[i % R, i // R]) for i in range(n_total)]


class ShardedRetrievalPipeline(RetrievalPipeline):
    def __init__(self, config_path: Optional[str] = None, config: Optional[PipelineConfig] = None,
                 group=None):
        super().__init__(config_path=config_path, config=config)
        self.group = group
        self._dist, self.world_size, self.rank = _world(group)
        self._indexed = False

    def _comm_device(self):
        backend = self._dist.get_backend(self.group) if self.world_size > 1 else "gloo"
        return torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")

    # -- indexing: each rank encodes and stores only its row shard ---------------
    def add_documents(self, documents: List[str], metadata: Optional[List[Dict[str, Any]]] = None):
        if self._indexed:
            raise ValueError("ShardedRetrievalPipeline takes the corpus in one add_documents call")
        if not self.stage1:
            self.initialize_stages()
        s1 = self.stage1
        n = len(documents)
        lo, hi = shard_bounds(n, self.world_size, self.rank)
        s1.documents.extend(documents)
        s1.doc_metadata.extend(metadata if metadata is not None else [{}] * n)
        mine = list(documents[lo:hi])
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
            s1.bm25_index.fit(s1.documents)
        if self.stage2 is not None and self.stage2.config.precompute_document_embeddings:
            self.stage2.index_documents(list(documents), 0)   # replicas keep the whole token store
        if self.stage3 is not None and self.config.stage3_cache_document_tokens:
            self.stage3.index_documents(list(documents), 0)
        self._install_data_parallel_scoring()
        self._indexed = True

    # -- stages 2 and 3: replicas, work split by candidate ------------------------
    def _install_data_parallel_scoring(self) -> None:
        if self.world_size == 1:
            return
        R, rank, group, dev = self.world_size, self.rank, self.group, self._comm_device()
        s2, s3 = self.stage2, self.stage3
        base2, base3 = s2.score_candidates, s3.raw_scores

        def score_candidates(query, candidates):
            mine = candidates[rank::R]
            local = base2(query, mine) if mine else []
            return gather_interleaved(local, len(candidates), group, dev)

        def raw_scores(query, documents):
            mine = documents[rank::R]
            local = [float(x) for x in base3(query, mine)] if mine else []
            return gather_interleaved(local, len(documents), group, dev)

        s2.score_candidates = score_candidates
        s3.raw_scores = raw_scores
        self._local_scoring = (base2, base3)
        self._parallel_scoring = (score_candidates, raw_scores)

    def search_many(self, queries: List[str], top_k: Optional[int] = None) -> List[Dict[str, Any]]:
        """Batched search over R ranks: stage 1 is collective (every rank sweeps its row shard
        for all queries, one all-gather + merge per 64 queries); stages 2 and 3 are split BY
        QUERY — rank r rescoring and reranking queries r, r+R, … with its replica of the
        encoders and of the token store, no collective inside — and one all-gather of the
        finished records puts every query's result on every rank."""
        if self.world_size == 1:
            return super().search_many(queries, top_k)
        import time
        if not self.stage1 or not self.stage2 or not self.stage3:
            self.initialize_stages()
        top_k = top_k or self.config.stage3_top_k
        queries = list(queries)
        if not queries:
            return []
        n, R, rank = len(queries), self.world_size, self.rank
        fast = self._search_many_arrays_sharded(queries, top_k)
        if fast is not None:
            return fast
        total_start = self._now()
        t = self._now()
        s1 = self.stage1.search_many(queries, self.config.stage1_top_k)
        t1 = (time.time() - t) / n if t else None
        mine = list(range(rank, n, R))
        # the per-candidate collectives of search() must not run here: ranks work on different queries
        self.stage2.score_candidates, self.stage3.raw_scores = self._local_scoring
        try:
            s2m, s3m, t2, t3 = self._later_stages_many([queries[i] for i in mine], [s1[i] for i in mine])
        finally:
            self.stage2.score_candidates, self.stage3.raw_scores = self._parallel_scoring
        gathered: List[Any] = [None] * R
        self._dist.all_gather_object(gathered, (s2m, s3m, t2, t3), group=self.group)
        s2: List[Any] = [None] * n
        s3: List[Any] = [None] * n
        for r, (a, b, _, _) in enumerate(gathered):
            for j, i in enumerate(range(r, n, R)):
                s2[i], s3[i] = a[j], b[j]
        t2 = max((g[2] or 0.0) for g in gathered) * len(mine) / n if t2 is not None else None
        t3 = max((g[3] or 0.0) for g in gathered) * len(mine) / n if t3 is not None else None
        total = (time.time() - total_start) / n if total_start else None
        return self._assemble_many(queries, top_k, s1, s2, s3, t1, t2, t3, total)

    def _search_many_arrays_sharded(self, queries: List[str], top_k: int):
        """The array path of RetrievalPipeline.search_many over R ranks: stage 1 collective (id / score matrices of
        ALL queries on every rank), stages 2 and 3 of queries r, r+R, ... on rank r, then ONE all-gather of four small
        arrays per rank instead of pickled record lists.  Every rank must reach the same decision (array path or
        not): the preconditions are properties of the replicated stores, and a rank-local failure afterwards is
        agreed on through the gather itself."""
        import time
        ready = self._arrays_ready()
        flags: List[Any] = [None] * self.world_size
        self._dist.all_gather_object(flags, bool(ready), group=self.group)
        if not all(flags):
            return None
        n, R, rank = len(queries), self.world_size, self.rank
        total_start = t = self._tick()
        got = self._arrays_stage1(queries)          # collective inside (ShardedFlatIPIndex.search)
        t1 = (self._tick() - t) / n if t is not None else None
        mine = list(range(rank, n, R))
        later = None
        if got is not None and mine:
            import torch
            ids1_dev, sc1 = got
            sel = torch.as_tensor(mine, device=ids1_dev.device)
            later = self._arrays_stage23([queries[i] for i in mine], ids1_dev[sel])
        payload = None if got is None or (mine and later is None) else (later if mine else ())
        gathered: List[Any] = [None] * R
        self._dist.all_gather_object(gathered, payload, group=self.group)
        if any(g is None for g in gathered):
            return None                              # some rank could not take the array path: all take the record path
        import numpy as np
        first = next(g for g in gathered if g)
        pos2 = np.zeros((n,) + first[0].shape[1:], dtype=first[0].dtype)
        sc2 = np.zeros((n,) + first[1].shape[1:], dtype=first[1].dtype)
        pos3 = np.zeros((n,) + first[2].shape[1:], dtype=first[2].dtype)
        sc3 = np.zeros((n,) + first[3].shape[1:], dtype=first[3].dtype)
        t2 = t3 = 0.0
        for r, g in enumerate(gathered):
            if not g:
                continue
            rows = list(range(r, n, R))
            pos2[rows], sc2[rows], pos3[rows], sc3[rows] = g[0], g[1], g[2], g[3]
            t2, t3 = max(t2, g[4] or 0.0), max(t3, g[5] or 0.0)
        ids1_dev, sc1 = got
        ids1_h = ids1_dev.cpu().numpy()
        sc1_h = sc1.cpu().numpy() if hasattr(sc1, "cpu") else sc1
        total = (time.time() - total_start) / n if total_start is not None else None
        timing = self.config.enable_timing
        return self._records_from_arrays(queries, top_k, ids1_h, sc1_h, pos2, sc2, pos3, sc3, t1,
                                         t2 / n if timing else None, t3 / n if timing else None, total)
