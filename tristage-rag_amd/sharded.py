"""Row-sharded stage-1 index: one process per GPU, RCCL all-gather of partial top-k.

The reference has no multi-device path at all (no
torch.distributed, batch_search is a sequential loop at
src/retrieval_pipeline.py:444-448); this is the scale-out BASELINE.json's
north_star specifies for the same `faiss_index.search` call
(src/stage1_retriever.py:380):

  * the corpus matrix is split row-wise and contiguously: rank r of R holds rows
    [r*ceil(N/R), min(N, (r+1)*ceil(N/R)))  and reports global ids (local row +
    offset) straight from the kernel;
  * every rank scans its shard for the full query batch (queries are replicated,
    64 x 768 fp16 = 96 KiB);
  * ONE collective per batch: an all-gather of each rank's packed partial result
    (B*k*(4+8) bytes, 768 KB at B=64, k=1000).  xGMI is point-to-point, so this is
    a direct peer exchange, latency-bound and far below the shard scan time; no
    ring all-reduce anywhere;
  * each rank merges the R sorted lists with the HIP select kernel
    (ts_merge_topk), so every rank ends with the identical global top-k, in the
    canonical (score desc, id asc) order — the result does not depend on R.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple


def shard_bounds(n_total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Rows [lo, hi) held by `rank`; contiguous, ceil-divided."""
    per = -(-int(n_total) // int(world_size)) if n_total > 0 else 0
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


class ShardedFlatIPIndex:
    """FAISS-shaped index over R row shards (this process owns one of them).

    ``local_index`` is a tristage_rag_amd.index.FlatIPIndex on this rank's GPU;
    ``merge_fn(scores[R,B,k], ids[R,B,k]) -> (D[B,k], I[B,k])`` defaults to the
    HIP merge kernel.  Both are attributes so the process-group plumbing can be
    exercised on CPU (gloo) with stand-ins in tests.
    """

    def __init__(self, d: int, n_total: int, dtype: str = "f16", device: Optional[int] = None,
                 group=None, local_index=None, merge_fn: Optional[Callable] = None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.d = int(d)
        self.n_total = int(n_total)
        self.lo, self.hi = shard_bounds(self.n_total, self.world_size, self.rank)
        if local_index is None:
            from .index import FlatIPIndex
            local_index = FlatIPIndex(d, dtype=dtype, device=0 if device is None else device)
            local_index.reserve(max(self.hi - self.lo, 1))
        self.local_index = local_index
        self.local_index.set_id_offset(self.lo)
        if merge_fn is None:
            from .index import merge_topk
            merge_fn = merge_topk
        self.merge_fn = merge_fn
        self._pending = []
        self.always_exchange = False  # tests: run the collective + merge even for one rank

    # -- FAISS duck type ----------------------------------------------------
    @property
    def ntotal(self) -> int:
        return self.n_total

    def add_local(self, rows) -> None:
        """Append rows of THIS rank's shard (in shard order)."""
        self.local_index.add(rows)
        if self.local_index.ntotal > self.hi - self.lo:
            raise ValueError("more rows added than this rank's shard holds")

    def add_global(self, rows) -> None:
        """Every rank passes the same full matrix; each keeps its own slice."""
        if rows.shape[0] != self.n_total:
            raise ValueError(f"expected {self.n_total} rows, got {rows.shape[0]}")
        self.local_index.add(rows[self.lo:self.hi])

    def search(self, q, k: int, async_: bool = False):
        """Global top-k for the replicated query batch `q` (tensor).  Returns
        tensors (D float32 [B,k], I int64 [B,k]) identical on every rank.

        ``async_=True``: local search, all-gather and merge are only enqueued (all
        three are stream-ordered); call :meth:`finish` before reading the result."""
        import torch
        if async_ and len(self._pending) >= 48:
            self.finish()
        if self.hi > self.lo:
            D, I = (self.local_index.search(q, k, async_=True) if async_ else self.local_index.search(q, k))
            if not torch.is_tensor(D):
                D, I = torch.from_numpy(D), torch.from_numpy(I)
        else:  # an empty shard contributes only padding
            B = q.shape[0]
            D = torch.full((B, k), -3.4028234663852886e38, dtype=torch.float32, device=q.device)
            I = torch.full((B, k), -1, dtype=torch.int64, device=q.device)
        if self.world_size == 1 and not self.always_exchange:
            return D, I
        Dg, Ig = self._all_gather(D, I)
        out = self.merge_fn(Dg, Ig)
        if async_:
            self._pending.append((q, k, D, I, out))
        return out

    def finish(self):
        """Complete asynchronous searches on every rank.  If ANY rank had to repeat a
        local search, all ranks repeat the exchange for that batch (collectively)."""
        import torch
        redone_local = set(self.local_index.finish()) if hasattr(self.local_index, "finish") else set()
        if (self.world_size == 1 and not self.always_exchange) or not self._pending:
            self._pending.clear()
            return
        # the i-th pending entry of every rank is the same batch; local tickets are
        # consecutive, so "repeated" maps to positions from the end
        last = self._last_ticket()
        flags = torch.zeros(len(self._pending), dtype=torch.int32, device=self._pending[0][4][0].device)
        for t in redone_local:
            pos = len(self._pending) - 1 - (last - t)
            if 0 <= pos < len(self._pending):
                flags[pos] = 1
        self._dist.all_reduce(flags, op=self._dist.ReduceOp.MAX, group=self.group)
        for pos in torch.nonzero(flags).flatten().tolist():
            q, k, D, I, out = self._pending[pos]
            Dg, Ig = self._all_gather(D, I)          # D, I were corrected in place by local finish()
            Dn, In = self.merge_fn(Dg, Ig)
            out[0].copy_(Dn)
            out[1].copy_(In)
        torch.cuda.current_stream().synchronize() if flags.is_cuda else None
        self._pending.clear()

    def _last_ticket(self) -> int:
        li = self.local_index
        return int(li._lib.ts_index_last_ticket(li._h)) if hasattr(li, "_lib") else -1

    def _all_gather(self, D, I):
        """One collective: [scores | ids] packed as bytes -> [R, B, k] pair."""
        import torch
        B, k = D.shape
        packed = torch.cat([D.contiguous().view(torch.uint8).reshape(-1),
                            I.contiguous().view(torch.uint8).reshape(-1)])
        flat = torch.empty((self.world_size * packed.numel(),), dtype=torch.uint8, device=packed.device)
        self._dist.all_gather_into_tensor(flat, packed, group=self.group)  # 1-D: gloo and RCCL both take it
        out = flat.view(self.world_size, packed.numel())
        ns = B * k * 4
        Dg = out[:, :ns].contiguous().view(torch.float32).reshape(self.world_size, B, k)
        Ig = out[:, ns:].contiguous().view(torch.int64).reshape(self.world_size, B, k)
        return Dg, Ig
