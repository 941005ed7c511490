"""FlatIPIndex — the object that stands where the reference keeps its FAISS index.

Reference seam: ``Stage1Retriever.faiss_index`` (reference
src/stage1_retriever.py:126), built by ``faiss.IndexFlatIP(d)`` /
``IndexIVFFlat`` (:256-283) and used through ``.add`` (:270,277,313),
``.search`` (:380), ``.ntotal``.  This class keeps that duck type
(``add``, ``search``, ``ntotal``, ``d``, ``reset``, ``reconstruct_n``) on top of
the C ABI of libtristage.so; the search is always exact (FAISS ``IndexFlatIP``
semantics — the reference's IVF variant above 1000 rows is an approximation of
this result).

Inputs may be numpy arrays (host pointers, FAISS style) or torch tensors on the
index's GPU (device pointers, no copies).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np

from . import _lib

_NP_DTYPES = {np.dtype(np.float32): _lib.TS_F32, np.dtype(np.float16): _lib.TS_F16}
_NAME_TO_DTYPE = {"f32": _lib.TS_F32, "fp32": _lib.TS_F32, "float32": _lib.TS_F32,
                  "f16": _lib.TS_F16, "fp16": _lib.TS_F16, "float16": _lib.TS_F16,
                  "bf16": _lib.TS_BF16, "bfloat16": _lib.TS_BF16}
_DTYPE_NAME = {_lib.TS_F32: "f32", _lib.TS_F16: "f16", _lib.TS_BF16: "bf16"}


def _torch():
    import torch
    return torch


def _is_tensor(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def _tensor_dtype(t) -> int:
    torch = _torch()
    if t.dtype == torch.float32:
        return _lib.TS_F32
    if t.dtype == torch.float16:
        return _lib.TS_F16
    if t.dtype == torch.bfloat16:
        return _lib.TS_BF16
    raise TypeError(f"unsupported tensor dtype {t.dtype}")


def _stream_ptr(device_index: int) -> int:
    torch = _torch()
    return int(torch.cuda.current_stream(device_index).cuda_stream)


class FlatIPIndex:
    """Exact inner-product index resident in MI355X HBM."""

    supports_out = True  # search(..., out=(D, I)) writes into caller tensors
    PENDING_PASSES = 240  # passes of <= 32 queries that may wait for finish() (the library tracks 256)
    MAX_KERNEL_K = 16384  # the select kernels hold 16384 keys in LDS; larger k: _search_large_k
    MAX_ASYNC_QUERIES = 128  # per asynchronous library call (4 passes of >= 32 queries)

    def __init__(self, d: int, dtype: str = "f32", device: int = 0):
        if dtype not in _NAME_TO_DTYPE:
            raise ValueError(f"unknown storage dtype {dtype!r}")
        self._lib = _lib.load()
        self.d = int(d)
        self.device = int(device)
        self.storage_dtype = _DTYPE_NAME[_NAME_TO_DTYPE[dtype]]
        self._h = ctypes.c_void_p()
        _lib.check(self._lib.ts_index_create(self.d, _NAME_TO_DTYPE[dtype],
                                             _lib.TS_METRIC_INNER_PRODUCT, self.device,
                                             ctypes.byref(self._h)))
        self.is_trained = True  # FAISS attribute; a flat index needs no training
        self._pending = {}      # ticket -> (q, k, D, I) of unfinished async searches
        self._pending_passes = 0
        self._auto_redone = []  # tickets repeated by an internal finish() the caller has not seen yet
        self.auto_finish = True  # False: the owner (ShardedFlatIPIndex) calls finish() itself, collectively
        self.classic_filter = False  # True: every search takes the five-launch filter path (A/B measurements)
        self.one_launch_filter = False  # True: the one-launch scan wherever it is valid (also pipelined / large corpora)

    # -- lifetime ---------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.ts_index_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- FAISS duck type --------------------------------------------------
    @property
    def ntotal(self) -> int:
        return int(self._lib.ts_index_ntotal(self._h))

    def train(self, x) -> None:  # IndexIVFFlat API used at reference :267; no-op here
        return None

    def reset(self) -> None:
        _lib.check(self._lib.ts_index_reset(self._h))

    def reserve(self, nrows: int) -> None:
        _lib.check(self._lib.ts_index_reserve(self._h, int(nrows)))

    def set_id_offset(self, offset: int) -> None:
        _lib.check(self._lib.ts_index_set_id_offset(self._h, int(offset)))
        self._id_offset = int(offset)

    def add(self, x, normalize: bool = False) -> None:
        """Append rows ``x`` [n, d] (numpy float32/float16 or a CUDA tensor)."""
        flags = _lib.TS_FLAG_NORMALIZE if normalize else 0
        if _is_tensor(x):
            if x.dim() != 2 or x.shape[1] != self.d:
                raise ValueError(f"expected [n, {self.d}] rows, got {tuple(x.shape)}")
            if not x.is_cuda:
                return self.add(x.detach().float().numpy(), normalize=normalize)
            if x.device.index != self.device:
                raise ValueError("rows live on a different GPU than the index")
            x = x.contiguous()
            _lib.check(self._lib.ts_index_add(self._h, ctypes.c_void_p(x.data_ptr()), x.shape[0],
                                              _tensor_dtype(x), flags,
                                              ctypes.c_void_p(_stream_ptr(self.device))))
            return None
        x = np.ascontiguousarray(x)
        if x.dtype not in _NP_DTYPES:
            x = x.astype(np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"expected [n, {self.d}] rows, got {x.shape}")
        _lib.check(self._lib.ts_index_add(self._h, x.ctypes.data_as(ctypes.c_void_p), x.shape[0],
                                          _NP_DTYPES[x.dtype], flags | _lib.TS_FLAG_HOST_PTR,
                                          None))
        return None

    def search(self, q, k: int, exact_dense: bool = False, async_: bool = False, out=None,
               inputs_ready: bool = False, classic: bool = False, one_launch: bool = False):
        """Top-``k`` inner products.  numpy in -> ``(D float32[B,k], I int64[B,k])``
        numpy out (FAISS convention, -1 padded); CUDA tensor in -> tensors out.

        ``async_=True`` (CUDA tensors only): the work is only enqueued on the current
        stream and the output tensors are returned at once; they are complete and
        verified after :meth:`finish`.  Batches issued this way run back to back on
        the GPU without a host round trip between them.
        ``out=(D, I)``: CUDA tensors [B,k] float32 / int64 to write into.
        ``inputs_ready=True`` (with ``async_``): the caller guarantees ``q`` is already
        complete in memory (not the result of work still pending on the stream); the
        library then pipelines this search's small kernels beside its neighbours' scans
        (TS_FLAG_PIPELINE)."""
        k = int(k)
        if k <= 0:
            raise ValueError("k must be positive")
        if k > self.MAX_KERNEL_K and self.ntotal > self.MAX_KERNEL_K:
            return self._search_large_k(q, k, out)
        if async_ and _is_tensor(q) and q.is_cuda and q.shape[0] > self.MAX_ASYNC_QUERIES:
            # the library takes at most 4 passes (256 queries) per asynchronous call: larger batches go in slices
            torch = _torch()
            B = q.shape[0]
            D, I = out if out is not None else (torch.empty((B, k), dtype=torch.float32, device=q.device),
                                                torch.empty((B, k), dtype=torch.int64, device=q.device))
            for s in range(0, B, self.MAX_ASYNC_QUERIES):
                e = min(B, s + self.MAX_ASYNC_QUERIES)
                self.search(q[s:e], k, exact_dense=exact_dense, async_=True, out=(D[s:e], I[s:e]),
                            inputs_ready=inputs_ready, classic=classic, one_launch=one_launch)
            return D, I
        flags = _lib.TS_FLAG_NO_FILTER if exact_dense else 0
        if classic or self.classic_filter:   # the five-launch filter path even where the one-launch scan is the default (A/B, tests)
            flags |= _lib.TS_FLAG_CLASSIC
        elif one_launch or self.one_launch_filter:   # the one-launch scan also where it is not the default
            flags |= _lib.TS_FLAG_ONE_LAUNCH
        if async_:
            if not (_is_tensor(q) and q.is_cuda):
                raise ValueError("async_ search needs a CUDA tensor")
            # the library tracks at most 64 unfinished passes (one per <= 64 queries)
            if self.pending_room(q.shape[0]) < 0:
                if not self.auto_finish:
                    raise RuntimeError("too many unfinished asynchronous searches: the owner of this index "
                                       "must call finish() (see pending_room())")
                self._auto_redone = self.finish()   # (finish() hands back what an earlier internal one repeated)
            flags |= _lib.TS_FLAG_ASYNC
            if inputs_ready:
                flags |= _lib.TS_FLAG_PIPELINE
        if _is_tensor(q) and q.is_cuda:
            torch = _torch()
            if q.dim() != 2 or q.shape[1] != self.d:
                raise ValueError(f"expected [B, {self.d}] queries, got {tuple(q.shape)}")
            q = q.contiguous()
            B = q.shape[0]
            if out is not None:
                D, I = out
                if (D.shape != (B, k) or I.shape != (B, k) or D.dtype != torch.float32 or
                        I.dtype != torch.int64 or not D.is_contiguous() or not I.is_contiguous()):
                    raise ValueError("out must be contiguous (float32[B,k], int64[B,k]) CUDA tensors")
            else:
                D = torch.empty((B, k), dtype=torch.float32, device=q.device)
                I = torch.empty((B, k), dtype=torch.int64, device=q.device)
            self._search_raw(q.data_ptr(), B, _tensor_dtype(q), k, D.data_ptr(), I.data_ptr(),
                             flags, _stream_ptr(self.device))
            if async_:
                self._pending[int(self._lib.ts_index_last_ticket(self._h))] = (q, k, D, I)
                self._pending_passes += (B + 31) // 32
            return D, I
        if _is_tensor(q):
            q = q.detach().float().numpy()
        q = np.ascontiguousarray(q)
        if q.dtype not in _NP_DTYPES:
            q = q.astype(np.float32)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError(f"expected [B, {self.d}] queries, got {q.shape}")
        B = q.shape[0]
        D = np.empty((B, k), dtype=np.float32)
        I = np.empty((B, k), dtype=np.int64)
        self._search_raw(q.ctypes.data, B, _NP_DTYPES[q.dtype], k, D.ctypes.data, I.ctypes.data,
                         flags | _lib.TS_FLAG_HOST_PTR, 0)
        return D, I

    def _search_large_k(self, q, k: int, out=None):
        """k > 16384 on a corpus of more than 16384 rows (the select kernels keep 16384 keys in LDS; FAISS itself takes
        any k on the CPU, reference src/stage1_retriever.py:380): every inner product from the HIP dense scan
        (ts_index_scores), then ONE stable descending device sort per slice of queries — equal scores keep ascending
        row order, the canonical tie rule — and the FAISS padding (-1 / -FLT_MAX) beyond ntotal.  Exact; not a fast path:
        it materialises 4 B x queries x rows."""
        torch = _torch()
        was_np = not (_is_tensor(q) and q.is_cuda)
        dev = torch.device("cuda", self.device)
        qt = torch.as_tensor(np.ascontiguousarray(q.detach().cpu().numpy() if _is_tensor(q) else q, dtype=np.float32)).to(dev) if was_np else q
        if qt.dim() != 2 or qt.shape[1] != self.d:
            raise ValueError(f"expected [B, {self.d}] queries, got {tuple(qt.shape)}")
        n, B = self.ntotal, qt.shape[0]
        if n == 0:
            raise ValueError("No documents indexed. Call add_documents() first.")
        kk = min(k, n)
        if out is not None and not was_np:
            D, I = out
        else:
            D = torch.empty((B, k), dtype=torch.float32, device=dev)
            I = torch.empty((B, k), dtype=torch.int64, device=dev)
        D[:, kk:] = -3.4028234663852886e38
        I[:, kk:] = -1
        step = max(1, int(2e9 // (4 * max(n, 1))))          # <= 2 GB of scores per slice
        off = int(self._id_offset_value())
        for s in range(0, B, step):
            sc = self.scores(qt[s: s + step])
            srt, idx = torch.sort(sc, dim=1, descending=True, stable=True)
            D[s: s + step, :kk] = srt[:, :kk]
            I[s: s + step, :kk] = idx[:, :kk] + off
        if was_np:
            return D.cpu().numpy(), I.cpu().numpy()
        return D, I

    def _id_offset_value(self) -> int:
        return getattr(self, "_id_offset", 0)

    def scores(self, q):
        """All inner products, in row order: float32 [B, ntotal] (a CUDA tensor for tensor
        input, numpy for numpy input).  No selection; the dense scan writes the matrix."""
        torch = _torch()
        was_np = not _is_tensor(q)
        dev = torch.device("cuda", self.device)
        qt = torch.as_tensor(np.ascontiguousarray(q, dtype=np.float32)) if was_np else q
        if qt.dim() != 2 or qt.shape[1] != self.d:
            raise ValueError(f"expected [B, {self.d}] queries, got {tuple(qt.shape)}")
        if qt.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            qt = qt.float()
        qt = qt.to(dev).contiguous()
        n = self.ntotal
        if n == 0:
            raise ValueError("No documents indexed. Call add_documents() first.")
        ld = (n + 31) // 32 * 32
        out = torch.empty((qt.shape[0], ld), dtype=torch.float32, device=dev)
        _lib.check(self._lib.ts_index_scores(self._h, ctypes.c_void_p(qt.data_ptr()), qt.shape[0],
                                             _tensor_dtype(qt), ctypes.c_void_p(out.data_ptr()), ld,
                                             ctypes.c_void_p(_stream_ptr(self.device)) if _stream_ptr(self.device) else None))
        out = out[:, :n]
        return out.cpu().numpy() if was_np else out

    def _search_raw(self, q_ptr: int, B: int, q_dtype: int, k: int, d_ptr: int, i_ptr: int,
                    flags: int, stream: int) -> None:
        code = self._lib.ts_index_search(self._h, ctypes.c_void_p(q_ptr), B, q_dtype, k,
                                         ctypes.c_void_p(d_ptr), ctypes.c_void_p(i_ptr), flags,
                                         ctypes.c_void_p(stream) if stream else None)
        if code == _lib.TS_ERR_EMPTY:
            # same exception type and text as reference src/stage1_retriever.py:370-371
            raise ValueError("No documents indexed. Call add_documents() first.")
        _lib.check(code)

    def search_in_stream_order(self, q, k: int):
        """search() for a caller that consumes (D, I) on the current stream only: when the dense path will be taken
        (small corpora, k > 2048 — exact by construction) the search is just ENQUEUED and the host does not wait; on the
        filter path it is the ordinary synchronous, verified search.  CUDA tensor queries only."""
        path = int(self._lib.ts_index_filter_path(self._h, int(k)))
        if path != 0 or not (_is_tensor(q) and q.is_cuda) or not self.auto_finish:
            return self.search(q, k)
        return self.search(q, k, async_=True)     # (its ticket is retired by the next finish(); nothing to verify)

    def pending_room(self, n_queries: int) -> int:
        """>= 0 while another asynchronous search of `n_queries` queries fits before a finish()."""
        return self.PENDING_PASSES - self._pending_passes - (int(n_queries) + 31) // 32

    def finish(self):
        """Complete every asynchronous search: one stream sync, then the (rare)
        batches whose fused filter could not prove exactness are repeated on the
        exact dense path, in place.  Returns the tickets that were repeated since the
        caller's previous finish() (including those an internal finish() — issued when
        too many searches were pending — had to repeat)."""
        failed = (ctypes.c_int64 * 256)()
        nf = ctypes.c_int32(0)
        _lib.check(self._lib.ts_index_finish(self._h, ctypes.c_void_p(_stream_ptr(self.device)) if
                                             _stream_ptr(self.device) else None, failed, 256, ctypes.byref(nf)))
        redone = []
        for i in range(nf.value):
            q, k, D, I = self._pending[int(failed[i])]
            self._search_raw(q.data_ptr(), q.shape[0], _tensor_dtype(q), k, D.data_ptr(), I.data_ptr(),
                             _lib.TS_FLAG_NO_FILTER, _stream_ptr(self.device))
            redone.append(int(failed[i]))
        self._pending.clear()
        self._pending_passes = 0
        if self._auto_redone:
            redone = self._auto_redone + redone
            self._auto_redone = []
        return redone

    def reconstruct_n(self, i0: int = 0, n: Optional[int] = None) -> np.ndarray:
        """Rows [i0, i0+n) as float32 (after storage rounding)."""
        if n is None:
            n = self.ntotal - i0
        out = np.empty((n, self.d), dtype=np.float32)
        if n:
            _lib.check(self._lib.ts_index_reconstruct(self._h, int(i0), int(n),
                                                      out.ctypes.data_as(ctypes.c_void_p),
                                                      _lib.TS_FLAG_HOST_PTR, None))
        return out

    PHASES = ("qprep", "sample_scan", "tau", "filter_scan", "select", "dense", "_6", "_7")

    def set_profiling(self, on: bool = True, every: int = 1) -> None:
        """HIP-event timing of the search phases; ``every=N`` times each N-th search only."""
        _lib.check(self._lib.ts_index_set_profiling(self._h, (max(int(every), 1) if on else 0)))

    def timings(self, reset: bool = True) -> dict:
        """{phase: (total_ms, count)} measured with HIP events on the search stream."""
        ms = (ctypes.c_double * 8)()
        cnt = (ctypes.c_int64 * 8)()
        _lib.check(self._lib.ts_index_get_timings(self._h, ms, cnt, 1 if reset else 0))
        return {self.PHASES[i]: (float(ms[i]), int(cnt[i])) for i in range(6)}

    def read_probe(self, reps: int = 5) -> dict:
        """Read-only pass over the index's own tiled corpus with the scan's access pattern (no LDS, no MFMA):
        the streaming ceiling of THIS box, measured with HIP events.  {"bytes", "ms_avg", "ms_best", "gbps_avg",
        "gbps_best"}."""
        ms_avg, ms_best, nbytes = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0)
        st = _stream_ptr(self.device)
        _lib.check(self._lib.ts_index_read_probe(self._h, int(reps), ctypes.byref(ms_avg), ctypes.byref(ms_best),
                                                 ctypes.byref(nbytes), ctypes.c_void_p(st) if st else None))
        return {"bytes": int(nbytes.value), "ms_avg": ms_avg.value, "ms_best": ms_best.value,
                "gbps_avg": nbytes.value / (ms_avg.value * 1e-3) / 1e9, "gbps_best": nbytes.value / (ms_best.value * 1e-3) / 1e9}

    def last_search_info(self) -> dict:
        arr = (ctypes.c_int64 * 4)()
        _lib.check(self._lib.ts_index_last_search_info(self._h, arr))
        return {"path": ("dense", "filter", "filter+dense-fallback")[arr[0] & 15],
                "one_launch": bool(arr[0] & 16),   # query image + thresholds + scan+filter in ONE kernel
                "max_candidates": int(arr[1]), "sample_rows": int(arr[2]),
                "sample_rank": int(arr[3])}


def merge_topk(scores, ids, k: Optional[int] = None):
    """Merge per-shard sorted lists ``scores``/``ids`` [R, B, k] (CUDA tensors)
    into the global top-k [B, k] with the canonical (score desc, id asc) order."""
    torch = _torch()
    lib = _lib.load()
    if scores.dim() != 3 or ids.shape != scores.shape:
        raise ValueError("expected scores/ids of shape [R, B, k]")
    R, B, kk = scores.shape
    if k is not None and k != kk:
        raise ValueError("k must equal the list length")
    scores = scores.contiguous().float()
    ids = ids.contiguous().to(torch.int64)
    out_s = torch.empty((B, kk), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((B, kk), dtype=torch.int64, device=scores.device)
    dev = scores.device.index
    _lib.check(lib.ts_merge_topk(ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(ids.data_ptr()),
                                 R, B, kk, ctypes.c_void_p(out_s.data_ptr()),
                                 ctypes.c_void_p(out_i.data_ptr()), dev,
                                 ctypes.c_void_p(_stream_ptr(dev))))
    return out_s, out_i


def packed_layout(B: int, k: int):
    """(offset of the int64 id block, bytes per rank) of one rank's packed partial result:
    float32 scores [B,k], padded to a multiple of 8 bytes, then int64 ids [B,k]."""
    ids_at = (4 * B * k + 7) & ~7
    return ids_at, ids_at + 8 * B * k


def merge_topk_packed(gathered, R: int, B: int, k: int):
    """Merge straight out of an all-gather buffer: `gathered` is a uint8 CUDA tensor of R
    blocks, each laid out as packed_layout(B, k) says."""
    torch = _torch()
    lib = _lib.load()
    ids_at, nbytes = packed_layout(B, k)
    if gathered.dtype != torch.uint8 or gathered.numel() != R * nbytes or gathered.data_ptr() % 8:
        raise ValueError("bad packed buffer")
    out_s = torch.empty((B, k), dtype=torch.float32, device=gathered.device)
    out_i = torch.empty((B, k), dtype=torch.int64, device=gathered.device)
    base = gathered.data_ptr()
    dev = gathered.device.index
    _lib.check(lib.ts_merge_topk_strided(ctypes.c_void_p(base), ctypes.c_void_p(base + ids_at), R, B, k,
                                         nbytes // 4, nbytes // 8, ctypes.c_void_p(out_s.data_ptr()),
                                         ctypes.c_void_p(out_i.data_ptr()), dev,
                                         ctypes.c_void_p(_stream_ptr(dev))))
    return out_s, out_i


def maxsim(q, docs, doc_offsets, mode: str = "maxsim"):
    """Stage-2 scores of every candidate for one query (CUDA tensors).

    q [Lq, H]; docs [sum(Ld), H] packed token embeddings; doc_offsets int32
    [n_docs+1].  mode 'maxsim' | 'colbert' (reference src/stage2_rescorer.py:167-201)."""
    torch = _torch()
    lib = _lib.load()
    if q.dtype != docs.dtype:
        docs = docs.to(q.dtype)
    q = q.contiguous()
    docs = docs.contiguous()
    doc_offsets = doc_offsets.to(device=q.device, dtype=torch.int32).contiguous()
    n_docs = doc_offsets.numel() - 1
    out = torch.empty((max(n_docs, 0),), dtype=torch.float32, device=q.device)
    if n_docs <= 0:
        return out
    dev = q.device.index
    _lib.check(lib.ts_maxsim(ctypes.c_void_p(q.data_ptr()), q.shape[0],
                             ctypes.c_void_p(docs.data_ptr()),
                             ctypes.c_void_p(doc_offsets.data_ptr()), n_docs, q.shape[1],
                             _tensor_dtype(q), 0 if mode == "maxsim" else 1,
                             ctypes.c_void_p(out.data_ptr()), dev,
                             ctypes.c_void_p(_stream_ptr(dev))))
    return out


def maxsim_indexed(q, store, starts, lens, mode: str = "maxsim"):
    """Stage-2 scores for candidates that live in a resident token store: document i
    is rows [starts[i], starts[i]+lens[i]) of ``store`` [rows, H] (CUDA tensors; starts
    int64, lens int32).  The kernel reads the store in place."""
    torch = _torch()
    lib = _lib.load()
    if q.dtype != store.dtype:
        q = q.to(store.dtype)
    q = q.contiguous()
    if not store.is_contiguous():
        raise ValueError("token store must be contiguous")
    starts = starts.to(device=q.device, dtype=torch.int64).contiguous()
    lens = lens.to(device=q.device, dtype=torch.int32).contiguous()
    n = starts.numel()
    out = torch.empty((n,), dtype=torch.float32, device=q.device)
    if n == 0:
        return out
    dev = q.device.index
    _lib.check(lib.ts_maxsim_indexed(ctypes.c_void_p(q.data_ptr()), q.shape[0],
                                     ctypes.c_void_p(store.data_ptr()), ctypes.c_void_p(starts.data_ptr()),
                                     ctypes.c_void_p(lens.data_ptr()), n, q.shape[1], _tensor_dtype(q),
                                     0 if mode == "maxsim" else 1, ctypes.c_void_p(out.data_ptr()), dev,
                                     ctypes.c_void_p(_stream_ptr(dev))))
    return out


def maxsim_indexed_batch(q_packed, q_offsets, store, starts, lens, cand_offsets, mode: str = "maxsim"):
    """Stage-2 scores for SEVERAL queries in one launch: query j has tokens
    q_packed[q_offsets[j]:q_offsets[j+1]] and candidates starts/lens[cand_offsets[j]:cand_offsets[j+1]]
    (rows of the resident token ``store``).  q_offsets / cand_offsets: host int sequences of
    nq+1 values starting at 0.  Returns float32 [cand_offsets[-1]] on the GPU."""
    import numpy as np
    torch = _torch()
    lib = _lib.load()
    if q_packed.dtype != store.dtype:
        q_packed = q_packed.to(store.dtype)
    q_packed = q_packed.contiguous()
    if not store.is_contiguous():
        raise ValueError("token store must be contiguous")
    qo = np.ascontiguousarray(np.asarray(q_offsets, dtype=np.int32))
    co = np.ascontiguousarray(np.asarray(cand_offsets, dtype=np.int32))
    if qo.ndim != 1 or qo.shape != co.shape or qo.size < 1 or qo[0] != 0 or co[0] != 0:
        raise ValueError("q_offsets / cand_offsets must be 1-D, of equal length nq+1, starting at 0")
    nq = qo.size - 1
    if int(qo[-1]) != q_packed.shape[0]:
        raise ValueError("q_offsets[-1] must equal the number of packed query tokens")
    starts = starts.to(device=q_packed.device, dtype=torch.int64).contiguous()
    lens = lens.to(device=q_packed.device, dtype=torch.int32).contiguous()
    n = int(co[-1])
    if starts.numel() != n or lens.numel() != n:
        raise ValueError("starts / lens must hold cand_offsets[-1] entries")
    out = torch.empty((n,), dtype=torch.float32, device=q_packed.device)
    if n == 0 or nq == 0:
        return out
    dev = q_packed.device.index
    _lib.check(lib.ts_maxsim_indexed_batch(ctypes.c_void_p(q_packed.data_ptr()), qo.ctypes.data_as(ctypes.c_void_p), nq,
                                           ctypes.c_void_p(store.data_ptr()), ctypes.c_void_p(starts.data_ptr()),
                                           ctypes.c_void_p(lens.data_ptr()), co.ctypes.data_as(ctypes.c_void_p),
                                           q_packed.shape[1], _tensor_dtype(q_packed), 0 if mode == "maxsim" else 1,
                                           ctypes.c_void_p(out.data_ptr()), dev, ctypes.c_void_p(_stream_ptr(dev))))
    return out


def add_layernorm(x, residual, gamma, beta, eps: float, lp_dtype=None, want_f32: bool = True, prenorm: bool = False):
    """``LayerNorm(x + residual) * gamma + beta`` over the last dimension in ONE pass on the GPU (ts_add_layernorm):
    x [..., H] (fp32 / fp16 / bf16), residual fp32 of the same shape or None, gamma fp32 [H], beta fp32 [H] or None.
    Returns (y as fp32 or None, y in ``lp_dtype`` or None) — the next residual and the next GEMM's input.
    ``prenorm=True`` (ts_add_prenorm, pre-LN models): the fp32 result is ``x + residual`` — the residual stream — and the
    ``lp_dtype`` one its LayerNorm."""
    torch = _torch()
    lib = _lib.load()
    H = int(x.shape[-1])
    x = x.contiguous()
    rows = x.numel() // H
    if residual is not None:
        residual = residual.contiguous()
        if residual.dtype != torch.float32 or residual.shape != x.shape:
            raise ValueError("residual must be float32 with x's shape")
    gamma = gamma.detach().contiguous()
    beta = beta.detach().contiguous() if beta is not None else None
    for t in (gamma, beta):
        if t is not None and (t.dtype != torch.float32 or t.numel() != H or t.device != x.device):
            raise ValueError("gamma / beta must be float32 [H] on x's device")
    out32 = torch.empty(x.shape, dtype=torch.float32, device=x.device) if want_f32 else None
    outlp = torch.empty(x.shape, dtype=lp_dtype, device=x.device) if lp_dtype is not None else None
    dev = x.device.index
    ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    fn = lib.ts_add_prenorm if prenorm else lib.ts_add_layernorm
    _lib.check(fn(ptr(x), _tensor_dtype(x), ptr(residual), ptr(gamma), ptr(beta), float(eps), rows, H, ptr(out32), ptr(outlp),
                  _tensor_dtype(outlp) if outlp is not None else _lib.TS_BF16, dev, ctypes.c_void_p(_stream_ptr(dev))))
    return out32, outlp


def attention_varlen(qkv, lens, heads: int, out=None, scale: Optional[float] = None, window: int = 0, rope=None,
                     offs=None, max_len: Optional[int] = None):
    """Self-attention of a right-padded batch on the GPU (ts_attention_varlen): ``qkv`` [B, L, 3*heads*dh] (fp16 / bf16,
    the fused projection's output, read in place), ``lens`` int32 [B] on the device.  Returns [B, L, heads*dh]; rows at
    padded positions are zeros (``out`` given: left as they are).  ``window`` > 0: keys within that distance only.
    ``rope`` = (cos, sin) float32 [L, dh]: rotary embedding applied to q and k on the fly (qkv is not modified).
    PACKED batches: ``qkv`` [T, 3*heads*dh] with ``offs`` int32 [B] (first token of each sequence, on the device) and
    ``max_len`` >= every length; returns [T, heads*dh]."""
    torch = _torch()
    lib = _lib.load()
    packed = offs is not None
    if packed:
        if qkv.dim() != 2 or max_len is None:
            raise ValueError("a packed batch is qkv [T, 3 * heads * head_dim] with offs and max_len")
        T, W = (int(v) for v in qkv.shape)
        B, L = int(lens.numel()), int(max_len)
        if offs.dtype != torch.int32 or offs.numel() != B or offs.device != qkv.device:
            raise ValueError("offs must be int32 [B] on qkv's device")
    else:
        B, L, W = (int(v) for v in qkv.shape)
    if W % (3 * heads):
        raise ValueError("last dimension must be 3 * heads * head_dim")
    dh = W // (3 * heads)
    if not qkv.is_contiguous() or lens.dtype != torch.int32 or lens.numel() != B or lens.device != qkv.device:
        raise ValueError("qkv must be contiguous and lens int32 [B] on the same device")
    if out is None:
        out = torch.zeros(((T,) if packed else (B, L)) + (heads * dh,), dtype=qkv.dtype, device=qkv.device)
    dev = qkv.device.index
    cos = sin = None
    if rope is not None:
        cos, sin = rope
        for t in (cos, sin):
            if (t.dtype != torch.float32 or t.dim() != 2 or t.shape[0] < L or t.shape[1] != dh or not t.is_contiguous()
                    or t.device != qkv.device):
                raise ValueError("rope tables must be contiguous float32 [>= L, head_dim] on qkv's device")
    ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    _lib.check(lib.ts_attention_varlen(ptr(qkv), ptr(lens), B, L, heads, dh, _tensor_dtype(qkv),
                                       float(scale if scale is not None else dh ** -0.5), int(window), ptr(cos), ptr(sin),
                                       ptr(offs), ptr(out), dev, ctypes.c_void_p(_stream_ptr(dev))))
    return out


def rope_inplace(qkv, cos, sin, heads: int):
    """Rotary position embedding of the q and k thirds of ``qkv`` [B, L, 3*heads*dh] (fp16 / bf16) in place
    (ts_rope_inplace); ``cos`` / ``sin`` float32 [L, dh]."""
    torch = _torch()
    lib = _lib.load()
    B, L, W = (int(v) for v in qkv.shape)
    dh = W // (3 * heads)
    if W != 3 * heads * dh or not qkv.is_contiguous():
        raise ValueError("qkv must be contiguous [B, L, 3 * heads * head_dim]")
    for t in (cos, sin):
        if t.dtype != torch.float32 or tuple(t.shape) != (L, dh) or not t.is_contiguous() or t.device != qkv.device:
            raise ValueError("cos / sin must be contiguous float32 [L, head_dim] on qkv's device")
    dev = qkv.device.index
    _lib.check(lib.ts_rope_inplace(ctypes.c_void_p(qkv.data_ptr()), _tensor_dtype(qkv), ctypes.c_void_p(cos.data_ptr()),
                                   ctypes.c_void_p(sin.data_ptr()), B, L, heads, dh, dev, ctypes.c_void_p(_stream_ptr(dev))))
    return qkv


def geglu(u):
    """``gelu(u[..., :I]) * u[..., I:]`` for u [..., 2 I] (fp16 / bf16) in one pass on the GPU (ts_geglu)."""
    torch = _torch()
    lib = _lib.load()
    if not u.is_contiguous() or u.shape[-1] % 2:
        raise ValueError("u must be contiguous with an even last dimension")
    I = int(u.shape[-1]) // 2
    out = torch.empty(tuple(u.shape[:-1]) + (I,), dtype=u.dtype, device=u.device)
    dev = u.device.index
    _lib.check(lib.ts_geglu(ctypes.c_void_p(u.data_ptr()), _tensor_dtype(u), u.numel() // (2 * I), I,
                            ctypes.c_void_p(out.data_ptr()), dev, ctypes.c_void_p(_stream_ptr(dev))))
    return out


def embed_layernorm(ids, pos_ids, type_ids, word, pos, typ, gamma, beta, eps: float, lp_dtype=None, want_f32: bool = True):
    """The embedding layer of a BERT-family model in one pass on the GPU (ts_embed_layernorm):
    ``LayerNorm((word[ids] + typ[type_ids]) + pos[pos_ids])``; ids / pos_ids / type_ids int64 of one shape (type_ids None =
    type 0), tables fp32 [*, H].  Indices must be in range (the caller's tokenizer guarantees it).  Returns (fp32 or None,
    ``lp_dtype`` copy or None) of shape ids.shape + (H,)."""
    torch = _torch()
    lib = _lib.load()
    H = int(word.shape[-1])
    tabs = [t.detach() if t is not None else None for t in (word, pos, typ, gamma, beta)]
    if any(t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.device != ids.device) for t in tabs):
        raise ValueError("embedding tables and LayerNorm parameters must be contiguous float32 on the ids' device")
    if any(t.dim() != 2 or t.shape[-1] != H for t in tabs[:3]) or tabs[3].numel() != H or (tabs[4] is not None and tabs[4].numel() != H):
        raise ValueError("embedding tables must be [*, H] and gamma / beta [H]")
    idx = [ids.contiguous(), pos_ids.contiguous(), type_ids.contiguous() if type_ids is not None else None]
    if any(t is not None and (t.dtype != torch.int64 or t.shape != ids.shape) for t in idx):
        raise ValueError("ids, pos_ids and type_ids must be int64 of one shape")
    shape = tuple(ids.shape) + (H,)
    out32 = torch.empty(shape, dtype=torch.float32, device=ids.device) if want_f32 else None
    outlp = torch.empty(shape, dtype=lp_dtype, device=ids.device) if lp_dtype is not None else None
    dev = ids.device.index
    ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    _lib.check(lib.ts_embed_layernorm(ptr(idx[0]), ptr(idx[1]), ptr(idx[2]), ptr(tabs[0]), ptr(tabs[1]), ptr(tabs[2]),
                                      ptr(tabs[3]), ptr(tabs[4]), float(eps), ids.numel(), H, ptr(out32), ptr(outlp),
                                      _tensor_dtype(outlp) if outlp is not None else _lib.TS_BF16, dev,
                                      ctypes.c_void_p(_stream_ptr(dev))))
    return out32, outlp


def mlp_add_layernorm(up: "TiledLinear", down: "TiledLinear", x, residual, gamma, beta, eps: float, want_f32: bool = True):
    """``LayerNorm(down(gelu(up(x))) + residual) * gamma + beta`` — BertIntermediate + BertOutput — in ONE kernel
    (ts_mlp_add_layernorm): the intermediate never goes to HBM.  ``up`` [I, H] / ``down`` [H, I] are TiledLinear weights
    (``mlp_usable(up, down)``), x [..., H] of their dtype, residual fp32 [..., H] or None, gamma / beta fp32 [H].  Returns
    (y fp32 or None, y in the weights' dtype) — bit-identical to ``down.add_layernorm(up(x, gelu=True), residual, ...)``."""
    torch = _torch()
    if not mlp_usable(up, down):
        raise ValueError("mlp_add_layernorm takes an up weight [I, 384] and a down weight [384, I] with I % 384 == 0 (mlp_usable)")
    H = up.K
    if x.dtype != up.dtype or x.shape[-1] != H or not x.is_contiguous() or x.device != up.device:
        raise ValueError("x must be a contiguous [..., H] tensor of the weights' dtype on their device")
    if residual is not None:
        residual = residual.contiguous()
        if residual.dtype != torch.float32 or residual.shape != x.shape or residual.device != x.device:
            raise ValueError("residual must be float32 with x's shape on x's device")
    gamma = gamma.detach().contiguous()
    beta = beta.detach().contiguous() if beta is not None else None
    for t in (gamma, beta):
        if t is not None and (t.dtype != torch.float32 or t.numel() != H or t.device != x.device):
            raise ValueError("gamma / beta must be float32 [H] on x's device")
    out32 = torch.empty(x.shape, dtype=torch.float32, device=x.device) if want_f32 else None
    outlp = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    dev = x.device.index
    ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    _lib.check(_lib.load().ts_mlp_add_layernorm(ptr(up.tiled), ptr(up.bias), ptr(down.tiled), ptr(down.bias), ptr(x), ptr(residual),
                                                ptr(gamma), ptr(beta), float(eps), _tensor_dtype(x), x.numel() // H, H, up.N,
                                                ptr(out32), ptr(outlp), dev, ctypes.c_void_p(_stream_ptr(dev))))
    return out32, outlp


def mlp_usable(up, down) -> bool:
    """Whether ``mlp_add_layernorm`` takes this pair of TiledLinear weights: hidden size 384 (a workgroup owns whole rows),
    intermediate size a multiple of 384, same dtype and device."""
    return (up is not None and down is not None and up.K == 384 and down.N == 384 and up.N == down.K and up.N % 384 == 0
            and up.dtype == down.dtype and up.device == down.device)


class TiledLinear:
    """A torch.nn.Linear-style weight [N, K] (fp16 / bf16, on the GPU) re-tiled once for ts_linear_act: ``y = act(x W^T + b)``
    with the weight streamed from L2 and x's rows in LDS — for reduction dimensions up to 384 (MiniLM-class encoders), where
    the library GEMM re-reads its operands many times and this kernel is 1.2-1.5x faster; at K = 768 the library's
    stream-K GEMMs (1.16 PFLOP/s) win and the encoders keep them.  ``TiledLinear.usable(N, K)`` says
    whether a shape qualifies."""

    @staticmethod
    def usable(N: int, K: int) -> bool:
        return N % 32 == 0 and K % 128 == 0 and K <= 384

    @staticmethod
    def usable_with_layernorm(N: int, K: int) -> bool:
        """Shapes ``add_layernorm`` takes: a workgroup owns whole rows of the output (N <= 384), the reduction is walked
        in chunks of 384 (the attention-output and feed-forward down projections of MiniLM-class encoders)."""
        return N % 32 == 0 and N <= 384 and K % 384 == 0

    def __init__(self, weight, bias=None, with_layernorm: bool = False):
        torch = _torch()
        lib = _lib.load()
        w = weight.detach().contiguous()
        self.N, self.K = int(w.shape[0]), int(w.shape[1])
        ok = self.usable_with_layernorm(self.N, self.K) if with_layernorm else self.usable(self.N, self.K)
        if not w.is_cuda or w.dtype not in (torch.float16, torch.bfloat16) or not ok:
            raise ValueError("TiledLinear takes a 16-bit CUDA weight [N, K] with N % 32 == 0, K % 128 == 0, K <= 384 "
                             "(with_layernorm: N <= 384, K % 384 == 0)")
        self.dtype, self.device = w.dtype, w.device
        self.bias = bias.detach().to(w.dtype).contiguous() if bias is not None else None
        self.tiled = torch.empty_like(w)
        dev = w.device.index
        _lib.check(lib.ts_linear_tile_weight(ctypes.c_void_p(w.data_ptr()), _tensor_dtype(w), self.N, self.K,
                                             ctypes.c_void_p(self.tiled.data_ptr()), dev, ctypes.c_void_p(_stream_ptr(dev))))

    def __call__(self, x, gelu: bool = False):
        torch = _torch()
        if x.dtype != self.dtype or x.shape[-1] != self.K or not x.is_contiguous() or x.device != self.device:
            raise ValueError("x must be a contiguous [..., K] tensor of the weight's dtype on its device")
        out = torch.empty(tuple(x.shape[:-1]) + (self.N,), dtype=x.dtype, device=x.device)
        dev = x.device.index
        _lib.check(_lib.load().ts_linear_act(ctypes.c_void_p(self.tiled.data_ptr()), ctypes.c_void_p(x.data_ptr()),
                                             ctypes.c_void_p(self.bias.data_ptr()) if self.bias is not None else None,
                                             _tensor_dtype(x), x.numel() // self.K, self.N, self.K, 1 if gelu else 0,
                                             ctypes.c_void_p(out.data_ptr()), dev, ctypes.c_void_p(_stream_ptr(dev))))
        return out

    def add_layernorm(self, x, residual, gamma, beta, eps: float, want_f32: bool = True, gelu_input: bool = False):
        """``LayerNorm(linear(x) + residual) * gamma + beta`` in ONE kernel (ts_linear_add_layernorm; BertSelfOutput /
        BertOutput): x [..., K] of the weight's dtype, residual fp32 [..., N] or None, gamma / beta fp32 [N].  Returns
        (y fp32 or None, y in the weight's dtype) — bit-identical to ``add_layernorm(self(x), residual, ...)``.
        ``gelu_input=True``: ``linear(gelu(x))`` — x is the up projection's output BEFORE its activation, the erf GELU (and
        its rounding to the 16-bit type) is applied while the rows are staged: bit-identical to passing
        ``F.gelu``-of-x as produced by ``TiledLinear.__call__(..., gelu=True)``."""
        torch = _torch()
        if not self.usable_with_layernorm(self.N, self.K):
            raise ValueError("this weight's shape has no fused LayerNorm kernel (TiledLinear.usable_with_layernorm)")
        if x.dtype != self.dtype or x.shape[-1] != self.K or not x.is_contiguous() or x.device != self.device:
            raise ValueError("x must be a contiguous [..., K] tensor of the weight's dtype on its device")
        shape = tuple(x.shape[:-1]) + (self.N,)
        if residual is not None:
            residual = residual.contiguous()
            if residual.dtype != torch.float32 or tuple(residual.shape) != shape or residual.device != x.device:
                raise ValueError("residual must be float32 [..., N] on x's device")
        gamma = gamma.detach().contiguous()
        beta = beta.detach().contiguous() if beta is not None else None
        for t in (gamma, beta):
            if t is not None and (t.dtype != torch.float32 or t.numel() != self.N or t.device != x.device):
                raise ValueError("gamma / beta must be float32 [N] on x's device")
        out32 = torch.empty(shape, dtype=torch.float32, device=x.device) if want_f32 else None
        outlp = torch.empty(shape, dtype=x.dtype, device=x.device)
        dev = x.device.index
        ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(_lib.load().ts_linear_add_layernorm(ptr(self.tiled), ptr(x), ptr(self.bias), ptr(residual), ptr(gamma), ptr(beta),
                                                       float(eps), _tensor_dtype(x), x.numel() // self.K, self.N, self.K,
                                                       1 if gelu_input else 0, ptr(out32), ptr(outlp), dev,
                                                       ctypes.c_void_p(_stream_ptr(dev))))
        return out32, outlp
