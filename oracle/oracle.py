"""CPU oracle for the TriStage-RAG retrieval hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Nothing in tristage-rag_amd/ does; the product path is the HIP
library and fails loudly without it.

Every function cites the reference file:line it restates (reference =
/root/reference, NoliNobdon/TriStage-RAG).  Pure functions are pinned against
outputs of the reference's own code (tests/golden/reference_kat.json, produced by
tests/golden/make_golden.py in the build container).  The FAISS / sentence-
transformers / transformers calls are third-party code absent from the
reference tree: "parity unpinned" there, contract stated in oracle.c.
"""
from __future__ import annotations

import ctypes
import math
import os
import re
import subprocess
from collections import defaultdict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build() -> str:
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.PIPE,
                   stderr=subprocess.STDOUT)
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oracle_ip_topk.restype = ctypes.c_int
        L.oracle_ip_topk.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p,
                                     ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_int32, ctypes.c_int32, ctypes.c_int64]
        L.oracle_scores_f64.restype = None
        L.oracle_scores_f64.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                        ctypes.c_void_p, ctypes.c_void_p]
        L.oracle_normalize.restype = None
        L.oracle_normalize.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p]
        L.oracle_maxsim.restype = None
        L.oracle_maxsim.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
        L.oracle_minmax.restype = None
        L.oracle_minmax.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        L.oracle_max_threads.restype = ctypes.c_int
        _lib = L
    return _lib


# --------------------------------------------------------------------- dtypes
def quantize(x: np.ndarray, dtype: str) -> np.ndarray:
    """Round float32 values to the storage dtype and back (round-to-nearest-even),
    i.e. the values the GPU index actually holds."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if dtype in ("f32", "fp32", "float32"):
        return x
    if dtype in ("f16", "fp16", "float16"):
        return x.astype(np.float16).astype(np.float32)
    if dtype in ("bf16", "bfloat16"):
        u = x.view(np.uint32).astype(np.uint64)
        rounded = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        out = rounded.astype(np.uint32).view(np.float32)
        nan = np.isnan(x)
        if nan.any():
            out = out.copy()
            out[nan] = np.nan
        return out.reshape(x.shape)
    raise ValueError(dtype)


# --------------------------------------------------------------------- stage 1
def normalize_embeddings(e: np.ndarray) -> np.ndarray:
    """reference src/stage1_retriever.py:285-288"""
    e = np.asarray(e)
    norms = np.linalg.norm(e, axis=1, keepdims=True)
    return e / (norms + 1e-8)


def ip_topk(corpus: np.ndarray, queries: np.ndarray, k: int, f64: bool = True,
            nthreads: int = 0, id_offset: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Exact inner-product top-k (IndexFlatIP.search as used at reference
    src/stage1_retriever.py:380), canonical order (score desc, id asc), padded
    with (-FLT_MAX, -1).  Inputs are float32 (already storage-quantised)."""
    corpus = np.ascontiguousarray(corpus, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    n, d = corpus.shape
    nq = queries.shape[0]
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    lib().oracle_ip_topk(corpus.ctypes.data, n, d, queries.ctypes.data, nq, k, D.ctypes.data,
                         I.ctypes.data, 1 if f64 else 0, nthreads, id_offset)
    return D, I


def scores_f64(corpus: np.ndarray, query: np.ndarray) -> np.ndarray:
    corpus = np.ascontiguousarray(corpus, dtype=np.float32)
    query = np.ascontiguousarray(query, dtype=np.float32)
    out = np.empty((corpus.shape[0],), dtype=np.float64)
    lib().oracle_scores_f64(corpus.ctypes.data, corpus.shape[0], corpus.shape[1],
                            query.ctypes.data, out.ctypes.data)
    return out


def ip_topk_blas(corpus: np.ndarray, queries: np.ndarray, k: int,
                 chunk: int = 262144) -> Tuple[np.ndarray, np.ndarray]:
    """The CPU baseline that is TIMED: blocked fp32 SGEMM + partial sort, which is
    how faiss-cpu's IndexFlatIP.search computes a query batch (BLAS for nq>=20).
    Same results as ip_topk up to fp32 accumulation order."""
    n = corpus.shape[0]
    nq = queries.shape[0]
    kk = min(k, n)
    best_s = np.full((nq, 0), -np.inf, dtype=np.float32)
    best_i = np.zeros((nq, 0), dtype=np.int64)
    for r0 in range(0, n, chunk):
        s = queries @ corpus[r0:r0 + chunk].T  # [nq, c]
        c = s.shape[1]
        if c > kk:
            part = np.argpartition(-s, kk - 1, axis=1)[:, :kk]
            ps = np.take_along_axis(s, part, axis=1)
        else:
            part = np.broadcast_to(np.arange(c), (nq, c))
            ps = s
        best_s = np.concatenate([best_s, ps], axis=1)
        best_i = np.concatenate([best_i, part + r0], axis=1)
        if best_s.shape[1] > kk:
            sel = np.argpartition(-best_s, kk - 1, axis=1)[:, :kk]
            best_s = np.take_along_axis(best_s, sel, axis=1)
            best_i = np.take_along_axis(best_i, sel, axis=1)
    order = np.lexsort((best_i, -best_s), axis=1)
    D = np.take_along_axis(best_s, order, axis=1)
    I = np.take_along_axis(best_i, order, axis=1)
    if kk < k:
        D = np.concatenate([D, np.full((nq, k - kk), -np.finfo(np.float32).max, np.float32)], axis=1)
        I = np.concatenate([I, np.full((nq, k - kk), -1, np.int64)], axis=1)
    return D.astype(np.float32), I


def merge_topk(scores: np.ndarray, ids: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Global top-k of per-shard lists [R, B, k']: canonical order,
    id < 0 entries are padding."""
    R, B, kk = scores.shape
    s = np.transpose(scores, (1, 0, 2)).reshape(B, R * kk).astype(np.float64)
    i = np.transpose(ids, (1, 0, 2)).reshape(B, R * kk).astype(np.int64)
    s = np.where(i < 0, -np.inf, s)
    big = np.where(i < 0, np.iinfo(np.int64).max, i)
    order = np.lexsort((big, -s), axis=1)[:, :k]
    D = np.take_along_axis(s, order, axis=1)
    I = np.take_along_axis(i, order, axis=1)
    D = np.where(I < 0, -np.finfo(np.float32).max, D)
    return D.astype(np.float32), I


# BM25 — reference src/stage1_retriever.py:35-112
class BM25Index:
    def __init__(self, k1: float = 1.2, b: float = 0.75):
        self.k1, self.b = k1, b
        self.doc_freqs: List[Dict[str, int]] = []
        self.idf: Dict[str, float] = {}
        self.doc_lens: List[int] = []
        self.avg_doc_len = 0.0
        self.corpus_size = 0
        self.vocabulary = set()
        self.documents: List[str] = []

    @staticmethod
    def tokenize(text: str) -> List[str]:
        return re.sub(r"[^a-z0-9\s]", " ", text.lower()).split()  # :49-54

    def fit(self, documents: Sequence[str]) -> None:  # :56-81 (appends doc_freqs: reference quirk kept)
        self.documents = list(documents)
        self.corpus_size = len(documents)
        for doc in documents:
            toks = self.tokenize(doc)
            self.vocabulary.update(toks)
            tf: Dict[str, int] = defaultdict(int)
            for t in toks:
                tf[t] += 1
            self.doc_freqs.append(tf)
            self.doc_lens.append(len(toks))
        self.avg_doc_len = sum(self.doc_lens) / self.corpus_size if self.corpus_size > 0 else 0
        for tok in self.vocabulary:
            df = sum(1 for f in self.doc_freqs if tok in f)
            self.idf[tok] = math.log((self.corpus_size - df + 0.5) / (df + 0.5) + 1.0)

    def score(self, query: str, doc_idx: int) -> float:  # :83-101
        if doc_idx >= len(self.doc_freqs):
            return 0.0
        f, dl, s = self.doc_freqs[doc_idx], self.doc_lens[doc_idx], 0.0
        for tok in self.tokenize(query):
            if tok in f and tok in self.idf:
                tf = f[tok]
                s += self.idf[tok] * (tf * (self.k1 + 1)) / (
                    tf + self.k1 * (1 - self.b + self.b * dl / self.avg_doc_len))
        return s

    def search(self, query: str, top_k: int = 10) -> List[Tuple[int, float]]:  # :103-112
        scores = [(i, self.score(query, i)) for i in range(len(self.documents))]
        scores.sort(key=lambda x: x[1], reverse=True)  # stable: ties keep ascending doc order
        return scores[:top_k]


def reciprocal_rank_fusion(dense, bm25, rrf_k: int = 60):
    """reference src/stage1_retriever.py:326-343"""
    scores = defaultdict(float)
    for rank, (idx, _) in enumerate(dense):
        scores[idx] += 1.0 / (rrf_k + rank + 1)
    for rank, (idx, _) in enumerate(bm25):
        scores[idx] += 1.0 / (rrf_k + rank + 1)
    fused = list(scores.items())
    fused.sort(key=lambda x: x[1], reverse=True)
    return fused


def weighted_fusion(dense, bm25, dense_weight: float = 0.7, bm25_weight: float = 0.3):
    """reference src/stage1_retriever.py:345-366"""
    scores = defaultdict(float)
    if dense:
        mx = max(s for _, s in dense)
        for idx, s in dense:
            scores[idx] += dense_weight * (s / mx)
    if bm25:
        mx = max(s for _, s in bm25)
        for idx, s in bm25:
            scores[idx] += bm25_weight * (s / mx)
    fused = list(scores.items())
    fused.sort(key=lambda x: x[1], reverse=True)
    return fused


# --------------------------------------------------------------------- stage 2
def maxsim_scores(q: np.ndarray, docs: Sequence[np.ndarray], mode: str = "maxsim") -> np.ndarray:
    """reference src/stage2_rescorer.py:167-201 for every candidate; q [Lq,H],
    docs: list of [Ld_i,H].  float64 accumulation."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    H = q.shape[1]
    off = np.zeros(len(docs) + 1, dtype=np.int32)
    for i, d in enumerate(docs):
        off[i + 1] = off[i] + d.shape[0]
    packed = (np.ascontiguousarray(np.concatenate([np.asarray(d, np.float32).reshape(-1, H) for d in docs], 0))
              if len(docs) and off[-1] > 0 else np.zeros((1, H), np.float32))
    out = np.zeros(len(docs), dtype=np.float32)
    lib().oracle_maxsim(q.ctypes.data, q.shape[0], packed.ctypes.data, off.ctypes.data, len(docs), H,
                        0 if mode == "maxsim" else 1, out.ctypes.data)
    return out


def maxsim_numpy(q: np.ndarray, d: np.ndarray, mode: str = "maxsim") -> float:
    """Independent numpy restatement of the same two functions (cross-check of the C one)."""
    q = q.astype(np.float64)
    d = d.astype(np.float64)
    qn = q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-12)
    dn = d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-12)
    m = (qn @ dn.T).max(axis=1)
    if mode == "maxsim":
        return float(m.mean())
    w = np.exp(m - m.max())
    w /= w.sum()
    return float((w * m).sum())


def stable_sort_desc(items: List[dict], key: str) -> List[dict]:
    """list.sort(key=..., reverse=True) as at src/stage2_rescorer.py:294 and
    src/stage3_reranker.py:257: descending, equal keys keep their input order."""
    return sorted(items, key=lambda x: x[key], reverse=True)


# --------------------------------------------------------------------- stage 3
def minmax_normalize(scores: Sequence[float]) -> List[float]:
    """reference src/stage3_reranker.py:212-228"""
    if not len(scores):
        return list(scores)
    a = np.array(scores)
    mn, mx = a.min(), a.max()
    return ((a - mn) / (mx - mn) if mx > mn else np.zeros_like(a)).tolist()


def adaptive_batch_size(texts: Sequence[str], batch_size: int) -> int:
    """reference src/stage3_reranker.py:328-344"""
    if not texts:
        return batch_size
    avg = sum(len(t.split()) for t in texts) / len(texts)
    if avg > 200:
        return max(4, batch_size // 4)
    if avg > 100:
        return max(8, batch_size // 2)
    if avg > 50:
        return max(16, batch_size // 1)
    return batch_size


def cosine_similarity(query_embedding: np.ndarray, document_embeddings: np.ndarray) -> np.ndarray:
    """reference src/embedding_service.py:228-237"""
    qn = query_embedding / np.linalg.norm(query_embedding)
    dn = document_embeddings / np.linalg.norm(document_embeddings, axis=1, keepdims=True)
    return np.dot(dn, qn).reshape(1, -1)


# --------------------------------------------------------------------- metric
def ndcg_at_k(qrels: Dict[str, Dict[str, int]], results: Dict[str, Dict[str, float]], k: int = 10) -> float:
    """nDCG@k with trec_eval semantics (gain = rel, discount 1/log2(rank+1), ties
    broken by doc id descending), the measure MTEB reports for retrieval tasks
    (reference benchmark/run_mteb_evaluation.py:343-392 reads ndcg_at_10)."""
    vals = []
    for qid, rels in qrels.items():
        run = results.get(qid, {})
        ranked = sorted(run.items(), key=lambda kv: (kv[1], kv[0]), reverse=True)[:k]
        dcg = sum(rels.get(doc, 0) / math.log2(r + 2) for r, (doc, _) in enumerate(ranked))
        ideal = sorted((v for v in rels.values() if v > 0), reverse=True)[:k]
        idcg = sum(g / math.log2(r + 2) for r, g in enumerate(ideal))
        vals.append(dcg / idcg if idcg > 0 else 0.0)
    return float(np.mean(vals)) if vals else 0.0


# --------------------------------------------------------------------- parity checks
def check_topk(D, I, corpus, queries, k, id_offset=0, score_tol=1e-3, tie_tol=2e-6):
    """(D, I) from the HIP path vs the float64 oracle on the SAME quantised inputs.

    Bar (BASELINE.json north_star): bit-exact top-k doc ids, scores within 1e-3.
    fp32 accumulation order on the GPU differs from float64, so two scores closer than
    `tie_tol` may legitimately swap; a difference in ids is accepted ONLY if the oracle's own
    float64 scores of the ids involved differ by < tie_tol — never by count.
    Returns the number of positions where ids differed (all explained)."""
    D = np.asarray(D)
    I = np.asarray(I)
    D0, I0 = ip_topk(corpus, queries, k, f64=True, id_offset=id_offset)
    n = corpus.shape[0]
    kk = min(k, n)
    assert D.shape == D0.shape and I.shape == I0.shape
    assert (I[:, kk:] == -1).all()          # padding
    assert (D[:, kk:] <= -3.0e38).all()
    swaps = 0
    for q in range(queries.shape[0]):
        if np.array_equal(I[q, :kk], I0[q, :kk]):
            np.testing.assert_allclose(D[q, :kk], D0[q, :kk], atol=score_tol, rtol=0)
            continue
        s = scores_f64(corpus, queries[q])
        got = I[q, :kk] - id_offset
        assert got.min() >= 0 and got.max() < n, "id out of range"
        assert len(set(got.tolist())) == kk, "duplicate ids in the result"
        sg = s[got]
        # returned scores match the oracle's score of the SAME id
        np.testing.assert_allclose(D[q, :kk], sg, atol=score_tol, rtol=0)
        # order is descending up to near-ties; exact ties by ascending id
        dif = np.diff(sg)
        assert (dif <= tie_tol).all(), f"query {q}: result not sorted (max inversion {dif.max()})"
        # nothing better than the boundary was left out
        kth = np.sort(s)[::-1][kk - 1]
        assert sg.min() >= kth - tie_tol, f"query {q}: a returned id is below the k-th best score"
        left_out = np.setdiff1d(I0[q, :kk] - id_offset, got)
        assert (s[left_out] <= sg.min() + tie_tol).all(), f"query {q}: a better id was left out"
        swaps += int((I[q, :kk] != I0[q, :kk]).sum())
    return swaps


def check_topk_sparse(d_row, i_row, ref_ids, fetch_rows, query, score_tol=1e-3, tie_tol=2e-6):
    """The same rule where a float64 oracle over the whole corpus is too slow (10 M / 50 M rows):
    `ref_ids` = top-k ids of an INDEPENDENT float32 reference (e.g. a rocBLAS GEMM + torch.topk on
    the same quantised data); `fetch_rows(ids) -> float32 [len(ids), d]` returns those corpus rows.
    Float64 oracle scores are computed for the union of both id sets only; every id on which the
    two results differ must sit within `tie_tol` (float64) of the k-th boundary.  Any row outside
    the union scored below the reference's k-th entry in the reference's own arithmetic, i.e. it
    cannot beat the boundary by more than that reference's rounding error.
    Returns the number of ids that differed (all explained)."""
    d_row = np.asarray(d_row, dtype=np.float32)
    got = np.asarray(i_row, dtype=np.int64)
    ref = np.asarray(ref_ids, dtype=np.int64)
    k = got.shape[0]
    assert len(set(got.tolist())) == k, "duplicate ids in the result"
    union = np.union1d(got, ref)
    s = scores_f64(fetch_rows(union), query)          # float64 accumulation, as in ip_topk
    pos = {int(u): j for j, u in enumerate(union)}
    sg = np.array([s[pos[int(i)]] for i in got])
    np.testing.assert_allclose(d_row, sg, atol=score_tol, rtol=0)
    dif = np.diff(sg)
    assert (dif <= tie_tol).all(), f"result not sorted (max inversion {dif.max()})"
    eq = np.nonzero(dif == 0.0)[0]
    assert (got[eq + 1] > got[eq]).all(), "exact ties must come in ascending id order"
    only_got = np.setdiff1d(got, ref)
    only_ref = np.setdiff1d(ref, got)
    if only_got.size or only_ref.size:
        kth = np.sort(s)[::-1][k - 1]                 # boundary among everything either side returned
        for i in only_ref:
            assert s[pos[int(i)]] <= kth + tie_tol, f"id {int(i)} (score {s[pos[int(i)]]}) left out above the boundary {kth}"
        for i in only_got:
            assert s[pos[int(i)]] >= kth - tie_tol, f"id {int(i)} (score {s[pos[int(i)]]}) returned from below the boundary {kth}"
    return int(only_got.size)
