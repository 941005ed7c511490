/*
 * oracle.c — CPU restatement of the TriStage-RAG retrieval hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under tristage-rag_amd/ may import, link
 * or call this file; it is the checker for tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py.
 *
 * What it restates (reference = /root/reference, NoliNobdon/TriStage-RAG):
 *   oracle_ip_topk      faiss.IndexFlatIP.search as called at
 *                       src/stage1_retriever.py:380 and filtered at :383 —
 *                       exact inner product of every query with every row,
 *                       k best, descending, -1 padded.  FAISS itself
 *                       (requirements.txt:10 "faiss-cpu>=1.7.0", unpinned) is
 *                       not in the reference tree; its published IndexFlatIP
 *                       contract is restated here with one addition the build
 *                       fixes: ties are ordered by ascending id.
 *   oracle_normalize    Stage1Retriever._normalize_embeddings,
 *                       src/stage1_retriever.py:285-288: x / (|x|_2 + 1e-8)
 *   oracle_maxsim       ColBERTScorer._maxsim_score / _colbert_score,
 *                       src/stage2_rescorer.py:167-183 and :185-201, with
 *                       torch.nn.functional.normalize = x / max(|x|_2, 1e-12)
 *   oracle_minmax       CrossEncoderReranker._normalize_scores,
 *                       src/stage3_reranker.py:212-228
 *
 * Pinning: the reference ships no tests and no golden vectors for this path
 * (SURVEY.md §4).  The pure functions among the above are pinned against
 * outputs of the reference's own code imported in the build container
 * (tests/golden/make_golden.py -> tests/golden/*.json); the FAISS search has
 * no reference-side fixture at all => "parity unpinned" for that call, the
 * contract above is the build's own.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- heap of the k best, worst on top: worse = lower score, then higher id */
typedef struct { double s; int64_t id; } ent_t;

static inline int worse(const ent_t* a, const ent_t* b) {
  if (a->s != b->s) return a->s < b->s;
  return a->id > b->id;
}

static void sift_down(ent_t* h, int n, int i) {
  for (;;) {
    int l = 2 * i + 1, r = l + 1, m = i;
    if (l < n && worse(&h[l], &h[m])) m = l;
    if (r < n && worse(&h[r], &h[m])) m = r;
    if (m == i) return;
    ent_t t = h[i]; h[i] = h[m]; h[m] = t;
    i = m;
  }
}

static void sift_up(ent_t* h, int i) {
  while (i > 0) {
    int p = (i - 1) / 2;
    if (!worse(&h[i], &h[p])) return;
    ent_t t = h[i]; h[i] = h[p]; h[p] = t;
    i = p;
  }
}

static int cmp_best_first(const void* a, const void* b) {
  const ent_t* x = (const ent_t*)a; const ent_t* y = (const ent_t*)b;
  if (worse(x, y)) return 1;
  if (worse(y, x)) return -1;
  return 0;
}

/*
 * corpus [n,d], queries [nq,d] float32 row-major.  accumulate_f64 != 0: dot
 * products summed in double (the parity oracle); 0: summed in float in index
 * order (what a scalar fp32 CPU port computes; used for the timed baseline).
 * out_scores [nq,k] float32, out_ids [nq,k] int64; tail padded (-FLT_MAX, -1).
 * nthreads <= 0: all OpenMP threads.  Returns the thread count used.
 */
int oracle_ip_topk(const float* corpus, int64_t n, int32_t d, const float* queries,
                   int32_t nq, int32_t k, float* out_scores, int64_t* out_ids,
                   int32_t accumulate_f64, int32_t nthreads, int64_t id_offset) {
  int used = 1;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  used = nthreads < nq ? nthreads : (nq > 0 ? nq : 1);
#else
  (void)nthreads;
#endif
  const int kk = (int64_t)k < n ? k : (int)n;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(used)
#endif
  for (int32_t q = 0; q < nq; ++q) {
    const float* qv = queries + (int64_t)q * d;
    ent_t* heap = (ent_t*)malloc(sizeof(ent_t) * (size_t)(kk > 0 ? kk : 1));
    int hn = 0;
    for (int64_t i = 0; i < n; ++i) {
      const float* row = corpus + i * d;
      double s;
      if (accumulate_f64) {
        double acc = 0.0;
        for (int32_t j = 0; j < d; ++j) acc += (double)row[j] * (double)qv[j];
        s = acc;
      } else {
        float acc = 0.f;
#pragma omp simd reduction(+ : acc)
        for (int32_t j = 0; j < d; ++j) acc += row[j] * qv[j];
        s = (double)acc;
      }
      ent_t e = {s, i};
      if (hn < kk) {
        heap[hn] = e;
        sift_up(heap, hn);
        ++hn;
      } else if (kk > 0 && worse(&heap[0], &e)) {
        heap[0] = e;
        sift_down(heap, hn, 0);
      }
    }
    qsort(heap, (size_t)hn, sizeof(ent_t), cmp_best_first);
    for (int32_t r = 0; r < k; ++r) {
      if (r < hn) {
        out_scores[(int64_t)q * k + r] = (float)heap[r].s;
        out_ids[(int64_t)q * k + r] = heap[r].id + id_offset;
      } else {
        out_scores[(int64_t)q * k + r] = -FLT_MAX;
        out_ids[(int64_t)q * k + r] = -1;
      }
    }
    free(heap);
  }
  return used;
}

/* all scores of one query in double: lets a test look at the gap around rank k */
void oracle_scores_f64(const float* corpus, int64_t n, int32_t d, const float* query,
                       double* out) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int64_t i = 0; i < n; ++i) {
    const float* row = corpus + i * d;
    double acc = 0.0;
    for (int32_t j = 0; j < d; ++j) acc += (double)row[j] * (double)query[j];
    out[i] = acc;
  }
}

/* x / (|x|_2 + 1e-8), in float32 like numpy does for float32 input */
void oracle_normalize(const float* x, int64_t n, int32_t d, float* out) {
  for (int64_t i = 0; i < n; ++i) {
    const float* r = x + i * d;
    double ss = 0.0;
    for (int32_t j = 0; j < d; ++j) ss += (double)r[j] * (double)r[j];
    const float den = (float)sqrt(ss) + 1e-8f;
    for (int32_t j = 0; j < d; ++j) out[i * d + j] = r[j] / den;
  }
}

/*
 * q [Lq,H], docs [sum Ld,H] packed, off [n_docs+1].  mode 0 maxsim, 1 colbert.
 * A document with no tokens scores 0.0 (reference src/stage2_rescorer.py:285-291).
 */
void oracle_maxsim(const float* q, int32_t Lq, const float* docs, const int32_t* off,
                   int32_t n_docs, int32_t H, int32_t mode, float* out) {
  double* qn = (double*)malloc(sizeof(double) * (size_t)(Lq > 0 ? Lq : 1));
  for (int32_t i = 0; i < Lq; ++i) {
    double ss = 0.0;
    for (int32_t k = 0; k < H; ++k) ss += (double)q[(int64_t)i * H + k] * (double)q[(int64_t)i * H + k];
    qn[i] = fmax(sqrt(ss), 1e-12);
  }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
  for (int32_t dI = 0; dI < n_docs; ++dI) {
    const int32_t d0 = off[dI], Ld = off[dI + 1] - d0;
    if (Ld <= 0 || Lq <= 0) { out[dI] = 0.f; continue; }
    double* dn = (double*)malloc(sizeof(double) * (size_t)Ld);
    double* m = (double*)malloc(sizeof(double) * (size_t)Lq);
    for (int32_t j = 0; j < Ld; ++j) {
      const float* r = docs + (int64_t)(d0 + j) * H;
      double ss = 0.0;
      for (int32_t k = 0; k < H; ++k) ss += (double)r[k] * (double)r[k];
      dn[j] = fmax(sqrt(ss), 1e-12);
    }
    for (int32_t i = 0; i < Lq; ++i) {
      double best = -DBL_MAX;
      for (int32_t j = 0; j < Ld; ++j) {
        const float* r = docs + (int64_t)(d0 + j) * H;
        double acc = 0.0;
        for (int32_t k = 0; k < H; ++k) acc += (double)q[(int64_t)i * H + k] * (double)r[k];
        acc /= (qn[i] * dn[j]);
        if (acc > best) best = acc;
      }
      m[i] = best;
    }
    double res;
    if (mode == 0) {
      double s = 0.0;
      for (int32_t i = 0; i < Lq; ++i) s += m[i];
      res = s / (double)Lq;
    } else {
      double mx = -DBL_MAX, den = 0.0, num = 0.0;
      for (int32_t i = 0; i < Lq; ++i) if (m[i] > mx) mx = m[i];
      for (int32_t i = 0; i < Lq; ++i) { double e = exp(m[i] - mx); den += e; num += e * m[i]; }
      res = num / den;
    }
    out[dI] = (float)res;
    free(dn);
    free(m);
  }
  free(qn);
}

/* (s-min)/(max-min); all-equal -> zeros */
void oracle_minmax(const double* s, int32_t n, double* out) {
  if (n <= 0) return;
  double mn = s[0], mx = s[0];
  for (int32_t i = 1; i < n; ++i) { if (s[i] < mn) mn = s[i]; if (s[i] > mx) mx = s[i]; }
  for (int32_t i = 0; i < n; ++i) out[i] = (mx > mn) ? (s[i] - mn) / (mx - mn) : 0.0;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
