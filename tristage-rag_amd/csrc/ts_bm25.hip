// BM25 scoring on the GPU — the lexical half of the reference's default stage 1
// (BM25Index, reference src/stage1_retriever.py:35-112, used at :385-400 and fused
// with the dense list by RRF / weighted fusion).  The reference scores every
// document with a per-document Python loop (O(N*|q|) per query); here the
// postings live in HBM as CSR and a query touches only the postings of its terms.
//
// Arithmetic = the reference's, in float64:  for query tokens in order,
//     acc[d] += idf[t] * (tf*(k1+1)) / (tf + k1*(1 - b + b*len[d]/avg_len))
// One launch per query token (a token's postings hit distinct documents, so no
// atomics on the accumulator and every document receives its additions in query
// order: bit-identical sums).  The first launch that touches a document appends
// it to the query's "touched" list; the top-k (score desc, doc id asc — the
// reference's stable descending sort) is selected from that list by a 96-bit
// MSB-first radix select (64 score bits, then 32 id bits) and a bitonic sort of
// the k survivors.  Documents nobody touched score 0.0 and are filled in by the
// host in ascending id order when fewer than k documents were touched.
//
// That select is one workgroup making up to 12 gathering passes over the touched list, which
// is fine for thousands of documents and far too slow for millions (a common term touches a
// large share of the corpus).  Above BM_PRE_MIN touched documents the list is first cut down
// the way the stage-1 scan does it: score keys -> a strided sample -> the m-th best sample key
// as threshold -> a chip-wide filter into a candidate list (~4k expected) -> the exact select on
// the candidates.  The threshold only decides how fast: if the candidate list comes out short
// or overflows (massive ties), the exact select runs over the whole touched list instead.
#include "ts_common.h"

#include <algorithm>
#include <new>
#include <vector>

#define BM_SEL_THREADS 1024
#define BM_MAX_K 2048
#define BM_PRE_MIN 8192     // posting-list total below which the select runs on the touched list directly
#define BM_SAMPLE 4096      // score keys sampled for the threshold
#define BM_CAND_CAP 16384   // candidate list of the pre-filter

struct ts_bm25 {
  int device = 0;
  int64_t N = 0, V = 0, nnz = 0;
  double k1p1 = 2.2;
  int32_t* post_doc = nullptr;
  float* post_tf = nullptr;
  double* idf = nullptr;       // [V]
  double* len_norm = nullptr;  // [N]  k1*(1-b+b*len/avg)
  double* acc = nullptr;       // [N], all zero between queries
  int32_t* touched = nullptr;  // [N]
  uint32_t* counters = nullptr;  // [0] n_touched, [1] n_out, [2] n_cand, [3] 1 = candidates usable
  uint64_t* keys = nullptr;    // [N] score keys of the touched documents (pre-filter)
  int32_t* cand = nullptr;     // [BM_CAND_CAP]
  uint64_t* tau = nullptr;     // [1]
  double* out_s = nullptr;     // [BM_MAX_K]
  int32_t* out_i = nullptr;    // [BM_MAX_K]
  int64_t* term_off = nullptr;  // host copy [V+1]
  double* idf_host = nullptr;
  void* batch_buf = nullptr;   // results of a query batch (device), grown on demand
  size_t batch_bytes = 0;
};

namespace {
struct Guard {
  int prev = -1;
  explicit Guard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~Guard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};
}  // namespace

__global__ void bm25_accumulate(const int32_t* __restrict__ post_doc, const float* __restrict__ post_tf,
                                int64_t off, int64_t df, double idf, double k1p1,
                                const double* __restrict__ len_norm, double* __restrict__ acc,
                                int32_t* __restrict__ touched, uint32_t* __restrict__ n_touched) {
  // no fused multiply-add here: the reference computes idf*(num/den) and the running
  // sum with one rounding per operation (CPython floats), and hipcc contracts by default
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= df) return;
  const int32_t d = post_doc[off + i];
  const double tf = (double)post_tf[off + i];
  const double c = idf * ((tf * k1p1) / (tf + len_norm[d]));
  const double old = acc[d];
  acc[d] = old + c;
  if (old == 0.0) touched[atomicAdd(n_touched, 1u)] = d;  // contributions are > 0: first touch
}

__global__ void bm25_reset(const int32_t* __restrict__ touched, const uint32_t* __restrict__ n_touched,
                           double* __restrict__ acc) {
  const uint32_t n = *n_touched;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    acc[touched[i]] = 0.0;
}

__device__ __forceinline__ uint64_t d2key(double v) {  // orderable; scores are >= 0 but stay general
  uint64_t u = __builtin_bit_cast(uint64_t, v + 0.0);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// digit `pass` (0..11) of the 96-bit key (score key, then ~doc so that smaller ids rank higher)
__device__ __forceinline__ uint32_t key_digit(uint64_t sk, uint32_t dk, int pass) {
  return pass < 8 ? (uint32_t)(sk >> (56 - 8 * pass)) & 0xFFu : (dk >> (24 - 8 * (pass - 8))) & 0xFFu;
}

__device__ __forceinline__ bool prefix_match(uint64_t sk, uint32_t dk, uint64_t psk, uint32_t pdk, int passes) {
  if (passes == 0) return true;
  if (passes <= 8) return passes == 8 ? sk == psk : (sk >> (64 - 8 * passes)) == (psk >> (64 - 8 * passes));
  if (sk != psk) return false;
  const int b = 8 * (passes - 8);
  return b == 32 ? dk == pdk : (dk >> (32 - b)) == (pdk >> (32 - b));
}

// one vote per (thread, key); waves whose lanes agree on the bin add once (the top
// bytes of the score key are shared by almost all documents)
__device__ __forceinline__ void bm_vote(uint32_t* hist, bool in, uint32_t digit, int tid) {
  const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
  if (m == 0ull) return;
  const int src = __builtin_ctzll(m);
  const uint32_t d0 = (uint32_t)__shfl((int)digit, src, 64);
  const unsigned long long same = __builtin_amdgcn_ballot_w64(in && digit == d0);
  if (same == m) {
    if ((tid & 63) == src) atomicAdd(&hist[d0], (uint32_t)__builtin_popcountll(m));
  } else if (in) {
    atomicAdd(&hist[digit], 1u);
  }
}

__device__ __forceinline__ bool key_ge(uint64_t sk, uint32_t dk, uint64_t tsk, uint32_t tdk) {
  return sk > tsk || (sk == tsk && dk >= tdk);
}

// ---- pre-filter for long touched lists --------------------------------------------------
__global__ void bm25_keys(const int32_t* __restrict__ touched, const uint32_t* __restrict__ n_touched,
                          const double* __restrict__ acc, uint64_t* __restrict__ keys) {
  const uint32_t n = *n_touched;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    keys[i] = d2key(acc[touched[i]]);
}

// one workgroup: tau = the m-th best of BM_SAMPLE strided keys, m ~ 4k * sample / n
__global__ __launch_bounds__(BM_SEL_THREADS) void bm25_tau(const uint64_t* __restrict__ keys,
                                                           const uint32_t* __restrict__ n_touched, int k,
                                                           uint64_t* __restrict__ tau,
                                                           uint32_t* __restrict__ counters) {
  __shared__ uint64_t sk[BM_SAMPLE];
  const int tid = threadIdx.x;
  const uint32_t n = *n_touched;
  if (tid == 0) { counters[2] = 0; counters[3] = 0; }
  if (n <= BM_PRE_MIN) {          // short list: the exact select takes it as it is
    if (tid == 0) *tau = ~0ull;
    return;
  }
  for (uint32_t j = tid; j < BM_SAMPLE; j += BM_SEL_THREADS)
    sk[j] = keys[(uint64_t)j * n / BM_SAMPLE];
  __syncthreads();
  for (uint32_t size = 2; size <= BM_SAMPLE; size <<= 1) {   // bitonic, descending
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t t = tid; t < BM_SAMPLE / 2; t += BM_SEL_THREADS) {
        const uint32_t i = 2 * t - (t & (stride - 1)), j = i + stride;
        const bool desc = (i & size) == 0;
        const uint64_t a = sk[i], b = sk[j];
        if (desc ? (b > a) : (a > b)) { sk[i] = b; sk[j] = a; }
      }
      __syncthreads();
    }
  }
  if (tid == 0) {
    uint64_t m = (4ull * (uint64_t)k * BM_SAMPLE + n - 1) / n;
    if (m < 8) m = 8;
    if (m > BM_SAMPLE) m = BM_SAMPLE;
    *tau = sk[m - 1];
    counters[3] = 1;
  }
}

__global__ void bm25_filter(const int32_t* __restrict__ touched, const uint32_t* __restrict__ n_touched,
                            const uint64_t* __restrict__ keys, const uint64_t* __restrict__ tau,
                            int32_t* __restrict__ cand, uint32_t* __restrict__ counters) {
  if (counters[3] == 0) return;
  const uint32_t n = *n_touched;
  const uint64_t t = *tau;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (keys[i] >= t) {
      const uint32_t pos = atomicAdd(&counters[2], 1u);
      if (pos < BM_CAND_CAP) cand[pos] = touched[i];
    }
  }
}

// one workgroup: exact top-k of the documents in `touched` (the whole touched list, or the
// candidate list of the pre-filter).  use_cand: 1 = run only if the candidate list is usable
// (it holds at least min(k, n_touched) documents and did not overflow); 0 = run only if it is not.
__global__ __launch_bounds__(BM_SEL_THREADS) void bm25_select(const int32_t* __restrict__ touched,
                                                              const uint32_t* __restrict__ n_touched,
                                                              const double* __restrict__ acc, int k,
                                                              double* __restrict__ out_s,
                                                              int32_t* __restrict__ out_i,
                                                              uint32_t* __restrict__ n_out,
                                                              const uint32_t* __restrict__ counters,
                                                              int use_cand) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sh[8];
  __shared__ uint64_t ssk[BM_MAX_K];
  __shared__ uint32_t sdk[BM_MAX_K];
  const int tid = threadIdx.x;
  const uint32_t n_all = counters[0];
  const uint32_t kk = (uint32_t)k < n_all ? (uint32_t)k : n_all;
  const bool cand_ok = counters[3] != 0 && counters[2] >= kk && counters[2] <= BM_CAND_CAP;
  if ((use_cand != 0) != cand_ok) return;   // (uniform) the other launch does the work
  const uint32_t n = *n_touched;
  if (tid == 0) *n_out = kk;
  if (kk == 0) return;
  uint64_t psk = 0;
  uint32_t pdk = 0;
  int passes = 0;
  uint32_t krem = kk;
  bool whole = (n <= kk);  // everything is wanted: skip the select
  if (!whole) {
    for (int pass = 0; pass < 12; ++pass) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      for (uint32_t base = 0; base < n; base += BM_SEL_THREADS) {  // wave-uniform trip count
        const uint32_t i = base + tid;
        const int32_t d = (i < n) ? touched[i] : 0;
        const uint64_t sk = (i < n) ? d2key(acc[d]) : 0ull;
        const uint32_t dk = ~(uint32_t)d;
        bm_vote(hist, (i < n) && prefix_match(sk, dk, psk, pdk, passes), key_digit(sk, dk, pass), tid);
      }
      __syncthreads();
      if (tid == 0) {
        uint32_t cum = 0;
        int dg = 255;
        for (; dg > 0; --dg) {
          if (cum + hist[dg] >= krem) break;
          cum += hist[dg];
        }
        sh[0] = (uint32_t)dg;
        sh[1] = krem - cum;
        sh[2] = (hist[dg] == krem - cum) ? 1u : 0u;
      }
      __syncthreads();
      const uint32_t dg = sh[0];
      krem = sh[1];
      if (pass < 8) psk |= (uint64_t)dg << (56 - 8 * pass);
      else pdk |= dg << (24 - 8 * (pass - 8));
      passes = pass + 1;
      const bool done = sh[2] != 0;
      __syncthreads();
      if (done) break;
    }
  }
  // survivors: key >= threshold prefix (lower bits zero)
  if (tid == 0) sh[3] = 0;
  __syncthreads();
  for (uint32_t i = tid; i < n; i += BM_SEL_THREADS) {
    const int32_t d = touched[i];
    const uint64_t sk = d2key(acc[d]);
    const uint32_t dk = ~(uint32_t)d;
    if (whole || key_ge(sk, dk, psk, pdk)) {
      const uint32_t pos = atomicAdd(&sh[3], 1u);
      if (pos < BM_MAX_K) { ssk[pos] = sk; sdk[pos] = dk; }
    }
  }
  __syncthreads();
  const uint32_t cnt = sh[3] < BM_MAX_K ? sh[3] : BM_MAX_K;
  uint32_t P = 2;
  while (P < cnt) P <<= 1;
  for (uint32_t i = cnt + tid; i < P; i += BM_SEL_THREADS) { ssk[i] = 0; sdk[i] = 0; }
  __syncthreads();
  for (uint32_t size = 2; size <= P; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t t = tid; t < (P >> 1); t += BM_SEL_THREADS) {
        const uint32_t i = 2 * t - (t & (stride - 1)), j = i + stride;
        const bool desc = (i & size) == 0;
        const uint64_t a = ssk[i], b = ssk[j];
        const uint32_t ad = sdk[i], bd = sdk[j];
        const bool a_before_b = a > b || (a == b && ad > bd);
        const bool b_before_a = b > a || (a == b && bd > ad);
        if (desc ? b_before_a : a_before_b) { ssk[i] = b; ssk[j] = a; sdk[i] = bd; sdk[j] = ad; }
      }
      __syncthreads();
    }
  }
  for (uint32_t i = tid; i < kk && i < cnt; i += BM_SEL_THREADS) {
    const int32_t d = (int32_t)~sdk[i];
    out_i[i] = d;
    out_s[i] = acc[d];
  }
}

extern "C" int ts_bm25_create(int32_t device, ts_bm25** out) {
  if (!out) { ts_set_error("out is null"); return TS_ERR_INVALID; }
  *out = nullptr;
  int ndev = 0;
  TS_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) { ts_set_error("device %d not present", device); return TS_ERR_INVALID; }
  ts_bm25* h = new (std::nothrow) ts_bm25();
  if (!h) { ts_set_error("out of host memory"); return TS_ERR_OOM; }
  h->device = device;
  *out = h;
  return TS_OK;
}

static void bm25_free(ts_bm25* h) {
  void* bufs[] = {h->post_doc, h->post_tf, h->idf, h->len_norm, h->acc, h->touched, h->counters, h->out_s, h->out_i,
                  h->keys, h->cand, h->tau, h->batch_buf};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  h->batch_buf = nullptr; h->batch_bytes = 0;
  h->post_doc = nullptr; h->post_tf = nullptr; h->idf = nullptr; h->len_norm = nullptr; h->acc = nullptr;
  h->touched = nullptr; h->counters = nullptr; h->out_s = nullptr; h->out_i = nullptr;
  h->keys = nullptr; h->cand = nullptr; h->tau = nullptr;
  delete[] h->term_off; h->term_off = nullptr;
  delete[] h->idf_host; h->idf_host = nullptr;
}

extern "C" int ts_bm25_destroy(ts_bm25* h) {
  if (!h) return TS_OK;
  Guard g(h->device);
  (void)hipDeviceSynchronize();
  bm25_free(h);
  delete h;
  return TS_OK;
}

// All pointers are host memory.  term_off[V+1] (CSR), post_doc/post_tf[nnz] sorted by
// term, idf[V], len_norm[N] = k1*(1-b+b*len/avg), k1p1 = k1+1.
extern "C" int ts_bm25_set_index(ts_bm25* h, int64_t N, int64_t V, int64_t nnz, const int64_t* term_off,
                                 const int32_t* post_doc, const float* post_tf, const double* idf,
                                 const double* len_norm, double k1p1) {
  if (!h || N < 0 || V < 0 || nnz < 0 || N >= (1LL << 31) || (nnz && (!term_off || !post_doc || !post_tf)) ||
      (V && !idf) || (N && !len_norm)) {
    ts_set_error("bad arguments to bm25_set_index");
    return TS_ERR_INVALID;
  }
  Guard g(h->device);
  bm25_free(h);
  h->N = N; h->V = V; h->nnz = nnz; h->k1p1 = k1p1;
  h->term_off = new (std::nothrow) int64_t[V + 1];
  h->idf_host = new (std::nothrow) double[V > 0 ? V : 1];
  if (!h->term_off || !h->idf_host) { ts_set_error("out of host memory"); return TS_ERR_OOM; }
  for (int64_t i = 0; i <= V; ++i) h->term_off[i] = term_off ? term_off[i] : 0;
  for (int64_t i = 0; i < V; ++i) h->idf_host[i] = idf[i];
  const size_t n1 = (size_t)(N > 0 ? N : 1), z1 = (size_t)(nnz > 0 ? nnz : 1);
  TS_HIP(hipMalloc((void**)&h->post_doc, z1 * 4));
  TS_HIP(hipMalloc((void**)&h->post_tf, z1 * 4));
  TS_HIP(hipMalloc((void**)&h->len_norm, n1 * 8));
  TS_HIP(hipMalloc((void**)&h->acc, n1 * 8));
  TS_HIP(hipMalloc((void**)&h->touched, n1 * 4));
  TS_HIP(hipMalloc((void**)&h->counters, 64));
  TS_HIP(hipMalloc((void**)&h->out_s, BM_MAX_K * 8));
  TS_HIP(hipMalloc((void**)&h->out_i, BM_MAX_K * 4));
  TS_HIP(hipMalloc((void**)&h->keys, n1 * 8));
  TS_HIP(hipMalloc((void**)&h->cand, (size_t)BM_CAND_CAP * 4));
  TS_HIP(hipMalloc((void**)&h->tau, 8));
  if (nnz) {
    TS_HIP(hipMemcpy(h->post_doc, post_doc, (size_t)nnz * 4, hipMemcpyHostToDevice));
    TS_HIP(hipMemcpy(h->post_tf, post_tf, (size_t)nnz * 4, hipMemcpyHostToDevice));
  }
  if (N) TS_HIP(hipMemcpy(h->len_norm, len_norm, (size_t)N * 8, hipMemcpyHostToDevice));
  TS_HIP(hipMemset(h->acc, 0, n1 * 8));
  TS_HIP(hipMemset(h->counters, 0, 64));
  return TS_OK;
}

// term_ids: the query's tokens mapped to vocabulary ids, in query order (unknown
// tokens dropped, repeats kept).  Writes up to k (score, doc) pairs, best first;
// *n_out < k means every document with a non-zero score is in the output.
// One query's launches; results land in out_s / out_i (device, k entries) and its count in *cnt_out (device); the
// accumulator, the touched list and the counters are back to zero afterwards.  Everything is stream-ordered.
static int bm25_enqueue_query(ts_bm25* h, const int32_t* term_ids, int32_t n_terms, int32_t k, double* out_s,
                              int32_t* out_i, uint32_t* cnt_out, hipStream_t s) {
  int64_t total_df = 0;
  for (int i = 0; i < n_terms; ++i) {
    const int32_t t = term_ids[i];
    if (t < 0 || t >= h->V) { ts_set_error("term id %d out of range", t); return TS_ERR_INVALID; }
    const int64_t off = h->term_off[t], df = h->term_off[t + 1] - off;
    if (df <= 0) continue;
    total_df += df;
    const int64_t blocks = (df + 255) / 256;
    hipLaunchKernelGGL(bm25_accumulate, dim3((unsigned)blocks), dim3(256), 0, s, h->post_doc, h->post_tf,
                       off, df, h->idf_host[t], h->k1p1, h->len_norm, h->acc, h->touched, h->counters);
  }
  if (total_df > BM_PRE_MIN) {
    // long touched list (its length is only known on the device; the postings' total bounds it)
    const int blocks = (int)std::min<int64_t>(1024, (total_df + 255) / 256);
    hipLaunchKernelGGL(bm25_keys, dim3(blocks), dim3(256), 0, s, h->touched, h->counters, h->acc, h->keys);
    hipLaunchKernelGGL(bm25_tau, dim3(1), dim3(BM_SEL_THREADS), 0, s, h->keys, h->counters, k, h->tau, h->counters);
    hipLaunchKernelGGL(bm25_filter, dim3(blocks), dim3(256), 0, s, h->touched, h->counters, h->keys, h->tau,
                       h->cand, h->counters);
    hipLaunchKernelGGL(bm25_select, dim3(1), dim3(BM_SEL_THREADS), 0, s, h->cand, h->counters + 2, h->acc, k,
                       out_s, out_i, h->counters + 1, h->counters, 1);
  }
  // (runs only when the candidate list was not made or is not usable)
  hipLaunchKernelGGL(bm25_select, dim3(1), dim3(BM_SEL_THREADS), 0, s, h->touched, h->counters, h->acc, k,
                     out_s, out_i, h->counters + 1, h->counters, 0);
  TS_HIP(hipGetLastError());
  TS_HIP(hipMemcpyAsync(cnt_out, h->counters + 1, 4, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(bm25_reset, dim3(256), dim3(256), 0, s, h->touched, h->counters, h->acc);
  TS_HIP(hipMemsetAsync(h->counters, 0, 16, s));
  return TS_OK;
}

extern "C" int ts_bm25_search_batch(ts_bm25* h, const int32_t* term_ids, const int64_t* term_off, int32_t nq, int32_t k,
                                    double* out_scores, int64_t* out_ids, int32_t* n_out, void* stream) {
  if (!h || !out_scores || !out_ids || !n_out || !term_off || nq < 0 || k <= 0) {
    ts_set_error("bad arguments to bm25_search_batch");
    return TS_ERR_INVALID;
  }
  if (k > BM_MAX_K) { ts_set_error("bm25 top_k %d exceeds %d", k, BM_MAX_K); return TS_ERR_UNSUPPORTED; }
  for (int q = 0; q < nq; ++q) {
    n_out[q] = 0;
    if (term_off[q + 1] < term_off[q] || (term_off[q + 1] > term_off[q] && !term_ids)) {
      ts_set_error("bad term offsets in bm25_search_batch");
      return TS_ERR_INVALID;
    }
  }
  if (nq == 0 || h->N == 0) return TS_OK;
  Guard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  // per-batch result buffers (device): [nq, k] scores, [nq, k] ids, [nq] counts — one copy back and ONE sync per batch
  const size_t need = (size_t)nq * k * 12 + (size_t)nq * 4;
  if (need > h->batch_bytes) {
    if (h->batch_buf) (void)hipFree(h->batch_buf);
    h->batch_buf = nullptr; h->batch_bytes = 0;
    TS_HIP(hipMalloc(&h->batch_buf, need));
    h->batch_bytes = need;
  }
  double* bs = reinterpret_cast<double*>(h->batch_buf);
  int32_t* bi = reinterpret_cast<int32_t*>(bs + (size_t)nq * k);
  uint32_t* bc = reinterpret_cast<uint32_t*>(bi + (size_t)nq * k);
  TS_HIP(hipMemsetAsync(bc, 0, (size_t)nq * 4, s));
  int st = TS_OK;
  for (int q = 0; q < nq && st == TS_OK; ++q) {
    const int64_t nt = term_off[q + 1] - term_off[q];
    if (nt == 0) continue;
    st = bm25_enqueue_query(h, term_ids + term_off[q], (int32_t)nt, k, bs + (size_t)q * k, bi + (size_t)q * k, bc + q, s);
  }
  if (st != TS_OK) {   // a bad term id part-way: leave the handle clean (accumulator / counters) before reporting it
    hipLaunchKernelGGL(bm25_reset, dim3(256), dim3(256), 0, s, h->touched, h->counters, h->acc);
    (void)hipMemsetAsync(h->counters, 0, 16, s);
    (void)hipStreamSynchronize(s);
    return st;
  }
  std::vector<int32_t> ids32((size_t)nq * k);
  std::vector<uint32_t> cnt((size_t)nq);
  TS_HIP(hipMemcpyAsync(out_scores, bs, (size_t)nq * k * 8, hipMemcpyDeviceToHost, s));
  TS_HIP(hipMemcpyAsync(ids32.data(), bi, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s));
  TS_HIP(hipMemcpyAsync(cnt.data(), bc, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
  TS_HIP(hipStreamSynchronize(s));
  for (int q = 0; q < nq; ++q) {
    const int32_t n = (int32_t)std::min<uint32_t>(cnt[q], (uint32_t)k);
    for (int32_t i = 0; i < n; ++i) out_ids[(size_t)q * k + i] = ids32[(size_t)q * k + i];
    n_out[q] = n;
  }
  return TS_OK;
}

extern "C" int ts_bm25_search(ts_bm25* h, const int32_t* term_ids, int32_t n_terms, int32_t k,
                              double* out_scores, int64_t* out_ids, int32_t* n_out, void* stream) {
  if (!h || !out_scores || !out_ids || !n_out || n_terms < 0 || k <= 0 || (n_terms && !term_ids)) {
    ts_set_error("bad arguments to bm25_search");
    return TS_ERR_INVALID;
  }
  const int64_t off[2] = {0, n_terms};
  return ts_bm25_search_batch(h, term_ids, off, 1, k, out_scores, out_ids, n_out, stream);
}
