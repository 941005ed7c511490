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
  // LANES: `lanes` independent sets of the per-query workspace, so that a batch of queries runs as ONE launch per token
  // position (blockIdx.y = lane) instead of one chain of launches per query; lane y's arrays start at y * N (acc, touched,
  // keys), y * 4 (counters), y * BM_CAND_CAP (cand), y (tau).
  int lanes = 0;
  double* acc = nullptr;       // [lanes][N], all zero between queries
  int32_t* touched = nullptr;  // [lanes][N]
  uint32_t* counters = nullptr;  // [lanes][4]: [0] n_touched, [1] n_out, [2] n_cand, [3] 1 = candidates usable
  uint64_t* keys = nullptr;    // [lanes][N] score keys of the touched documents (pre-filter)
  int32_t* cand = nullptr;     // [lanes][BM_CAND_CAP]
  uint64_t* tau = nullptr;     // [lanes]
  void* steps = nullptr;       // device copy of a batch's per-(token position, lane) posting ranges
  size_t steps_bytes = 0;
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

// One token position of every lane's query: lane y = blockIdx.y adds the postings of ITS token (st[y]: offset, length, idf;
// length 0 = this query has no token at this position).  Within a lane the launches follow each other in query order and a
// token's postings hit distinct documents: the same additions in the same order as one query at a time.
struct BmStep { int64_t off, df; double idf; };
__global__ void bm25_accumulate(const int32_t* __restrict__ post_doc, const float* __restrict__ post_tf,
                                const BmStep* __restrict__ st, double k1p1, int64_t N,
                                const double* __restrict__ len_norm, double* __restrict__ acc,
                                int32_t* __restrict__ touched, uint32_t* __restrict__ counters) {
  // no fused multiply-add here: the reference computes idf*(num/den) and the running
  // sum with one rounding per operation (CPython floats), and hipcc contracts by default
#pragma clang fp contract(off)
  const int y = blockIdx.y;
  const int64_t off = st[y].off, df = st[y].df;
  const double idf = st[y].idf;
  acc += (size_t)y * N; touched += (size_t)y * N;
  uint32_t* n_touched = counters + 4 * y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= df) return;
  const int32_t d = post_doc[off + i];
  const double tf = (double)post_tf[off + i];
  const double c = idf * ((tf * k1p1) / (tf + len_norm[d]));
  const double old = acc[d];
  acc[d] = old + c;
  if (old == 0.0) touched[atomicAdd(n_touched, 1u)] = d;  // contributions are > 0: first touch
}

__global__ void bm25_reset(const int32_t* __restrict__ touched, const uint32_t* __restrict__ counters,
                           double* __restrict__ acc, int64_t N) {
  const int y = blockIdx.y;
  touched += (size_t)y * N; acc += (size_t)y * N;
  const uint32_t n = counters[4 * y];
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
__global__ void bm25_keys(const int32_t* __restrict__ touched, const uint32_t* __restrict__ counters,
                          const double* __restrict__ acc, uint64_t* __restrict__ keys, int64_t N) {
  const int y = blockIdx.y;
  touched += (size_t)y * N; acc += (size_t)y * N; keys += (size_t)y * N;
  const uint32_t n = counters[4 * y];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    keys[i] = d2key(acc[touched[i]]);
}

// one workgroup: tau = the m-th best of BM_SAMPLE strided keys, m ~ 4k * sample / n
__global__ __launch_bounds__(BM_SEL_THREADS) void bm25_tau(const uint64_t* __restrict__ keys, int k,
                                                           uint64_t* __restrict__ tau,
                                                           uint32_t* __restrict__ counters, int64_t N) {
  __shared__ uint64_t sk[BM_SAMPLE];
  const int tid = threadIdx.x;
  const int y = blockIdx.y;
  keys += (size_t)y * N; tau += y; counters += 4 * y;
  const uint32_t n = counters[0];
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

__global__ void bm25_filter(const int32_t* __restrict__ touched, const uint64_t* __restrict__ keys,
                            const uint64_t* __restrict__ tau, int32_t* __restrict__ cand,
                            uint32_t* __restrict__ counters, int64_t N) {
  const int y = blockIdx.y;
  touched += (size_t)y * N; keys += (size_t)y * N; tau += y; cand += (size_t)y * BM_CAND_CAP; counters += 4 * y;
  if (counters[3] == 0) return;
  const uint32_t n = counters[0];
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
// Lane y = blockIdx.y: the list is the lane's candidate list (use_cand 1) or its touched list (0); results go to row y of
// out_s / out_i [lanes, k] and the count to n_out[y].
__global__ __launch_bounds__(BM_SEL_THREADS) void bm25_select(const int32_t* __restrict__ touched_all,
                                                              const int32_t* __restrict__ cand_all,
                                                              const double* __restrict__ acc, int k,
                                                              double* __restrict__ out_s,
                                                              int32_t* __restrict__ out_i,
                                                              uint32_t* __restrict__ n_out,
                                                              const uint32_t* __restrict__ counters,
                                                              int use_cand, int64_t N) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sh[8];
  __shared__ uint64_t ssk[BM_MAX_K];
  __shared__ uint32_t sdk[BM_MAX_K];
  const int tid = threadIdx.x;
  const int y = blockIdx.y;
  counters += 4 * y; acc += (size_t)y * N; out_s += (size_t)y * k; out_i += (size_t)y * k; n_out += y;
  const int32_t* touched = use_cand ? cand_all + (size_t)y * BM_CAND_CAP : touched_all + (size_t)y * N;
  const uint32_t* n_touched = use_cand ? counters + 2 : counters;
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
                  h->keys, h->cand, h->tau, h->batch_buf, h->steps};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  h->batch_buf = nullptr; h->batch_bytes = 0;
  h->steps = nullptr; h->steps_bytes = 0; h->lanes = 0;
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

// (Re)allocates the per-query workspace for `lanes` lanes, zeroed.  No launch may be pending on the handle.
#define BM_LANE_BUDGET (4ull << 30)   // bytes of lane workspace a handle may hold (20 bytes per document and lane)
static int bm25_set_lanes(ts_bm25* h, int lanes) {
  void* old[] = {h->acc, h->touched, h->counters, h->keys, h->cand, h->tau};
  for (void* b : old)
    if (b) (void)hipFree(b);
  h->acc = nullptr; h->touched = nullptr; h->counters = nullptr; h->keys = nullptr; h->cand = nullptr; h->tau = nullptr;
  h->lanes = 0;
  const size_t n1 = (size_t)(h->N > 0 ? h->N : 1) * (size_t)lanes;
  TS_HIP(hipMalloc((void**)&h->acc, n1 * 8));
  TS_HIP(hipMalloc((void**)&h->touched, n1 * 4));
  TS_HIP(hipMalloc((void**)&h->counters, (size_t)lanes * 16));
  TS_HIP(hipMalloc((void**)&h->keys, n1 * 8));
  TS_HIP(hipMalloc((void**)&h->cand, (size_t)lanes * BM_CAND_CAP * 4));
  TS_HIP(hipMalloc((void**)&h->tau, (size_t)lanes * 8));
  TS_HIP(hipMemset(h->acc, 0, n1 * 8));
  TS_HIP(hipMemset(h->counters, 0, (size_t)lanes * 16));
  h->lanes = lanes;
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
  TS_HIP(hipMalloc((void**)&h->out_s, BM_MAX_K * 8));
  TS_HIP(hipMalloc((void**)&h->out_i, BM_MAX_K * 4));
  TS_CHECK(bm25_set_lanes(h, 1));
  if (nnz) {
    TS_HIP(hipMemcpy(h->post_doc, post_doc, (size_t)nnz * 4, hipMemcpyHostToDevice));
    TS_HIP(hipMemcpy(h->post_tf, post_tf, (size_t)nnz * 4, hipMemcpyHostToDevice));
  }
  if (N) TS_HIP(hipMemcpy(h->len_norm, len_norm, (size_t)N * 8, hipMemcpyHostToDevice));
  return TS_OK;
}

// term_ids: the query's tokens mapped to vocabulary ids, in query order (unknown
// tokens dropped, repeats kept).  Writes up to k (score, doc) pairs, best first;
// *n_out < k means every document with a non-zero score is in the output.
// A chunk of up to h->lanes queries (query q0 + y in lane y): one bm25_accumulate launch per token position, then the
// pre-filter (if some lane's postings are long enough to need it) and the two select launches, all with blockIdx.y = lane;
// results land in rows q0 .. of bs / bi / bc (device).  Accumulators, touched lists and counters are back to zero afterwards.
// Everything is stream-ordered; `st_host` / `st_dev` hold this chunk's [positions][lanes] posting ranges.
static int bm25_enqueue_chunk(ts_bm25* h, const int32_t* term_ids, const int64_t* term_off, int q0, int nl, int32_t k,
                              double* bs, int32_t* bi, uint32_t* bc, BmStep* st_host, BmStep* st_dev, hipStream_t s) {
  const int L = h->lanes;
  int64_t max_terms = 0, max_total = 0;
  for (int y = 0; y < nl; ++y) max_terms = std::max<int64_t>(max_terms, term_off[q0 + y + 1] - term_off[q0 + y]);
  std::vector<int64_t> max_df((size_t)max_terms, 0);
  for (int y = 0; y < nl; ++y) {
    const int32_t* t = term_ids + term_off[q0 + y];
    const int64_t nt = term_off[q0 + y + 1] - term_off[q0 + y];
    int64_t total = 0;
    for (int64_t i = 0; i < max_terms; ++i) {
      BmStep& e = st_host[(size_t)i * L + y];
      e.off = 0; e.df = 0; e.idf = 0.0;
      if (i >= nt) continue;
      if (t[i] < 0 || t[i] >= h->V) { ts_set_error("term id %d out of range", t[i]); return TS_ERR_INVALID; }
      e.off = h->term_off[t[i]];
      e.df = h->term_off[t[i] + 1] - e.off;
      e.idf = h->idf_host[t[i]];
      if (e.df < 0) e.df = 0;
      total += e.df;
      max_df[(size_t)i] = std::max(max_df[(size_t)i], e.df);
    }
    max_total = std::max(max_total, total);
  }
  for (int y = nl; y < L; ++y)
    for (int64_t i = 0; i < max_terms; ++i) st_host[(size_t)i * L + y] = BmStep{0, 0, 0.0};
  if (max_terms)
    TS_HIP(hipMemcpyAsync(st_dev, st_host, (size_t)max_terms * L * sizeof(BmStep), hipMemcpyHostToDevice, s));
  for (int64_t i = 0; i < max_terms; ++i) {
    if (max_df[(size_t)i] <= 0) continue;
    const int64_t blocks = (max_df[(size_t)i] + 255) / 256;
    hipLaunchKernelGGL(bm25_accumulate, dim3((unsigned)blocks, (unsigned)nl), dim3(256), 0, s, h->post_doc, h->post_tf,
                       st_dev + (size_t)i * L, h->k1p1, h->N, h->len_norm, h->acc, h->touched, h->counters);
  }
  if (max_total > BM_PRE_MIN) {
    // long touched lists (their lengths are only known on the device; a lane's postings' total bounds its list; a lane
    // whose list is short is left alone by bm25_tau: tau = all ones, "candidates usable" stays 0)
    const int blocks = (int)std::min<int64_t>(1024, (max_total + 255) / 256);
    hipLaunchKernelGGL(bm25_keys, dim3(blocks, nl), dim3(256), 0, s, h->touched, h->counters, h->acc, h->keys, h->N);
    hipLaunchKernelGGL(bm25_tau, dim3(1, nl), dim3(BM_SEL_THREADS), 0, s, h->keys, k, h->tau, h->counters, h->N);
    hipLaunchKernelGGL(bm25_filter, dim3(blocks, nl), dim3(256), 0, s, h->touched, h->keys, h->tau, h->cand, h->counters, h->N);
    hipLaunchKernelGGL(bm25_select, dim3(1, nl), dim3(BM_SEL_THREADS), 0, s, h->touched, h->cand, h->acc, k,
                       bs + (size_t)q0 * k, bi + (size_t)q0 * k, bc + q0, h->counters, 1, h->N);
  }
  // (a lane runs this one only when its candidate list was not made or is not usable)
  hipLaunchKernelGGL(bm25_select, dim3(1, nl), dim3(BM_SEL_THREADS), 0, s, h->touched, h->cand, h->acc, k,
                     bs + (size_t)q0 * k, bi + (size_t)q0 * k, bc + q0, h->counters, 0, h->N);
  hipLaunchKernelGGL(bm25_reset, dim3(256, nl), dim3(256), 0, s, h->touched, h->counters, h->acc, h->N);
  TS_HIP(hipGetLastError());
  TS_HIP(hipMemsetAsync(h->counters, 0, (size_t)L * 16, s));
  return TS_OK;
}

extern "C" int ts_bm25_search_batch(ts_bm25* h, const int32_t* term_ids, const int64_t* term_off, int32_t nq, int32_t k,
                                    double* out_scores, int64_t* out_ids, int32_t* n_out, void* stream) {
  if (!h || !out_scores || !out_ids || !n_out || !term_off || nq < 0 || k <= 0) {
    ts_set_error("bad arguments to bm25_search_batch");
    return TS_ERR_INVALID;
  }
  if (k > BM_MAX_K) { ts_set_error("bm25 top_k %d exceeds %d", k, BM_MAX_K); return TS_ERR_UNSUPPORTED; }
  int64_t max_terms = 0;
  for (int q = 0; q < nq; ++q) {
    n_out[q] = 0;
    if (term_off[q + 1] < term_off[q] || (term_off[q + 1] > term_off[q] && !term_ids)) {
      ts_set_error("bad term offsets in bm25_search_batch");
      return TS_ERR_INVALID;
    }
    max_terms = std::max(max_terms, term_off[q + 1] - term_off[q]);
  }
  if (nq == 0 || h->N == 0) return TS_OK;
  Guard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  // lanes: as many as the batch has queries, within the workspace budget (64 at most)
  {
    const size_t per_lane = (size_t)h->N * 20 + (size_t)BM_CAND_CAP * 4 + 32;
    int want = (int)std::min<int64_t>(std::min<int64_t>(nq, 64), std::max<int64_t>(1, (int64_t)(BM_LANE_BUDGET / per_lane)));
    if (want > h->lanes) {
      TS_HIP(hipStreamSynchronize(s));
      TS_CHECK(bm25_set_lanes(h, want));
    }
  }
  const int L = h->lanes;
  // per-batch result buffers (device): [nq, k] scores, [nq, k] ids, [nq] counts — one copy back and ONE sync per batch
  const size_t need = (size_t)nq * k * 12 + (size_t)nq * 4;
  if (need > h->batch_bytes) {
    if (h->batch_buf) (void)hipFree(h->batch_buf);
    h->batch_buf = nullptr; h->batch_bytes = 0;
    TS_HIP(hipMalloc(&h->batch_buf, need));
    h->batch_bytes = need;
  }
  const int nchunks = (nq + L - 1) / L;
  const size_t st_count = (size_t)std::max<int64_t>(max_terms, 1) * L;
  if ((size_t)nchunks * st_count * sizeof(BmStep) > h->steps_bytes) {
    if (h->steps) (void)hipFree(h->steps);
    h->steps = nullptr; h->steps_bytes = 0;
    TS_HIP(hipMalloc(&h->steps, (size_t)nchunks * st_count * sizeof(BmStep)));
    h->steps_bytes = (size_t)nchunks * st_count * sizeof(BmStep);
  }
  std::vector<BmStep> st_host((size_t)nchunks * st_count);   // (alive until the synchronisation below: source of async copies)
  double* bs = reinterpret_cast<double*>(h->batch_buf);
  int32_t* bi = reinterpret_cast<int32_t*>(bs + (size_t)nq * k);
  uint32_t* bc = reinterpret_cast<uint32_t*>(bi + (size_t)nq * k);
  TS_HIP(hipMemsetAsync(bc, 0, (size_t)nq * 4, s));
  int st = TS_OK;
  for (int c = 0; c < nchunks && st == TS_OK; ++c) {
    const int q0 = c * L, nl = std::min(L, nq - q0);
    st = bm25_enqueue_chunk(h, term_ids, term_off, q0, nl, k, bs, bi, bc, st_host.data() + (size_t)c * st_count,
                            reinterpret_cast<BmStep*>(h->steps) + (size_t)c * st_count, s);
  }
  if (st != TS_OK) {   // a bad term id part-way: leave the handle clean (accumulators / counters) before reporting it
    hipLaunchKernelGGL(bm25_reset, dim3(256, L), dim3(256), 0, s, h->touched, h->counters, h->acc, h->N);
    (void)hipMemsetAsync(h->counters, 0, (size_t)L * 16, s);
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
