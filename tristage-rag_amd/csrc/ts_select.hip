// Exact top-k selection for gfx950: the part of faiss_index.search
// (reference src/stage1_retriever.py:380) that turns scores into the k best
// (score desc, id asc) — plus the threshold estimate that lets the scan kernel
// drop >99.9 % of the scores on the fly, and the merge of per-shard lists.
//
// One 1024-thread workgroup per query.  Everything is ordered by a 64-bit key
//     key = orderable(score) << 32 | (0xFFFFFFFF - tiebreak)
// so "larger key" == "better" and keys are unique (tiebreak = row id, or the
// position in the concatenated lists for the int64-id merge, where exact
// score ties are then resolved by comparing the real ids).
//   phase 1  entries -> LDS.  If they do not fit (dense scores of a large
//            chunk), an MSB-first 8-bit radix select over global memory finds
//            the key of the k-th best first and only the survivors are kept.
//   phase 2  if far more entries than k sit in LDS (the usual case after the
//            fused scan+filter: ~5 k candidates for k = 1000), a radix select
//            IN LDS finds the k-th key and the k survivors are compacted;
//   phase 3  bitonic sort of what is left (~k entries), write the first k.
#include "ts_common.h"

#define SEL_THREADS 1024
#define SEL_OUT_CAP 2048   // survivors kept by the in-LDS select (k <= 2048)
#define SEL_TIE_CAP 1024   // MERGE64: entries whose score equals the k-th score
#define NEG_MAX (-3.402823466e38f)

__device__ __forceinline__ uint32_t f2key(float f) {
  if (f != f) return 0u;  // NaN ranks last
  f = f + 0.0f;           // -0 -> +0 so equal scores have equal keys
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __builtin_bit_cast(float, u);
}

// entry e of a query lives at qbase + e, or — when the entries are the
// concatenation of `seg_len`-long lists that are `seg_stride` apart (the
// [nlists, nq, k] layout an all-gather produces) — at the address below
__device__ __forceinline__ int64_t entry_addr(const SelParams& p, int64_t qbase, uint32_t e) {
  if (p.seg_len == 0) return qbase + e;
  return qbase + (int64_t)(e / p.seg_len) * p.seg_stride + (e % p.seg_len);
}
// the int64 ids of a merge may sit at another list pitch than the scores (packed
// [scores | ids] per rank, as the all-gather delivers them)
__device__ __forceinline__ int64_t id_addr(const SelParams& p, int64_t qbase, uint32_t e) {
  if (p.seg_len == 0) return qbase + e;
  const int64_t pitch = p.seg_stride_ids ? p.seg_stride_ids : p.seg_stride;
  return qbase + (int64_t)(e / p.seg_len) * pitch + (e % p.seg_len);
}

template <int MODE>
__device__ __forceinline__ uint64_t load_key(const SelParams& p, int64_t qbase, uint32_t i) {
  const int64_t a = entry_addr(p, qbase, i);
  const float s = p.scores[a];
  uint32_t tb;
  if constexpr (MODE == SEL_DENSE) {
    tb = i + (uint32_t)p.id_base;
  } else if constexpr (MODE == SEL_PAIRS32) {
    const int32_t id = p.ids32[a];
    if (id < 0) return 0ull;  // padding entry of a short list
    tb = (uint32_t)id;
  } else {
    if (p.ids64[id_addr(p, qbase, i)] < 0) return 0ull;
    tb = i;
  }
  return ((uint64_t)f2key(s) << 32) | (uint64_t)(0xFFFFFFFFu - tb);
}

template <int MODE>
__device__ __forceinline__ bool key_before(const SelParams& p, int64_t qbase, uint64_t a,
                                           uint64_t b) {
  // true when a must come before b in the output (a is "better")
  if constexpr (MODE == SEL_MERGE64) {
    const uint32_t sa = (uint32_t)(a >> 32), sb = (uint32_t)(b >> 32);
    if (sa != sb) return sa > sb;
    if (a == 0ull || b == 0ull) return a > b;
    const int64_t ia = p.ids64[id_addr(p, qbase, 0xFFFFFFFFu - (uint32_t)a)];
    const int64_t ib = p.ids64[id_addr(p, qbase, 0xFFFFFFFFu - (uint32_t)b)];
    return ia < ib;
  } else {
    return a > b;
  }
}

// ---- shared pieces ----------------------------------------------------------
// one histogram vote per (thread, key); whole waves that agree on the digit
// (concentrated scores) add once
__device__ __forceinline__ void hist_vote(uint32_t* hist, bool in, uint32_t digit, int tid) {
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

// hist[256] holds the votes of one radix pass; finds the bin that contains the
// krem-th largest entry.  sh[0] = bin, sh[1] = rank still wanted inside the bin,
// sh[2] = 1 if the whole bin is wanted.  Must be called by all SEL_THREADS
// threads, after a barrier that made hist visible; ends with a barrier.
__device__ __forceinline__ void find_digit(const uint32_t* hist, uint32_t krem, uint32_t* sh,
                                           uint32_t* wtot, int tid) {
  uint32_t v = 0, own = 0;
  const int lane = tid & 63, wv = tid >> 6;
  if (tid < 256) {
    own = hist[tid];
    v = own;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = (uint32_t)__shfl_down((int)v, off, 64);
      if (lane + off < 64) v += t;
    }
    if (lane == 0) wtot[wv] = v;  // votes in bins [64*wv, 64*wv+63]
  }
  __syncthreads();
  if (tid < 256) {
    for (int w2 = wv + 1; w2 < 4; ++w2) v += wtot[w2];  // v = votes in bins >= tid
    const uint32_t above = v - own;
    if (v >= krem && above < krem) {
      sh[0] = (uint32_t)tid;
      sh[1] = krem - above;
      sh[2] = (own == krem - above) ? 1u : 0u;
    }
  }
  __syncthreads();
}

template <int MODE>
__device__ __forceinline__ void bitonic_desc(const SelParams& p, int64_t qbase, uint64_t* keys,
                                             uint32_t P, int tid) {
  for (uint32_t size = 2; size <= P; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      for (uint32_t t = tid; t < (P >> 1); t += SEL_THREADS) {
        const uint32_t i = 2 * t - (t & (stride - 1));
        const uint32_t j = i + stride;
        const bool desc = (i & size) == 0;
        const uint64_t a = keys[i], b = keys[j];
        const bool swap = desc ? key_before<MODE>(p, qbase, b, a) : key_before<MODE>(p, qbase, a, b);
        if (swap) { keys[i] = b; keys[j] = a; }
      }
      __syncthreads();
    }
  }
}

// Same network for P <= 1024 with one key per thread held in a register: partners
// closer than a wave (stride < 64) are exchanged with shuffles, so only the ten
// stages with stride >= 64 go through LDS and a barrier (instead of all 55).
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask, 64);
  const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask, 64);
  return ((uint64_t)hi << 32) | lo;
}

template <int MODE>
__device__ __forceinline__ void bitonic_desc_reg(const SelParams& p, int64_t qbase, uint64_t* keys,
                                                 uint32_t P, int tid) {
  // P <= SEL_THREADS; keys[0..P) in LDS on entry and on exit (sorted, best first)
  uint64_t mine = ((uint32_t)tid < P) ? keys[tid] : 0ull;
  for (uint32_t size = 2; size <= P; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      uint64_t other;
      if (stride < 64) {
        other = shfl_xor_u64(mine, (int)stride);
      } else {
        __syncthreads();
        if ((uint32_t)tid < P) keys[tid] = mine;
        __syncthreads();
        other = ((uint32_t)tid < P) ? keys[tid ^ stride] : 0ull;
      }
      const bool lower = (tid & stride) == 0;          // this thread keeps the element of the lower index
      const bool desc = (tid & size) == 0;
      // in a descending pair the lower index keeps the better key
      const bool want_better = (lower == desc);
      const bool other_better = key_before<MODE>(p, qbase, other, mine);
      if ((uint32_t)tid < P && (want_better == other_better)) mine = other;
    }
  }
  __syncthreads();
  if ((uint32_t)tid < P) keys[tid] = mine;
  __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(SEL_THREADS) void select_kernel(SelParams p, uint32_t lds_keys) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem);                 // [lds_keys]
  uint64_t* outk = keys + lds_keys;                                   // [SEL_OUT_CAP]
  uint64_t* ties = outk + SEL_OUT_CAP;                                // [SEL_TIE_CAP] (MERGE64)
  uint32_t* hist = reinterpret_cast<uint32_t*>(ties + (MODE == SEL_MERGE64 ? SEL_TIE_CAP : 0));
  uint32_t* sh = hist + 256;   // [8]
  uint32_t* wtot = sh + 8;     // [4]
  const int q = blockIdx.x;
  const int tid = threadIdx.x;
  const int64_t qbase = (int64_t)q * p.stride;

  uint32_t n = p.n;
  if (p.n_per_q) {
    const uint32_t c = p.n_per_q[q];
    n = c < p.n_cap ? c : p.n_cap;
    __syncthreads();   // every thread has read the count before it is given back
    if (tid == 0) {
      if (p.clear_counts) p.clear_counts[q] = 0u;
      uint32_t st = 0;
      if (c > p.n_cap) st |= TS_STATUS_OVERFLOW;
      if (c < p.need) st |= TS_STATUS_SHORT;
      if (st && p.status) atomicOr(p.status, st);
      if (p.host_report) {  // mapped pinned host memory: [0..63] counts, [64] status
        p.host_report[q] = c;
        if (st) atomicOr(&p.host_report[64], st);
      }
    }
  }
  const uint32_t kk = (uint32_t)p.k < n ? (uint32_t)p.k : n;

  // ---- phase 1: entries -> LDS (through a global radix select if too many)
  uint32_t count;
  if (n > lds_keys) {
    uint64_t prefix = 0ull;
    int bits_done = 0;
    uint32_t krem = kk;
    for (int pass = 0; pass < 8; ++pass) {
      const int shift = 56 - 8 * pass;
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      for (uint32_t base = 0; base < n; base += 4 * SEL_THREADS) {  // wave-uniform trip count
        uint64_t kx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t i = base + u * SEL_THREADS + tid;
          kx[u] = (i < n) ? load_key<MODE>(p, qbase, i) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint32_t i = base + u * SEL_THREADS + tid;
          const bool in = (i < n) && (bits_done == 0 || (kx[u] >> (64 - bits_done)) == prefix);
          hist_vote(hist, in, (uint32_t)(kx[u] >> shift) & 0xFFu, tid);
        }
      }
      __syncthreads();
      find_digit(hist, krem, sh, wtot, tid);
      prefix = (prefix << 8) | (uint64_t)sh[0];
      krem = sh[1];
      bits_done += 8;
      const bool done = sh[2] != 0;
      __syncthreads();
      if (done) break;
    }
    const uint64_t T = (bits_done == 64) ? prefix : (prefix << (64 - bits_done));
    if (tid == 0) sh[3] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += SEL_THREADS) {
      const uint64_t key = load_key<MODE>(p, qbase, i);
      if (key >= T && key != 0ull) {
        const uint32_t pos = atomicAdd(&sh[3], 1u);
        if (pos < lds_keys) keys[pos] = key;
      }
    }
    __syncthreads();
    count = sh[3] < lds_keys ? sh[3] : lds_keys;
  } else {
    for (uint32_t i = tid; i < n; i += SEL_THREADS) keys[i] = load_key<MODE>(p, qbase, i);
    count = n;
    __syncthreads();
  }

  // ---- phase 2: many more entries than wanted -> radix select inside LDS
  uint64_t* fin = keys;
  uint32_t pk = 2;
  while (pk < kk) pk <<= 1;
  if (kk >= 1 && pk <= SEL_OUT_CAP && count > 2 * pk) {
    // Scores that survived the filter share their sign, exponent and often the first
    // mantissa bits: skip the radix passes over the bytes every key has in common.
    uint64_t all_and = ~0ull, all_or = 0ull;
    for (uint32_t i = tid; i < count; i += SEL_THREADS) {
      const uint64_t key = keys[i];
      all_and &= key;
      all_or |= key;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      all_and &= shfl_xor_u64(all_and, off);
      all_or |= shfl_xor_u64(all_or, off);
    }
    if (tid == 0) { sh[4] = 0xFFFFFFFFu; sh[5] = 0xFFFFFFFFu; sh[6] = 0u; sh[7] = 0u; }
    __syncthreads();
    if ((tid & 63) == 0) {
      atomicAnd(&sh[4], (uint32_t)(all_and >> 32));
      atomicAnd(&sh[5], (uint32_t)all_and);
      atomicOr(&sh[6], (uint32_t)(all_or >> 32));
      atomicOr(&sh[7], (uint32_t)all_or);
    }
    __syncthreads();
    const uint64_t g_and = ((uint64_t)sh[4] << 32) | sh[5], g_or = ((uint64_t)sh[6] << 32) | sh[7];
    const uint64_t diff = g_and ^ g_or;
    int skip = diff ? (__builtin_clzll(diff) >> 3) : 7;   // whole bytes all keys agree on
    if (MODE == SEL_MERGE64 && skip > 3) skip = 3;        // keep the pass that ends the score bits
    __syncthreads();
    uint64_t prefix = skip ? (g_or >> (64 - 8 * skip)) : 0ull;
    int bits_done = 8 * skip;
    uint32_t krem = kk;
    for (int pass = skip; pass < 8; ++pass) {
      const int shift = 56 - 8 * pass;
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      for (uint32_t base = 0; base < count; base += SEL_THREADS) {
        const uint32_t i = base + tid;
        const uint64_t key = (i < count) ? keys[i] : 0ull;
        const bool in = (i < count) && (bits_done == 0 || (key >> (64 - bits_done)) == prefix);
        hist_vote(hist, in, (uint32_t)(key >> shift) & 0xFFu, tid);
      }
      __syncthreads();
      find_digit(hist, krem, sh, wtot, tid);
      prefix = (prefix << 8) | (uint64_t)sh[0];
      krem = sh[1];
      bits_done += 8;
      const bool done = sh[2] != 0;
      __syncthreads();
      if (done) break;
      if (MODE == SEL_MERGE64 && bits_done == 32) break;  // score decided; ties by real id below
    }
    const uint64_t T = (bits_done == 64) ? prefix : (prefix << (64 - bits_done));
    const bool whole = sh[2] != 0;  // every entry of the last bin is wanted: survivors = key >= T
    __syncthreads();
    if (tid == 0) { sh[3] = 0; sh[4] = 0; }
    __syncthreads();
    if constexpr (MODE == SEL_MERGE64) {
      // survivors = scores above the k-th score, plus, of the entries AT that score,
      // the ones with the smallest ids (the key's low half is only a position)
      const uint32_t sT = (uint32_t)(T >> 32);
      for (uint32_t i = tid; i < count; i += SEL_THREADS) {
        const uint64_t key = keys[i];
        if (key == 0ull) continue;
        const uint32_t sk = (uint32_t)(key >> 32);
        if (whole ? (key >= T) : (sk > sT)) {
          const uint32_t pos = atomicAdd(&sh[3], 1u);
          if (pos < SEL_OUT_CAP) outk[pos] = key;
        } else if (!whole && sk == sT) {
          const uint32_t pos = atomicAdd(&sh[4], 1u);
          if (pos < SEL_TIE_CAP) ties[pos] = key;
        }
      }
      __syncthreads();
      uint32_t c1 = sh[3], nt = sh[4];
      if (nt > SEL_TIE_CAP) nt = SEL_TIE_CAP;  // > 1024 exact ties at the boundary: by position
      if (nt > 0) {
        uint32_t pt = 2;
        while (pt < nt) pt <<= 1;
        for (uint32_t i = nt + tid; i < pt; i += SEL_THREADS) ties[i] = 0ull;
        __syncthreads();
        bitonic_desc<MODE>(p, qbase, ties, pt, tid);
        const uint32_t want = kk > c1 ? kk - c1 : 0;
        for (uint32_t i = tid; i < nt && i < want; i += SEL_THREADS)
          if (c1 + i < SEL_OUT_CAP) outk[c1 + i] = ties[i];
        c1 += (want < nt ? want : nt);
      }
      count = c1 < SEL_OUT_CAP ? c1 : SEL_OUT_CAP;
    } else {
      for (uint32_t i = tid; i < count; i += SEL_THREADS) {
        const uint64_t key = keys[i];
        if (key >= T && key != 0ull) {
          const uint32_t pos = atomicAdd(&sh[3], 1u);
          if (pos < SEL_OUT_CAP) outk[pos] = key;
        }
      }
      __syncthreads();
      count = sh[3] < SEL_OUT_CAP ? sh[3] : SEL_OUT_CAP;
    }
    fin = outk;
    __syncthreads();
  }

  // ---- phase 3: bitonic sort of the remaining entries, best first
  uint32_t P = 2;
  while (P < count) P <<= 1;
  for (uint32_t i = count + tid; i < P; i += SEL_THREADS) fin[i] = 0ull;
  __syncthreads();
  if (P <= SEL_THREADS) bitonic_desc_reg<MODE>(p, qbase, fin, P, tid);
  else bitonic_desc<MODE>(p, qbase, fin, P, tid);

  // ---- output
  float* os = p.out_scores + (int64_t)q * p.out_stride;
  for (uint32_t i = tid; i < (uint32_t)p.k; i += SEL_THREADS) {
    const uint64_t key = (i < count) ? fin[i] : 0ull;
    float s = NEG_MAX;
    int64_t id = -1;
    if (i < kk && key != 0ull) {
      s = key2f((uint32_t)(key >> 32));
      const uint32_t tb = 0xFFFFFFFFu - (uint32_t)key;
      if constexpr (MODE == SEL_MERGE64) id = p.ids64[id_addr(p, qbase, tb)];
      else id = (int64_t)tb;
    }
    os[i] = s;
    if (p.out_ids64)
      p.out_ids64[(int64_t)q * p.out_stride + i] =
          (id < 0) ? -1 : (MODE == SEL_MERGE64 ? id : id + p.id_offset);
    if (p.out_ids32) p.out_ids32[(int64_t)q * p.out_stride + i] = (int32_t)id;
  }
}

template <int MODE>
static int launch_select_t(const SelParams& p, int nq, hipStream_t stream) {
  // LDS: enough 64-bit keys for the entries (or for k when radix-selecting in global)
  uint32_t nmax = p.n_per_q ? p.n_cap : p.n;
  uint32_t need = nmax;
  const uint32_t lds_cap = TS_SEL_LDS_KEYS;
  if (need > lds_cap) need = lds_cap;
  uint32_t lds_keys = 2;
  while (lds_keys < need) lds_keys <<= 1;
  if (nmax > lds_cap) {
    uint32_t kk = (uint32_t)p.k;
    if (kk > TS_SEL_LDS_KEYS) {
      ts_set_error("k=%d exceeds the supported maximum %d for %u entries", p.k, TS_SEL_LDS_KEYS,
                   nmax);
      return TS_ERR_UNSUPPORTED;
    }
    lds_keys = 2;
    while (lds_keys < kk) lds_keys <<= 1;
  }
  const size_t lds = ((size_t)lds_keys + SEL_OUT_CAP + (MODE == SEL_MERGE64 ? SEL_TIE_CAP : 0)) * 8 +
                     (256 + 8 + 4) * 4;
  auto kern = select_kernel<MODE>;
  static TsDeviceOnce lds_attr;  // per instantiation, per device (ts_common.h)
  TS_CHECK(ts_allow_max_lds(lds_attr, reinterpret_cast<const void*>(kern)));
  hipLaunchKernelGGL(kern, dim3(nq), dim3(SEL_THREADS), lds, stream, p, lds_keys);
  TS_HIP(hipGetLastError());
  return TS_OK;
}

int ts_launch_select(const SelParams& p, int nq, hipStream_t stream) {
  if (nq <= 0 || p.k <= 0) return TS_OK;
  switch (p.mode) {
    case SEL_DENSE: return launch_select_t<SEL_DENSE>(p, nq, stream);
    case SEL_PAIRS32: return launch_select_t<SEL_PAIRS32>(p, nq, stream);
    case SEL_MERGE64: return launch_select_t<SEL_MERGE64>(p, nq, stream);
  }
  ts_set_error("bad select mode %d", p.mode);
  return TS_ERR_INVALID;
}

// ------------------------------------------------------------------ tau
// Per-query threshold from the dense sample scores: roughly the m-th largest.
// Every thread keeps the KEEP largest keys of its share; the m-th largest of
// those 1024*KEEP survivors is a LOWER bound of the exact m-th largest (large
// values are only lost when one thread sees more than KEEP of them), which is
// the safe direction: a lower threshold admits more candidates, never fewer.
// The exactness of the search never depends on this value (ts_index.hip
// verifies the candidate counts).
template <int KEEP>
__global__ __launch_bounds__(SEL_THREADS) void tau_kernel(const float* sample, int64_t ld,
                                                          uint32_t n, uint32_t m, int nq,
                                                          float* tau) {
  __shared__ __attribute__((aligned(16))) uint32_t keys[KEEP * SEL_THREADS];
  const int q = blockIdx.x;
  const int tid = threadIdx.x;
  if (q >= nq) {
    if (tid == 0) tau[q] = 3.402823466e38f;
    return;
  }
  const float* s = sample + (int64_t)q * ld;
  const bool dbg_inf = (n == 0xFFFFFFFFu);  // tuning experiments: nothing passes
  if (dbg_inf) n = 0;
  uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  auto take = [&](float f) {
    const uint32_t k = f2key(f);
    if constexpr (KEEP == 1) {
      a0 = k > a0 ? k : a0;
    } else {
      if (k > a3) {
        if (k > a0) { a3 = a2; a2 = a1; a1 = a0; a0 = k; }
        else if (k > a1) { a3 = a2; a2 = a1; a1 = k; }
        else if (k > a2) { a3 = a2; a2 = k; }
        else a3 = k;
      }
    }
  };
  const uint32_t n4 = ((ld & 3) == 0) ? (n & ~3u) : 0u;  // float4 part (rows are 16-byte aligned)
  // 8 independent 16-byte loads in flight per thread: one workgroup per query
  // means this loop is latency-bound, not bandwidth-bound
  for (uint32_t i0 = 0; i0 < n4; i0 += 32 * SEL_THREADS) {
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t i = i0 + 4 * (tid + j * SEL_THREADS);
      v[j] = (i < n4) ? *reinterpret_cast<const float4*>(s + i)
                      : make_float4(NEG_MAX, NEG_MAX, NEG_MAX, NEG_MAX);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t i = i0 + 4 * (tid + j * SEL_THREADS);
      if (i < n4) { take(v[j].x); take(v[j].y); take(v[j].z); take(v[j].w); }
    }
  }
  for (uint32_t i = n4 + tid; i < n; i += SEL_THREADS) take(s[i]);

  keys[tid] = a0;
  if constexpr (KEEP == 4) {
    keys[SEL_THREADS + tid] = a1;
    keys[2 * SEL_THREADS + tid] = a2;
    keys[3 * SEL_THREADS + tid] = a3;
  }
  __syncthreads();
  constexpr uint32_t P = KEEP * SEL_THREADS;
  uint32_t idx = m ? m - 1 : 0;
  if (idx >= P) idx = P - 1;
  if constexpr (KEEP == 1) {
    // one key per thread: shuffles for stride < 64, LDS only for the ten wider stages
    uint32_t mine = a0;
    for (uint32_t size = 2; size <= P; size <<= 1) {
      for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
        uint32_t other;
        if (stride < 64) {
          other = (uint32_t)__shfl_xor((int)mine, (int)stride, 64);
        } else {
          __syncthreads();
          keys[tid] = mine;
          __syncthreads();
          other = keys[tid ^ stride];
        }
        const bool want_larger = (((tid & stride) == 0) == ((tid & size) == 0));
        if (want_larger == (other > mine)) mine = other;
      }
    }
    __syncthreads();
    keys[tid] = mine;
    __syncthreads();
  } else {
    for (uint32_t size = 2; size <= P; size <<= 1) {
      for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
        for (uint32_t t = tid; t < (P >> 1); t += SEL_THREADS) {
          const uint32_t i = 2 * t - (t & (stride - 1));
          const uint32_t j = i + stride;
          const bool desc = (i & size) == 0;
          const uint32_t a = keys[i], b = keys[j];
          if (desc ? (b > a) : (a > b)) { keys[i] = b; keys[j] = a; }
        }
        __syncthreads();
      }
    }
}
  if (tid == 0) {
    const uint32_t k = keys[idx];
    tau[q] = dbg_inf ? 3.402823466e38f : ((k == 0u) ? NEG_MAX : key2f(k));
  }
}

int ts_launch_tau(const float* sample, int64_t ld, uint32_t n, uint32_t m, int nq, float* tau,
                  hipStream_t stream) {
  // one key per thread is enough while the wanted rank is far below 1024
  if (m <= 64)
    hipLaunchKernelGGL(tau_kernel<1>, dim3(TS_MAX_Q), dim3(SEL_THREADS), 0, stream, sample, ld, n,
                       m, nq, tau);
  else
    hipLaunchKernelGGL(tau_kernel<4>, dim3(TS_MAX_Q), dim3(SEL_THREADS), 0, stream, sample, ld, n,
                       m, nq, tau);
  TS_HIP(hipGetLastError());
  return TS_OK;
}
