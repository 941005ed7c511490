// Device code shared by the scan kernels (ts_scan.hip: dense / filter scans; ts_fused.hip: the
// one-launch search).  Not part of the public ABI.
#pragma once
#include "ts_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef SCAN_THREADS
#define SCAN_THREADS 512
#endif
#define SCAN_WAVES (SCAN_THREADS / 64)

__device__ __forceinline__ u32x4 stream_load(const u32x4* p) {
#ifndef TS_PLAIN_LOADS  // non-temporal: the corpus is read once per batch, keep it out of L2/MALL
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

template <int DT>
__device__ __forceinline__ void mma_group(f32x16& acc, const u32x4& a,
                                          const u32x4& b) {
#if defined(TS_TUNING) && defined(DBG_NO_MFMA)  // ablation builds only: keep the operands live, skip the matrix op
  acc[0] += __uint_as_float((a[0] ^ b[0]) & 0x007fffffu);
  acc[1] += __uint_as_float((a[1] ^ b[1]) & 0x007fffffu);
  acc[2] += __uint_as_float((a[2] ^ b[2]) & 0x007fffffu);
  acc[3] += __uint_as_float((a[3] ^ b[3]) & 0x007fffffu);
  return;
#endif
  if constexpr (DT == TS_F16) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(
        __builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), acc, 0, 0, 0);
  } else if constexpr (DT == TS_BF16) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
        __builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc, 0, 0, 0);
  } else {
    const f32x4 af = __builtin_bit_cast(f32x4, a);
    const f32x4 bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], acc, 0, 0, 0);
  }
}

// row inside the 32-row block held by accumulator register r of this lane
// (C/D map of the 32x32 MFMA shapes: col = lane & 31, row below)
__device__ __forceinline__ int acc_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

__device__ __forceinline__ float acc_max(const f32x16& a) {
  float m0 = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
  float m1 = fmaxf(fmaxf(a[4], a[5]), fmaxf(a[6], a[7]));
  float m2 = fmaxf(fmaxf(a[8], a[9]), fmaxf(a[10], a[11]));
  float m3 = fmaxf(fmaxf(a[12], a[13]), fmaxf(a[14], a[15]));
  return fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
}

template <int QH>
__device__ __forceinline__ void epilogue_dense(const ScanParams& p,
                                               const f32x16 (&acc)[QH],
                                               int64_t w, int64_t blk,
                                               int lane) {
  const int j = lane & 31;
  const int64_t row_base = blk * TS_ROWS_PER_BLOCK;
#pragma unroll
  for (int hq = 0; hq < QH; ++hq) {
    const int q = hq * 32 + j;
    if (q < p.nq) {
      float* dst = p.dense + (int64_t)q * p.dense_ld + w * TS_ROWS_PER_BLOCK;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int i0 = 8 * r4 + 4 * (lane >> 5);
        float4 v;
        v.x = (row_base + i0 + 0 < p.ntotal) ? acc[hq][4 * r4 + 0] : -3.402823466e38f;
        v.y = (row_base + i0 + 1 < p.ntotal) ? acc[hq][4 * r4 + 1] : -3.402823466e38f;
        v.z = (row_base + i0 + 2 < p.ntotal) ? acc[hq][4 * r4 + 2] : -3.402823466e38f;
        v.w = (row_base + i0 + 3 < p.ntotal) ? acc[hq][4 * r4 + 3] : -3.402823466e38f;
        *reinterpret_cast<float4*>(dst + i0) = v;
      }
    }
  }
}

// Survivors of the threshold test are parked in LDS and written to the
// per-query candidate lists once, by the whole workgroup, when it has finished
// streaming.  (Appending from the hot loop with returning global atomics costs
// 27 % of the kernel at 10M x 768: each append stalls its wave for microseconds
// behind the HBM stream — measured, DESIGN.md "candidate staging".)
#define STAGE_CAP 2048
template <int CAP>
struct StageLdsT {
  static constexpr uint32_t kCap = CAP;
  uint32_t cnt;
  uint32_t tau_flag;        // ts_fused.hip: wave 0 has published this launch's thresholds in tauv[]
  uint32_t arrived;         // ts_fused.hip: waves of this workgroup that have delivered their sample
  uint32_t pad[1];
  float tauv[TS_MAX_Q];     // ts_fused.hip: the thresholds, once tau_flag is set
  uint32_t qcnt[TS_MAX_Q];
  uint32_t qbase[TS_MAX_Q];
  uint32_t qoff[TS_MAX_Q];
  float score[CAP];
  int32_t id[CAP];
  uint8_t q[CAP];
};
typedef StageLdsT<STAGE_CAP> StageLds;

template <int QH, class SL>
__device__ __forceinline__ void epilogue_filter(const ScanParams& p, SL* st,
                                                const f32x16 (&acc)[QH],
                                                const float (&tau)[QH],
                                                int64_t blk, int lane) {
  bool hit = false;
#pragma unroll
  for (int hq = 0; hq < QH; ++hq) hit |= (acc_max(acc[hq]) >= tau[hq]);
  if (__builtin_amdgcn_ballot_w64(hit) == 0ull) return;  // the common case
  // Rare path: some lane holds at least one surviving score.
  const int64_t row_base = blk * TS_ROWS_PER_BLOCK;
#pragma unroll
  for (int hq = 0; hq < QH; ++hq) {
    const int q = hq * 32 + (lane & 31);
    uint32_t mask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool ok = (acc[hq][r] >= tau[hq]) &&
                      (row_base + acc_row(r, lane) < p.ntotal);
      mask |= ok ? (1u << r) : 0u;
    }
    if (mask) {
      uint32_t slot = atomicAdd(&st->cnt, (uint32_t)__builtin_popcount(mask));  // LDS
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (mask & (1u << r)) {
          const int32_t id = (int32_t)(row_base + acc_row(r, lane));
          if (slot < SL::kCap) {
            st->score[slot] = acc[hq][r];
            st->id[slot] = id;
            st->q[slot] = (uint8_t)q;
          } else {
            // staging area full (heavily clustered hits): append directly
            const uint32_t g = atomicAdd(&p.cand_cnt[q], 1u);
            if (g < p.cand_cap) {
              p.cand_score[(size_t)q * p.cand_cap + g] = acc[hq][r];
              p.cand_id[(size_t)q * p.cand_cap + g] = id;
            }
          }
          ++slot;
        }
      }
    }
  }
}

// Workgroup-wide: move the staged survivors to the per-query lists.  One global
// atomic per (workgroup, query) reserves the slots.
template <class SL>
__device__ __forceinline__ void flush_stage(const ScanParams& p, SL* st, int tid) {
  __syncthreads();
  const uint32_t n = st->cnt < SL::kCap ? st->cnt : SL::kCap;
  if (n == 0) return;  // uniform: cnt is final after the barrier
  if (tid < TS_MAX_Q) { st->qcnt[tid] = 0; st->qoff[tid] = 0; }
  __syncthreads();
  for (uint32_t e = tid; e < n; e += SCAN_THREADS) atomicAdd(&st->qcnt[st->q[e]], 1u);
  __syncthreads();
  if (tid < TS_MAX_Q && st->qcnt[tid] > 0)
    st->qbase[tid] = atomicAdd(&p.cand_cnt[tid], st->qcnt[tid]);
  __syncthreads();
  for (uint32_t e = tid; e < n; e += SCAN_THREADS) {
    const uint32_t q = st->q[e];
    const uint32_t slot = st->qbase[q] + atomicAdd(&st->qoff[q], 1u);
    if (slot < p.cand_cap) {
      p.cand_score[(size_t)q * p.cand_cap + slot] = st->score[e];
      p.cand_id[(size_t)q * p.cand_cap + slot] = st->id[e];
    }
  }
}


// ------------------------------------------------------------------ layout
template <typename T> struct ElemIO;
template <> struct ElemIO<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
};
template <> struct ElemIO<_Float16> {
  static __device__ __forceinline__ float ld(const _Float16* p) { return (float)*p; }
};
template <> struct ElemIO<__bf16> {
  static __device__ __forceinline__ float ld(const __bf16* p) {
    uint32_t u = (uint32_t)(*reinterpret_cast<const uint16_t*>(p)) << 16;
    return __builtin_bit_cast(float, u);
  }
};

// 8 consecutive elements as floats (p 16-byte aligned for 16-bit types, 32 for float rows
// whose dim is a multiple of 8)
template <typename T> __device__ __forceinline__ void ld8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void ld8<float>(const float* p, float (&v)[8]) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void ld8<_Float16>(const _Float16* p, float (&v)[8]) {
  const h8 x = *reinterpret_cast<const h8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)x[i];
}
template <> __device__ __forceinline__ void ld8<__bf16>(const __bf16* p, float (&v)[8]) {
  const uint4 x = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(x.x << 16); v[1] = __uint_as_float(x.x & 0xffff0000u);
  v[2] = __uint_as_float(x.y << 16); v[3] = __uint_as_float(x.y & 0xffff0000u);
  v[4] = __uint_as_float(x.z << 16); v[5] = __uint_as_float(x.z & 0xffff0000u);
  v[6] = __uint_as_float(x.w << 16); v[7] = __uint_as_float(x.w & 0xffff0000u);
}

__device__ __forceinline__ uint16_t f32_to_storage16(float f, int dt) {
  if (dt == TS_F16) {
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
  }
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}

// k index of element e (0..epl-1) of lane half h in group g
__device__ __forceinline__ int frag_k(int dt, int g, int h, int e) {
  // f32: two consecutive units (g even, g odd) hold 8 consecutive k of a lane's row — k = 16(g/2) + 8h + 4(g&1) + e —
  // so that a pair of units is the A operand of one 16-wide k step of the bf16x3 split scan (ts_scan_f32s.hip);
  // the exact-f32 MFMA path does not care about the order (the query image uses the same map)
  return (dt == TS_F32) ? (16 * (g >> 1) + 8 * h + 4 * (g & 1) + e) : (16 * g + 8 * h + e);
}

