// fp32-storage scan on the 16-bit matrix cores: the bf16x3 split (gfx950).
//
// The reference keeps its corpus in fp32 (faiss.IndexFlatIP, reference src/stage1_retriever.py:263-277) and
// TS_F32 storage keeps those 32 bits.  The exact-f32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate:
// at d = 768 it is a co-bound of the scan (5.5 TB/s = 69 % of HBM peak where the 16-bit scans reach 87 %).  Here
// every fp32 value is split in registers into three bf16 terms
//       x = hi + mid + lo          hi = trunc16(x), mid = trunc16(x - hi), lo = trunc16(x - hi - mid)
// (8 + 8 + 8 significand bits: EXACT for normal numbers, the subtractions are exact in fp32), the query image
// holds the same three terms of every query, and a k step of 16 is six v_mfma_f32_32x32x16_bf16
//       ah*bh + ah*bm + am*bh   (one accumulator)      al*bh + ah*bl + am*bm   (a second one, added at the end)
// (products of bf16 values are exact; fp32 accumulation).  The three dropped terms (am*bl, al*bm, al*bl) are
// below 2^-24 |a||b| each: for unit-norm rows the score error they add is < 2e-7, inside the float64 near-tie
// rule of the parity tests (2e-6).  Matrix work per k step: 6 x 32 cycles instead of 8 x 64; the split costs
// ~44 VALU instructions per k step (and / sub / and / sub per value, v_perm_b32 to pack two upper halves).
// Measured at 10 M x 768 (32 queries per pass): 5.59 ms per pass with the exact-f32 MFMA (5.5 TB/s = 69 %),
// 4.99 ms with one accumulator chain and shift+or packing (6.15 TB/s), 4.59 ms as it stands (6.69 TB/s = 84 %).
//
// Scope: 32 queries per pass (three 16-bit query images of 32 queries are 144 KiB at d = 768) and only where the
// exact-f32 path could not take 64 queries per pass anyway (512 < d <= 768 — which includes the headline
// dimension); elsewhere TS_F32 keeps the exact-f32 MFMA kernel of ts_scan.hip.  The corpus layout is the same
// for both (ts_scan_dev.h frag_k: two consecutive 1 KiB units of a row block hold 8 consecutive k per lane).
#include "ts_scan_dev.h"

#define F32S_STAGE_CAP 1024   // (32 queries per pass: half the survivors per workgroup of a 64-query pass)
typedef StageLdsT<F32S_STAGE_CAP> StageLdsS;

__device__ __forceinline__ uint32_t f32s_trunc(float x) { return __builtin_bit_cast(uint32_t, x) & 0xFFFF0000u; }

// eight fp32 values (two ring units of one lane) -> their hi / mid / lo bf16 vectors (element j in bits 16(j&1))
__device__ __forceinline__ void f32s_split8(const u32x4& a0, const u32x4& a1, u32x4& hi, u32x4& mid, u32x4& lo) {
  const f32x4 f0 = __builtin_bit_cast(f32x4, a0), f1 = __builtin_bit_cast(f32x4, a1);
  uint32_t h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = j < 4 ? f0[j & 3] : f1[j & 3];
    h[j] = f32s_trunc(x);
    const float r1 = x - __builtin_bit_cast(float, h[j]);
    m[j] = f32s_trunc(r1);
    l[j] = __builtin_bit_cast(uint32_t, r1 - __builtin_bit_cast(float, m[j]));   // (its upper half is taken below)
  }
  // two upper halves -> one dword of two bf16 (v_perm_b32: bytes 2,3 of the even element, bytes 2,3 of the odd one)
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    hi[d] = __builtin_amdgcn_perm(h[2 * d + 1], h[2 * d], 0x07060302u);
    mid[d] = __builtin_amdgcn_perm(m[2 * d + 1], m[2 * d], 0x07060302u);
    lo[d] = __builtin_amdgcn_perm(l[2 * d + 1], l[2 * d], 0x07060302u);
  }
}

__device__ __forceinline__ void f32s_mma(f32x16& acc, const u32x4& a, const u32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc, 0, 0, 0);
}

// one k step of 16: ring units (a0, a1) of this lane's row against query-image group G (three terms).  Two
// accumulators — the small cross terms and the large ones — so that consecutive MFMAs are not one dependency
// chain (a wave issues in order: six dependent 32-cycle MFMAs stall it for ~190 cycles per k step).
__device__ __forceinline__ void f32s_step(f32x16& acc, f32x16& acc_small, const u32x4& a0, const u32x4& a1,
                                          const u32x4* ql, int G) {
  u32x4 ah, am, al;
  f32s_split8(a0, a1, ah, am, al);
  const u32x4 bh = ql[(size_t)(G * 3 + 0) * 64], bm = ql[(size_t)(G * 3 + 1) * 64], bl = ql[(size_t)(G * 3 + 2) * 64];
  f32s_mma(acc, ah, bh);
  f32s_mma(acc_small, al, bh);
  f32s_mma(acc, ah, bm);
  f32s_mma(acc_small, ah, bl);
  f32s_mma(acc, am, bh);
  f32s_mma(acc_small, am, bm);
}

template <int MODE>
__global__ __launch_bounds__(SCAN_THREADS) void scan_f32s_kernel(ScanParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* qlds = reinterpret_cast<u32x4*>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int kg = p.kg;            // 1 KiB units per row block (even)
  const int ng = kg >> 1;         // k steps of 16

  const int64_t nwaves = (int64_t)gridDim.x * SCAN_WAVES;
  int64_t w = (int64_t)blockIdx.x * SCAN_WAVES + wave;
  const bool active = w < p.nwork;
  const u32x4* base = reinterpret_cast<const u32x4*>(p.corpus) + lane;
  const size_t blk_units = (size_t)kg * 64;
  int64_t blk = active ? p.blk0 + w * p.blk_stride : p.blk0;
  const u32x4* cur = base + (size_t)blk * blk_units;
  u32x4 ring[TS_RING];
  if (active) {
#pragma unroll
    for (int i = 0; i < TS_RING; ++i) ring[i] = stream_load(cur + (size_t)i * 64);
  }
  // ---- prologue: the three-term query image global(L2) -> LDS, once per workgroup
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(p.qimg);
    const int units = ng * 3 * 64;
    for (int i0 = tid; i0 < units; i0 += 8 * SCAN_THREADS) {
      u32x4 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + j * SCAN_THREADS;
        t[j] = (i < units) ? src[i] : u32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + j * SCAN_THREADS;
        if (i < units) qlds[i] = t[j];
      }
    }
  }
  StageLdsS* st = reinterpret_cast<StageLdsS*>(smem + (size_t)ng * 3 * 1024);
  if constexpr (MODE == SCAN_FILTER) {
    if (tid == 0) st->cnt = 0;
  }
  __syncthreads();

  if (active) {
    float tau[1];
    if constexpr (MODE == SCAN_FILTER) tau[0] = p.tau[lane & 31];
    const u32x4* ql = qlds + lane;
    while (true) {
      const int64_t wn = w + nwaves;
      const bool has_next = wn < p.nwork;
      const int64_t blkn = has_next ? (p.blk0 + wn * p.blk_stride) : blk;
      const u32x4* nxt = base + (size_t)blkn * blk_units;
      f32x16 acc[1], acc_small;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc_small[r] = 0.f; }
      int g0 = 0;
      for (; g0 < kg - TS_RING; g0 += TS_RING) {
#pragma unroll
        for (int i = 0; i < TS_RING; i += 2) {
          f32s_step(acc[0], acc_small, ring[i], ring[i + 1], ql, (g0 + i) >> 1);
          ring[i] = stream_load(cur + (size_t)(g0 + i + TS_RING) * 64);
          ring[i + 1] = stream_load(cur + (size_t)(g0 + i + 1 + TS_RING) * 64);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int i = 0; i < TS_RING; i += 2) {   // tail: the ring is refilled from the wave's next row block
        f32s_step(acc[0], acc_small, ring[i], ring[i + 1], ql, (g0 + i) >> 1);
        ring[i] = stream_load(nxt + (size_t)i * 64);
        ring[i + 1] = stream_load(nxt + (size_t)(i + 1) * 64);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][r] += acc_small[r];
      if constexpr (MODE == SCAN_DENSE)
        epilogue_dense<1>(p, acc, w, blk, lane);
      else
        epilogue_filter<1>(p, st, acc, tau, blk, lane);
      if (!has_next) break;
      w = wn;
      blk = blkn;
      cur = nxt;
    }
  }
  if constexpr (MODE == SCAN_FILTER) flush_stage(p, st, tid);
}

// LDS of the split scan for 32 queries; 0 if this layout is not one it takes
size_t ts_scan_f32s_lds_bytes(const TsLayout& L) {
  if (L.dtype != TS_F32 || (L.kg & 1)) return 0;
  return (size_t)(L.kg / 2) * 3 * 1024 + sizeof(StageLdsS);
}

int ts_launch_scan_f32s(const TsLayout& L, int mode, const ScanParams& p, int num_cus, hipStream_t stream) {
  const size_t lds = (size_t)(L.kg / 2) * 3 * 1024 + (mode == SCAN_FILTER ? sizeof(StageLdsS) : 0);
  int64_t want = (p.nwork + SCAN_WAVES - 1) / SCAN_WAVES;
  int grid = (int)(want < num_cus ? want : num_cus);
  if (grid < 1) grid = 1;
  if (mode == SCAN_DENSE) {
    auto kern = scan_f32s_kernel<SCAN_DENSE>;
    static TsDeviceOnce lds_attr;
    TS_CHECK(ts_allow_max_lds(lds_attr, reinterpret_cast<const void*>(kern)));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SCAN_THREADS), lds, stream, p);
  } else {
    auto kern = scan_f32s_kernel<SCAN_FILTER>;
    static TsDeviceOnce lds_attr;
    TS_CHECK(ts_allow_max_lds(lds_attr, reinterpret_cast<const void*>(kern)));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SCAN_THREADS), lds, stream, p);
  }
  TS_HIP(hipGetLastError());
  return TS_OK;
}

// ---- query image: thread per 16-byte unit; unit (G*3 + term)*64 + l = term `term` of query (l & 31), k = 16G + 8(l>>5) + 0..7
template <typename TIN>
__global__ void qprep_f32s_kernel(const TIN* q, int nq, int dim, int ng, uint4* qimg, uint32_t* cand_cnt, uint32_t* status) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < TS_MAX_Q && cand_cnt) cand_cnt[t] = 0;
  if (t == 0 && status) status[0] = 0;
  if (t >= ng * 3 * 64) return;
  const int l = t & 63;
  const int term = (t >> 6) % 3;
  const int G = (t >> 6) / 3;
  const int qi = l & 31, h = l >> 5;
  const TIN* src = q + (int64_t)qi * dim;
  uint32_t b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 16 * G + 8 * h + e;
    const float x = (qi < nq && k < dim) ? ElemIO<TIN>::ld(src + k) : 0.f;
    const uint32_t hb = f32s_trunc(x);
    const float r1 = x - __builtin_bit_cast(float, hb);
    const uint32_t mb = f32s_trunc(r1);
    const float r2 = r1 - __builtin_bit_cast(float, mb);
    b[e] = term == 0 ? hb : (term == 1 ? mb : f32s_trunc(r2));
  }
  u32x4 out;
#pragma unroll
  for (int d = 0; d < 4; ++d) out[d] = (b[2 * d] >> 16) | b[2 * d + 1];
  reinterpret_cast<u32x4*>(qimg)[t] = out;
}

int ts_launch_qprep_f32s(const TsLayout& L, const void* q, int q_dtype, int nq, uint4* qimg, uint32_t* cand_cnt,
                         uint32_t* status, hipStream_t stream) {
  const int ng = L.kg / 2;
  const int units = ng * 3 * 64;
  const int blocks = (units + 255) / 256;
  switch (q_dtype) {
    case TS_F32:
      hipLaunchKernelGGL(qprep_f32s_kernel<float>, dim3(blocks), dim3(256), 0, stream, (const float*)q, nq, L.dim, ng, qimg, cand_cnt, status);
      break;
    case TS_F16:
      hipLaunchKernelGGL(qprep_f32s_kernel<_Float16>, dim3(blocks), dim3(256), 0, stream, (const _Float16*)q, nq, L.dim, ng, qimg, cand_cnt, status);
      break;
    case TS_BF16:
      hipLaunchKernelGGL(qprep_f32s_kernel<__bf16>, dim3(blocks), dim3(256), 0, stream, (const __bf16*)q, nq, L.dim, ng, qimg, cand_cnt, status);
      break;
    default:
      ts_set_error("bad query dtype %d", q_dtype);
      return TS_ERR_INVALID;
  }
  TS_HIP(hipGetLastError());
  return TS_OK;
}
