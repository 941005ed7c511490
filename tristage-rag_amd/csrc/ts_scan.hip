// Stage-1 brute-force inner-product scan for gfx950 (MI355X, CDNA4).
//
// Replaces the arithmetic the reference reaches through
// faiss.IndexFlatIP.add / .search (reference src/stage1_retriever.py:263-277,
// 313, 380): every query of a batch (<= 64) against every corpus row.
//
// Design:
//   * HBM-bound streaming kernel.  The corpus is stored pre-tiled in MFMA
//     A-fragment order (ts_common.h), so each wave-level load is one
//     contiguous 1 KiB global_load_dwordx4 straight into VGPRs; no LDS round
//     trip and no barrier for the streamed operand.
//   * The query batch (the stationary operand, <= 128 KiB) is staged once per
//     workgroup into LDS in B-fragment order; every ds_read_b128 is
//     lane-linear, hence bank-conflict free.
//   * Each wave owns a 32-row x 64-query score tile held in two 32x32 MFMA
//     accumulators; a TS_RING-deep register ring keeps 8 KiB of corpus loads in
//     flight per wave across row-block boundaries (persistent waves).
//   * Epilogue, dense mode: scores are written out (small corpora, the sample
//     that seeds the thresholds, fallback).  Filter mode: a score survives
//     only if it is >= the per-query threshold; survivors (a few thousand per
//     query out of millions) are appended to per-query candidate lists, so the
//     B x N score matrix is never materialised.
#include "ts_scan_dev.h"
#include <stdlib.h>

// One persistent wave streams row blocks gw, gw+W, gw+2W, ... (W = waves in
// the grid).  No barrier after the prologue.
template <int DT, int QH, int MODE>
__global__ __launch_bounds__(SCAN_THREADS) void scan_kernel(ScanParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* qlds = reinterpret_cast<u32x4*>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int kg = p.kg;

  // ---- the first corpus loads go out before anything else, so the HBM round trip
  // overlaps the Q-image copy below
  const int64_t nwaves = (int64_t)gridDim.x * SCAN_WAVES;
  int64_t w = (int64_t)blockIdx.x * SCAN_WAVES + wave;
  const bool active = w < p.nwork;  // (waves without work still join the final flush)
  const u32x4* base = reinterpret_cast<const u32x4*>(p.corpus) + lane;
  const size_t blk_units = (size_t)kg * 64;
  int64_t blk = active ? p.blk0 + w * p.blk_stride : p.blk0;
  const u32x4* cur = base + (size_t)blk * blk_units;
  u32x4 ring[TS_RING];
  if (active) {
#pragma unroll
    for (int i = 0; i < TS_RING; ++i) ring[i] = stream_load(cur + (size_t)i * 64);
  }

  // ---- prologue: Q image global(L2) -> LDS, once per workgroup
  {
    const u32x4* src = reinterpret_cast<const u32x4*>(p.qimg);
    const int units = kg * QH * 64;
    // 8 independent loads in flight per thread (a one-load-at-a-time copy of the
    // 96 KiB image costs ~18 us of serial L2 latency per workgroup)
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
  StageLds* st = reinterpret_cast<StageLds*>(smem + (size_t)kg * QH * 1024);
  if constexpr (MODE == SCAN_FILTER) {
    if (tid == 0) st->cnt = 0;
  }
  __syncthreads();

  if (active) {

  float tau[QH];
  if constexpr (MODE == SCAN_FILTER) {
#pragma unroll
    for (int hq = 0; hq < QH; ++hq) tau[hq] = p.tau[hq * 32 + (lane & 31)];
  }

  const u32x4* ql = qlds + lane;

  while (true) {
    const int64_t wn = w + nwaves;
    const bool has_next = wn < p.nwork;
    const int64_t blkn = has_next ? (p.blk0 + wn * p.blk_stride) : blk;
    const u32x4* nxt = base + (size_t)blkn * blk_units;

    f32x16 acc[QH];
#pragma unroll
    for (int hq = 0; hq < QH; ++hq)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[hq][r] = 0.f;

    // main part: prefetch stays inside the current row block
    int g0 = 0;
    for (; g0 < kg - TS_RING; g0 += TS_RING) {
#pragma unroll
      for (int i = 0; i < TS_RING; ++i) {
#pragma unroll
        for (int hq = 0; hq < QH; ++hq) {
#if defined(TS_TUNING) && defined(DBG_NO_LDS)  // ablation builds only
          const u32x4 b = ring[(i + 1) % TS_RING];
#else
          const u32x4 b = ql[(size_t)((g0 + i) * QH + hq) * 64];
#endif
          mma_group<DT>(acc[hq], ring[i], b);
        }
        // refill the slot just consumed; the barrier keeps the compiler from
        // clustering the ring's loads (which would drain vmcnt to 0 mid-loop)
        ring[i] = stream_load(cur + (size_t)(g0 + i + TS_RING) * 64);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // tail: the ring is refilled from the start of the wave's next row block
#pragma unroll
    for (int i = 0; i < TS_RING; ++i) {
#pragma unroll
      for (int hq = 0; hq < QH; ++hq) {
#if defined(TS_TUNING) && defined(DBG_NO_LDS)  // ablation builds only
        const u32x4 b = ring[(i + 1) % TS_RING];
#else
        const u32x4 b = ql[(size_t)((g0 + i) * QH + hq) * 64];
#endif
        mma_group<DT>(acc[hq], ring[i], b);
      }
      ring[i] = stream_load(nxt + (size_t)i * 64);
      __builtin_amdgcn_sched_barrier(0);
    }

    if constexpr (MODE == SCAN_DENSE)
      epilogue_dense<QH>(p, acc, w, blk, lane);
    else
      epilogue_filter<QH>(p, st, acc, tau, blk, lane);

    if (!has_next) break;
    w = wn;
    blk = blkn;
    cur = nxt;
  }
  }  // active
  if constexpr (MODE == SCAN_FILTER) flush_stage(p, st, tid);
}

bool ts_use_f32_split(const TsLayout& L, int qh) {
  if (L.dtype != TS_F32 || qh != 1) return false;
#ifdef TS_TUNING   // A/B: TS_NO_F32_SPLIT=1 keeps the exact-f32 MFMA kernel
  static const bool off = getenv("TS_NO_F32_SPLIT") != nullptr;
  if (off) return false;
#endif
  const size_t s = ts_scan_f32s_lds_bytes(L);
  if (s == 0 || s > 160 * 1024) return false;
  // only where the exact-f32 kernel could not take 64 queries per pass anyway
  return (size_t)L.kg * 2 * 1024 + sizeof(StageLds) > 160 * 1024;
}

size_t ts_scan_lds_bytes(const TsLayout& L, int qh) {
  if (ts_use_f32_split(L, qh)) return ts_scan_f32s_lds_bytes(L);
  return (size_t)L.kg * qh * 1024 + sizeof(StageLds);
}

template <int DT, int QH, int MODE>
static int launch_scan_t(const TsLayout& L, const ScanParams& p, int num_cus,
                         hipStream_t stream) {
  const size_t lds = (size_t)L.kg * QH * 1024 + (MODE == SCAN_FILTER ? sizeof(StageLds) : 0);
  auto kern = scan_kernel<DT, QH, MODE>;
  static TsDeviceOnce lds_attr;  // per instantiation, per device (ts_common.h)
  TS_CHECK(ts_allow_max_lds(lds_attr, reinterpret_cast<const void*>(kern)));
  // The fused scan is HBM-bound with one 8-wave workgroup per CU (more waves measured
  // slower: 16 waves/CU -1.5 %); the short dense scans (sample, small corpora) are
  // latency-bound and take a second workgroup per CU when LDS allows.
  int wg_per_cu = 1;
  if (MODE == SCAN_DENSE && 2 * lds <= 160 * 1024) wg_per_cu = 2;
  int64_t want = (p.nwork + SCAN_WAVES - 1) / SCAN_WAVES;
  int64_t cap = (int64_t)num_cus * wg_per_cu;
  int grid = (int)(want < cap ? want : cap);
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(SCAN_THREADS), lds, stream, p);
  TS_HIP(hipGetLastError());
  return TS_OK;
}

template <int DT>
static int launch_scan_dt(const TsLayout& L, int mode, int qh,
                          const ScanParams& p, int num_cus, hipStream_t s) {
  if (qh == 1) {
    return mode == SCAN_DENSE ? launch_scan_t<DT, 1, SCAN_DENSE>(L, p, num_cus, s)
                              : launch_scan_t<DT, 1, SCAN_FILTER>(L, p, num_cus, s);
  }
  return mode == SCAN_DENSE ? launch_scan_t<DT, 2, SCAN_DENSE>(L, p, num_cus, s)
                            : launch_scan_t<DT, 2, SCAN_FILTER>(L, p, num_cus, s);
}

int ts_launch_scan(const TsLayout& L, int mode, int qh, const ScanParams& p,
                   int num_cus, hipStream_t stream) {
  if (p.nwork <= 0) return TS_OK;
  if (ts_use_f32_split(L, qh)) return ts_launch_scan_f32s(L, mode, p, num_cus, stream);
  if (ts_scan_lds_bytes(L, qh) > 160 * 1024) {
    ts_set_error("dimension %d too large for the LDS-resident query image", L.dim);
    return TS_ERR_UNSUPPORTED;
  }
  switch (L.dtype) {
    case TS_F16: return launch_scan_dt<TS_F16>(L, mode, qh, p, num_cus, stream);
    case TS_BF16: return launch_scan_dt<TS_BF16>(L, mode, qh, p, num_cus, stream);
    case TS_F32: return launch_scan_dt<TS_F32>(L, mode, qh, p, num_cus, stream);
  }
  ts_set_error("bad dtype %d", L.dtype);
  return TS_ERR_INVALID;
}

// ------------------------------------------------------------------ layout
// den[i] = |row_i| + 1e-8  (reference src/stage1_retriever.py:285-288)
template <typename TIN>
__global__ void row_den_kernel(const TIN* rows, int64_t n, int dim, float* den) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const TIN* src = rows + row * dim;
  float s = 0.f;
  for (int k = lane; k < dim; k += 64) {
    float v = ElemIO<TIN>::ld(src + k);
    s += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) den[row] = sqrtf(s) + 1e-8f;
}

// One wave per (row block, k group): 64 lanes write one contiguous 1 KiB unit row.
template <typename TIN>
__global__ void relayout_kernel(const TIN* rows, int64_t n, int dim,
                                int64_t row0, int64_t blk_first, int64_t nunits_wave,
                                uint4* tiled, int kg, int dt, const float* den, int vec) {
  const int lane = threadIdx.x & 63;
  const int64_t wv = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (wv >= nunits_wave) return;
  const int64_t b = blk_first + wv / kg;
  const int g = (int)(wv % kg);
  const int r = lane & 31, h = lane >> 5;
  const int64_t row = b * TS_ROWS_PER_BLOCK + r;
  if (row < row0 || row >= row0 + n) return;
  const TIN* src = rows + (row - row0) * dim;
  const float d = den ? den[row - row0] : 1.0f;
  u32x4 out;
  if (dt == TS_F32) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = frag_k(dt, g, h, e);
      float v = (k < dim) ? ElemIO<TIN>::ld(src + k) : 0.f;
      if (den) v = v / d;
      out[e] = __builtin_bit_cast(uint32_t, v);
    }
  } else if (vec) {
    // 16-bit storage: the unit's 8 values are consecutive in the row -> one (f32 rows: two)
    // 16-byte load instead of eight scalar ones (808 -> ~300 us per 500 k x 768 rows)
    const int k0 = frag_k(dt, g, h, 0);
    float v[8];
    if (k0 < dim) {
      ld8<TIN>(src + k0, v);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v0 = v[2 * e], v1 = v[2 * e + 1];
      if (den) { v0 = v0 / d; v1 = v1 / d; }
      out[e] = (uint32_t)f32_to_storage16(v0, dt) | ((uint32_t)f32_to_storage16(v1, dt) << 16);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k0 = frag_k(dt, g, h, 2 * e), k1 = k0 + 1;
      float v0 = (k0 < dim) ? ElemIO<TIN>::ld(src + k0) : 0.f;
      float v1 = (k1 < dim) ? ElemIO<TIN>::ld(src + k1) : 0.f;
      if (den) { v0 = v0 / d; v1 = v1 / d; }
      out[e] = (uint32_t)f32_to_storage16(v0, dt) |
               ((uint32_t)f32_to_storage16(v1, dt) << 16);
    }
  }
  reinterpret_cast<u32x4*>(tiled)[(size_t)(b * kg + g) * 64 + lane] = out;
}

template <typename TIN>
static int relayout_t(const TsLayout& L, const TIN* rows, int64_t n, int64_t row0,
                      uint4* tiled, bool normalize, float* den, hipStream_t s) {
  if (normalize) {
    int blocks = (int)((n + 3) / 4);
    hipLaunchKernelGGL(row_den_kernel<TIN>, dim3(blocks), dim3(256), 0, s, rows, n,
                       L.dim, den);
    TS_HIP(hipGetLastError());
  }
  const int64_t blk_first = row0 / TS_ROWS_PER_BLOCK;
  const int64_t blk_last = (row0 + n - 1) / TS_ROWS_PER_BLOCK;
  const int64_t nwave = (blk_last - blk_first + 1) * L.kg;
  const int64_t blocks = (nwave + 3) / 4;
  if (blocks > 0x7fffffffLL) {
    ts_set_error("add: too many rows in one call");
    return TS_ERR_INVALID;
  }
  hipLaunchKernelGGL(relayout_kernel<TIN>, dim3((unsigned)blocks), dim3(256), 0, s,
                     rows, n, L.dim, row0, blk_first, nwave, tiled, L.kg, L.dtype,
                     normalize ? den : (const float*)nullptr,
                     (int)(L.dtype != TS_F32 && (L.dim % 8) == 0 &&
                           (reinterpret_cast<uintptr_t>(rows) % (4 * sizeof(TIN) >= 16 ? 32 : 16)) == 0));
  TS_HIP(hipGetLastError());
  return TS_OK;
}

int ts_launch_relayout(const TsLayout& L, const void* rows, int in_dtype, int64_t n,
                       int64_t row0, uint4* tiled, bool normalize,
                       float* den_scratch, hipStream_t stream) {
  if (n <= 0) return TS_OK;
  switch (in_dtype) {
    case TS_F32: return relayout_t<float>(L, (const float*)rows, n, row0, tiled, normalize, den_scratch, stream);
    case TS_F16: return relayout_t<_Float16>(L, (const _Float16*)rows, n, row0, tiled, normalize, den_scratch, stream);
    case TS_BF16: return relayout_t<__bf16>(L, (const __bf16*)rows, n, row0, tiled, normalize, den_scratch, stream);
  }
  ts_set_error("bad rows dtype %d", in_dtype);
  return TS_ERR_INVALID;
}

__global__ void reconstruct_kernel(const uint4* tiled, int64_t row0, int64_t n,
                                   int dim, int kg, int dt, float* out) {
  // thread per (row, 16-byte unit)
  const int upr = kg * 2;  // units per row
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * upr) return;
  const int64_t row = row0 + t / upr;
  const int u = (int)(t % upr);
  const int g = u >> 1, h = u & 1;
  const int64_t b = row / TS_ROWS_PER_BLOCK;
  const int lane = h * 32 + (int)(row % TS_ROWS_PER_BLOCK);
  const u32x4 v = reinterpret_cast<const u32x4*>(tiled)[(size_t)(b * kg + g) * 64 + lane];
  float* dst = out + (row - row0) * dim;
  if (dt == TS_F32) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = frag_k(dt, g, h, e);
      const uint32_t w = v[e];
      if (k < dim) dst[k] = __uint_as_float(w);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = frag_k(dt, g, h, e);
      const uint16_t bits = (uint16_t)(v[e >> 1] >> (16 * (e & 1)));
      float f;
      if (dt == TS_F16) f = (float)__builtin_bit_cast(_Float16, bits);
      else f = __builtin_bit_cast(float, (uint32_t)bits << 16);
      if (k < dim) dst[k] = f;
    }
  }
}

int ts_launch_reconstruct(const TsLayout& L, const uint4* tiled, int64_t row0,
                          int64_t n, float* out, hipStream_t stream) {
  if (n <= 0) return TS_OK;
  const int64_t total = n * L.kg * 2;
  const int64_t blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffLL) {
    ts_set_error("reconstruct: range too large");
    return TS_ERR_INVALID;
  }
  hipLaunchKernelGGL(reconstruct_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                     tiled, row0, n, L.dim, L.kg, L.dtype, out);
  TS_HIP(hipGetLastError());
  return TS_OK;
}

// thread per 16-byte unit of the Q image
template <typename TIN>
__global__ void qprep_kernel(const TIN* q, int nq, int dim, int kg, int qh, int dt,
                             uint4* qimg, uint32_t* cand_cnt, uint32_t* status) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < TS_MAX_Q && cand_cnt) cand_cnt[t] = 0;
  if (t == 0 && status) status[0] = 0;
  if (t >= kg * qh * 64) return;
  const int lane = t & 63;
  const int hq = (t >> 6) % qh;
  const int g = (t >> 6) / qh;
  const int j = lane & 31, h = lane >> 5;
  const int qi = hq * 32 + j;
  const TIN* src = q + (int64_t)qi * dim;
  u32x4 out;
  if (dt == TS_F32) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = frag_k(dt, g, h, e);
      float v = (qi < nq && k < dim) ? ElemIO<TIN>::ld(src + k) : 0.f;
      out[e] = __builtin_bit_cast(uint32_t, v);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k0 = frag_k(dt, g, h, 2 * e), k1 = k0 + 1;
      float v0 = (qi < nq && k0 < dim) ? ElemIO<TIN>::ld(src + k0) : 0.f;
      float v1 = (qi < nq && k1 < dim) ? ElemIO<TIN>::ld(src + k1) : 0.f;
      out[e] = (uint32_t)f32_to_storage16(v0, dt) |
               ((uint32_t)f32_to_storage16(v1, dt) << 16);
    }
  }
  reinterpret_cast<u32x4*>(qimg)[t] = out;
}

int ts_launch_qprep(const TsLayout& L, const void* q, int q_dtype, int nq, int qh,
                    uint4* qimg, uint32_t* cand_cnt, uint32_t* status,
                    hipStream_t stream) {
  if (ts_use_f32_split(L, qh)) return ts_launch_qprep_f32s(L, q, q_dtype, nq, qimg, cand_cnt, status, stream);
  const int units = L.kg * qh * 64;
  const int blocks = (units + 255) / 256;
  switch (q_dtype) {
    case TS_F32:
      hipLaunchKernelGGL(qprep_kernel<float>, dim3(blocks), dim3(256), 0, stream,
                         (const float*)q, nq, L.dim, L.kg, qh, L.dtype, qimg, cand_cnt, status);
      break;
    case TS_F16:
      hipLaunchKernelGGL(qprep_kernel<_Float16>, dim3(blocks), dim3(256), 0, stream,
                         (const _Float16*)q, nq, L.dim, L.kg, qh, L.dtype, qimg, cand_cnt, status);
      break;
    case TS_BF16:
      hipLaunchKernelGGL(qprep_kernel<__bf16>, dim3(blocks), dim3(256), 0, stream,
                         (const __bf16*)q, nq, L.dim, L.kg, qh, L.dtype, qimg, cand_cnt, status);
      break;
    default:
      ts_set_error("bad query dtype %d", q_dtype);
      return TS_ERR_INVALID;
  }
  TS_HIP(hipGetLastError());
  return TS_OK;
}
