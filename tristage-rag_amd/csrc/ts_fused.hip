// One-launch stage-1 search for gfx950: query preparation, threshold estimation and the fused
// scan+filter of ts_scan.hip in ONE kernel (the part of faiss_index.search, reference
// src/stage1_retriever.py:380, that precedes the final selection).
//
// Before: query prep -> sample scan -> thresholds -> scan+filter -> select = five launches; the three
// in front of the scan cost ~45 us per 64-query batch, which is nothing at 10 M rows but a sixth of
// the per-rank step of an 8-way sharded corpus (1.25 M rows, 290 us of streaming).  Here:
//
//   * workgroups [0, scan_wgs) are the persistent scan waves of ts_scan.hip.  Each builds the LDS
//     query image straight from the caller's query rows (no prepared image in global memory).
//   * A wave's first `sample_rounds` row blocks double as the threshold SAMPLE: the blocks of round
//     r are w + r*W for wave w, i.e. a contiguous prefix of W*32 rows per round … which is why the
//     host side hands the waves a STRIDED block order for those rounds (see blk_of()).  For every
//     (lane, query half) the maximum of the lane's 16 scores — its order-preserving 32-bit key — is
//     stored into the query's slot array with ONE agent-scope atomic store (a slot holds 0 or its
//     final key; key 0 is never produced).
//   * workgroups [scan_wgs, scan_wgs + tau_wgs) run on the CUs the scan grid leaves free.  They
//     reload a query's slots until none is 0 (so readiness does not rely on any ordering between
//     different addresses), select the m-th largest key EXACTLY (radix select in LDS), give the slots
//     back as zeros and publish (generation, threshold) as ONE 64-bit atomic store per query.
//   * A scan wave never waits in the common case: until its lanes see this launch's generation in
//     the thresholds it keeps streaming and parks the dense 32x64 score tiles of its blocks in a
//     private spill area (8 KiB per block, re-read by the same wave only), then switches to the
//     fused filter epilogue and re-filters the parked tiles after its last block.  Only if the
//     spill area is exhausted, or at the very end, does it spin — every spin is bounded, and a
//     wave that gives up poisons the candidate count so that the exactness verification of
//     ts_index.hip redoes the batch on the dense path: every wave reaches its exit.
//
// Any threshold is a SAFE threshold (ts_index.hip verifies the candidate counts); the group maxima (the
// m-th largest of per-16-row maxima is the m-th largest sample score as long as m is far below the
// number of groups; the host only takes this path while 4 m <= groups) only decide how many
// candidates survive.  (A first version estimated the threshold from a 14-bit histogram of the keys:
// with real embedding models, whose scores sit in a narrow band, one bin held a third of the corpus.)
#include "ts_scan_dev.h"
#include <algorithm>


__device__ __forceinline__ uint32_t fz_key(float f) {
  if (f != f) return 0u;
  f = f + 0.0f;
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fz_unkey(uint32_t k) {
  const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __builtin_bit_cast(float, u);
}

// 16-byte loads / stores that are coherent at agent scope (sc0 sc1: they bypass the non-coherent
// per-XCD L2 state) — the histogram is only ever touched by atomics and by these.
__device__ __forceinline__ u32x4 coherent_load16(const uint32_t* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void coherent_store16(uint32_t* p, const u32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}

struct FusedParams {
  ScanParams sp;            // corpus / kg / nq / nwork = row blocks / ntotal / candidate lists (tau, qimg, dense unused)
  const void* queries;      // [nq, dim] rows of q_dtype (device)
  int q_dtype;
  int dim;
  int q_vec;                // query rows can be read 8 elements at a time
  int scan_wgs, tau_wgs;
  int64_t n_sample;         // row blocks that feed the threshold sample (each wave's first rounds)
  int spill_rounds;         // capacity of the spill area, in rounds
  int64_t sample_stride;    // block stride of the sample rounds (see blk_of)
  uint32_t m;               // wanted rank among the group maxima
  uint32_t expect;          // slots per query of a complete sample (<= TS_FUSED_MAX_KEYS)
  uint32_t keys_ld;         // slots between consecutive queries
  uint32_t gen;             // generation tag of this launch (never 0)
  uint32_t arrive_goal;     // value of *arrive when every sample wave has reported (hint only)
  uint32_t wait_iters;      // bound of every spin loop
  uint32_t* skeys;          // [64][keys_ld] sample keys, all-zero between launches
  uint32_t* arrive;         // monotonic hint counter
  unsigned long long* tau64;  // [64] (generation << 32) | float bits
  float* spill;             // [spill_rounds][scan waves][64][32]
};

// ---- block order ------------------------------------------------------------------------------
// Work item j of wave w is "round j".  Rounds >= R walk the corpus interleaved (block = item index),
// exactly like ts_scan.hip.  The R sample rounds must not be a contiguous prefix of the corpus
// (documents are often added grouped by topic): sample item s = r*W + w is mapped to block
// s*stride, and the interleaved rounds skip those blocks.  To keep the index arithmetic trivial the
// host chooses stride so that the sample blocks are exactly the multiples of `stride` below
// n_sample*stride, and the remaining blocks are enumerated by skipping them.
struct BlockOrder {
  int64_t nblk, n_sample, stride, n_rest;
};
__device__ __forceinline__ int64_t rest_block(const BlockOrder& o, int64_t t) {
  // t-th block (in ascending order) that is not a sample block.  Below n_sample*stride every run of
  // `stride` blocks holds stride-1 of them.
  if (o.stride <= 1) return o.n_sample + t;
  const int64_t per = o.stride - 1;
  const int64_t full = o.n_sample * per;
  if (t >= full) return o.n_sample * o.stride + (t - full);
  return (t / per) * o.stride + 1 + (t % per);
}

template <int QH>
__device__ __forceinline__ void spill_store(float* dst, const f32x16 (&acc)[QH], int lane) {
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int hq = 0; hq < QH; ++hq) {
    float* q = dst + (hq * 32 + j) * 32 + 4 * h;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      float4 v;
      v.x = acc[hq][4 * r4 + 0]; v.y = acc[hq][4 * r4 + 1]; v.z = acc[hq][4 * r4 + 2]; v.w = acc[hq][4 * r4 + 3];
      *reinterpret_cast<float4*>(q + 8 * r4) = v;
    }
  }
}
template <int QH>
__device__ __forceinline__ void spill_load(const float* src, f32x16 (&acc)[QH], int lane) {
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int hq = 0; hq < QH; ++hq) {
    const float* q = src + (hq * 32 + j) * 32 + 4 * h;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const float4 v = *reinterpret_cast<const float4*>(q + 8 * r4);
      acc[hq][4 * r4 + 0] = v.x; acc[hq][4 * r4 + 1] = v.y; acc[hq][4 * r4 + 2] = v.z; acc[hq][4 * r4 + 3] = v.w;
    }
  }
}

// this lane's thresholds, once every lane of the wave sees the launch's generation
template <int QH>
__device__ __forceinline__ bool tau_ready(const FusedParams& p, const unsigned long long (&tq)[QH], float (&tau)[QH]) {
  bool ok = true;
#pragma unroll
  for (int hq = 0; hq < QH; ++hq) ok &= ((uint32_t)(tq[hq] >> 32) == p.gen);
  if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) return false;
#pragma unroll
  for (int hq = 0; hq < QH; ++hq) tau[hq] = __builtin_bit_cast(float, (uint32_t)tq[hq]);
  return true;
}
template <int QH>
__device__ __forceinline__ void tau_fetch(const FusedParams& p, unsigned long long (&tq)[QH], int lane) {
#pragma unroll
  for (int hq = 0; hq < QH; ++hq)
    tq[hq] = __hip_atomic_load(p.tau64 + hq * 32 + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int QH>
__device__ __forceinline__ bool tau_wait(const FusedParams& p, float (&tau)[QH], int lane) {
  unsigned long long tq[QH];
  for (uint32_t it = 0; it < p.wait_iters; ++it) {
    tau_fetch<QH>(p, tq, lane);
    if (tau_ready<QH>(p, tq, tau)) return true;
    __builtin_amdgcn_s_sleep(64);
  }
  return false;
}

// ---- threshold workgroups ---------------------------------------------------------------------
__device__ void tau_role(const FusedParams& p, int tw, unsigned char* smem) {
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem);    // [256]
  uint32_t* wtot = hist + 256;                           // [4]
  uint32_t* res = wtot + 8;                              // [0] digit, [1] rank left, [2] flags
  uint32_t* keys = res + 8;                              // [expect]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t n = p.expect;
  // the hint: all sample waves have reported (bounded; the loads below re-check slot by slot)
  if (tid == 0) {
    for (uint32_t it = 0; it < p.wait_iters; ++it) {
      const uint32_t cur = __hip_atomic_load(p.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((int32_t)(cur - p.arrive_goal) >= 0) break;
      __builtin_amdgcn_s_sleep(64);
    }
  }
  __syncthreads();
  for (int q = tw; q < TS_MAX_Q; q += p.tau_wgs) {
    if (q >= p.sp.nq) {   // no such query: nothing may pass
      if (tid == 0)
        __hip_atomic_store(p.tau64 + q, ((unsigned long long)p.gen << 32) | 0x7f7fffffull, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    uint32_t* src = p.skeys + (size_t)q * p.keys_ld;
    // ---- slots -> LDS, until every one of them holds a key
    bool complete = false;
    for (uint32_t it = 0; it < p.wait_iters && !complete; ++it) {
      __syncthreads();
      if (tid == 0) res[2] = 0u;
      __syncthreads();
      bool hole = false;
      for (uint32_t i = 4 * tid; i < n; i += 4 * SCAN_THREADS) {   // (n is a multiple of 4: two slots per sample item, items even)
        const u32x4 v = coherent_load16(src + i);
        keys[i] = v[0]; keys[i + 1] = v[1]; keys[i + 2] = v[2]; keys[i + 3] = v[3];
        hole |= (v[0] == 0u) | (v[1] == 0u) | (v[2] == 0u) | (v[3] == 0u);
      }
      if (__builtin_amdgcn_ballot_w64(hole) != 0ull && lane == 0) res[2] = 1u;
      __syncthreads();
      complete = res[2] == 0u;
      if (!complete) __builtin_amdgcn_s_sleep(64);
    }
    uint32_t tkey = 0u;
    if (complete) {
      // ---- the m-th largest key: MSB-first radix select, 8 bits per pass, in LDS
      uint32_t prefix = 0u, krem = p.m < n ? p.m : n;
      int bits = 0;
      for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        __syncthreads();
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        for (uint32_t i = tid; i < n; i += SCAN_THREADS) {
          const uint32_t key = keys[i];
          if (bits == 0 || (key >> (32 - bits)) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        // bin holding the krem-th largest: suffix sums over the 256 bins (4 waves x 64 lanes)
        uint32_t own = 0u, v = 0u;
        if (tid < 256) {
          own = hist[tid];
          v = own;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_down((int)v, o, 64);
            if (lane + o < 64) v += t;
          }
          if (lane == 0) wtot[wave] = v;
        }
        __syncthreads();
        if (tid < 256) {
          for (int w2 = wave + 1; w2 < 4; ++w2) v += wtot[w2];   // entries in bins >= tid
          const uint32_t above = v - own;
          if (v >= krem && above < krem) { res[0] = (uint32_t)tid; res[1] = krem - above; }
        }
        __syncthreads();
        prefix = (prefix << 8) | res[0];
        krem = res[1];
        bits += 8;
      }
      tkey = prefix;
      // give the slots back as zeros (the sample is complete: nobody writes them any more)
      for (uint32_t i = 4 * tid; i < n; i += 4 * SCAN_THREADS) coherent_store16(src + i, u32x4{0u, 0u, 0u, 0u});
    }
    if (tid == 0) {
      // scores >= the m-th largest group maximum pass.  No complete sample within the bound (e.g. the scan
      // workgroups never became resident): nothing passes and the verification redoes the batch.
      const uint32_t bits32 = complete ? __builtin_bit_cast(uint32_t, fz_unkey(tkey)) : 0x7f7fffffu;
      __hip_atomic_store(p.tau64 + q, ((unsigned long long)p.gen << 32) | (unsigned long long)bits32, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- the kernel -------------------------------------------------------------------------------
template <typename TIN>
__device__ __forceinline__ u32x4 q_unit(const TIN* q, int nq, int dim, int dt, int g, int h, int qi, int vec) {
  u32x4 out;
  const TIN* src = q + (int64_t)qi * dim;
  if (dt == TS_F32) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = frag_k(dt, g, h, e);
      const float v = (qi < nq && k < dim) ? ElemIO<TIN>::ld(src + k) : 0.f;
      out[e] = __builtin_bit_cast(uint32_t, v);
    }
  } else if (vec) {
    const int k0 = frag_k(dt, g, h, 0);
    float v[8];
    if (qi < nq && k0 < dim) {
      ld8<TIN>(src + k0, v);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      out[e] = (uint32_t)f32_to_storage16(v[2 * e], dt) | ((uint32_t)f32_to_storage16(v[2 * e + 1], dt) << 16);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k0 = frag_k(dt, g, h, 2 * e), k1 = k0 + 1;
      const float v0 = (qi < nq && k0 < dim) ? ElemIO<TIN>::ld(src + k0) : 0.f;
      const float v1 = (qi < nq && k1 < dim) ? ElemIO<TIN>::ld(src + k1) : 0.f;
      out[e] = (uint32_t)f32_to_storage16(v0, dt) | ((uint32_t)f32_to_storage16(v1, dt) << 16);
    }
  }
  return out;
}

// 8 independent units per thread and trip: their loads are in flight together (one unit at a time
// is a chain of ~100 dependent L2 round trips per thread: it cost ~100 us per launch)
template <typename TIN, int DT, int QH>
__device__ __forceinline__ void build_qimage(const FusedParams& p, u32x4* qlds, int tid) {
  const int units = p.sp.kg * QH * 64;
  const TIN* q = reinterpret_cast<const TIN*>(p.queries);
  for (int t0 = tid; t0 < units; t0 += 8 * SCAN_THREADS) {
    u32x4 u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = t0 + i * SCAN_THREADS;
      const int tt = t < units ? t : t0;          // (always a valid unit: no branch around the loads)
      const int l = tt & 63;
      const int hq = (tt >> 6) % QH;
      const int g = (tt >> 6) / QH;
      u[i] = q_unit<TIN>(q, p.sp.nq, p.dim, DT, g, l >> 5, hq * 32 + (l & 31), p.q_vec);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = t0 + i * SCAN_THREADS;
      if (t < units) qlds[t] = u[i];
    }
  }
}

template <int DT, int QH>
__global__ __launch_bounds__(SCAN_THREADS) void fused_kernel(FusedParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if ((int)blockIdx.x >= p.scan_wgs) {
    tau_role(p, (int)blockIdx.x - p.scan_wgs, smem);
    return;
  }
  u32x4* qlds = reinterpret_cast<u32x4*>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int kg = p.sp.kg;
  const int64_t nwaves = (int64_t)p.scan_wgs * SCAN_WAVES;
  const int64_t w = (int64_t)blockIdx.x * SCAN_WAVES + wave;
  BlockOrder order;
  order.nblk = p.sp.nwork;
  order.stride = p.sample_stride;
  order.n_sample = p.n_sample;   // (host: n_sample * stride <= nblk)
  order.n_rest = order.nblk - order.n_sample;
  // work item j of this wave: sample item w + j*W while that is < n_sample, then the rest blocks
  const int64_t my_sample = w < order.n_sample ? (order.n_sample - w + nwaves - 1) / nwaves : 0;
  const int64_t my_rest = w < order.n_rest ? (order.n_rest - w + nwaves - 1) / nwaves : 0;
  const int64_t my_items = my_sample + my_rest;
  auto blk_of = [&](int64_t j) -> int64_t {
    return j < my_sample ? (w + j * nwaves) * order.stride : rest_block(order, w + (j - my_sample) * nwaves);
  };
  const bool active = my_items > 0;

  // ---- the first corpus loads go out before anything else
  const u32x4* base = reinterpret_cast<const u32x4*>(p.sp.corpus) + lane;
  const size_t blk_units = (size_t)kg * 64;
  int64_t blk = active ? blk_of(0) : 0;
  const u32x4* cur = base + (size_t)blk * blk_units;
  u32x4 ring[TS_RING];
  if (active) {
#pragma unroll
    for (int i = 0; i < TS_RING; ++i) ring[i] = stream_load(cur + (size_t)i * 64);
  }

  // ---- prologue: the query image is built in LDS from the caller's rows
  if (p.q_dtype == TS_F32) build_qimage<float, DT, QH>(p, qlds, tid);
  else if (p.q_dtype == TS_F16) build_qimage<_Float16, DT, QH>(p, qlds, tid);
  else build_qimage<__bf16, DT, QH>(p, qlds, tid);
  StageLds* st = reinterpret_cast<StageLds*>(smem + (size_t)kg * QH * 1024);
  if (tid == 0) st->cnt = 0;
  // (the staging loop's loads sit under lane predicates: make the compiler's scoreboard forget them
  // here instead of in front of the first ring consumer)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();

  bool failed = false;
  if (active) {
    float tau[QH];
#pragma unroll
    for (int hq = 0; hq < QH; ++hq) tau[hq] = 3.402823466e38f;
    unsigned long long tq[QH];
    tau_fetch<QH>(p, tq, lane);          // first look, consumed after the first block
    bool filtering = false;
    int64_t n_spilled = 0;               // items [0, n_spilled) were parked in the spill area
    float* myspill = p.spill + (size_t)w * (TS_MAX_Q * 32);
    const size_t spill_round = (size_t)nwaves * (TS_MAX_Q * 32);
    const u32x4* ql = qlds + lane;

    for (int64_t j = 0; j < my_items; ++j) {
      const bool has_next = j + 1 < my_items;
      const int64_t blkn = has_next ? blk_of(j + 1) : blk;
      const u32x4* nxt = base + (size_t)blkn * blk_units;

      f32x16 acc[QH];
#pragma unroll
      for (int hq = 0; hq < QH; ++hq)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[hq][r] = 0.f;

      int g0 = 0;
      for (; g0 < kg - TS_RING; g0 += TS_RING) {
#pragma unroll
        for (int i = 0; i < TS_RING; ++i) {
#pragma unroll
          for (int hq = 0; hq < QH; ++hq) mma_group<DT>(acc[hq], ring[i], ql[(size_t)((g0 + i) * QH + hq) * 64]);
          ring[i] = stream_load(cur + (size_t)(g0 + i + TS_RING) * 64);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int i = 0; i < TS_RING; ++i) {
#pragma unroll
        for (int hq = 0; hq < QH; ++hq) mma_group<DT>(acc[hq], ring[i], ql[(size_t)((g0 + i) * QH + hq) * 64]);
        ring[i] = stream_load(nxt + (size_t)i * 64);
        __builtin_amdgcn_sched_barrier(0);
      }

      // ---- epilogue
      if (j < my_sample) {
        // sample round: the lane's group maximum -> its slot of the query's key array
        const int64_t slot = 2 * (w + j * nwaves) + (lane >> 5);
#pragma unroll
        for (int hq = 0; hq < QH; ++hq) {
          const int q = hq * 32 + (lane & 31);
          if (q < p.sp.nq) {
            uint32_t key = fz_key(acc_max(acc[hq]));
            key = key ? key : 1u;   // (0 means "not written yet"; only a NaN score maps there)
            __hip_atomic_store(p.skeys + (size_t)q * p.keys_ld + slot, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        if (j + 1 == my_sample && lane == 0) atomicAdd(p.arrive, 1u);   // (a hint; see tau_role)
      }
      if (!filtering) filtering = tau_ready<QH>(p, tq, tau);
      if (!filtering && j >= p.spill_rounds) {
        // spill area exhausted: the only place a wave waits in mid-stream
        filtering = tau_wait<QH>(p, tau, lane);
        if (!filtering) { failed = true; break; }
      }
      if (filtering) {
        epilogue_filter<QH>(p.sp, st, acc, tau, blk, lane);
      } else {
        spill_store<QH>(myspill + (size_t)j * spill_round, acc, lane);
        n_spilled = j + 1;
        tau_fetch<QH>(p, tq, lane);      // next look, consumed after the next block
      }
      blk = blkn;
      cur = nxt;
    }
    if (!failed && !filtering) {
      filtering = tau_wait<QH>(p, tau, lane);
      failed = !filtering;
    }
    if (!failed) {
      // the parked tiles, now that the thresholds are known
      for (int64_t j = 0; j < n_spilled; ++j) {
        f32x16 acc[QH];
        spill_load<QH>(myspill + (size_t)j * spill_round, acc, lane);
        epilogue_filter<QH>(p.sp, st, acc, tau, blk_of(j), lane);
      }
    }
  }
  if (__builtin_amdgcn_ballot_w64(failed) != 0ull && lane == 0)
    atomicOr(&p.sp.cand_cnt[0], 0x80000000u);   // "overflow": ts_index.hip redoes this batch exactly
  flush_stage(p.sp, st, tid);
}

// ---- host side --------------------------------------------------------------------------------
struct FusedLaunch {
  FusedParams p;
  int grid;
  size_t lds;
};

template <int DT, int QH>
static int launch_fused_t(const FusedParams& p, int grid, size_t lds, hipStream_t s) {
  auto kern = fused_kernel<DT, QH>;
  static TsDeviceOnce lds_attr;
  TS_CHECK(ts_allow_max_lds(lds_attr, reinterpret_cast<const void*>(kern)));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(SCAN_THREADS), lds, s, p);
  TS_HIP(hipGetLastError());
  return TS_OK;
}

size_t ts_fused_keys_bytes() { return (size_t)TS_MAX_Q * TS_FUSED_MAX_KEYS * sizeof(uint32_t); }
size_t ts_fused_spill_bytes(int scan_wgs, int spill_rounds) {
  return (size_t)spill_rounds * scan_wgs * SCAN_WAVES * TS_MAX_Q * 32 * sizeof(float);
}

int ts_launch_fused(const TsLayout& L, int qh, const TsFusedArgs& a, hipStream_t stream) {
  FusedParams p{};
  p.sp.corpus = a.corpus;
  p.sp.kg = L.kg;
  p.sp.nq = a.nq;
  p.sp.nwork = a.nblk;
  p.sp.blk0 = 0;
  p.sp.blk_stride = 1;
  p.sp.ntotal = a.ntotal;
  p.sp.cand_cnt = a.cand_cnt;
  p.sp.cand_score = a.cand_score;
  p.sp.cand_id = a.cand_id;
  p.sp.cand_cap = a.cand_cap;
  p.queries = a.queries;
  p.q_dtype = a.q_dtype;
  p.dim = L.dim;
  const size_t esz = a.q_dtype == TS_F32 ? 4 : 2;
  p.q_vec = (L.dtype != TS_F32 && (L.dim % 8) == 0 &&
             (reinterpret_cast<uintptr_t>(a.queries) % (esz == 4 ? 32 : 16)) == 0) ? 1 : 0;
  p.scan_wgs = a.scan_wgs;
  p.tau_wgs = a.tau_wgs;
  p.n_sample = a.n_sample;
  p.spill_rounds = a.spill_rounds;
  p.sample_stride = a.sample_stride;
  p.m = a.m;
  p.expect = a.expect;
  p.gen = a.gen;
  p.arrive_goal = a.arrive_goal;
  p.wait_iters = a.wait_iters;
  p.skeys = a.skeys;
  p.keys_ld = TS_FUSED_MAX_KEYS;
  p.arrive = a.arrive;
  p.tau64 = a.tau64;
  p.spill = a.spill;
  // every workgroup gets the same dynamic LDS: the scan's query image + staging, or the threshold role's keys
  const size_t lds = std::max(ts_scan_lds_bytes(L, qh), (size_t)a.expect * 4 + 4096);
  if (a.expect > TS_FUSED_MAX_KEYS || (a.expect & 3u) || lds > 160 * 1024 || a.expect != 2 * a.n_sample ||
      a.n_sample * a.sample_stride > a.nblk || a.n_sample < 1 || a.sample_stride < 1) {
    ts_set_error("one-launch search: bad sample geometry (%u slots)", a.expect);
    return TS_ERR_INVALID;
  }
  const int grid = a.scan_wgs + a.tau_wgs;
  if (qh == 1) {
    switch (L.dtype) {
      case TS_F16: return launch_fused_t<TS_F16, 1>(p, grid, lds, stream);
      case TS_BF16: return launch_fused_t<TS_BF16, 1>(p, grid, lds, stream);
      case TS_F32: return launch_fused_t<TS_F32, 1>(p, grid, lds, stream);
    }
  } else {
    switch (L.dtype) {
      case TS_F16: return launch_fused_t<TS_F16, 2>(p, grid, lds, stream);
      case TS_BF16: return launch_fused_t<TS_BF16, 2>(p, grid, lds, stream);
      case TS_F32: return launch_fused_t<TS_F32, 2>(p, grid, lds, stream);
    }
  }
  ts_set_error("bad dtype %d", L.dtype);
  return TS_ERR_INVALID;
}
