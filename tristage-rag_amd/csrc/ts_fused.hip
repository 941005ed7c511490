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
//   * Until a scan wave sees this launch's thresholds it keeps streaming and PARKS the 32x64 score tiles
//     of its blocks in registers (two tiles, 64 VGPRs); when the thresholds arrive it filters the parked
//     tiles and goes on with the fused filter epilogue.  A third tile before the thresholds, or the end
//     of its blocks, makes it spin — every spin is bounded, and a wave that gives up poisons the
//     candidate count so that the exactness verification of ts_index.hip redoes the batch on the dense
//     path: every wave reaches its exit.  (Parking in global memory was tried first: on gfx9 stores share
//     vmcnt with the corpus ring, every parked block drained the ring, 20-30 us per block instead of
//     13, tools/trace_fused.py.)
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
  int tau_waves;            // waves of a threshold workgroup that take queries (their key arrays share its LDS)
  int64_t n_sample;         // row blocks that feed the threshold sample (each wave's first rounds)
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
};

// ---- block order ------------------------------------------------------------------------------
// Work item j of wave w is "round j".  Rounds >= R walk the corpus interleaved (block = item index),
// exactly like ts_scan.hip.  The R sample rounds must not be a contiguous prefix of the corpus
// (documents are often added grouped by topic): sample item s = r*W + w is mapped to block
// s*stride, and the interleaved rounds skip those blocks.  To keep the index arithmetic trivial the
// host chooses stride so that the sample blocks are exactly the multiples of `stride` below
// n_sample*stride, and the remaining blocks are enumerated by skipping them.
//
// The sample blocks come in GROUPS of SAMPLE_GROUP consecutive blocks (the 8 waves of a workgroup read one
// contiguous 8-block run): 1792 waves each opening its own 2 MiB page somewhere in the corpus was a TLB-miss
// storm at kernel start (first block done after 30-56 us instead of ~14, tools/trace_fused.py); groups keep
// the sample spread over n_sample/8 positions with an eighth of the page walks.
#define SAMPLE_GROUP 8
struct BlockOrder {
  int64_t nblk, n_sample, stride, n_rest;   // stride: distance between the first blocks of consecutive groups
};
__device__ __forceinline__ int64_t sample_block(const BlockOrder& o, int64_t s) {
  return (s / SAMPLE_GROUP) * o.stride + (s % SAMPLE_GROUP);
}
__device__ __forceinline__ int64_t rest_block(const BlockOrder& o, int64_t t) {
  // t-th block (in ascending order) that is not a sample block.  Below n_groups*stride every run of
  // `stride` blocks starts with SAMPLE_GROUP sample blocks.
  const int64_t n_groups = o.n_sample / SAMPLE_GROUP;
  const int64_t per = o.stride - SAMPLE_GROUP;
  if (per <= 0) return o.n_sample + t;
  const int64_t full = n_groups * per;
  if (t >= full) return n_groups * o.stride + (t - full);
  return (t / per) * o.stride + SAMPLE_GROUP + (t % per);
}

// The thresholds reach a scan workgroup through ONE wave: wave 0 polls the 64 published (generation,
// threshold) words with agent-scope loads and hands them to its seven siblings through LDS.  (Every
// wave polling by itself — 1792 waves x 64 lanes hitting the same 512 bytes with L2-bypassing loads
// once per block — made each block of the parking phase 3x slower than a streaming block: the loads
// queue up at one memory channel and the ring's loads return in order behind them, tools/trace_fused.py.)
template <int QH>
__device__ __forceinline__ void tau_fetch(const FusedParams& p, unsigned long long (&tq)[2], int lane) {
#pragma unroll
  for (int hq = 0; hq < 2; ++hq)   // (all 64 thresholds, whatever QH is: the siblings' lanes need theirs)
    tq[hq] = __hip_atomic_load(p.tau64 + hq * 32 + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wave 0 only: true (and the thresholds are in LDS, flag set) once every lane sees the launch's generation
__device__ __forceinline__ bool tau_publish_lds(const FusedParams& p, StageLds* st, const unsigned long long (&tq)[2], int lane) {
  const bool ok = ((uint32_t)(tq[0] >> 32) == p.gen) && ((uint32_t)(tq[1] >> 32) == p.gen);
  if (__builtin_amdgcn_ballot_w64(!ok) != 0ull) return false;
  // values first, flag second — both LDS: the LDS unit performs one wave's operations in the order they were
  // issued, so a sibling that reads the flag as 1 and then the values gets the values.  NO fence builtin here:
  // a workgroup-scope fence orders every address space, i.e. it waits for vmcnt(0) and drains the corpus
  // ring (it made each block of the parking phase twice as long, tools/trace_fused.py).
  volatile float* tv = st->tauv;
  if (lane < 32) {
    tv[lane] = __builtin_bit_cast(float, (uint32_t)tq[0]);
    tv[32 + lane] = __builtin_bit_cast(float, (uint32_t)tq[1]);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) *(volatile uint32_t*)&st->tau_flag = 1u;
  return true;
}
// any wave: the thresholds from LDS if wave 0 has published them
template <int QH>
__device__ __forceinline__ bool tau_from_lds(StageLds* st, float (&tau)[QH], int lane) {
  if (*(volatile uint32_t*)&st->tau_flag == 0u) return false;
  asm volatile("" ::: "memory");
  volatile float* tv = st->tauv;
#pragma unroll
  for (int hq = 0; hq < QH; ++hq) tau[hq] = tv[hq * 32 + (lane & 31)];
  return true;
}
// bounded wait (spill area exhausted, or nothing left to stream)
template <int QH>
__device__ __forceinline__ bool tau_wait(const FusedParams& p, StageLds* st, float (&tau)[QH], int lane, int wave) {
  for (uint32_t it = 0; it < p.wait_iters; ++it) {
    if (wave == 0 && *(volatile uint32_t*)&st->tau_flag == 0u) {
      unsigned long long tq[2];
      tau_fetch<QH>(p, tq, lane);
      (void)tau_publish_lds(p, st, tq, lane);
    }
    if (tau_from_lds<QH>(st, tau, lane)) return true;
    __builtin_amdgcn_s_sleep(32);
  }
  return false;
}

#if defined(TS_TUNING) && defined(FZ_TRACE)   // diagnostic builds only: per-wave phase time stamps (100 MHz)
__device__ unsigned long long fz_trace_buf[4096 * 8];
#define FZ_STAMP(row, i) do { if ((threadIdx.x & 63) == 0 && (row) < 4096) fz_trace_buf[(row) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int ts_debug_fused_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fz_trace_buf), sizeof(fz_trace_buf)) == hipSuccess ? 0 : -2;
}
#else
#define FZ_STAMP(row, i) do { } while (0)
#endif

__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
// ---- threshold workgroups ---------------------------------------------------------------------
// ONE WAVE PER QUERY, no workgroup barrier and no LDS atomics: wave v of threshold workgroup t owns queries
// t*tau_waves + v, +tau_wgs*tau_waves, ...  Its keys live in a private LDS region (lane L holds keys L, L+64, ...:
// conflict-free reads); the m-th largest is found by BISECTION on the 32-bit key — count the keys >= pivot,
// one wave reduction per step, the bits every key shares skipped.  (Tried before: a radix select by the whole
// workgroup — ~20 barriers, 14 us per query pair; a radix select by one wave — bank-conflict-bound LDS atomics,
// 25 us.  tools/trace_fused.py.)
__device__ __forceinline__ void coherent_load4x8(const uint32_t* p0, int stride, uint32_t (&v)[8]) {
  // eight independent agent-coherent dword loads in flight, then ONE wait
  asm volatile(
      "global_load_dword %0, %8, off sc0 sc1\n\t"
      "global_load_dword %1, %9, off sc0 sc1\n\t"
      "global_load_dword %2, %10, off sc0 sc1\n\t"
      "global_load_dword %3, %11, off sc0 sc1\n\t"
      "global_load_dword %4, %12, off sc0 sc1\n\t"
      "global_load_dword %5, %13, off sc0 sc1\n\t"
      "global_load_dword %6, %14, off sc0 sc1\n\t"
      "global_load_dword %7, %15, off sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(p0), "v"(p0 + stride), "v"(p0 + 2 * stride), "v"(p0 + 3 * stride), "v"(p0 + 4 * stride),
        "v"(p0 + 5 * stride), "v"(p0 + 6 * stride), "v"(p0 + 7 * stride)
      : "memory");
}
__device__ __forceinline__ void tau_role(const FusedParams& p, int tw, unsigned char* smem) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave >= p.tau_waves) return;
  const uint32_t n = p.expect;                            // (a multiple of 512: see the host side)
  uint32_t* keys = reinterpret_cast<uint32_t*>(smem) + (size_t)wave * n;
  [[maybe_unused]] const int trow = 3000 + tw * SCAN_WAVES + wave;   // (FZ_TRACE builds)
  FZ_STAMP(trow, 0);
  // the hint: all sample workgroups have reported (bounded; the loads below re-check slot by slot)
  for (uint32_t it = 0; it < p.wait_iters; ++it) {
    const uint32_t cur = __hip_atomic_load(p.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int32_t)(cur - p.arrive_goal) >= 0) break;
    __builtin_amdgcn_s_sleep(48);   // (~1 us: 64 waves hammering one address with L2-bypassing loads disturb the stream)
  }
  FZ_STAMP(trow, 1);
  const int first = tw * p.tau_waves + wave, step = p.tau_wgs * p.tau_waves;
  for (int q = first; q < TS_MAX_Q; q += step) {
    if (q >= p.sp.nq) {   // no such query: nothing may pass
      if (lane == 0)
        __hip_atomic_store(p.tau64 + q, ((unsigned long long)p.gen << 32) | 0x7f7fffffull, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    uint32_t* src = p.skeys + (size_t)q * p.keys_ld;
    // ---- slots -> LDS until every one of them holds a key; the bits all keys share on the way
    bool complete = false;
    uint32_t k_and = 0xFFFFFFFFu, k_or = 0u;
    for (uint32_t it = 0; it < p.wait_iters && !complete; ++it) {
      bool hole = false;
      k_and = 0xFFFFFFFFu; k_or = 0u;
      for (uint32_t i0 = 0; i0 < n; i0 += 512) {
        uint32_t v[8];
        coherent_load4x8(src + i0 + lane, 64, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          keys[i0 + 64 * j + lane] = v[j];
          hole |= v[j] == 0u;
          k_and &= v[j]; k_or |= v[j];
        }
      }
      complete = __builtin_amdgcn_ballot_w64(hole) == 0ull;
      if (!complete) __builtin_amdgcn_s_sleep(8);
    }
    if (q == first) FZ_STAMP(trow, 2);
    uint32_t bits32 = 0x7f7fffffu;   // no complete sample within the bound: nothing passes, the verification redoes the batch
    if (complete) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        k_and &= (uint32_t)__shfl_xor((int)k_and, o, 64);
        k_or |= (uint32_t)__shfl_xor((int)k_or, o, 64);
      }
      wave_lds_sync();
      // the largest T with  #{key >= T} >= m  is the m-th largest key: build it bit by bit from the first bit the
      // keys disagree on (the bits above it are common to all of them)
      const uint32_t want = p.m < n ? p.m : n;
      const uint32_t diff = k_and ^ k_or;
      uint32_t T = k_and;                                     // common leading bits (and zeros below them, filled in next)
      if (diff) {
        const int top = 31 - __builtin_clz(diff);
        T = k_and & ~((2u << top) - 1u);                      // keep only the bits above the first disagreement
        // 14 bits below the first disagreement are plenty (2^-14 of the sample's score spread); the bits left
        // at zero make T a little LOWER than the m-th largest key, which is the safe side
        const int last = top > 13 ? top - 13 : 0;
        for (int b = top; b >= last; --b) {
          const uint32_t cand = T | (1u << b);
          uint32_t cnt = 0;
          for (uint32_t i = lane; i < n; i += 512) {           // (n is a multiple of 512: eight LDS reads in flight)
            uint32_t kk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) kk[u] = keys[i + 64 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) cnt += kk[u] >= cand ? 1u : 0u;
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o, 64);
          if (cnt >= want) T = cand;
        }
      }
      bits32 = __builtin_bit_cast(uint32_t, fz_unkey(T));   // scores >= the m-th largest group maximum pass
    }
    if (lane == 0)
      __hip_atomic_store(p.tau64 + q, ((unsigned long long)p.gen << 32) | (unsigned long long)bits32, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    if (q == first) FZ_STAMP(trow, 3);
    if (complete) {
      // give the slots back as zeros (the sample is complete: nobody writes them any more)
      for (uint32_t i = 4 * lane; i < n; i += 4 * 64) coherent_store16(src + i, u32x4{0u, 0u, 0u, 0u});
    }
  }
  FZ_STAMP(trow, 4);
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
  // The threshold workgroups come FIRST in the grid: they are dispatched before the scan workgroups and so always
  // find a CU at once — in pipelined mode the previous search's select (64 workgroups, 128 KiB of LDS each) becomes
  // runnable about when this kernel starts and would otherwise take the free CUs for its first 30 us.
  if ((int)blockIdx.x < p.tau_wgs) {
    tau_role(p, (int)blockIdx.x, smem);
    return;
  }
  const int sblock = (int)blockIdx.x - p.tau_wgs;   // index among the scan workgroups
  u32x4* qlds = reinterpret_cast<u32x4*>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int kg = p.sp.kg;
  const int64_t nwaves = (int64_t)p.scan_wgs * SCAN_WAVES;
  const int64_t w = (int64_t)sblock * SCAN_WAVES + wave;
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
    return j < my_sample ? sample_block(order, w + j * nwaves) : rest_block(order, w + (j - my_sample) * nwaves);
  };
  const bool active = my_items > 0;

  [[maybe_unused]] const int srow = w < 3000 ? (int)w : 1 << 20;   // (FZ_TRACE builds)
  FZ_STAMP(srow, 0);
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
  if (tid == 0) { st->cnt = 0; st->tau_flag = 0; st->arrived = 0; }
  // (the staging loop's loads sit under lane predicates: make the compiler's scoreboard forget them
  // here instead of in front of the first ring consumer)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  FZ_STAMP(srow, 1);

  bool failed = false;
  if (active) {
    float tau[QH];
#pragma unroll
    for (int hq = 0; hq < QH; ++hq) tau[hq] = 3.402823466e38f;
    unsigned long long tq[2] = {0ull, 0ull};   // (wave 0 only: the last look at the published thresholds)
    bool filtering = false;
    f32x16 parkA[QH], parkB[QH];         // score tiles waiting for the thresholds
    int64_t blkA = 0, blkB = 0;
    int npark = 0;
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
      // (still parking tiles: look at the thresholds now — the answer is consumed one ring's worth of MFMAs
      // later, ~2 us, instead of a whole block later)
      if (!filtering && wave == 0) tau_fetch<QH>(p, tq, lane);
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
        // the arrival hint (see tau_role): ONE global atomic per workgroup, by its last wave to get here
        if (j + 1 == my_sample && lane == 0 && atomicAdd(&st->arrived, 1u) == SCAN_WAVES - 1) atomicAdd(p.arrive, 1u);
      }
      if (j == 0) FZ_STAMP(srow, 2);
      if (j == 1) FZ_STAMP(srow, 7);
      if (!filtering) {
        if (wave == 0) (void)tau_publish_lds(p, st, tq, lane);
        filtering = tau_from_lds<QH>(st, tau, lane);
        if (!filtering && npark == 2) {
          // two tiles parked and a third one ready: the only place a wave waits in mid-stream
          filtering = tau_wait<QH>(p, st, tau, lane, wave);
          if (!filtering) { failed = true; break; }
        }
        if (filtering) {
          FZ_STAMP(srow, 3);
          if (npark >= 1) epilogue_filter<QH>(p.sp, st, parkA, tau, blkA, lane);
          if (npark == 2) epilogue_filter<QH>(p.sp, st, parkB, tau, blkB, lane);
          npark = 0;
        }
      }
      if (filtering) {
        epilogue_filter<QH>(p.sp, st, acc, tau, blk, lane);
      } else if (npark == 0) {
#pragma unroll
        for (int hq = 0; hq < QH; ++hq) parkA[hq] = acc[hq];
        blkA = blk;
        npark = 1;
      } else {
#pragma unroll
        for (int hq = 0; hq < QH; ++hq) parkB[hq] = acc[hq];
        blkB = blk;
        npark = 2;
      }
      blk = blkn;
      cur = nxt;
    }
    FZ_STAMP(srow, 4);
    if (!failed && !filtering) {
      filtering = tau_wait<QH>(p, st, tau, lane, wave);
      failed = !filtering;
      FZ_STAMP(srow, 3);
      if (filtering) {
        if (npark >= 1) epilogue_filter<QH>(p.sp, st, parkA, tau, blkA, lane);
        if (npark == 2) epilogue_filter<QH>(p.sp, st, parkB, tau, blkB, lane);
      }
    }
  }
  FZ_STAMP(srow, 5);
  if (__builtin_amdgcn_ballot_w64(failed) != 0ull && lane == 0)
    atomicOr(&p.sp.cand_cnt[0], 0x80000000u);   // "overflow": ts_index.hip redoes this batch exactly
  flush_stage(p.sp, st, tid);
  FZ_STAMP(srow, 6);
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
  // every workgroup gets the same dynamic LDS: the scan's query image + staging, or the threshold role's keys
  // threshold role: every wave that takes queries keeps its keys + a 256-bin histogram in LDS
  const size_t per_wave = (size_t)a.expect * 4;
  int tau_waves = (int)std::min<size_t>(SCAN_WAVES, (160 * 1024) / per_wave);
  if (tau_waves < 1) { ts_set_error("one-launch search: sample too large for LDS"); return TS_ERR_INVALID; }
  // (no more workgroups x waves than queries; at least enough to leave each wave one or two queries)
  while (tau_waves > 1 && (a.tau_wgs * (tau_waves - 1)) >= TS_MAX_Q) --tau_waves;
  p.tau_waves = tau_waves;
  const size_t lds = std::max(ts_scan_lds_bytes(L, qh), per_wave * tau_waves);
  if (a.expect > TS_FUSED_MAX_KEYS || (a.expect & 511u) || lds > 160 * 1024 || a.expect != 2 * a.n_sample ||
      (a.n_sample % SAMPLE_GROUP) != 0 || (a.n_sample / SAMPLE_GROUP) * a.sample_stride > a.nblk ||
      a.n_sample < SAMPLE_GROUP || a.sample_stride < SAMPLE_GROUP) {
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
