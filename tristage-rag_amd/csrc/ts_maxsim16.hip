// Stage-2 MaxSim on gfx950: the HBM-bound streaming form (f16 / bf16 token matrices, and f32
// ones on the exact-f32 MFMA; rows are addressed in bytes, a k step is 32 bytes of a row — the
// "16" in the names is the 16-byte operand unit every lane loads).
//
// Same scores as ts_maxsim.hip (reference src/stage2_rescorer.py:167-201 applied to
// every candidate, loop at :268-276):  for query tokens Q[Lq,H], document tokens D[Ld,H]
//   S = normalize(Q) . normalize(D)^T,  m_i = max_j S_ij,
//   maxsim = mean_i m_i            colbert = sum_i softmax(m)_i * m_i
// but organised like the stage-1 scan, because with a resident token store this step
// is a pure read of the candidates' token rows (C * Ld * H * 2 bytes per query, ~0.2-0.3 GB
// at C = 1000): the roofline that bounds it is HBM, not MFMA.  ONE launch per query:
//
//   * work item = 32 consecutive token rows of one candidate ("tile").  The tiles of all
//     candidates form one sequence; persistent wave w owns the contiguous slice
//     [w*T/W, (w+1)*T/W) of it, so every wave streams the same number of bytes (+-1 tile)
//     whatever the candidates' lengths are.  Every workgroup computes the tile prefix sums
//     of the (<= 4096) candidates itself, in LDS, while its query image is in flight — a
//     separate "prepare" launch cost 8 us + a kernel boundary on a ~40 us kernel.
//   * the token rows stay ROW-MAJOR where PyTorch wrote them.  A wave reads them
//     directly in MFMA A-fragment order (lane (r,h): 16 B of row r at byte 32g+16h):
//     32 rows x 32 B per instruction.  Plain loads (NOT nt: each 128-byte line is
//     touched by four consecutive instructions; with nt the line is refetched, measured
//     3.3 vs 6.2 TB/s, tools/bw_probe.hip "rowfrag"), a 16-deep register ring per wave,
//     refilled 8 slots at a time.
//   * the query tokens (<= 64 per pass) are the stationary operand: staged once per
//     workgroup into LDS in B-fragment order, lane-linear ds_read_b128.
//   * v_mfma_f32_32x32x16_{f16,bf16}: accumulator column <-> query token (lane),
//     accumulator row <-> document token (register), so max over the document's
//     tokens is an in-lane max over registers.  The rows' squared norms come from the
//     operand values already in registers (v_dot2c); 1/|d| reaches the accumulator
//     layout through a 128-byte wave-private LDS transpose.
//   * per (candidate, query token) maxima are combined across tiles in registers while
//     a wave stays inside one candidate, then across waves with one uint atomicMax per
//     lane (order-preserving float key) into a scratch array.  When a wave has finished
//     streaming it adds its tile counts to per-candidate arrival counters; the wave that
//     completes a candidate reads the maxima back, writes the score and zeroes the cells,
//     so the scratch is all-zero again when the launch ends (no init / finish launches).
#include "ts_common.h"
#include <algorithm>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifndef M16_THREADS
#define M16_THREADS 256  // 4 waves per workgroup, one workgroup per CU: fewer, fatter waves get more
#endif                   // tiles each, so the last, partly filled round of tiles weighs less
                         // (8 waves x 8 KiB: 51.6 us per 1000-candidate query, 4 x 16 KiB: 48.4)
#define M16_WAVES (M16_THREADS / 64)
#ifndef M16_RING
#define M16_RING 16     // k steps (1 KiB each) in the register ring of a wave
#endif
#ifndef M16_GROUP
#define M16_GROUP 8     // ring slots refilled together.  Issuing the loads that share a 128-byte line
                        // set back to back matters: refilling one slot per step (GROUP 1) re-misses
                        // L1 for 3 of the 4 touches of a line: 4.35 TB/s vs 4.76 (GROUP 4) vs 5.09
                        // (GROUP 8) on 600 MB of candidates (tools/variants_maxsim.sh)
#endif
#define M16_NEG (-3.402823466e38f)

__device__ const u32x4 m16_zero16 = {0u, 0u, 0u, 0u};   // what the query image's padding lanes load

#define M16_MAX_DOCS 4096   // candidates per launch (their tile prefix sums live in LDS)
#define M16_MAX_BATCH 64    // queries per launch

struct Ms16Params {
  const unsigned char* q;  // [Lq, H] (row_bytes per row)
  int Lq, H;               // H is carried as row_bytes = H * element size: a k step is 32 bytes of a row
                           // (2 x 16-byte operand units) for every element type
  int s_pad;               // k steps per tile (16 elements each), multiple of M16_RING
  int lq_pad;              // row stride of `best` (= passes * NQT * 32)
  int passes;              // gridDim.y: query tokens are taken NQT*32 per pass
  const unsigned char* docs;  // [rows, H] row-major
  const int32_t* doc_off;  // packed candidates: [n_docs+1] (or null)
  const int64_t* starts;   // token store: [n_docs]
  const int32_t* lens;     //              [n_docs]
  int n_docs;              // <= M16_MAX_DOCS
  // several queries in one launch (blockIdx.z = query): query j owns query tokens
  // [q_off[j], q_off[j+1]) of q and candidates [cand_off[j], cand_off[j+1]) of starts/lens/out
  // (offsets travel in the kernel arguments: captured at launch, no host->device copy whose
  // source the caller could free too early)
  int nq;                             // 0 = single query (q / Lq / n_docs above are final)
  int32_t q_off[M16_MAX_BATCH + 1];
  int32_t cand_off[M16_MAX_BATCH + 1];
  // scratch, all-zero between launches
  uint32_t* cnt;           // [n_docs] tiles (x passes) that have been folded into `best`
  uint32_t* best;          // [n_docs][lq_pad] ordered-uint keys of max_j cos(q_i, d_j); 0 = none
  int mode;
  float* out;
  int eq_slices;           // single-query launches: equal tile counts on fewer waves (see "this wave's slice")
};

// Wave-uniform table reads inside the streaming loop must be SCALAR loads: as vector loads
// they share vmcnt with the ring and every candidate change would wait for vmcnt(0), i.e.
// drain the ring (the compiler does not pick s_load by itself once the loop holds atomics).
__device__ __forceinline__ int32_t m16_sload32(const int32_t* ptr) {
  int32_t v;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ptr) : "memory");
  return v;
}
__device__ __forceinline__ int64_t m16_sload64(const int64_t* ptr) {
  int64_t v;
  asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(ptr) : "memory");
  return v;
}
__device__ __forceinline__ int m16_len_s(const Ms16Params& p, int d) {
  return p.starts ? m16_sload32(p.lens + d) : (m16_sload32(p.doc_off + d + 1) - m16_sload32(p.doc_off + d));
}
__device__ __forceinline__ int64_t m16_start_s(const Ms16Params& p, int d) {
  return p.starts ? m16_sload64(p.starts + d) : (int64_t)m16_sload32(p.doc_off + d);
}
__device__ __forceinline__ int m16_len(const Ms16Params& p, int d) {
  return p.starts ? p.lens[d] : (p.doc_off[d + 1] - p.doc_off[d]);
}
// tiles of a candidate of `len` token rows.  Capped (8 M tokens) so that the int32 sums over
// <= 4096 candidates cannot overflow whatever a caller's `lens` array holds.
__device__ __forceinline__ int m16_tiles(int len) {
  return len > 0 ? min((len + 31) / 32, 1 << 18) : 0;
}
// float -> uint whose unsigned order is the float order; 0 is below every finite value
__device__ __forceinline__ uint32_t m16_key(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float m16_unkey(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// sum of squares of the 8 values of one operand register quad.  (Whole-vector casts and
// constant shuffles only: hipcc 7.2 miscompiles __builtin_bit_cast of a[i] inside an
// unrolled loop — every iteration reads element 0; see DESIGN.md "Notes".)
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int DT> __device__ __forceinline__ float m16_sumsq(const u32x4& a, float s) {
  if constexpr (DT == TS_F32) {
    const f32x4v v = __builtin_bit_cast(f32x4v, a);
    s = fmaf(v[0], v[0], s); s = fmaf(v[1], v[1], s); s = fmaf(v[2], v[2], s); s = fmaf(v[3], v[3], s);
  } else if constexpr (DT == TS_F16) {
    const h8 v = __builtin_bit_cast(h8, a);
    const h2 p0 = __builtin_shufflevector(v, v, 0, 1), p1 = __builtin_shufflevector(v, v, 2, 3);
    const h2 p2 = __builtin_shufflevector(v, v, 4, 5), p3 = __builtin_shufflevector(v, v, 6, 7);
    s = __builtin_amdgcn_fdot2(p0, p0, s, false);
    s = __builtin_amdgcn_fdot2(p1, p1, s, false);
    s = __builtin_amdgcn_fdot2(p2, p2, s, false);
    s = __builtin_amdgcn_fdot2(p3, p3, s, false);
  } else {
    const bf8 v = __builtin_bit_cast(bf8, a);
    const bf2 p0 = __builtin_shufflevector(v, v, 0, 1), p1 = __builtin_shufflevector(v, v, 2, 3);
    const bf2 p2 = __builtin_shufflevector(v, v, 4, 5), p3 = __builtin_shufflevector(v, v, 6, 7);
    s = __builtin_amdgcn_fdot2_f32_bf16(p0, p0, s, false);
    s = __builtin_amdgcn_fdot2_f32_bf16(p1, p1, s, false);
    s = __builtin_amdgcn_fdot2_f32_bf16(p2, p2, s, false);
    s = __builtin_amdgcn_fdot2_f32_bf16(p3, p3, s, false);
  }
  return s;
}
template <int DT>
__device__ __forceinline__ void m16_mma(f32x16& acc, const u32x4& a, const u32x4& b) {
  if constexpr (DT == TS_F32) {
    // fp32 rows: the unit holds 4 consecutive floats; the t-th exact-f32 MFMA (K = 2: this
    // lane's value and its h-partner's) contracts k = 8g + t and 8g + 4 + t — any pairing is
    // fine as long as the query image uses the same one, which it does by construction
    const f32x4v af = __builtin_bit_cast(f32x4v, a), bf = __builtin_bit_cast(f32x4v, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], acc, 0, 0, 0);
  } else if constexpr (DT == TS_F16)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), acc, 0, 0, 0);
  else
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc, 0, 0, 0);
}

// ---------------------------------------------------------------------------------
// 16 bytes of this lane's row for k step g (k = 16g + 8h .. +8).  The load is ALWAYS issued
// (a load under a lane predicate becomes a branch with s_waitcnt vmcnt(0) behind it, which
// drains the ring): steps past H re-read unit 0 of the row and are zeroed when consumed.
template <bool FULL>
__device__ __forceinline__ u32x4 m16_load(const unsigned char* rowp, int g, int h, int row_bytes) {
  if constexpr (FULL) return *reinterpret_cast<const u32x4*>(rowp + 32 * g);
  const int off = (32 * g + 16 * h < row_bytes) ? 32 * g : 0;
  return *reinterpret_cast<const u32x4*>(rowp + off);
}
template <bool FULL>
__device__ __forceinline__ u32x4 m16_use(const u32x4& v, int g, int h, int row_bytes) {
  if constexpr (FULL) return v;
  const bool ok = 32 * g + 16 * h < row_bytes;
  return u32x4{ok ? v[0] : 0u, ok ? v[1] : 0u, ok ? v[2] : 0u, ok ? v[3] : 0u};
}

// ---------------------------------------------------------------------------------
// The wave that completes candidate d: m_i -> score, then give the cells back as zeros.
__device__ __forceinline__ void m16_finish_doc(const Ms16Params& p, int d, int lane) {
  uint32_t* b = p.best + (size_t)d * p.lq_pad;
  float res;
  if (p.mode == 0) {
    float s = 0.f;
    for (int i = lane; i < p.Lq; i += 64)
      s += m16_unkey(__hip_atomic_load(b + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    res = s / (float)p.Lq;
  } else {
    float mx = M16_NEG;
    for (int i = lane; i < p.Lq; i += 64)
      mx = fmaxf(mx, m16_unkey(__hip_atomic_load(b + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float num = 0.f, den = 0.f;
    for (int i = lane; i < p.Lq; i += 64) {
      const float v = m16_unkey(__hip_atomic_load(b + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      const float e = expf(v - mx);
      den += e;
      num += e * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      num += __shfl_xor(num, o, 64);
      den += __shfl_xor(den, o, 64);
    }
    res = num / den;
  }
  if (lane == 0) p.out[d] = res;
  for (int i = lane; i < p.lq_pad; i += 64) __hip_atomic_store(b + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (lane == 0) __hip_atomic_store(p.cnt + d, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Candidates this wave has left (lane i holds record i): add their tile counts to the arrival
// counters; whoever brings a counter to its target finishes that candidate.
__device__ __forceinline__ void m16_flush_records(const Ms16Params& p, int nrec, int rdoc,
                                                  uint32_t rtiles, uint32_t rneed, int lane, uint32_t sink) {
  // "My maxima are performed" must precede "my counts are performed".  A release fence at agent
  // scope would do it inside the HIP memory model, but it is buffer_wbl2 sc1 — a write-back of the
  // whole L2 — and 2048 waves executing it made the kernel 3.5x slower (186 vs 53 us).  Instead every
  // atomicMax of this wave is a RETURNING atomic and `sink` is a value computed from all the
  // returned words: an atomic's return value comes from the place where the read-modify-write was
  // performed (for agent scope: memory-side, the one copy all 8 XCDs see), so once `sink` exists
  // every maximum of this wave has been performed, and the atomicAdd below — which consumes
  // `sink` — cannot issue earlier.  The finishing wave's own returning atomicAdd closes the chain:
  // the count it sees includes only adds issued after their waves' maxima were performed, and it
  // reads the maxima with agent-scope atomic loads issued after that add returned.  (Every word
  // involved is only ever touched by agent-scope atomics: no cached copy can go stale.)
  // (`sink` is an input of this empty statement: the compiler has to wait for the returns that feed it
  // before this point, and the atomicAdd below — whose operand passes through it — is issued after it)
  asm volatile("" : "+v"(rtiles) : "v"(sink) : "memory");
  uint32_t old = 0;
  if (lane < nrec) old = atomicAdd(p.cnt + rdoc, rtiles);
  const bool last = lane < nrec && old + rtiles == rneed;  // (consuming `old` waits for the atomic)
  uint64_t mask = __builtin_amdgcn_ballot_w64(last);
  asm volatile("" ::: "memory");
  while (mask) {
    const int bit = __builtin_ctzll(mask);
    mask &= mask - 1;
    m16_finish_doc(p, __builtin_amdgcn_readlane(rdoc, bit), lane);  // agent-scope atomic loads / stores
  }
}

#if defined(TS_TUNING) && defined(M16_TRACE)  // diagnostic builds only: per-wave phase time stamps (100 MHz)
__device__ unsigned long long m16_trace_buf[4096 * 8];
#define M16_STAMP(i) do { if (lane == 0 && gwt < 4096) m16_trace_buf[gwt * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int ts_debug_m16_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(m16_trace_buf), sizeof(m16_trace_buf)) == hipSuccess ? 0 : -2;
}
#else
#define M16_STAMP(i) do { } while (0)
#endif

// one k step: the row norms and NQT MFMAs on operand `a` of step g
#if defined(TS_TUNING) && defined(M16_DBG_NOCOMPUTE)  // ablation builds only: loads kept live, no math
#define M16_STEP(a, g) do { dsq += __uint_as_float(((a)[0] ^ (a)[1] ^ (a)[2] ^ (a)[3]) & 0x007fffffu); } while (0)
#else
#define M16_STEP(a, g)                                                                          \
  do {                                                                                          \
    dsq = m16_sumsq<DT>((a), dsq);                                                              \
    _Pragma("unroll") for (int t = 0; t < NQT; ++t)                                             \
        m16_mma<DT>(acc[t], (a), ql[(size_t)((g) * NQT + t) * 64]);                             \
  } while (0)
#endif

// SINGLE: the launch scores ONE query (its prologue is on the critical path: the query image comes by LDS-DMA before
// the ring's first loads, the slices are equal); otherwise many queries share the launch, a workgroup's prologue runs
// beside its neighbours' streaming, and the ring's loads go out first (measured both ways: the single-query order costs
// the 64-query launch 3-4 %, the batch order costs the single query ~2 us).
template <int DT, int NQT, bool FULL, bool SINGLE>
__global__ __launch_bounds__(M16_THREADS) void maxsim16_kernel(Ms16Params pin) {
  Ms16Params p = pin;
  if (pin.nq) {  // one of several queries: narrow every array to this query's part
    const int qj = blockIdx.z;
    const int qa = pin.q_off[qj], ca = pin.cand_off[qj];
    p.Lq = pin.q_off[qj + 1] - qa;
    p.n_docs = pin.cand_off[qj + 1] - ca;
    p.q = pin.q + (size_t)qa * pin.H;   // (H = row bytes)
    p.starts += ca; p.lens += ca; p.out += ca;
    p.cnt += ca;
    p.best += (size_t)ca * pin.lq_pad;
    p.passes = (p.Lq + NQT * 32 - 1) / (NQT * 32);
    if ((int)blockIdx.y >= p.passes || p.n_docs <= 0) return;  // (uniform for the workgroup)
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* qlds = reinterpret_cast<u32x4*>(smem);                                  // [s_pad][NQT][64]
  float* invl = reinterpret_cast<float*>(smem + (size_t)p.s_pad * NQT * 1024);    // [waves][32]
  int32_t* wsum = reinterpret_cast<int32_t*>(invl + M16_WAVES * 32);              // [waves] (+pad to 64 B)
  int32_t* prefix = wsum + 16;                                                    // [n_docs+1]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.y * (NQT * 32);  // first query token of this pass
  const int H = p.H;
  const int S = p.s_pad;
#if defined(TS_TUNING) && defined(M16_TRACE)
  const int gwt = (tid >> 6) * gridDim.x + blockIdx.x;
#endif
  M16_STAMP(0);

  // ---- tiles per candidate of this thread's share of the candidates (loads go out first)
  const int per = (p.n_docs + M16_THREADS - 1) / M16_THREADS;   // <= M16_MAX_DOCS / M16_THREADS
  const int d0 = min(tid * per, p.n_docs), d1 = min(d0 + per, p.n_docs);
  int mytiles = 0;
  for (int d = d0; d < d1; ++d) {
    const int len = m16_len(p, d);
    mytiles += m16_tiles(len);
    if (len <= 0 && blockIdx.x == 0 && blockIdx.y == 0)
      p.out[d] = 0.f;  // reference: a candidate that cannot be scored keeps 0.0 (:285-291)
  }

  // ---- exclusive prefix sums of tiles per candidate -> LDS (wave scan + the waves' totals).  This
  // comes FIRST: the wave's first tile is known ~2 us after the kernel starts, its ring loads go
  // out, and the query image below is staged while they cross the HBM latency (it used to be
  // staged first: first loads in flight at 7 us, tools/trace_maxsim.py).
  int incl = mytiles;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = 0, T = 0;
#pragma unroll
  for (int w = 0; w < M16_WAVES; ++w) {
    const int x = wsum[w];
    base += (w < wave) ? x : 0;
    T += x;
  }
  {
    int run = base + incl - mytiles;
    for (int d = d0; d < d1; ++d) {
      prefix[d] = run;
      const int len = m16_len(p, d);  // (L1/L2 hit: read a moment ago)
      run += m16_tiles(len);
    }
    if (tid == 0) prefix[p.n_docs] = T;
  }
  __syncthreads();
  M16_STAMP(1);

  // ---- this wave's slice of the tile sequence (wave-major numbering spreads the longer
  // slices over all workgroups)
  const int64_t n_waves = (int64_t)gridDim.x * M16_WAVES;
  int64_t gw = (int64_t)wave * gridDim.x + blockIdx.x;
  int64_t lo = gw * T / n_waves, hi = (gw + 1) * T / n_waves;
  if (SINGLE && p.eq_slices) {
    // ONE query: T / n_waves is a small number (4.3 at 1000 candidates on 1024 waves), so a third of the waves own one
    // tile more than the rest and stream it alone at the rate ONE ring sustains while the chip idles (slices done at
    // 32-44 us, tools/trace_maxsim.py).  The stream itself is HBM-bound, not wave-bound — so use FEWER waves, each with
    // the same t = ceil(T / n_waves) tiles: ceil(T / t) waves on the first workgroups (880 of 1024 at 5 tiles each),
    // the others leave.  Only while >= 3/4 of the waves stay (below that the launch is latency-bound anyway).
    const int64_t t = (T + n_waves - 1) / n_waves;
    const int64_t w_eff = t > 0 ? (T + t - 1) / t : 0;
    if (t > 0 && t <= 10 && 4 * w_eff >= 3 * n_waves) {   // (A/B at 600 ... 4000 candidates, tools/sessions/r03_maxsim_ab.sh: -3 % at 1000, +-1 % at 600 / 1500 / 2000, +1.5 % at 4000)
      const int64_t wg_eff = (w_eff + M16_WAVES - 1) / M16_WAVES;
      if ((int64_t)blockIdx.x >= wg_eff) return;     // (uniform for the workgroup; BEFORE the image's LDS-DMA is requested: nothing in flight, no barrier pending)
      gw = (int64_t)wave * wg_eff + blockIdx.x;
      lo = gw < w_eff ? gw * t : T;
      hi = lo + t < T ? lo + t : T;
    }
  }
  const bool has_work = lo < hi;   // (a wave without tiles still stages its share of the query image)

  // ---- Q image: unit (g*NQT + t)*64 + l = the 16 bytes at byte 32g + 16(l>>5) of query row q0 + 32t + (l&31), by
  // LDS-DMA (global_load_lds_dwordx4: per-lane source address, destination = wave-uniform LDS base + lane * 16 — a
  // wave's 64 lanes fill 64 consecutive units), requested HERE, before the slice search and long before the ring:
  // no registers are parked, the L2 round trips run under the binary search, and the image is not queued behind the
  // first 16 MB of ring loads.  (Round 2 requested it after the ring — complete at 8.3 us, every wave idle on a full
  // ring for ~2 us; through registers before the ring it cost the ring 3 us; parked in registers across the search it
  // made the compiler re-order the ring's refills.  tools/trace_maxsim.py.)  Lanes outside the image's valid part
  // (query tokens >= Lq, bytes >= H) read 16 zero bytes.
  if constexpr (SINGLE) {
    const int units = S * NQT * 64;
    for (int u0 = wave * 64; u0 < units; u0 += M16_THREADS) {     // (uniform per wave: units is a multiple of 64)
      const int u = u0 + lane;
      const int l = u & 63, t = (u >> 6) % NQT, g = (u >> 6) / NQT;
      const int qi = q0 + 32 * t + (l & 31), k = 32 * g + 16 * (l >> 5);   // k: byte offset in the row
      const bool ok = qi < p.Lq && k < H;
      const unsigned char* src = ok ? p.q + (size_t)qi * H + k : reinterpret_cast<const unsigned char*>(&m16_zero16);
      // (inline asm, not __builtin_amdgcn_global_load_lds: the compiler's waitcnt pass books the builtin as a pending
      // "flat" access that only a vmcnt(0) it can see clears, and until then turns EVERY vector-memory wait into
      // vmcnt(0) — with the partial wait below that is the tile loop's vmcnt(15..9) ladder, i.e. the ring drained once
      // per tile.  Unknown to the compiler the requests are harmless: they are older than every load it counts and
      // vmcnt retires in order, so its waits can only come out stricter, never too lenient.)
      const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(qlds + u0);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // (m0 is "reserved": it is the instruction's LDS base, and is named so the compiler knows it changed)
      __asm__ volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off"
                       :: "s"(__builtin_amdgcn_readfirstlane(lds_base)), "v"(src) : "m0", "memory");
#pragma clang diagnostic pop
    }
  }


  int doc = 0, tile = 0, len = 1;
  int64_t start = 0;
  if (has_work) {
    // the last candidate a with prefix[a] <= lo (invariant prefix[a] <= lo < prefix[b]), 64 probes per LDS round trip:
    // two round trips up to 4096 candidates where a bisection takes twelve dependent ones (~1 us of the prologue)
    int a = 0, b = p.n_docs;
    while (b - a > 1) {
      const int step = (b - a + 63) >> 6;
      const int m = a + lane * step;
      const bool le = m < b && (int64_t)prefix[min(m, b)] <= lo;   // (true on a prefix of the lanes: prefix[] is non-decreasing; lane 0 by the invariant)
      const int cnt = __builtin_popcountll(__ballot(le));
      a = a + (cnt - 1) * step;
      b = min(a + step, b);
    }
    doc = __builtin_amdgcn_readfirstlane(a);
    tile = __builtin_amdgcn_readfirstlane((int)(lo - prefix[a]));
    len = m16_len_s(p, doc);
    start = m16_start_s(p, doc);
  }
  const unsigned char* cur = p.docs;
  u32x4 ring[M16_RING];
  if (has_work) {
    const int rows = min(32, len - tile * 32);
    cur = p.docs + ((size_t)(start + tile * 32 + min(r, rows - 1)) * H + 16 * h);
#pragma unroll
    for (int i = 0; i < M16_RING; ++i) ring[i] = m16_load<FULL>(cur, i, h, H);
  }
  M16_STAMP(2);
  if constexpr (SINGLE) {
    // the query image requested above (LDS-DMA) has to be complete, for every wave of the workgroup, before the first
    // tile is multiplied.  Its requests are OLDER than the ring's M16_RING loads and vmcnt retires in order, so
    // vmcnt(M16_RING) is exactly "the image has landed" and the ring stays in flight across the barrier.
    static_assert(M16_RING == 16, "the wait below is vmcnt(16)");
    if (has_work) __builtin_amdgcn_s_waitcnt(0x4F70);  // vmcnt(16): [3:0] = 0, [15:14] = 1; expcnt/lgkmcnt untouched
    else __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): no ring behind the image
    // A bare s_barrier: __syncthreads() (and a workgroup fence, even an LDS-only one) put vmcnt(0) in front of the
    // barrier = the ring drained.  What the barrier orders here is only the image's LDS-DMA writes (complete: the
    // vmcnt wait above, in every wave) against the ds_reads below; the asm clobbers keep the compiler from moving
    // LDS accesses across it.
    __asm__ volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __asm__ volatile("" ::: "memory");
  }
  if constexpr (!SINGLE) {
    // ---- Q image through registers, AFTER the ring's first loads (a launch of many queries: other workgroups stream
    // meanwhile): 8 independent L2 reads in flight per thread (one at a time costs ~1 us each)
    const int units = S * NQT * 64;
    for (int u0 = tid; u0 < units; u0 += 8 * M16_THREADS) {
      u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int u = u0 + j * M16_THREADS;
        const int l = u & 63, t = (u >> 6) % NQT, g = (u >> 6) / NQT;
        const int qi = q0 + 32 * t + (l & 31), k = 32 * g + 16 * (l >> 5);   // k: byte offset in the row
        const bool ok = u < units && qi < p.Lq && k < H;
        // always-issued load from a valid address, zeroed afterwards (no branch around the load)
        const u32x4 x = *reinterpret_cast<const u32x4*>(p.q + (ok ? (size_t)qi * H + k : 0));
        v[j] = u32x4{ok ? x[0] : 0u, ok ? x[1] : 0u, ok ? x[2] : 0u, ok ? x[3] : 0u};
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int u = u0 + j * M16_THREADS;
        if (u < units) qlds[u] = v[j];
      }
    }
    // Explicit vmcnt(0): the stores above sit under a lane predicate, so on their skip path the compiler's scoreboard
    // still counts the staging loads as pending and would put a vmcnt(0) in front of the first reuse of their
    // registers — inside the tile loop, where it drains the ring once per tile.  (The ring's first loads are older
    // than the staging loads and are needed next anyway.)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched
    __syncthreads();
  }

  if (!has_work) return;  // (no block-level barrier below)

  M16_STAMP(3);
  const u32x4* ql = qlds + lane;
  // ---- 1/|q_i| of this lane's query token(s), from the image (fixed order: deterministic)
  float invq[NQT];
  {
    float qsq[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t) qsq[t] = 0.f;
    for (int g = 0; g < S; ++g)
#pragma unroll
      for (int t = 0; t < NQT; ++t) qsq[t] = m16_sumsq<DT>(ql[(size_t)(g * NQT + t) * 64], qsq[t]);
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
      qsq[t] += __shfl_xor(qsq[t], 32, 64);
      invq[t] = 1.0f / fmaxf(sqrtf(qsq[t]), 1e-12f);
    }
  }

  M16_STAMP(4);
  float* myinv = invl + wave * 32;
  float best[NQT];
#pragma unroll
  for (int t = 0; t < NQT; ++t) best[t] = M16_NEG;
  // candidates left behind: lane i keeps record i
  int nrec = 0, rdoc = 0;
  uint32_t rtiles = 0, rneed = 0, run_tiles = 0;
  uint32_t sink = 0;        // folds the words returned by this wave's atomicMax (see m16_flush_records)
  bool from_start = (tile == 0);   // this wave has seen the current candidate from its first tile on

  for (int64_t it = lo; it < hi; ++it) {
    // ---- next item of the slice (wave-uniform)
    const bool has_next = it + 1 < hi;
    int ndoc = doc, ntile = tile + 1, nlen = len;
    int64_t nstart = start;
    if (has_next && ntile >= m16_tiles(len)) {
      ntile = 0;
      ndoc = __builtin_amdgcn_readfirstlane(ndoc);
      do { ++ndoc; nlen = m16_len_s(p, ndoc); } while (nlen <= 0);  // a later tile exists: terminates
      nstart = m16_start_s(p, ndoc);
    }
    const unsigned char* nxt = cur;
    if (has_next) {
      const int nrows = min(32, nlen - ntile * 32);
      nxt = p.docs + ((size_t)(nstart + ntile * 32 + min(r, nrows - 1)) * H + 16 * h);
    }

    f32x16 acc[NQT];
#pragma unroll
    for (int t = 0; t < NQT; ++t)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[t][x] = 0.f;
    float dsq = 0.f;

    int g0 = 0;
    for (; g0 < S - M16_RING; g0 += M16_RING) {
#pragma unroll
      for (int i = 0; i < M16_RING; ++i) {
        const u32x4 a = m16_use<FULL>(ring[i], g0 + i, h, H);
        M16_STEP(a, g0 + i);
        if ((i + 1) % M16_GROUP == 0) {  // refill the group of slots just consumed
#pragma unroll
          for (int j = i + 1 - M16_GROUP; j <= i; ++j) ring[j] = m16_load<FULL>(cur, g0 + j + M16_RING, h, H);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < M16_RING; ++i) {  // tail: refill from the next tile
      const u32x4 a = m16_use<FULL>(ring[i], g0 + i, h, H);
      M16_STEP(a, g0 + i);
      if ((i + 1) % M16_GROUP == 0 && has_next) {  // (wave-uniform: nothing is read past the slice)
#pragma unroll
        for (int j = i + 1 - M16_GROUP; j <= i; ++j) ring[j] = m16_load<FULL>(nxt, j, h, H);
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- 1/|d_j| of the tile's rows: lane (r, *) knows row r, accumulator register x of
    // a lane holds row (x&3) + 8(x>>2) + 4h -> transpose through wave-private LDS
    dsq += __shfl_xor(dsq, 32, 64);
    if (h == 0) myinv[r] = 1.0f / fmaxf(sqrtf(dsq), 1e-12f);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int x4 = 0; x4 < 4; ++x4) {
      const float4 iv = *reinterpret_cast<const float4*>(myinv + 8 * x4 + 4 * h);
#pragma unroll
      for (int t = 0; t < NQT; ++t) {
        const float a0 = acc[t][4 * x4 + 0] * iv.x, a1 = acc[t][4 * x4 + 1] * iv.y;
        const float a2 = acc[t][4 * x4 + 2] * iv.z, a3 = acc[t][4 * x4 + 3] * iv.w;
        best[t] = fmaxf(best[t], fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    ++run_tiles;
    if (it == lo) M16_STAMP(5);

    // ---- leaving this candidate (or the slice)
    if (!has_next || ndoc != doc) {
      const bool to_end = tile + 1 >= m16_tiles(len);   // its last tile was this wave's
      if (from_start && to_end && p.passes == 1) {
        // The whole candidate went through THIS wave: its maxima are complete in registers — score it
        // here, no scratch, no atomics.  Same values and the same summation tree as m16_finish_doc (lane L
        // holds query token L: token 32t + r sits in lane (r, h = t)), hence bit-identical scores.
        float v = 0.f;
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
          const float m = fmaxf(best[t], __shfl_xor(best[t], 32, 64)) * invq[t];
          if (h == t && 32 * t + r < p.Lq) v = m;
          best[t] = M16_NEG;
        }
        float res;
        const bool has = (32 * h + r < p.Lq) && (h < NQT);
        if (p.mode == 0) {
          float sacc = has ? v : 0.f;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 64);
          res = sacc / (float)p.Lq;
        } else {
          float mx = has ? v : M16_NEG;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
          const float e = has ? expf(v - mx) : 0.f;
          float den = e, num = has ? e * v : 0.f;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            num += __shfl_xor(num, o, 64);
            den += __shfl_xor(den, o, 64);
          }
          res = num / den;
        }
        if (lane == 0) p.out[doc] = res;
      } else {
        // shared with other waves (or other passes): publish the maxima, remember the candidate
#pragma unroll
        for (int t = 0; t < NQT; ++t) {
          const float m = fmaxf(best[t], __shfl_xor(best[t], 32, 64)) * invq[t];
          const int qi = q0 + 32 * t + r;
          if (h == 0 && qi < p.Lq)
            sink |= atomicMax(p.best + (size_t)doc * p.lq_pad + qi, m16_key(m));   // RETURNING: see m16_flush_records
          best[t] = M16_NEG;
        }
        if (lane == nrec) {
          rdoc = doc;
          rtiles = run_tiles;
          rneed = (uint32_t)p.passes * (uint32_t)m16_tiles(len);
        }
        if (++nrec == 64) {  // (only with very many tiny candidates per wave)
          m16_flush_records(p, nrec, rdoc, rtiles, rneed, lane, sink);
          nrec = 0;
        }
      }
      run_tiles = 0;
      from_start = true;   // the next candidate (if any) starts at its tile 0
    }
    doc = ndoc; tile = ntile; len = nlen; start = nstart; cur = nxt;
  }
  M16_STAMP(6);
  if (nrec) m16_flush_records(p, nrec, rdoc, rtiles, rneed, lane, sink);
  M16_STAMP(7);
}

// ---------------------------------------------------------------------------------
// scratch per (device, stream): calls on one stream are ordered, so the buffer can be
// reused; calls on different streams get different buffers.  Invariant: all-zero whenever
// no launch is pending on it (the kernel returns every cell it used to zero).
#include <map>
#include <mutex>
#include <stdlib.h>
#include <utility>
namespace {
struct Scratch { void* ptr = nullptr; size_t bytes = 0; };
std::mutex g_mu;
std::map<std::pair<int, hipStream_t>, Scratch> g_scratch;
int g_cus[64];

// (callers hold g_mu from here until their launches are enqueued: another thread growing the
// same stream's buffer must not free it between this call and the launch that uses it)
int scratch_get(int device, hipStream_t stream, size_t bytes, void** out) {
  Scratch& s = g_scratch[{device, stream}];
  if (s.bytes < bytes) {
    if (s.ptr) {
      TS_HIP(hipStreamSynchronize(stream));  // earlier launches may still use it
      TS_HIP(hipFree(s.ptr));
      s.ptr = nullptr; s.bytes = 0;
    }
    const size_t want = bytes + bytes / 2 + 4096;
    TS_HIP(hipMalloc(&s.ptr, want));
    s.bytes = want;
    TS_HIP(hipMemsetAsync(s.ptr, 0, want, stream));
  }
  *out = s.ptr;
  return TS_OK;
}
}  // namespace

// frees the (device, stream) scratch buffers of `device` (all of them for device < 0); the caller
// guarantees no MaxSim launch is pending there
extern "C" int ts_maxsim_release_scratch(int32_t device) {
  std::lock_guard<std::mutex> lk(g_mu);
  int prev = -1;
  (void)hipGetDevice(&prev);
  for (auto it = g_scratch.begin(); it != g_scratch.end();) {
    if (device < 0 || it->first.first == device) {
      if (it->second.ptr) {
        (void)hipSetDevice(it->first.first);
        (void)hipStreamSynchronize(it->first.second);
        (void)hipFree(it->second.ptr);
      }
      it = g_scratch.erase(it);
    } else {
      ++it;
    }
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  return TS_OK;
}

template <int DT, int NQT, bool FULL, bool SINGLE>
static int launch_main_s(const Ms16Params& p, int grid, size_t lds, hipStream_t s, int nq) {
  auto kern = maxsim16_kernel<DT, NQT, FULL, SINGLE>;
  static TsDeviceOnce lds_attr;  // per instantiation, per device (ts_common.h)
  TS_CHECK(ts_allow_max_lds(lds_attr, reinterpret_cast<const void*>(kern)));
  hipLaunchKernelGGL(kern, dim3(grid, p.passes, nq), dim3(M16_THREADS), lds, s, p);
  TS_HIP(hipGetLastError());
  return TS_OK;
}
template <int DT, int NQT, bool FULL>
static int launch_main(const Ms16Params& p, int grid, size_t lds, hipStream_t s, int nq = 1) {
  return nq <= 1 ? launch_main_s<DT, NQT, FULL, true>(p, grid, lds, s, nq) : launch_main_s<DT, NQT, FULL, false>(p, grid, lds, s, nq);
}

// returns TS_ERR_UNSUPPORTED (without setting an error) when the shape is not one this
// kernel takes; the caller then uses the general kernel of ts_maxsim.hip
int ts_launch_maxsim16(const void* q, int Lq, const void* docs, const int32_t* doc_off,
                       const int64_t* starts, const int32_t* lens, int n_docs, int H, int dtype,
                       int mode, float* out, int device, hipStream_t stream) {
  if (dtype != TS_F16 && dtype != TS_BF16 && dtype != TS_F32) return TS_ERR_UNSUPPORTED;
  const int row_bytes = H * (dtype == TS_F32 ? 4 : 2);
  if (Lq <= 0 || n_docs <= 0 || (row_bytes % 16) != 0) return TS_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(docs)) & 15) return TS_ERR_UNSUPPORTED;
  const int s_real = (row_bytes + 31) / 32;
  const int s_pad = ((s_real + M16_RING - 1) / M16_RING) * M16_RING;
  const size_t lds_cap = 156 * 1024;
  const size_t extra = M16_WAVES * 32 * sizeof(float) + 64 + ((size_t)M16_MAX_DOCS + 1) * 4 + 12;
  const int nqt = (Lq > 32 && (size_t)s_pad * 2 * 1024 + extra <= lds_cap) ? 2 : 1;
  const size_t lds = (size_t)s_pad * nqt * 1024 + extra;
  if (lds > lds_cap) return TS_ERR_UNSUPPORTED;
  const int passes = (Lq + nqt * 32 - 1) / (nqt * 32);
  const int lq_pad = passes * nqt * 32;

  if (device < 0 || device >= 64) return TS_ERR_UNSUPPORTED;
  if (g_cus[device] == 0) {
    int n = 0;
    TS_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device));
    g_cus[device] = n > 0 ? n : 256;
  }
  int grid = g_cus[device];
#ifdef TS_TUNING
  if (const char* e = getenv("TS_M16_GRIDMUL")) grid = (int)(grid * atof(e));
#endif

  Ms16Params p;
  p.q = (const unsigned char*)q; p.Lq = Lq; p.H = row_bytes; p.s_pad = s_pad; p.lq_pad = lq_pad; p.passes = passes;
  p.docs = (const unsigned char*)docs; p.mode = mode; p.nq = 0;
  p.eq_slices = 1;
#ifdef TS_TUNING
  if (getenv("TS_M16_NO_EQ")) p.eq_slices = 0;
#endif
  const int chunk_max = M16_MAX_DOCS;
  const size_t cells = (size_t)std::min(n_docs, chunk_max) * (1 + (size_t)lq_pad);
  std::lock_guard<std::mutex> lk(g_mu);
  void* ws = nullptr;
  TS_CHECK(scratch_get(device, stream, cells * 4, &ws));
  const bool full = (row_bytes % (32 * M16_RING)) == 0;  // no k step past the row: unpredicated loads
  for (int c0 = 0; c0 < n_docs; c0 += chunk_max) {   // (one launch unless > 4096 candidates)
    const int n = std::min(chunk_max, n_docs - c0);
    p.n_docs = n;
    p.doc_off = doc_off ? doc_off + c0 : nullptr;
    p.starts = starts ? starts + c0 : nullptr;
    p.lens = lens ? lens + c0 : nullptr;
    p.out = out + c0;
    p.cnt = (uint32_t*)ws;
    p.best = p.cnt + n;
#define M16_GO(DT_, NQT_)                                                       \
  (full ? launch_main<DT_, NQT_, true>(p, grid, lds, stream)                    \
        : launch_main<DT_, NQT_, false>(p, grid, lds, stream))
    if (dtype == TS_F16) {
      if (nqt == 2) TS_CHECK(M16_GO(TS_F16, 2)); else TS_CHECK(M16_GO(TS_F16, 1));
    } else if (dtype == TS_BF16) {
      if (nqt == 2) TS_CHECK(M16_GO(TS_BF16, 2)); else TS_CHECK(M16_GO(TS_BF16, 1));
    } else {
      if (nqt == 2) TS_CHECK(M16_GO(TS_F32, 2)); else TS_CHECK(M16_GO(TS_F32, 1));
    }
#undef M16_GO
  }
  return TS_OK;
}

// Several queries, one launch (ts_maxsim_indexed_batch): the per-query fixed costs (launch,
// query image, prefix sums, completion round trips: ~25 us of a ~50 us single-query launch)
// overlap with other queries' streaming.  q_off / cand_off are HOST arrays; they reach the kernel
// inside its arguments, 64 queries per launch.
int ts_launch_maxsim16_batch(const void* q, const int32_t* q_off, int nq, const void* store,
                             const int64_t* starts, const int32_t* lens, const int32_t* cand_off,
                             int H, int dtype, int mode, float* out, int device, hipStream_t stream) {
  if (dtype != TS_F16 && dtype != TS_BF16 && dtype != TS_F32) return TS_ERR_UNSUPPORTED;
  const int row_bytes = H * (dtype == TS_F32 ? 4 : 2);
  if (nq <= 0 || (row_bytes % 16) != 0) return TS_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(store)) & 15) return TS_ERR_UNSUPPORTED;
  int max_lq = 0, max_cand = 0;
  for (int j = 0; j < nq; ++j) {
    const int lq = q_off[j + 1] - q_off[j], nc = cand_off[j + 1] - cand_off[j];
    if (lq <= 0 || nc < 0 || nc > M16_MAX_DOCS) return TS_ERR_UNSUPPORTED;
    max_lq = std::max(max_lq, lq);
    max_cand = std::max(max_cand, nc);
  }
  const int64_t n_pairs = (int64_t)cand_off[nq] - cand_off[0];
  if (n_pairs <= 0) return TS_OK;
  if (q_off[0] != 0 || cand_off[0] != 0) return TS_ERR_UNSUPPORTED;
  const int s_real = (row_bytes + 31) / 32;
  const int s_pad = ((s_real + M16_RING - 1) / M16_RING) * M16_RING;
  const size_t lds_cap = 156 * 1024;
  const size_t extra = M16_WAVES * 32 * sizeof(float) + 64 + ((size_t)M16_MAX_DOCS + 1) * 4 + 12;
  const int nqt = (max_lq > 32 && (size_t)s_pad * 2 * 1024 + extra <= lds_cap) ? 2 : 1;
  const size_t lds = (size_t)s_pad * nqt * 1024 + extra;
  if (lds > lds_cap) return TS_ERR_UNSUPPORTED;
  const int passes = (max_lq + nqt * 32 - 1) / (nqt * 32);
  const int lq_pad = passes * nqt * 32;
  if (device < 0 || device >= 64) return TS_ERR_UNSUPPORTED;
  if (g_cus[device] == 0) {
    int n = 0;
    TS_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device));
    g_cus[device] = n > 0 ? n : 256;
  }
  // workgroups per query: ~8 tiles per wave (a candidate is ~4 tiles), at most one per CU
  int grid = nq == 1 ? g_cus[device] : std::min(g_cus[device], std::max(1, (max_cand + 15) / 16));
#ifdef TS_TUNING
  if (const char* e = getenv("TS_M16_BATCH_GRID")) grid = atoi(e);
#endif

  Ms16Params p;
  p.H = row_bytes; p.s_pad = s_pad; p.lq_pad = lq_pad; p.passes = passes;
  p.docs = (const unsigned char*)store; p.doc_off = nullptr; p.mode = mode;
  p.Lq = max_lq; p.n_docs = max_cand;
  p.eq_slices = (nq == 1 && grid == g_cus[device]) ? 1 : 0;   // ONE query in the batch form (RetrievalPipeline.search)
#ifdef TS_TUNING
  if (getenv("TS_M16_NO_EQ")) p.eq_slices = 0;
#endif
  const bool full = (row_bytes % (32 * M16_RING)) == 0;
  std::lock_guard<std::mutex> lk(g_mu);
  for (int j0 = 0; j0 < nq; j0 += M16_MAX_BATCH) {   // (one launch unless > 64 queries)
    const int nb = std::min(M16_MAX_BATCH, nq - j0);
    const int qa = q_off[j0], ca = cand_off[j0];
    const int64_t pairs = (int64_t)cand_off[j0 + nb] - ca;
    if (pairs == 0) continue;
    for (int j = 0; j <= nb; ++j) {
      p.q_off[j] = q_off[j0 + j] - qa;
      p.cand_off[j] = cand_off[j0 + j] - ca;
    }
    p.nq = nb;
    p.q = (const unsigned char*)q + (size_t)qa * row_bytes;
    p.starts = starts + ca; p.lens = lens + ca; p.out = out + ca;
    void* ws = nullptr;
    TS_CHECK(scratch_get(device, stream, (size_t)pairs * (1 + (size_t)lq_pad) * 4, &ws));
    p.cnt = (uint32_t*)ws;
    p.best = p.cnt + pairs;
#define M16_GO(DT_, NQT_)                                                       \
  (full ? launch_main<DT_, NQT_, true>(p, grid, lds, stream, nb)                \
        : launch_main<DT_, NQT_, false>(p, grid, lds, stream, nb))
    if (dtype == TS_F16) {
      if (nqt == 2) TS_CHECK(M16_GO(TS_F16, 2)); else TS_CHECK(M16_GO(TS_F16, 1));
    } else if (dtype == TS_BF16) {
      if (nqt == 2) TS_CHECK(M16_GO(TS_BF16, 2)); else TS_CHECK(M16_GO(TS_BF16, 1));
    } else {
      if (nqt == 2) TS_CHECK(M16_GO(TS_F32, 2)); else TS_CHECK(M16_GO(TS_F32, 1));
    }
#undef M16_GO
  }
  return TS_OK;
}
