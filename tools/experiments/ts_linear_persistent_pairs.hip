// Linear layers of the encoder forwards with a SHORT reduction dimension (K <= 384: the Q/K/V, attention-output and
// feed-forward "up" projections of MiniLM-class models; at K = 768 the library's stream-K GEMMs are faster and the host
// keeps them), optionally with the erf GELU in the epilogue
// (BertIntermediate) — the GEMMs the reference reaches through CrossEncoder.predict / SentenceTransformer.encode
// (reference src/stage3_reranker.py:127-131, src/stage1_retriever.py:241-249), for gfx950.
//
// hipBLASLt runs these shapes at 450-730 TFLOP/s (M = 172 032 tokens: Q/K/V 384 -> 1152 in 0.32 ms, up 384 -> 1536 in
// 0.28 ms + 0.18 ms for the separate GELU pass): with K this short a tiled GEMM re-reads its operands from L2 many
// times per byte of HBM traffic (2 M N K (1/BM + 1/BN) bytes; tools/experiments/README.md).  This kernel has the
// STAGE-1 SCAN's structure instead:
//   * the weight matrix W [N, K] is pre-tiled once into the scan's corpus layout ([N/32][K/16][64 lanes] x 16 bytes: a
//     32-row block is one contiguous K/16 KiB run) and streamed by every wave through an 8-deep register ring as the
//     MFMA A operand — from L2, where its ~1 MB stays;
//   * the workgroup's 32*QH activation rows are the "queries": their whole K extent sits in LDS as the B-operand image,
//     built once from the row-major activations (the only HBM read of the kernel);
//   * each wave takes weight blocks w, w+8, ...: 32 output features x 32*QH rows per block, bias + GELU in the epilogue,
//     stored as 16-byte pieces (8 consecutive features of one row, after one exchange between the half-waves).
// The rounding points are those of linear followed by gelu: sum + bias rounded to the 16-bit type, GELU in fp32 on that
// value (erf to 1.5e-7, Abramowitz & Stegun 7.1.26: after the rounding to 8 / 11 mantissa bits a few values per million
// differ from torch's in the last bit), rounded again.  Without the activation the results were bit-identical to
// hipBLASLt's on every shape tried.
// No K loop with barriers, no operand double buffers: the activations are read from HBM once, W traffic from L2 is
// M / (32 QH) x |W|.
#include "ts_scan_dev.h"
#include <stdlib.h>
#include <algorithm>

#define FS_THREADS 512   // 8 waves: each takes weight blocks w, w + 8, ...
#define FS_WAVES 8
#ifndef FS_RING
#define FS_RING 4        // k groups in flight per wave and weight block of its pair (2 x 4 x 1 KiB, from L2)
#endif

__device__ __forceinline__ float fs_erf(float x) {   // Abramowitz & Stegun 7.1.26, |error| < 1.5e-7
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * ax * ax);
  return copysignf(fmaf(-poly, e, 1.0f), x);
}
template <int DT> __device__ __forceinline__ float fs_to_f32(uint16_t v) {
  if constexpr (DT == TS_F16) return (float)__builtin_bit_cast(_Float16, v);
  else return __uint_as_float((uint32_t)v << 16);
}
template <int DT> __device__ __forceinline__ uint16_t fs_from_f32(float v) {
  if constexpr (DT == TS_F16) return __builtin_bit_cast(uint16_t, (_Float16)v);
  else return __builtin_bit_cast(uint16_t, (__bf16)v);
}

#if defined(TS_TUNING) && defined(FS_TRACE)   // diagnostic builds only: per-wave phase time stamps (100 MHz)
__device__ unsigned long long fs_trace_buf[4096 * 8];
#define FS_STAMP(i) do { if ((threadIdx.x & 63) == 0 && fs_row < 4096) fs_trace_buf[fs_row * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int ts_debug_fs_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fs_trace_buf), sizeof(fs_trace_buf)) == hipSuccess ? 0 : -2;
}
#else
#define FS_STAMP(i) do { } while (0)
#endif

struct FsParams {
  const u32x4* w_tiled;    // [N/32][kg][64]
  const uint16_t* x;       // [M, K]
  const uint16_t* bias;    // [N] or null
  uint16_t* out;           // [M, N]
  int64_t M;
  int N, K, kg, gelu;
  int dbl;                 // two activation images fit LDS: the next tile is staged while the current one is computed
};

typedef float fs_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 fs_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 fs_f16x2 __attribute__((ext_vector_type(2)));
// two fp32 values -> one 32-bit word of two 16-bit values, round to nearest even (v_cvt_pk_bf16_f32 on gfx950)
template <int DT> __device__ __forceinline__ uint32_t fs_pack2(float a, float b) {
  if constexpr (DT == TS_F16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(fs_f32x2{a, b}, fs_f16x2));
  else return __builtin_bit_cast(uint32_t, __builtin_convertvector(fs_f32x2{a, b}, fs_bf16x2));
}
template <int DT> __device__ __forceinline__ float fs_lo(uint32_t w) {
  if constexpr (DT == TS_F16) return (float)__builtin_bit_cast(_Float16, (uint16_t)w);
  else return __uint_as_float(w << 16);
}
template <int DT> __device__ __forceinline__ float fs_hi(uint32_t w) {
  if constexpr (DT == TS_F16) return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
  else return __uint_as_float(w & 0xffff0000u);
}
__device__ __forceinline__ float fs_gelu(float u) { return (u * 0.5f) * (1.0f + fs_erf(u * 0.70710678118654752440f)); }

// Round 3: PERSISTENT workgroups, and every LDS read feeds TWO MFMAs.
//
// Round 2 launched one workgroup per tile of 32*QH activation rows; a wave took weight blocks w, w + 8, ... and issued,
// per k step, one 1 KiB weight load and QH x (ds_read_b128 of the activation fragment, MFMA).  What bounded it
// (M = 172 032, Q/K/V shape 384 -> 1152; tools/sessions/r03_linear*.sh, tools/trace_linear.py):
//   * 0.258 ms whatever the rows per workgroup (64 ... 192), two 4-waves-per-SIMD workgroups per CU or one persistent
//     2-waves-per-SIMD workgroup, with a 5.8 us or a 1.5 us epilogue: ~70 cycles per MFMA per SIMD instead of 32.  One
//     ds_read_b128 per MFMA is 1 KiB of LDS traffic per 32 matrix-pipe cycles and SIMD: four SIMDs ask for 128 B/clk,
//     the LDS's whole peak, and b128 reads deliver about half of it (SQ_LDS_IDX_ACTIVE, in quad-cycles, = 95 % of the
//     kernel's duration): the kernel was LDS-bandwidth-bound at 27 % matrix-pipe utilisation;
//   * a tile's life was 11 us of prologue (its rows fetched from HBM and re-tiled into LDS) in front of ~45 us of work;
//   * the epilogue branched on the activation flag around every value and loaded the bias from global memory — a
//     vector load whose wait, vmcnt(0), drained the weight ring once per block.
// Now: a wave takes PAIRS of adjacent weight blocks (64 output features) and each activation fragment read from LDS
// is the B operand of two MFMAs (half the LDS traffic; a row's 128 bytes of a pair are written by one wave); the
// workgroup is persistent — it walks tiles t, t + G, ... with the NEXT tile's rows parked in registers while the
// current one is computed (requested before the tile's main loop: the requests are older than the weight rings'
// refills, so nothing waits for them but the rings' second lap), barrier - LDS write - barrier between tiles; the
// weight rings run across units and tiles; the epilogue is straight-line code (activation = template parameter,
// v_pk_add_f32 / v_cvt_pk_bf16_f32, bias from LDS).  N / 64 pairs rarely divide by 8 waves (18 for Q/K/V): whole rounds
// of pairs first, the remaining pairs split by row quarter (unit = pair x one of the QH quarters) over all waves.
template <int DT, int QH, int NHQ, bool GELU>
__device__ __forceinline__ void fs_unit(const FsParams& p, const u32x4* ql, const float* bias_lds, int kg,
                                        const u32x4* curA, const u32x4* curB, const u32x4* nxtA, const u32x4* nxtB,
                                        u32x4 (&ringA)[FS_RING], u32x4 (&ringB)[FS_RING], int pair, int hq0, int64_t m0,
                                        int lane, [[maybe_unused]] int fs_row) {
  f32x16 accA[NHQ], accB[NHQ];
#pragma unroll
  for (int q = 0; q < NHQ; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) { accA[q][r] = 0.f; accB[q][r] = 0.f; }
  // One k step = NHQ activation fragments from LDS (read ONE STEP AHEAD, beside the previous step's second half), the
  // step's two weight registers, 2 NHQ MFMAs, two ring refills.  The halves are separated by scheduling barriers so
  // that loads are issued in the order they are consumed — A_i, B_i, A_i+1, ... — and every wait is vmcnt(7): left to
  // itself the compiler clustered the refills and waited vmcnt(1) once per ring lap (the whole ring drained).
  u32x4 bc[NHQ], bn[NHQ];
#pragma unroll
  for (int q = 0; q < NHQ; ++q) bc[q] = ql[(size_t)(hq0 + q) * 64];
  int g0 = 0;
  for (; g0 < kg - FS_RING; g0 += FS_RING) {
#pragma unroll
    for (int i = 0; i < FS_RING; ++i) {
#pragma unroll
      for (int q = 0; q < NHQ; ++q) mma_group<DT>(accA[q], ringA[i], bc[q]);
      ringA[i] = curA[(size_t)(g0 + i + FS_RING) * 64];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NHQ; ++q) bn[q] = ql[(size_t)((g0 + i + 1) * QH + hq0 + q) * 64];
#pragma unroll
      for (int q = 0; q < NHQ; ++q) mma_group<DT>(accB[q], ringB[i], bc[q]);
      ringB[i] = curB[(size_t)(g0 + i + FS_RING) * 64];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NHQ; ++q) bc[q] = bn[q];
    }
  }
#pragma unroll
  for (int i = 0; i < FS_RING; ++i) {
#pragma unroll
    for (int q = 0; q < NHQ; ++q) mma_group<DT>(accA[q], ringA[i], bc[q]);
    ringA[i] = nxtA[(size_t)i * 64];
    __builtin_amdgcn_sched_barrier(0);
    if (i + 1 < FS_RING) {
#pragma unroll
      for (int q = 0; q < NHQ; ++q) bn[q] = ql[(size_t)((g0 + i + 1) * QH + hq0 + q) * 64];
    }
#pragma unroll
    for (int q = 0; q < NHQ; ++q) mma_group<DT>(accB[q], ringB[i], bc[q]);
    ringB[i] = nxtB[(size_t)i * 64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NHQ; ++q) bc[q] = bn[q];
  }
  FS_STAMP(5);     // (FS_TRACE builds: fs_row < 4096 only for the unit that is traced)
  // ---- epilogue: rows of acc = 32 output features of a block, column = activation row j of quarter hq.  A lane holds 4
  // consecutive features per register group; groups 2p and 2p+1 are exchanged with lane ^ 32 so that each lane ends up
  // with 8 consecutive features (16 bytes) per pair of groups: half 0 gets features 16p + 0..7, half 1 features 16p + 8..15
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ab = 0; ab < 2; ++ab) {
    const int blk = 2 * pair + ab;
    const float* bl = bias_lds + blk * 32 + 4 * h;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bl + 8 * (2 * pr));
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(bl + 8 * (2 * pr + 1));
#pragma unroll
      for (int q = 0; q < NHQ; ++q) {
        const f32x16& acc = ab ? accB[q] : accA[q];
        uint32_t w[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 bb = t ? b1 : b0;
          const int r0 = 4 * (2 * pr + t);
          // sum + bias rounded to the 16-bit type (the linear's output) ...
          uint32_t lo = fs_pack2<DT>(acc[r0 + 0] + bb[0], acc[r0 + 1] + bb[1]);
          uint32_t hi = fs_pack2<DT>(acc[r0 + 2] + bb[2], acc[r0 + 3] + bb[3]);
          if constexpr (GELU) {   // ... the activation in fp32 on that value, rounded again
            lo = fs_pack2<DT>(fs_gelu(fs_lo<DT>(lo)), fs_gelu(fs_hi<DT>(lo)));
            hi = fs_pack2<DT>(fs_gelu(fs_lo<DT>(hi)), fs_gelu(fs_hi<DT>(hi)));
          }
          w[t][0] = lo;
          w[t][1] = hi;
        }
        const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
        const int64_t m = m0 + 32 * (hq0 + q) + j;
#if defined(TS_TUNING) && defined(FS_DBG_NOSTORE)   // ablation builds only: the epilogue's arithmetic without its stores
        if (m < p.M && s0[0] == 0x7fc1u && s1[1] == 0x12345u) {
#else
        if (m < p.M) {
#endif
          const u32x4 pk = {s0[0], s1[0], s0[1], s1[1]};
#if defined(TS_TUNING) && defined(FS_DBG_LINESTORE)   // ablation builds only: the same number of stores, each covering 8 WHOLE lines (wrong placement)
          *reinterpret_cast<u32x4*>(p.out + (m0 + 32 * (hq0 + q) + 8 * (2 * ab + pr) + (lane >> 3)) * p.N + pair * 64 + 8 * (lane & 7)) = pk;
#else
          *reinterpret_cast<u32x4*>(p.out + m * p.N + blk * 32 + 16 * pr + 8 * h) = pk;
#endif
        }
      }
    }
  }
}

template <int DT, int QH, bool GELU>
__global__ __launch_bounds__(FS_THREADS, 2) void ffn_stream_kernel(FsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kg = p.kg, K = p.K;
  const size_t img_bytes = (size_t)kg * QH * 1024;
  // LDS: [image 0][image 1 (when it fits: the next tile is staged while this one is read)][bias fp32 [N]]
  const bool dbl = p.dbl != 0;              // (uniform)
  float* bias_lds = reinterpret_cast<float*>(smem + (dbl ? 2 : 1) * img_bytes);
  const int npairs = p.N / 64;
  const int64_t ntiles = (p.M + 32 * QH - 1) / (32 * QH);
  [[maybe_unused]] const int fs_row = (int)(blockIdx.x * FS_WAVES + wave);   // (FS_TRACE builds)
  FS_STAMP(0);
  // ---- this wave's units of a tile: `full` whole pairs (pair = t * 8 + wave), then its share of the remaining pairs'
  // quarters (unit u = wave + 8 s of rem * QH: pair full * 8 + u / QH, quarter u % QH)
  const int full = npairs / FS_WAVES, rem = npairs % FS_WAVES;
  const int nsub = rem * QH > wave ? (rem * QH - wave + FS_WAVES - 1) / FS_WAVES : 0;
  const int nu = full + nsub;               // (0: a wave without work still fetches rows and keeps the barriers)
  auto unit_pair = [&](int t) -> int {
    return t < full ? t * FS_WAVES + wave : full * FS_WAVES + (wave + (t - full) * FS_WAVES) / QH;
  };
  // ---- the first weight loads go out before anything else
  const u32x4* base = p.w_tiled + lane;
  const size_t blk_units = (size_t)kg * 64;
  const u32x4* curA = base + (size_t)(2 * (nu ? unit_pair(0) : 0)) * blk_units;
  const u32x4* curB = curA + blk_units;
  u32x4 ringA[FS_RING], ringB[FS_RING];
#pragma unroll
  for (int i = 0; i < FS_RING; ++i) {
    ringA[i] = curA[(size_t)i * 64];
    ringB[i] = curB[(size_t)i * 64];
  }
  FS_STAMP(1);
  // ---- a tile's rows: the 16 bytes at (row, chunk c) of the tile's 32*QH x K/8 chunks are LDS unit
  // ((c >> 1) * QH + (row >> 5)) * 64 + 32 * (c & 1) + (row & 31) of the image.  A thread's chunks come in
  // NPIECE pieces of PIECE chunks; while tile t is computed, piece j of tile t + G is REQUESTED at the start of the
  // wave's j-th unit and WRITTEN to the other image at the start of the next one — the requests are older than every
  // weight load issued after them, so waiting for them does not drain the rings, and only PIECE quads are parked.
  constexpr int NPIECE = 3, PIECE = QH;     // NPIECE * PIECE * 512 chunks cover K = 384 (the prefetch's largest K)
  const int cpr = K / 8;                    // 16-byte chunks per row
  const int nchunk = 32 * QH * cpr;
  u32x4 pc[PIECE];
  // (a thread's chunk u = tid + c * FS_THREADS IS the LDS unit index: the 64 lanes of a wave write 64 consecutive
  // units — conflict-free — and read 32 rows x 32 bytes each; round 2 numbered the chunks row-major: coalesced reads,
  // but all lanes of a write landed in one bank group, SQ_LDS_BANK_CONFLICT = 27 % of the LDS's active cycles)
  auto chunk_src = [&](int u, int64_t m0) -> const u32x4* {
    const int l = u & 63, slot = u >> 6, hq = slot % QH, cc = slot / QH;
    const int row = 32 * hq + (l & 31), c = 2 * cc + (l >> 5);
    const bool ok = u < nchunk;
    const int64_t m = (ok && m0 + row < p.M) ? m0 + row : p.M - 1;
    return reinterpret_cast<const u32x4*>(p.x + m * K + 8 * (ok ? c : 0));
  };
  auto fetch_piece = [&](int64_t tile, int jp) {          // requests only
    const int64_t m0 = tile * (32 * QH);
#pragma unroll
    for (int i = 0; i < PIECE; ++i) pc[i] = *chunk_src(tid + (jp * PIECE + i) * FS_THREADS, m0);   // always issued, valid address
  };
  auto stage_piece = [&](u32x4* img, int jp) {            // registers -> image
#pragma unroll
    for (int i = 0; i < PIECE; ++i) {
      const int u = tid + (jp * PIECE + i) * FS_THREADS;
      if (u < nchunk) img[u] = pc[i];
    }
  };
  auto load_tile_sync = [&](u32x4* img, int64_t tile) {   // a whole tile, nothing else in flight (first tile; K > 384)
    const int64_t m0 = tile * (32 * QH);
    for (int u0 = tid; u0 < nchunk; u0 += 8 * FS_THREADS) {
      u32x4 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = *chunk_src(u0 + j * FS_THREADS, m0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int u = u0 + j * FS_THREADS;
        if (u < nchunk) img[u] = t[j];
      }
    }
  };
  int64_t tile = blockIdx.x;
  if (tile >= ntiles) return;               // (uniform; the host launches at most ntiles workgroups)
  u32x4* img_cur = reinterpret_cast<u32x4*>(smem);
  u32x4* img_nxt = reinterpret_cast<u32x4*>(smem + (dbl ? img_bytes : 0));
  load_tile_sync(img_cur, tile);
  for (int n = tid; n < p.N; n += FS_THREADS) bias_lds[n] = p.bias ? fs_to_f32<DT>(p.bias[n]) : 0.f;
  FS_STAMP(2);
  __syncthreads();
  FS_STAMP(3);
  [[maybe_unused]] int fs_first = 1;
  for (;;) {
    const int64_t m0 = tile * (32 * QH);
    const int64_t tile_n = tile + gridDim.x;
    const bool more = tile_n < ntiles;      // (uniform)
    const bool pre = dbl && more;           // the next tile goes into the other image while this one is computed
    const u32x4* ql = img_cur + lane;
    for (int t = 0; t < nu; ++t) {
      if (pre && t >= 1 && t <= NPIECE) stage_piece(img_nxt, t - 1);
      if (pre && t < NPIECE) fetch_piece(tile_n, t);
      if (fs_first == 2 && t == 1) { FS_STAMP(7); fs_first = 0; }
      const int pair = unit_pair(t);
      const int pair_n = unit_pair(t + 1 < nu ? t + 1 : 0);     // (the next tile starts with unit 0 again)
#if defined(TS_TUNING) && defined(FS_DBG_SAMEW)   // ablation builds only: every unit streams the first pair's weights (cache hits)
      const u32x4* nxtA = base;
#else
      const u32x4* nxtA = base + (size_t)(2 * pair_n) * blk_units;
#endif
      const u32x4* nxtB = nxtA + blk_units;
      if (t < full) {
        fs_unit<DT, QH, QH, GELU>(p, ql, bias_lds, kg, curA, curB, nxtA, nxtB, ringA, ringB, pair, 0, m0, lane,
                                  fs_first == 1 ? fs_row : 1 << 30);
      } else {
        const int hq0 = (wave + (t - full) * FS_WAVES) % QH;
        fs_unit<DT, QH, 1, GELU>(p, ql, bias_lds, kg, curA, curB, nxtA, nxtB, ringA, ringB, pair, hq0, m0, lane,
                                 fs_first == 1 ? fs_row : 1 << 30);
      }
      if (fs_first == 1) { FS_STAMP(4); fs_first = 2; }
      curA = nxtA;
      curB = nxtB;
    }
    if (!more) break;
    if (pre) {
      // the piece requested in the last unit, and the pieces a wave with fewer than NPIECE units never got to
      if (nu >= 1 && nu <= NPIECE) stage_piece(img_nxt, nu - 1);
      for (int jp = nu; jp < NPIECE; ++jp) {
        fetch_piece(tile_n, jp);
        stage_piece(img_nxt, jp);
      }
      __syncthreads();                      // the next image is complete, and nobody reads the current one any more
      u32x4* sw = img_cur; img_cur = img_nxt; img_nxt = sw;
    } else {
      __syncthreads();                      // every wave has finished reading the (only) image
      load_tile_sync(img_cur, tile_n);
      __syncthreads();
    }
    tile = tile_n;
  }
  FS_STAMP(6);
}

template <int DT, int QH, bool GELU>
static int fs_launch_g(const FsParams& p, hipStream_t s) {
  auto kern = ffn_stream_kernel<DT, QH, GELU>;
  static TsDeviceOnce attr;
  TS_CHECK(ts_allow_max_lds(attr, reinterpret_cast<const void*>(kern)));
  const size_t lds = (size_t)p.kg * QH * 1024 * (p.dbl ? 2 : 1) + (size_t)p.N * 4;
  const int64_t ntiles = (p.M + 32 * QH - 1) / (32 * QH);
  // persistent: ONE workgroup per CU (the next tile's rows parked in registers put a wave at ~220 VGPRs: two waves per
  // SIMD; with 128 registers — two workgroups per CU — the prefetch spills to scratch, whose accesses drain the weight
  // ring), each walking tiles blockIdx.x, + gridDim.x, ...
  int dev = 0, cus = 256;
  TS_HIP(hipGetDevice(&dev));
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  int per_cu = 1;
#ifdef TS_TUNING
  if (const char* e = getenv("TS_FS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
#endif
  const int64_t grid = std::min<int64_t>(ntiles, (int64_t)cus * per_cu);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(FS_THREADS), lds, s, p);
  TS_HIP(hipGetLastError());
  return TS_OK;
}
template <int DT, int QH>
static int fs_launch(const FsParams& p, hipStream_t s) {
  return p.gelu ? fs_launch_g<DT, QH, true>(p, s) : fs_launch_g<DT, QH, false>(p, s);
}

template <int DT>
static int fs_launch_qh(int qh, const FsParams& p, hipStream_t s) {
  switch (qh) {
    case 1: return fs_launch<DT, 1>(p, s);
    case 2: return fs_launch<DT, 2>(p, s);
    case 3: return fs_launch<DT, 3>(p, s);
#ifdef TS_TUNING
    case 4: return fs_launch<DT, 4>(p, s);
    case 5: return fs_launch<DT, 5>(p, s);
    case 6: return fs_launch<DT, 6>(p, s);
#endif
  }
  return TS_ERR_INVALID;
}

// ---- one-time re-tiling of a torch.nn.Linear weight [N, K] into the streamed layout
__global__ void linear_tile_kernel(const uint16_t* __restrict__ w, u32x4* __restrict__ out, int N, int K) {
  const int kg = K / 16;
  const int64_t unit = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte unit: (block, k group, lane)
  if (unit >= (int64_t)(N / 32) * kg * 64) return;
  const int lane = (int)(unit & 63), g = (int)((unit >> 6) % kg);
  const int64_t blk = (unit >> 6) / kg;
  out[unit] = *reinterpret_cast<const u32x4*>(w + (blk * 32 + (lane & 31)) * (int64_t)K + 16 * g + 8 * (lane >> 5));
}

extern "C" int ts_linear_tile_weight(const void* w, int32_t dtype, int32_t N, int32_t K, void* out, int32_t device, void* stream) {
  if (!w || !out || N <= 0 || K <= 0 || (dtype != TS_F16 && dtype != TS_BF16)) {
    ts_set_error("bad arguments to linear_tile_weight");
    return TS_ERR_INVALID;
  }
  if ((N % 32) || (K % 128) || ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(out)) & 15)) {
    ts_set_error("linear_tile_weight: N = %d (multiple of 32), K = %d (multiple of 128) or alignment not supported", N, K);
    return TS_ERR_UNSUPPORTED;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != device) TS_HIP(hipSetDevice(device));
  const int64_t units = (int64_t)(N / 32) * (K / 16) * 64;
  hipLaunchKernelGGL(linear_tile_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)w, (u32x4*)out, N, K);
  const hipError_t e = hipGetLastError();
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  if (e != hipSuccess) { ts_set_error("linear_tile_weight launch failed: %s", hipGetErrorString(e)); return TS_ERR_HIP; }
  return TS_OK;
}

extern "C" int ts_linear_act(const void* w_tiled, const void* x, const void* bias, int32_t dtype, int64_t M, int32_t N, int32_t K,
                             int32_t act, void* out, int32_t device, void* stream) {
  if (M == 0 || N == 0) return TS_OK;
  if (!w_tiled || !x || !out || M < 0 || N < 0 || K <= 0 || (dtype != TS_F16 && dtype != TS_BF16) || (act != 0 && act != 1)) {
    ts_set_error("bad arguments to linear_act");
    return TS_ERR_INVALID;
  }
  // rows of x per workgroup: three quarters of 32 when their whole-K image fits LDS, else two, else one
  const size_t lds_cap = 160 * 1024 - (size_t)N * 4;   // (the bias sits behind the image as fp32)
  // rows per workgroup: three quarters of 32 when TWO images of them fit LDS (K <= 384: the next tile is staged
  // beside the current one), else as many as one image allows
  const size_t img1 = (size_t)(K / 16) * 1024;
  int qh = 2 * 3 * img1 <= lds_cap ? 3 : 3 * img1 <= lds_cap ? 3 : 2 * img1 <= lds_cap ? 2 : 1;
#ifdef TS_TUNING   // A/B: rows per workgroup = 32 * TS_FS_QH (1..6) wherever the image fits LDS
  if (const char* e = getenv("TS_FS_QH")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 6 && (size_t)(K / 16) * v * 1024 <= lds_cap) qh = v;
  }
#endif
  const int gelu = act;
  const uintptr_t al = reinterpret_cast<uintptr_t>(w_tiled) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out);
  if ((N % 64) || (K % 128) || (size_t)N * 4 >= 160 * 1024 || (size_t)(K / 16) * qh * 1024 > lds_cap || (al & 15) || (reinterpret_cast<uintptr_t>(bias) & 7) ||
      (M + 32 * qh - 1) / (32 * qh) > 0x7fffffff) {
    ts_set_error("linear_act: N = %d (multiple of 64), K = %d (multiple of 128, at most 2560) or alignment not supported", N, K);
    return TS_ERR_UNSUPPORTED;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != device) TS_HIP(hipSetDevice(device));
  FsParams p;
  p.w_tiled = (const u32x4*)w_tiled; p.x = (const uint16_t*)x; p.bias = (const uint16_t*)bias; p.out = (uint16_t*)out;
  p.M = M; p.N = N; p.K = K; p.kg = K / 16; p.gelu = gelu;
  p.dbl = (2 * (size_t)qh * img1 <= lds_cap && K <= 384) ? 1 : 0;   // (the piecewise prefetch covers K <= 384)
  hipStream_t s = (hipStream_t)stream;
  int st = TS_ERR_INVALID;
  if (dtype == TS_BF16) st = fs_launch_qh<TS_BF16>(qh, p, s);
  else st = fs_launch_qh<TS_F16>(qh, p, s);
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  return st;
}
