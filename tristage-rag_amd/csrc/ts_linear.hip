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
//   * each of the 8 compute waves takes weight blocks w, w+8, ...: 32 output features x 32*QH rows per block, bias + GELU in
//     the epilogue; the finished block goes to the wave's LDS slot, and 4 storer waves write the slots out — 128
//     contiguous bytes of every row, from their own vmcnt queue (loads and stores share vmcnt on gfx9 and retire in order:
//     a compute wave that stored its block itself had its next weight refills queued behind the stores);
//   * workgroups are persistent (tile = blockIdx.x, + gridDim.x, ...): the storers fetch the next tile's rows into registers
//     while this tile is multiplied and write them into the image between the two barriers that end the tile.
// The rounding points are those of linear followed by gelu: sum + bias rounded to the 16-bit type, GELU in fp32 on that
// value (erf to 1.5e-7, Abramowitz & Stegun 7.1.26: after the rounding to 8 / 11 mantissa bits a few values per million
// differ from torch's in the last bit), rounded again.  Without the activation the results were bit-identical to
// hipBLASLt's on every shape tried.
// No K loop with barriers, no operand double buffers: the activations are read from HBM once, W traffic from L2 is
// M / (32 QH) x |W|.  The second kernel of this file, proj_ln_kernel (ts_linear_add_layernorm), is the same idea for the
// two projections that are followed by residual add + LayerNorm, with the reduction in chunks.
#include "ts_scan_dev.h"
#include "ts_ln_dev.h"
#include "ts_linear_dev.h"
#include <atomic>

#if defined(TS_TUNING) && defined(DBG_ONE_B)   // ablation builds only (wrong results): ONE B operand per k group from LDS, used for every row tile
#define FS_HQ(hq) 0
#else
#define FS_HQ(hq) (hq)
#endif
#define FS_THREADS 512   // 8 waves: each takes weight blocks w, w + 8, ...
#define FS_WAVES 8

#if defined(TS_TUNING) && defined(FS_TRACE)   // diagnostic builds only: per-wave phase time stamps (100 MHz), tools/trace_linear.py
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
  int cus;                 // compute units of the device (the persistent grid)
};

// LDS of a workgroup: the image of 32 QH rows, the slots, the counters
static inline size_t fs_lds_bytes(int kg, int qh) { return (size_t)kg * qh * 1024 + (size_t)FS_WAVES * 32 * qh * 80 + 2 * FS_WAVES * sizeof(int); }

// Slots: compute wave w hands each finished block (32 features x 32 QH rows, 16-bit) to a storer wave through its own LDS
// slot [32 QH rows][64 + 16 bytes] (the pad makes the 16-byte pieces of 16 rows fall into 16 different bank groups) and
// two counters: slot_full[w] = blocks written, slot_free[w] = blocks read.  Plain LDS words, volatile accesses, the
// order "tile, then counter" kept by lgkmcnt(0) between them (LDS executes a wave's instructions in order) — no
// workgroup fence: a fence also waits for the wave's global loads, i.e. drains the ring.
#define FS_STORERS 4
#define FS_SLOT_ROW 80
#define FS_PARTS 4         // a storer takes a pair of slots in this many parts (ROWS / 8 / FS_PARTS quads per lane each)
#define FS_SPIN_LIMIT (1 << 22)   // a counter that never arrives (it cannot) ends the wait, not the GPU
typedef __attribute__((address_space(3))) int fs_lds_int;   // (an LDS pointer the compiler knows to be one: a generic volatile
                                                            // access becomes a FLAT instruction, and a pending flat access turns
                                                            // every later wait into vmcnt(0) lgkmcnt(0))
__device__ __forceinline__ void fs_wait_counter(fs_lds_int* c, int at_least) {
  for (int spin = 0; __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < at_least && spin < FS_SPIN_LIMIT; ++spin)
    __builtin_amdgcn_s_sleep(1);
  __asm__ volatile("" ::: "memory");
}
__device__ __forceinline__ void fs_post_counter(fs_lds_int* c, int v, int lane) {
  __asm__ volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS traffic so far is done; vmcnt untouched
  if (lane == 0) __hip_atomic_store(c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The image of one tile (32 QH rows of x, whole K) -> LDS.  Piece (row, c) — c = 16-byte piece of the row — goes to unit
// ((c >> 1) QH + (row >> 5)) 64 + ((32 (c & 1) + (row & 31)) ^ ((c & 3) << 2)): the MFMA B-operand order with bits 2-3 of
// the lane swizzled by the piece number (readers: lane l, k group g -> l ^ ((2 (g & 1) + (l >> 5)) << 2)).  16 lanes read
// 256 contiguous bytes of a row, 4 rows per instruction; the swizzle spreads such a write over all 16 bank groups
// (without it the pieces of a row are 512 B / 3 KiB apart = in one bank group: 16 passes per ds_write_b128 instead of 4).
template <int QH>
__device__ __forceinline__ int fs_image_unit(int row, int c) {
  return ((c >> 1) * QH + (row >> 5)) * 64 + ((32 * (c & 1) + (row & 31)) ^ ((c & 3) << 2));
}
template <int QH>
__device__ __forceinline__ void fs_load_image(const FsParams& p, u32x4* qlds, int64_t m0, int t, int nt) {
  const int cc = t & 15, cpr16 = p.K / 128;           // pieces per row / 16
  for (int row = t >> 4; row < 32 * QH; row += nt >> 4) {
    const int64_t m = m0 + row < p.M ? m0 + row : p.M - 1;
    const uint16_t* src = p.x + m * p.K + 8 * cc;
    for (int jc0 = 0; jc0 < cpr16; jc0 += 4) {
      u32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const u32x4*>(src + 128 * min(jc0 + j, cpr16 - 1));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (jc0 + j < cpr16) qlds[fs_image_unit<QH>(row, cc + 16 * (jc0 + j))] = v[j];
    }
  }
}
// The storers' version in two halves: the NEXT tile's image is requested into registers while this tile is multiplied and
// written between the two barriers that end the tile — its HBM latency never shows.  Thread lt of 256: piece column
// lt & 15 (+ 16 jc, jc < K / 128 <= 3), row lt >> 4 (+ 16 jr, jr < 2 QH): at most 18 pieces (QH = 3, K = 384).
#define FS_PRE_JC 3
template <int QH>
__device__ __forceinline__ void fs_request_image(const FsParams& p, u32x4 (&v)[2 * QH][FS_PRE_JC], int64_t m0, int lt) {
  const int cc = lt & 15, rr = lt >> 4, cpr16 = p.K / 128;
#pragma unroll
  for (int jr = 0; jr < 2 * QH; ++jr) {
    const int64_t m = m0 + rr + 16 * jr < p.M ? m0 + rr + 16 * jr : p.M - 1;
    const uint16_t* src = p.x + m * p.K + 8 * cc;
#pragma unroll
    for (int jc = 0; jc < FS_PRE_JC; ++jc) v[jr][jc] = *reinterpret_cast<const u32x4*>(src + 128 * min(jc, cpr16 - 1));
  }
}
template <int QH>
__device__ __forceinline__ void fs_write_image(const FsParams& p, const u32x4 (&v)[2 * QH][FS_PRE_JC], u32x4* qlds, int lt) {
  const int cc = lt & 15, rr = lt >> 4, cpr16 = p.K / 128;
  // fs_image_unit(rr + 16 jr, cc + 16 jc) = one per-thread base + constants (rr < 16: bit 4 of the unit is free for 16 (jr & 1);
  // 16 jc changes neither c & 1 nor c & 3): every ds_write takes an immediate offset
  u32x4* base = qlds + fs_image_unit<QH>(rr, cc);
#pragma unroll
  for (int jr = 0; jr < 2 * QH; ++jr)
#pragma unroll
    for (int jc = 0; jc < FS_PRE_JC; ++jc)
      if (jc < cpr16) base[jc * (8 * QH * 64) + (jr >> 1) * 64 + 16 * (jr & 1)] = v[jr][jc];
}

// PERSISTENT workgroups: tile t = blockIdx.x, + gridDim.x, ... (32 QH rows each).  Per tile the compute waves run their
// blocks exactly as a one-tile workgroup would; between tiles two barriers — (A) every wave is done with the image,
// (B) the next image is in LDS — with only the storers' LDS writes of the prefetched image between them.
template <int DT, int QH, bool GELU>
__global__ __launch_bounds__(FS_THREADS + 64 * FS_STORERS) void ffn_stream_kernel(FsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* qlds = reinterpret_cast<u32x4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = p.kg;
  constexpr int ROWS = 32 * QH;
  constexpr int NT = FS_THREADS + 64 * FS_STORERS;
  unsigned char* slots = smem + (size_t)kg * QH * 1024;                     // [FS_WAVES][ROWS][FS_SLOT_ROW]
  fs_lds_int* slot_full = (fs_lds_int*)(slots + (size_t)FS_WAVES * ROWS * FS_SLOT_ROW);
  fs_lds_int* slot_free = slot_full + FS_WAVES;
  const int nblk = p.N / 32;
  const int64_t ntiles = (p.M + ROWS - 1) / ROWS;
  const bool compute = wave < FS_WAVES;
  const bool prefetch = p.K / 128 <= FS_PRE_JC;          // the next image fits the storers' registers
  [[maybe_unused]] const int fs_row = (int)(blockIdx.x * 12 + wave);   // (FS_TRACE builds: the first tile of each workgroup)
  FS_STAMP(0);
  // ---- the first weight loads go out before anything else
  const u32x4* base = p.w_tiled + lane;
  const size_t blk_units = (size_t)kg * 64;
  const bool active = compute && wave < nblk;
  const u32x4* cur = base + (size_t)(active ? wave : 0) * blk_units;
  u32x4 ring[TS_RING];
  if (active) {
#pragma unroll
    for (int i = 0; i < TS_RING; ++i) ring[i] = cur[(size_t)i * 64];
  }
  fs_load_image<QH>(p, qlds, (int64_t)blockIdx.x * ROWS, tid, NT);
  if (tid < 2 * FS_WAVES) slot_full[tid] = 0;
  FS_STAMP(1);
  fs_barrier();
  FS_STAMP(2);
  const int j = lane & 31, h = lane >> 5;
  if (!compute) {
    // ---- storer wave s serves compute waves 2 s and 2 s + 1, whose blocks of a round are neighbours: 128 contiguous
    // bytes of every output row.  Lane l: piece l & 7 of the row (0-3 from the first slot, 4-7 from the second), rows
    // (l >> 3) + 8 it.  The tiles go LDS -> registers -> global in this wave's own vmcnt queue: the compute waves' ring
    // refills are never queued behind a store (on gfx9 loads and stores share vmcnt and retire in order).
    const int lt = tid - FS_THREADS;
    const int s2 = 2 * (wave - FS_WAVES);
    const int piece = lane & 7, r0 = lane >> 3;
    const unsigned char* my = slots + (size_t)(s2 + (piece >> 2)) * ROWS * FS_SLOT_ROW + 16 * (piece & 3);
    int seen = 0, seen_b = 0;                         // blocks of earlier tiles from wave 2 s / from wave 2 s + 1 (the counters run on;
                                                      // the second wave owns one block fewer per tile when N / 32 is odd)
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int64_t m0 = tile * ROWS;
      const bool more = tile + gridDim.x < ntiles;
      u32x4 pre[2 * QH][FS_PRE_JC];
      if (more && prefetch) fs_request_image<QH>(p, pre, (tile + gridDim.x) * ROWS, lt);
      int n = 0;
      for (; s2 + 8 * n < nblk; ++n) {
        const int blk_a = s2 + 8 * n;
        const bool two = blk_a + 1 < nblk;
        fs_wait_counter(slot_full + s2, seen + n + 1);
        if (two) fs_wait_counter(slot_full + s2 + 1, seen_b + n + 1);
        // in FS_PARTS parts of the rows (the next image is parked in this wave's registers as well: 18 + 12 quads do not fit)
        const int rows_here = (int)((p.M - m0 < ROWS ? p.M - m0 : ROWS)) - r0;     // this lane's rows r0 + 8 i exist while 8 i < rows_here
        uint16_t* orow = p.out + (m0 + r0) * p.N + blk_a * 32 + 8 * piece;
#pragma unroll
        for (int half = 0; half < FS_PARTS; ++half) {
          constexpr int HI = ROWS / (8 * FS_PARTS);
          u32x4 v[HI];
          if (two || piece < 4) {
#pragma unroll
            for (int it = 0; it < HI; ++it) v[it] = *reinterpret_cast<const u32x4*>(my + (size_t)(r0 + 8 * (half * HI + it)) * FS_SLOT_ROW);
          }
          if (half == FS_PARTS - 1) {
            fs_post_counter(slot_free + s2, seen + n + 1, lane);
            if (two) fs_post_counter(slot_free + s2 + 1, seen_b + n + 1, lane);
            if (seen == 0 && n == 0) FS_STAMP(3);
            if (seen == 0 && n == 1) FS_STAMP(4);
          }
          if (two || piece < 4) {
#pragma unroll
            for (int it = 0; it < HI; ++it) {
              const int i8 = 8 * (half * HI + it);
              if (i8 < rows_here) *reinterpret_cast<u32x4*>(orow + (size_t)i8 * p.N) = v[it];
            }
          }
        }
      }
      seen += n;
      seen_b += (s2 + 1 < nblk) ? (nblk - s2 - 1 + 7) / 8 : 0;
      if (!more) break;
      fs_barrier();                                   // (A) the compute waves are done with this tile's image
      if (prefetch) fs_write_image<QH>(p, pre, qlds, lt);
      else fs_load_image<QH>(p, qlds, (tile + gridDim.x) * ROWS, tid, NT);
      fs_barrier();                                   // (B) the next image is in LDS
    }
    FS_STAMP(7);
    return;
  }
  const u32x4* ql[2] = {qlds + (lane ^ (h << 2)), qlds + (lane ^ ((2 + h) << 2))};   // k group even / odd (the image's swizzle)
  unsigned char* slot = slots + (size_t)wave * ROWS * FS_SLOT_ROW;
  int nth = 0;   // blocks of this wave so far (all tiles)
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const bool more = tile + gridDim.x < ntiles;
    if (active) {
      u32x4 bq[2][QH];
#pragma unroll
      for (int hq = 0; hq < QH; ++hq) bq[0][hq] = ql[0][(size_t)FS_HQ(hq) * 64];   // k group 0 of THIS tile's image
      int blk = wave;
      while (true) {
        const int blkn = blk + FS_WAVES;
        const bool has_next = blkn < nblk;
        // (after the tile's last block the ring is refilled with the first units of this wave's FIRST block: the next tile's)
        const u32x4* nxt = base + (size_t)(has_next ? blkn : wave) * blk_units;
        // bias of this block, requested BEFORE the k loop (in the epilogue it would be a wait for everything older = the ring's
        // refills for the next block): lane (j, h) ends up with features 8 (2 pr + t) + 4 h .. + 3
        uint2 bb[2][2] = {};
        if (p.bias) {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr)
#pragma unroll
            for (int t = 0; t < 2; ++t) bb[pr][t] = *reinterpret_cast<const uint2*>(p.bias + blk * 32 + 8 * (2 * pr + t) + 4 * h);
        }
        f32x16 acc[QH];
#pragma unroll
        for (int hq = 0; hq < QH; ++hq)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[hq][r] = 0.f;
        // The B operands (this workgroup's rows) are read from LDS ONE k group ahead of the MFMAs that use them, into the
        // other half of bq.  (k group parity = parity of i: kg and TS_RING are even; after the last group comes group 0 of
        // the next block — the same image.)
        int g0 = 0;
        for (; g0 < kg - TS_RING; g0 += TS_RING) {
#pragma unroll
          for (int i = 0; i < TS_RING; ++i) {
#pragma unroll
            for (int hq = 0; hq < QH; ++hq) bq[(i + 1) & 1][hq] = ql[(i + 1) & 1][(size_t)((g0 + i + 1) * QH + FS_HQ(hq)) * 64];
#pragma unroll
            for (int hq = 0; hq < QH; ++hq) mma_group<DT>(acc[hq], ring[i], bq[i & 1][hq]);
            ring[i] = cur[(size_t)(g0 + i + TS_RING) * 64];
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int i = 0; i < TS_RING; ++i) {
          const int gn = i + 1 < TS_RING ? g0 + i + 1 : 0;
#pragma unroll
          for (int hq = 0; hq < QH; ++hq) bq[(i + 1) & 1][hq] = ql[(i + 1) & 1][(size_t)(gn * QH + FS_HQ(hq)) * 64];
#pragma unroll
          for (int hq = 0; hq < QH; ++hq) mma_group<DT>(acc[hq], ring[i], bq[i & 1][hq]);
          ring[i] = nxt[(size_t)i * 64];
          __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: rows of acc = 32 output features of this block, column = activation row j of quarter hq.  A lane
        // holds 4 consecutive features per register group; groups 2p and 2p+1 are exchanged with lane ^ 32 so that each lane
        // ends up with 8 consecutive features (16 bytes) per pair: half 0 gets features 16p + 0..7, half 1 features
        // 16p + 8..15.  The pieces go to this wave's slot (once the storer has read the previous block out of it).
        if (nth == 0) FS_STAMP(3);
        if (nth == 1) FS_STAMP(5);
        fs_wait_counter(slot_free + wave, nth);
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          float b[2][4];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            // (16-bit zeros when there is no bias: + 0.0f)
            b[t][0] = fs_to_f32<DT>((uint16_t)bb[pr][t].x); b[t][1] = fs_to_f32<DT>((uint16_t)(bb[pr][t].x >> 16));
            b[t][2] = fs_to_f32<DT>((uint16_t)bb[pr][t].y); b[t][3] = fs_to_f32<DT>((uint16_t)(bb[pr][t].y >> 16));
          }
#pragma unroll
          for (int hq = 0; hq < QH; ++hq) {
            uint32_t w[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              uint16_t o[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                o[e] = fs_from_f32<DT>(acc[hq][4 * (2 * pr + t) + e] + b[t][e]);
                if constexpr (GELU) {
                  // The activation stays with the compute waves.  (Tried on the storers — one wave per SIMD at priority 1, plain
                  // and packed arithmetic: a SIMD's vector issue is shared by its waves and the erf costs ~72 issue cycles per
                  // value against the 32 of the MFMA it would hide under; the storer fell a block behind and the compute waves
                  // waited for their slots: 0.42 vs 0.39 ms, tools/trace_linear.py.)
                  const float u = fs_to_f32<DT>(o[e]);
                  o[e] = fs_from_f32<DT>((u * 0.5f) * (1.0f + fs_erf(u * 0.70710678118654752440f)));
                }
              }
              w[t][0] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
              w[t][1] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
            }
            const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
            const u32x4 pk = {s0[0], s1[0], s0[1], s1[1]};
            *reinterpret_cast<u32x4*>(slot + (size_t)(32 * hq + j) * FS_SLOT_ROW + 32 * pr + 16 * h) = pk;
          }
        }
        ++nth;
        fs_post_counter(slot_full + wave, nth, lane);
        if (nth == 1) FS_STAMP(4);
        if (nth == 2) FS_STAMP(6);
        if (!has_next) break;
        blk = blkn;
        cur = nxt;
      }
      cur = base + (size_t)wave * blk_units;   // (the ring now holds this wave's first block again)
    }
    if (!more) break;
    fs_barrier();                                     // (A) this wave is done with the image
    if (!prefetch) fs_load_image<QH>(p, qlds, (tile + gridDim.x) * ROWS, tid, NT);
    fs_barrier();                                     // (B) the next image is in LDS
  }
  FS_STAMP(7);
}

template <int DT, int QH, bool GELU>
static int fs_launch_act(const FsParams& p, hipStream_t s) {
  auto kern = ffn_stream_kernel<DT, QH, GELU>;
  static TsDeviceOnce attr;
  TS_CHECK(ts_allow_max_lds(attr, reinterpret_cast<const void*>(kern)));
  const size_t lds = fs_lds_bytes(p.kg, QH);
  const int64_t tiles = (p.M + 32 * QH - 1) / (32 * QH);
  const int64_t grid = tiles < p.cus ? tiles : p.cus;      // persistent: one workgroup per CU (its LDS allows no second one)
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(FS_THREADS + 64 * FS_STORERS), lds, s, p);
  TS_HIP(hipGetLastError());
  return TS_OK;
}
template <int DT, int QH>
static int fs_launch(const FsParams& p, hipStream_t s) {
  return p.gelu ? fs_launch_act<DT, QH, true>(p, s) : fs_launch_act<DT, QH, false>(p, s);
}

// ---------------------------------------------------------------------------------------------------------------
// Projection + bias + residual add + LayerNorm in one kernel: BertSelfOutput / BertOutput of a post-LN encoder
// (dense -> LayerNorm(dense(x) + input)), N <= 384 output features — the attention-output (K = N) and the feed-forward
// "down" (K = 4 N) projections of MiniLM-class encoders.  A workgroup owns 96 rows WHOLE (all N features), so the row
// statistics never leave it and the projection's output never goes to HBM: per row 2 K bytes are read and, for the
// fp32 residual stream plus its 16-bit copy, 4 N read + 6 N written — against 2 K + 2 N for the projection alone and
// another 2 N + 4 N + 6 N for the separate LayerNorm pass.
//   * waves 0 .. N/32 - 1 ("compute") each own ONE 32-feature weight block for the whole reduction: the block is one
//     contiguous run of the tiled weight, streamed from L2 through the 8-deep register ring without a break; the 96 rows
//     are the three MFMA B tiles, so a wave keeps 3 x 16 accumulators.
//   * the reduction is walked in chunks of 384 (24 k groups): the rows' chunk is a 72 KiB B-operand image in LDS, double
//     buffered; 4 "loader" waves do nothing but fetch the next chunk (coalesced 16-byte reads of the row-major rows,
//     scattered LDS writes).  Loader waves, not loads from the compute waves: vmcnt retires in order, so an HBM read
//     issued by a compute wave would hold every later L2 refill of its ring behind ~2 us of HBM latency once per chunk.
//     Compute waves meet the others at a bare s_barrier (their LDS reads are consumed, their ring stays in flight).
//   * epilogue: sum + bias rounded to the 16-bit type (= the projection's own output rounding) into an LDS staging tile
//     [96][N] (row stride 2 N + 16 bytes: conflict-free 16-byte writes), which takes the image's place; then ALL waves
//     run the LayerNorm rows exactly as add_layernorm_kernel does (half a wave per row, ln_row of ts_ln_dev.h: the same
//     bits), reading the fp32 residual from HBM and writing the fp32 stream (non-temporal) and its 16-bit copy.
#define PL_QH 3
#define PL_ROWS (32 * PL_QH)
#define PL_KGC 24                              // k groups per chunk (384 elements)
#define PL_KC (16 * PL_KGC)
#define PL_LOADERS 4
#define PL_IMG_UNITS (PL_KGC * PL_QH * 64)     // 16-byte units of one chunk image (72 KiB)
#define PL_MAX_N 384

#if defined(TS_TUNING) && defined(PL_TRACE)   // diagnostic builds only: per-wave phase time stamps (100 MHz), tools/trace_proj_ln.py
__device__ unsigned long long pl_trace_buf[4096 * 8];
#define PL_STAMP(i) do { if ((threadIdx.x & 63) == 0 && pl_row < 4096) pl_trace_buf[pl_row * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int ts_debug_pl_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pl_trace_buf), sizeof(pl_trace_buf)) == hipSuccess ? 0 : -2;
}
#else
#define PL_STAMP(i) do { } while (0)
#endif

struct PlParams {
  const u32x4* w_tiled;    // [N/32][K/16][64]
  const uint16_t* x;       // [M, K]
  const uint16_t* bias;    // [N] or null
  const float* res;        // fp32 [M, N] or null
  const float *gamma, *beta;
  float eps;
  float* out_f32;          // [M, N] or null
  uint16_t* out_lp;        // [M, N] or null
  int64_t M;
  int N, K, kg, nblk, nchunk;
  int stage_stride;        // bytes between staging rows
};

// ---- the chunk image in LDS.  Piece (row, c) — c = 16-byte piece of the row's 768-byte chunk, k group g = c >> 1, half
// h = c & 1 — lives at unit (g * PL_QH + (row >> 5)) * 64 + ((32 h + (row & 31)) ^ ((c & 3) << 2)): the MFMA B-operand
// order (lane 32 h + j <- row j) with bits 2-3 of the lane swizzled by the piece number.  Readers: lane l of a compute
// wave wants piece c = 2 g + (l >> 5), so it reads unit base + (l ^ ((2 (g & 1) + (l >> 5)) << 2)) — a permutation of
// the 64 units of the group, conflict-free, two per-lane constants (g even / odd).  Writers read the rows coalesced
// (16 lanes = 256 contiguous bytes of a row, 4 rows per instruction): without the swizzle the 16 pieces of a row land
// 3 KiB / 512 B apart = in the SAME 16-byte bank group and a 64-lane ds_write_b128 takes 16 passes instead of 4.
// (the rows of x are read once: non-temporal, so that they do not push the weight out of L2 — 0.33 -> 0.32 ms at K = 1536)
#define PL_XLOAD(ptr) __builtin_nontemporal_load(ptr)
// erf GELU of the 8 16-bit values of a piece, each rounded back to 16 bits: what torch's gelu makes of the up projection's
// output — applied by whoever stages the piece (GELU_IN), so the activation never costs a pass of its own
template <int DT>
__device__ __forceinline__ u32x4 pl_gelu8(const u32x4& v) {
  u32x4 o;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const float u0 = fs_to_f32<DT>((uint16_t)v[d]), u1 = fs_to_f32<DT>((uint16_t)(v[d] >> 16));
    const float y0 = (u0 * 0.5f) * (1.0f + fs_erf(u0 * 0.70710678118654752440f));
    const float y1 = (u1 * 0.5f) * (1.0f + fs_erf(u1 * 0.70710678118654752440f));
    o[d] = (uint32_t)fs_from_f32<DT>(y0) | ((uint32_t)fs_from_f32<DT>(y1) << 16);
  }
  return o;
}
__device__ __forceinline__ int pl_unit_in_group(int h, int row31, int c) { return (32 * h + row31) ^ ((c & 3) << 2); }

// Chunk 0: every thread of the workgroup takes part (t of nt; nt a multiple of 16), three pieces of a row per thread.
template <int DT, bool GELU_IN>
__device__ __forceinline__ void pl_load_image0(const PlParams& p, u32x4* dst, int64_t m0, int t, int nt) {
  const int cc = t & 15;
  for (int row = t >> 4; row < PL_ROWS; row += nt >> 4) {
    const int64_t m = m0 + row < p.M ? m0 + row : p.M - 1;
    const uint16_t* src = p.x + m * p.K + 8 * cc;
    u32x4 v[3];
#pragma unroll
    for (int jc = 0; jc < 3; ++jc) v[jc] = PL_XLOAD(reinterpret_cast<const u32x4*>(src + 128 * jc));
#pragma unroll
    for (int jc = 0; jc < 3; ++jc) {
      const int c = cc + 16 * jc;
      dst[(size_t)((c >> 1) * PL_QH + (row >> 5)) * 64 + pl_unit_in_group(c & 1, row & 31, c)] = GELU_IN ? pl_gelu8<DT>(v[jc]) : v[jc];
    }
  }
}

// The loader waves' version in two halves — a chunk is requested one phase before it is written.  Thread lt of
// PL_LOADERS * 64: piece column cc = lt & 15 (+ 16 jc), row rr = lt >> 4 (+ 16 jr): 3 x 6 = PL_LD pieces, every global
// address = one of 6 row pointers + a constant, every LDS address = one base + a constant.
#define PL_LD 18
static_assert(PL_LD * PL_LOADERS * 64 == PL_IMG_UNITS && PL_LOADERS * 64 == 256 && PL_ROWS == 96 && PL_KC == 384, "the loaders' piece map");
__device__ __forceinline__ void pl_request(const PlParams& p, u32x4 (&v)[PL_LD], int64_t m0, int kc, int lt) {
  const int cc = lt & 15, rr = lt >> 4;
#pragma unroll
  for (int jr = 0; jr < 6; ++jr) {
    const int64_t m = m0 + rr + 16 * jr < p.M ? m0 + rr + 16 * jr : p.M - 1;
    const uint16_t* src = p.x + m * p.K + (size_t)kc * PL_KC + 8 * cc;
#pragma unroll
    for (int jc = 0; jc < 3; ++jc) v[3 * jr + jc] = PL_XLOAD(reinterpret_cast<const u32x4*>(src + 128 * jc));
  }
}
template <int DT, bool GELU_IN>
__device__ __forceinline__ void pl_write(const u32x4 (&v)[PL_LD], u32x4* dst, int lt) {
  const int cc = lt & 15, rr = lt >> 4;
  u32x4* base = dst + (cc >> 1) * (PL_QH * 64) + pl_unit_in_group(cc & 1, rr, cc);   // (rr < 16: bit 4 of the unit is free for 16 (jr & 1))
#pragma unroll
  for (int jr = 0; jr < 6; ++jr)
#pragma unroll
    for (int jc = 0; jc < 3; ++jc) base[jc * (8 * PL_QH * 64) + (jr >> 1) * 64 + 16 * (jr & 1)] = GELU_IN ? pl_gelu8<DT>(v[3 * jr + jc]) : v[3 * jr + jc];
}
template <int DT, bool GELU_IN>
__global__ __launch_bounds__(1024) void proj_ln_kernel(PlParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* img = reinterpret_cast<u32x4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = p.nblk, nwaves = nblk + PL_LOADERS;
  const bool compute = wave < nblk;
  const int64_t m0 = (int64_t)blockIdx.x * PL_ROWS;
  [[maybe_unused]] const int pl_row = (int)(blockIdx.x * 16 + wave);   // (PL_TRACE builds)
  PL_STAMP(0);
  // ---- chunk 0 of the image: every wave helps
  pl_load_image0<DT, GELU_IN>(p, img, m0, tid, nwaves * 64);
  PL_STAMP(1);
  unsigned char* stg = smem;       // the staging tile takes the images' place after the last chunk
  const int j = lane & 31, h = lane >> 5;
  // Two disjoint code paths (not one loop with a branch inside: accumulators and ring would be live through the loaders'
  // branch and the register allocator spills a ring slot).  Both execute nchunk + 1 barriers.
  if (compute) {
    // ---- the weight block's first loads
    const u32x4* cur = p.w_tiled + (size_t)wave * p.kg * 64 + lane;
    u32x4 ring[TS_RING];
#pragma unroll
    for (int i = 0; i < TS_RING; ++i) ring[i] = cur[(size_t)i * 64];
    // ---- bias of this block: lane (j, h) ends up with features 8 (2 pr + t) + 4 h .. + 3 (requested here, used in the epilogue)
    uint2 bb[2][2] = {};
    if (p.bias) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int t = 0; t < 2; ++t) bb[pr][t] = *reinterpret_cast<const uint2*>(p.bias + wave * 32 + 8 * (2 * pr + t) + 4 * h);
    }
    fs_barrier();   // chunk 0 is in LDS (this wave's share: the lgkmcnt(0) inside)
    PL_STAMP(2);
    f32x16 acc[PL_QH];
#pragma unroll
    for (int hq = 0; hq < PL_QH; ++hq)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[hq][r] = 0.f;
    int g = 0;
    for (int c = 0; c < p.nchunk; ++c) {
      const u32x4* buf = img + (size_t)(c & 1) * PL_IMG_UNITS;
      const u32x4* ql[2] = {buf + (lane ^ (h << 2)), buf + (lane ^ ((2 + h) << 2))};   // k group even / odd (the image's swizzle)
      // (B operands one k group ahead, as in ffn_stream_kernel; the chunk's first group after the barrier, the last group
      // reads group 0 again — of THIS buffer, nobody uses it: the next chunk's buffer is not ours before the barrier)
      u32x4 bq[2][PL_QH];
#pragma unroll
      for (int hq = 0; hq < PL_QH; ++hq) bq[0][hq] = ql[0][(size_t)FS_HQ(hq) * 64];
      for (int r = 0; r < PL_KGC; r += TS_RING) {
#pragma unroll
        for (int i = 0; i < TS_RING; ++i) {
          const int gn = r + i + 1 < PL_KGC ? r + i + 1 : 0;
#pragma unroll
          for (int hq = 0; hq < PL_QH; ++hq) bq[(i + 1) & 1][hq] = ql[(i + 1) & 1][(size_t)(gn * PL_QH + FS_HQ(hq)) * 64];
#pragma unroll
          for (int hq = 0; hq < PL_QH; ++hq) mma_group<DT>(acc[hq], ring[i], bq[i & 1][hq]);
#if !(defined(TS_TUNING) && defined(DBG_NO_REFILL))   // (ablation: the first 8 weight units over and over — no L2 stream)
          ring[i] = cur[(size_t)min(g + r + i + TS_RING, p.kg - 1) * 64];   // (always issued; past the block's end: a re-read nobody uses)
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      g += PL_KGC;
      if (c == 0) PL_STAMP(3);
      fs_barrier();   // (this wave's LDS reads of the chunk were consumed by the MFMAs above; it has written nothing)
    }
    PL_STAMP(4);
    // ---- (every wave is past the last chunk's barrier: the images are dead) projection output -> staging tile
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      float b[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        // (16-bit zeros when there is no bias: +0.0f)
        b[t][0] = fs_to_f32<DT>((uint16_t)bb[pr][t].x); b[t][1] = fs_to_f32<DT>((uint16_t)(bb[pr][t].x >> 16));
        b[t][2] = fs_to_f32<DT>((uint16_t)bb[pr][t].y); b[t][3] = fs_to_f32<DT>((uint16_t)(bb[pr][t].y >> 16));
      }
#pragma unroll
      for (int hq = 0; hq < PL_QH; ++hq) {
        uint32_t w[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          uint16_t o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = fs_from_f32<DT>(acc[hq][4 * (2 * pr + t) + e] + b[t][e]);
          w[t][0] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
          w[t][1] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
        }
        const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
        const u32x4 pk = {s0[0], s1[0], s0[1], s1[1]};
        *reinterpret_cast<u32x4*>(stg + (size_t)(32 * hq + j) * p.stage_stride + 2 * (wave * 32 + 16 * pr + 8 * h)) = pk;
      }
    }
    PL_STAMP(5);
  } else {
    // ---- loaders: chunk k + 1 is WRITTEN during phase k (its buffer was last read in phase k - 1) from registers it was
    // requested into one phase earlier, so the HBM latency of a chunk passes under a whole phase of multiplications
    const int lt = tid - nblk * 64;
    u32x4 pre[PL_LD];
    if (p.nchunk > 1) pl_request(p, pre, m0, 1, lt);
    fs_barrier();
    PL_STAMP(2);
    for (int c = 0; c < p.nchunk; ++c) {
      if (c + 1 < p.nchunk) {
        pl_write<DT, GELU_IN>(pre, img + (size_t)((c + 1) & 1) * PL_IMG_UNITS, lt);
        if (c + 2 < p.nchunk) pl_request(p, pre, m0, c + 2, lt);
      }
      if (c == 0) PL_STAMP(3);
      fs_barrier();
    }
    PL_STAMP(4);
  }
  // ---- LayerNorm rows: half a wave per row, lane lir of the half owns chunks c * 32 + lir (as add_layernorm_kernel<.., 3, 32>)
  const int lir = j, sub = h, H = p.N;
  f32x4 gm[3], bt[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int e = (c * 32 + lir) * 4;
    gm[c] = bt[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (e < H) {
      gm[c] = *reinterpret_cast<const f32x4*>(p.gamma + e);
      if (p.beta) bt[c] = *reinterpret_cast<const f32x4*>(p.beta + e);
    }
  }
  // the first rows' residual is requested before the barrier (its HBM latency passes while the slower waves stage)
  f32x4 rs[3], rn[3];
  int pi = wave;                                   // row pair of the tile: rows 2 pi, 2 pi + 1
  auto fetch_res = [&](int pair, f32x4 (&r)[3]) {
    const int64_t row = m0 + 2 * pair + sub;
    const int64_t base = (row < p.M ? row : p.M - 1) * H;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int e = (c * 32 + lir) * 4;
      r[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.res && e < H) r[c] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.res + base + e));   // read once
    }
  };
  if (pi < PL_ROWS / 2) fetch_res(pi, rs);
  __syncthreads();
  PL_STAMP(6);
  for (; pi < PL_ROWS / 2; pi += nwaves) {
    const int pn = pi + nwaves;
    if (pn < PL_ROWS / 2) fetch_res(pn, rn);
    const int lrow = 2 * pi + sub;
    const int64_t row = m0 + lrow;
    f32x4 v[3], y[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int e = (c * 32 + lir) * 4;
      v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (e < H) {
        v[c] = ln_load4<DT>(stg + (size_t)lrow * p.stage_stride, e);
        if (p.res) v[c] += rs[c];
      }
    }
    ln_row<3, 32>(v, gm, bt, H, lir, p.eps, y);
    if (row < p.M) {
      const int64_t base = row * H;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int e = (c * 32 + lir) * 4;
        if (e < H) {
          if (p.out_f32) __builtin_nontemporal_store(y[c], reinterpret_cast<f32x4*>(p.out_f32 + base + e));
          if (p.out_lp) {
            uint2 pk;
            pk.x = ln_pack2(y[c][0], y[c][1], DT);
            pk.y = ln_pack2(y[c][2], y[c][3], DT);
            *reinterpret_cast<uint2*>(p.out_lp + base + e) = pk;
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) rs[c] = rn[c];
  }
  PL_STAMP(7);
}

template <int DT, bool GELU_IN>
static int pl_launch(const PlParams& p, hipStream_t s) {
  auto kern = proj_ln_kernel<DT, GELU_IN>;
  static TsDeviceOnce attr;
  TS_CHECK(ts_allow_max_lds(attr, reinterpret_cast<const void*>(kern)));
  const size_t images = (size_t)(p.nchunk > 1 ? 2 : 1) * PL_IMG_UNITS * 16, staging = (size_t)PL_ROWS * p.stage_stride;
  const size_t lds = images > staging ? images : staging;
  const int64_t grid = (p.M + PL_ROWS - 1) / PL_ROWS;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * (p.nblk + PL_LOADERS)), lds, s, p);
  TS_HIP(hipGetLastError());
  return TS_OK;
}

extern "C" int ts_linear_add_layernorm(const void* w_tiled, const void* x, const void* bias, const float* residual,
                                       const float* gamma, const float* beta, float eps, int32_t dtype, int64_t M, int32_t N,
                                       int32_t K, int32_t act_in, float* out_f32, void* out_lp, int32_t device, void* stream) {
  if (M == 0) return TS_OK;
  if (!w_tiled || !x || !gamma || (!out_f32 && !out_lp) || M < 0 || N <= 0 || K <= 0 || (dtype != TS_F16 && dtype != TS_BF16) ||
      (act_in != 0 && act_in != 1)) {
    ts_set_error("bad arguments to linear_add_layernorm");
    return TS_ERR_INVALID;
  }
  const uintptr_t al = reinterpret_cast<uintptr_t>(w_tiled) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(residual) |
                       reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(out_f32) |
                       reinterpret_cast<uintptr_t>(out_lp);
  if ((N % 32) || N > PL_MAX_N || (K % PL_KC) || (al & 15) || (reinterpret_cast<uintptr_t>(bias) & 7) ||
      (M + PL_ROWS - 1) / PL_ROWS > 0x7fffffff) {
    ts_set_error("linear_add_layernorm: N = %d (multiple of 32, at most %d), K = %d (multiple of %d) or alignment not supported",
                 N, PL_MAX_N, K, PL_KC);
    return TS_ERR_UNSUPPORTED;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != device) TS_HIP(hipSetDevice(device));
  PlParams p;
  p.w_tiled = (const u32x4*)w_tiled; p.x = (const uint16_t*)x; p.bias = (const uint16_t*)bias; p.res = residual;
  p.gamma = gamma; p.beta = beta; p.eps = eps; p.out_f32 = out_f32; p.out_lp = (uint16_t*)out_lp;
  p.M = M; p.N = N; p.K = K; p.kg = K / 16; p.nblk = N / 32; p.nchunk = K / PL_KC; p.stage_stride = 2 * N + 16;
  hipStream_t s = (hipStream_t)stream;
  const int st = dtype == TS_BF16 ? (act_in ? pl_launch<TS_BF16, true>(p, s) : pl_launch<TS_BF16, false>(p, s))
                                  : (act_in ? pl_launch<TS_F16, true>(p, s) : pl_launch<TS_F16, false>(p, s));
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  return st;
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
  const int qh = fs_lds_bytes(K / 16, 3) <= 160 * 1024 ? 3 : fs_lds_bytes(K / 16, 2) <= 160 * 1024 ? 2 : 1;
  const int gelu = act;
  const uintptr_t al = reinterpret_cast<uintptr_t>(w_tiled) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out);
  if ((N % 32) || (K % 128) || fs_lds_bytes(K / 16, qh) > 160 * 1024 || (al & 15) || (reinterpret_cast<uintptr_t>(bias) & 7) ||
      (M + 32 * qh - 1) / (32 * qh) > 0x7fffffff) {
    ts_set_error("linear_act: N = %d (multiple of 32), K = %d (multiple of 128, at most 2176) or alignment not supported", N, K);
    return TS_ERR_UNSUPPORTED;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != device) TS_HIP(hipSetDevice(device));
  FsParams p;
  p.w_tiled = (const u32x4*)w_tiled; p.x = (const uint16_t*)x; p.bias = (const uint16_t*)bias; p.out = (uint16_t*)out;
  p.M = M; p.N = N; p.K = K; p.kg = K / 16; p.gelu = gelu;
  {
    hipDeviceProp_t prop;
    static std::atomic<int> cus_of[64];
    if (device >= 0 && device < 64 && cus_of[device].load(std::memory_order_relaxed) > 0) p.cus = cus_of[device].load(std::memory_order_relaxed);
    else {
      if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        ts_set_error("linear_act: hipGetDeviceProperties failed");
        if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
        return TS_ERR_HIP;
      }
      p.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
      if (device >= 0 && device < 64) cus_of[device].store(p.cus, std::memory_order_relaxed);
    }
  }
  hipStream_t s = (hipStream_t)stream;
  auto go = [&](const FsParams& q, int rows32) -> int {
    if (dtype == TS_BF16) return rows32 == 3 ? fs_launch<TS_BF16, 3>(q, s) : rows32 == 2 ? fs_launch<TS_BF16, 2>(q, s) : fs_launch<TS_BF16, 1>(q, s);
    return rows32 == 3 ? fs_launch<TS_F16, 3>(q, s) : rows32 == 2 ? fs_launch<TS_F16, 2>(q, s) : fs_launch<TS_F16, 1>(q, s);
  };
  // (Tried: the last, partial round of the persistent grid as a second launch with 64-row tiles — 6.7 rounds' worth of
  // tile time instead of 7 on paper, no difference in the stage-3 forward on one box: tools/sessions/r03_split.sh.)
  const int st = go(p, qh);
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  return st;
}
