// The feed-forward block of a post-LN encoder layer in ONE kernel (gfx950): BertIntermediate + BertOutput of the
// MiniLM-class cross-encoder the reference reaches through CrossEncoder.predict (reference src/stage3_reranker.py:127-131;
// transformers' modeling_bert BertIntermediate.forward / BertOutput.forward):
//     y = LayerNorm(round(gelu(round(x W1^T + b1)) W2^T + b2) + residual) * gamma + beta,   x [M, H], W1 [I, H], W2 [H, I]
// H = 384 (a workgroup owns 96 rows WHOLE), I a multiple of 384 (1536).  As three kernels — up projection + GELU, down
// projection, residual + LayerNorm — a layer writes the M x I intermediate to HBM and reads it back (968 MB of the ~2.3 GB
// a layer moves at M = 157 539) and each kernel pays its own prologue and epilogue; as two (ts_linear_act +
// ts_linear_add_layernorm, ts_linear.hip) 0.62-0.68 ms.  Here the intermediate never leaves the CU:
//   * 12 waves, wave w owns 32-feature block w of every 384-wide CHUNK of the intermediate (W1 block 12 c + w) and block w
//     of the output (W2 block w, whose reduction is walked in the same chunks); both weights are streamed from L2 through
//     ONE register ring (6 deep: ML_RING) in the order the phases need them — the structure of ffn_stream_kernel / proj_ln_kernel;
//   * up(c): 96 rows x 32 features from the LDS image of x (72 KiB, loaded once);  its output + bias, rounded, GELU, rounded,
//     is written as the wave's 12th of the chunk's B-operand image (a second 72 KiB buffer) — a lane's 16-byte piece after
//     the half-wave exchange IS a piece of that image;  down(c): the chunk image x W2's chunk c into the output accumulators;
//   * the erf GELU is vector-ALU work (~72 issue cycles per value): chunk c's runs INSIDE down(c - 1), two values per k
//     group between the MFMAs of the same wave, where an MFMA leaves 24 of its 32 cycles of vector issue free
//     (MI355X_MICROARCH.md, constants table), chunk 0's inside up(1) — so the phase order is up(0) [up(1) + G(0)] |
//     [down(0) + G(1)] up(2) | [down(1) + G(2)] up(3) | ... | down(last), with two barriers per chunk around the image write
//     (one for chunk 0);
//   * no loader waves (there is nothing to load after the prologue but the residual), so 12 waves = 3 per SIMD = 168 registers:
//     both accumulator sets (96), the ring (32) and the B operands fit;
//   * the output tile is staged in LDS as 16-bit values where the images were and the LayerNorm rows are run by all waves
//     with ln_row (ts_ln_dev.h) exactly as proj_ln_kernel does.
// The rounding points and the accumulation order (k groups in sequence) are those of the two-kernel path: the same bits.
#include "ts_scan_dev.h"
#include "ts_ln_dev.h"
#include "ts_linear_dev.h"
#include <atomic>

#define ML_QH 3
#define ML_ROWS (32 * ML_QH)
#define ML_H 384
#define ML_WAVES (ML_H / 32)                   // 12
#define ML_KGC 24                              // k groups per chunk (384 elements)
#define ML_IMG_UNITS (ML_KGC * ML_QH * 64)     // 16-byte units of one image (72 KiB)
#define ML_RING 6                              // weight units in flight per wave (8 in the other kernels: 8 more registers spill here)
#define ML_STAGE_STRIDE (2 * ML_H + 16)        // bytes between staging rows: 16-byte pieces of 16 rows in 16 different bank groups

#if defined(TS_TUNING) && defined(ML_TRACE)   // diagnostic builds only: per-wave phase time stamps (100 MHz), tools/trace_mlp.py
__device__ unsigned long long ml_trace_buf[4096 * 8];
#define ML_STAMP(i) do { if ((threadIdx.x & 63) == 0 && ml_row < 4096) ml_trace_buf[ml_row * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int ts_debug_ml_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ml_trace_buf), sizeof(ml_trace_buf)) == hipSuccess ? 0 : -2;
}
#else
#define ML_STAMP(i) do { } while (0)
#endif

struct MlParams {
  const u32x4 *w1_tiled, *w2_tiled;   // [I/32][H/16][64], [H/32][I/16][64]
  const uint16_t *b1, *b2;            // [I], [H] or null
  const uint16_t* x;                  // [M, H]
  const float* res;                   // fp32 [M, H] or null
  const float *gamma, *beta;
  float eps;
  float* out_f32;                     // [M, H] or null
  uint16_t* out_lp;                   // [M, H] or null
  int64_t M;
  int I, nchunk;
};

// Image unit of piece (row, c) — c = 16-byte piece of the row's 768-byte chunk: the MFMA B-operand order with bits 2-3 of
// the lane swizzled by the piece number (conflict-free row-major writes; ts_linear.hip).  Readers: lane l, k group g ->
// (g QH + hq) 64 + (l ^ ((2 (g & 1) + (l >> 5)) << 2)).
__device__ __forceinline__ int ml_unit(int row, int c) {
  return ((c >> 1) * ML_QH + (row >> 5)) * 64 + ((32 * (c & 1) + (row & 31)) ^ ((c & 3) << 2));
}

// The 32 bias values of a block as 16 packed dwords in SCALAR registers (the block is wave-uniform): no vector registers.
// Accumulator register r of lane half h is feature 8 (r >> 2) + 4 h + (r & 3) of the block.
struct MlBias { uint32_t w[16]; };
__device__ __forceinline__ MlBias ml_load_bias(const uint16_t* b, int block) {
  MlBias o;
  if (b) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(b + block * 32);   // (uniform address: s_load_dwordx16)
#pragma unroll
    for (int i = 0; i < 16; ++i) o.w[i] = __builtin_amdgcn_readfirstlane(src[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) o.w[i] = 0u;
  }
  return o;
}
template <int DT>
__device__ __forceinline__ float ml_bias(const MlBias& b, int r, int h) {
  const int f0 = 8 * (r >> 2) + (r & 3);                      // feature for h = 0; + 4 for h = 1
  const uint32_t w = h ? b.w[(f0 + 4) >> 1] : b.w[f0 >> 1];
  return fs_to_f32<DT>((uint16_t)(w >> (16 * (f0 & 1))));
}
// erf GELU of the two 16-bit values of a packed pair, each rounded back to 16 bits (what torch's gelu makes of the up
// projection's rounded output).  Computed: ~72 vector issue cycles per value — 0.16 ms per layer of pure vector-ALU time at
// 157 539 x 1536 values, THE cost of the feed-forward block once its traffic is gone.
template <int DT>
__device__ __forceinline__ uint32_t ml_gelu2_computed(uint32_t w) {
  const float u0 = fs_to_f32<DT>((uint16_t)w), u1 = fs_to_f32<DT>((uint16_t)(w >> 16));
  const float y0 = (u0 * 0.5f) * (1.0f + fs_erf(u0 * 0.70710678118654752440f));
  const float y1 = (u1 * 0.5f) * (1.0f + fs_erf(u1 * 0.70710678118654752440f));
  return (uint32_t)fs_from_f32<DT>(y0) | ((uint32_t)fs_from_f32<DT>(y1) << 16);
}
// bf16: a function of 16 bits is a table.  Every workgroup builds, with the computed formula (so: the same bits), the table
// of all inputs with 2^-12 <= |u| < 8 — 15 exponents x 128 mantissas x 2 signs = 3840 entries of 2 bytes in the LDS the
// images leave free — and a value costs a handful of integer instructions and one ds_read_u16.  Inputs outside the range
// (|u| < 2^-12, |u| >= 8, inf, nan: a few per ten thousand) take the computed path under a wave-level branch.
#define ML_TAB_LO ((127 - 12) << 7)
#define ML_TAB_N (15 << 7)                     // entries per sign
typedef __attribute__((address_space(3))) const uint16_t ml_lds_u16;
__device__ __forceinline__ uint32_t ml_gelu2_table(uint32_t w, ml_lds_u16* tab) {
  const uint32_t m0 = (w & 0x7fffu) - ML_TAB_LO, m1 = ((w >> 16) & 0x7fffu) - ML_TAB_LO;
  const bool in0 = m0 < (uint32_t)ML_TAB_N, in1 = m1 < (uint32_t)ML_TAB_N;
  const uint32_t i0 = (in0 ? m0 : 0u) + ((w & 0x8000u) ? ML_TAB_N : 0), i1 = (in1 ? m1 : 0u) + ((w & 0x80000000u) ? ML_TAB_N : 0);
  uint32_t r = (uint32_t)tab[i0] | ((uint32_t)tab[i1] << 16);
  if (__builtin_expect(!(in0 && in1), 0)) {
    const uint32_t c = ml_gelu2_computed<TS_BF16>(w);
    r = (in0 ? r & 0xffffu : c & 0xffffu) | (in1 ? r & 0xffff0000u : c & 0xffff0000u);
  }
  return r;
}
template <int DT>
__device__ __forceinline__ uint32_t ml_gelu2(uint32_t w, ml_lds_u16* tab) {
  if constexpr (DT == TS_BF16) return ml_gelu2_table(w, tab);
  else return ml_gelu2_computed<DT>(w);
}

// acc = a x b (no accumulator input: the instruction's C operand is the inline constant 0).  An up phase starts with it
// instead of zeroed accumulators: 48 zero registers are loop-invariant, the compiler hoists them out of the chunk loop and
// they stay live across every phase (seen: v1-v49 untouched through the down phases while pairs of `pk` were in scratch).
template <int DT>
__device__ __forceinline__ void ml_mma_first(f32x16& acc, const u32x4& a, const u32x4& b) {
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  typedef __bf16 ml_bf8 __attribute__((ext_vector_type(8)));
  typedef _Float16 ml_h8 __attribute__((ext_vector_type(8)));
  if constexpr (DT == TS_F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ml_h8, a), __builtin_bit_cast(ml_h8, b), z, 0, 0, 0);
  else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ml_bf8, a), __builtin_bit_cast(ml_bf8, b), z, 0, 0, 0);
}

template <int DT, bool MIX, bool FRESH = false>
__device__ __forceinline__ void ml_phase(f32x16 (&acc)[ML_QH], u32x4 (&ring)[ML_RING], const u32x4* cur, const u32x4* nxt,
                                         const u32x4* ql0, const u32x4* ql1, uint32_t (&pk)[ML_QH][8], int lane,
                                         ml_lds_u16* tab) {
  // (cur / nxt are WAVE-UNIFORM pointers: scalar registers, the lane's 16 bytes come in as the load's 32-bit offset — with
  // per-lane 64-bit pointers for this block, the next one and both weights the kernel spilled)
  const u32x4* tail = nxt ? nxt : cur + (size_t)(ML_KGC - ML_RING) * 64;   // (null: the last units of this block again)
#pragma unroll
  for (int r = 0; r < ML_KGC; r += ML_RING) {
#pragma unroll
    for (int i = 0; i < ML_RING; ++i) {
      const int s = r + i;
      const u32x4* ql = (i & 1) ? ql1 : ql0;
      // (bf16: the two table reads of pair s go out BEFORE the group's B-operand reads — LDS returns in order, so by the time
      // the MFMAs have their operands the table values are there too: no wait of their own)
      [[maybe_unused]] uint32_t t0 = 0, t1 = 0;
      [[maybe_unused]] bool in0 = true, in1 = true;
      if constexpr (MIX && DT == TS_BF16) {
        const uint32_t w = pk[s >> 3][s & 7];
        const uint32_t m0 = (w & 0x7fffu) - ML_TAB_LO, m1 = ((w >> 16) & 0x7fffu) - ML_TAB_LO;
        in0 = m0 < (uint32_t)ML_TAB_N; in1 = m1 < (uint32_t)ML_TAB_N;
#if defined(TS_TUNING) && defined(DBG_ML_TAB0)   // ablation (wrong results): conflict-free table addresses
        t0 = tab[lane]; t1 = tab[lane + 64];
#elif defined(TS_TUNING) && defined(DBG_ML_NOGELU)   // ablation (wrong results): no activation inside the down phases
        t0 = w & 0xffffu; t1 = w >> 16;
#else
        t0 = tab[(in0 ? m0 : 0u) + ((w & 0x8000u) ? ML_TAB_N : 0)];
        t1 = tab[(in1 ? m1 : 0u) + ((w & 0x80000000u) ? ML_TAB_N : 0)];
#endif
      }
#pragma unroll
      for (int hq = 0; hq < ML_QH; ++hq) {
        if (FRESH && s == 0) ml_mma_first<DT>(acc[hq], ring[i], ql[(size_t)(s * ML_QH + hq) * 64]);
        else mma_group<DT>(acc[hq], ring[i], ql[(size_t)(s * ML_QH + hq) * 64]);
      }
      ring[i] = s + ML_RING < ML_KGC ? cur[(size_t)(s + ML_RING) * 64 + lane] : tail[(size_t)(s + ML_RING - ML_KGC) * 64 + lane];
      if constexpr (MIX && DT == TS_BF16) {
        uint32_t r = t0 | (t1 << 16);
        if (__builtin_expect(!(in0 && in1), 0)) {
          const uint32_t c = ml_gelu2_computed<TS_BF16>(pk[s >> 3][s & 7]);
          r = (in0 ? r & 0xffffu : c & 0xffffu) | (in1 ? r & 0xffff0000u : c & 0xffff0000u);
        }
        pk[s >> 3][s & 7] = r;
      } else if constexpr (MIX) {
        pk[s >> 3][s & 7] = ml_gelu2_computed<DT>(pk[s >> 3][s & 7]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// NCH = I / 384 is a template parameter and the chunk loop is unrolled: as a run-time loop the output accumulators are
// loop-carried, the register allocator gave them a home (v2-v49) AND a working copy (24 v_mov_b64 in and out of every down
// phase), and with 96 registers for one accumulator set pairs of `pk` went to scratch — a scratch reload in every k group,
// each a vmcnt(0) = the ring drained.
template <int DT, int NCH>
__global__ __launch_bounds__(64 * ML_WAVES) void mlp_ln_kernel(MlParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* ximg = reinterpret_cast<u32x4*>(smem);
  u32x4* dimg = ximg + ML_IMG_UNITS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t m0 = (int64_t)blockIdx.x * ML_ROWS;
  const int j = lane & 31, h = lane >> 5;
  const int kg2 = p.I / 16;                                  // k groups of a W2 block
  [[maybe_unused]] const int ml_row = (int)(blockIdx.x * ML_WAVES + wave);   // (ML_TRACE builds)
  ML_STAMP(0);
  // ---- the first weight units (W1 block `wave` of chunk 0) go out before anything else
  const u32x4* w1 = p.w1_tiled;                              // (uniform) block b: w1 + b * ML_KGC * 64
  const u32x4* w2 = p.w2_tiled + (size_t)wave * kg2 * 64;    // (uniform) this wave's output block; chunk c: units 24 c ..
  u32x4 ring[ML_RING];
#pragma unroll
  for (int i = 0; i < ML_RING; ++i) ring[i] = w1[((size_t)wave * ML_KGC + i) * 64 + lane];
  // ---- the image of x: 16 lanes read 256 contiguous bytes of a row, three pieces per thread and pass
  {
    const int cc = tid & 15;
    for (int row = tid >> 4; row < ML_ROWS; row += (64 * ML_WAVES) >> 4) {
      const int64_t m = m0 + row < p.M ? m0 + row : p.M - 1;
      const uint16_t* src = p.x + m * ML_H + 8 * cc;
      u32x4 v[3];
#pragma unroll
      for (int jc = 0; jc < 3; ++jc) v[jc] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + 128 * jc));   // read once
#pragma unroll
      for (int jc = 0; jc < 3; ++jc) ximg[ml_unit(row, cc + 16 * jc)] = v[jc];
    }
  }
  // ---- bf16: the activation table (see ml_gelu2_table), behind the two images
  uint16_t* tab_w = reinterpret_cast<uint16_t*>(smem + 2 * (size_t)ML_IMG_UNITS * 16);
  ml_lds_u16* tab = (ml_lds_u16*)tab_w;
  if constexpr (DT == TS_BF16) {
    for (int e = tid; e < 2 * ML_TAB_N; e += 64 * ML_WAVES) {
      const uint32_t bits = (uint32_t)(ML_TAB_LO + (e < ML_TAB_N ? e : e - ML_TAB_N)) | (e < ML_TAB_N ? 0u : 0x8000u);
      tab_w[e] = (uint16_t)ml_gelu2_computed<TS_BF16>(bits);
    }
  }
  ML_STAMP(1);
  fs_barrier();
  ML_STAMP(2);
  const u32x4* xq0 = ximg + (lane ^ (h << 2));               // B operands of k group even / odd (the image's swizzle)
  const u32x4* xq1 = ximg + (lane ^ ((2 + h) << 2));
  const u32x4* dq0 = dimg + (lane ^ (h << 2));
  const u32x4* dq1 = dimg + (lane ^ ((2 + h) << 2));
  f32x16 accd[ML_QH], accu[ML_QH];   // (neither is zeroed: a set's first MFMAs take no accumulator input — FRESH)
  uint32_t pk[ML_QH][8];             // a chunk's up output (this wave's block) + bias, rounded, packed pairs; activated in place
  // + bias, rounded to the 16-bit type (the linear's output), packed: accu -> pk
  auto pack = [&](int c) {
    const MlBias b1 = ml_load_bias(p.b1, c * ML_WAVES + wave);   // bias of this chunk's block of the intermediate
#pragma unroll
    for (int hq = 0; hq < ML_QH; ++hq)
#pragma unroll
      for (int k = 0; k < 8; ++k)
        pk[hq][k] = (uint32_t)fs_from_f32<DT>(accu[hq][2 * k] + ml_bias<DT>(b1, 2 * k, h)) |
                    ((uint32_t)fs_from_f32<DT>(accu[hq][2 * k + 1] + ml_bias<DT>(b1, 2 * k + 1, h)) << 16);
  };
  // this wave's 12th of a chunk image: a lane's 8 consecutive features after the half-wave exchange are piece
  // 4 wave + 2 pr + h of row 32 hq + j (values 4 (2 pr + t) .. + 3 of the tile = pairs 2 (2 pr + t), 2 (2 pr + t) + 1)
  auto write_chunk = [&]() {
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int hq = 0; hq < ML_QH; ++hq) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(pk[hq][4 * pr], pk[hq][4 * pr + 2], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(pk[hq][4 * pr + 1], pk[hq][4 * pr + 3], false, false);
        dimg[ml_unit(32 * hq + j, 4 * wave + 2 * pr + h)] = u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
  };
  auto w1_block = [&](int c) { return w1 + (size_t)(c * ML_WAVES + wave) * ML_KGC * 64; };
  auto w2_chunk = [&](int c) { return w2 + (size_t)c * ML_KGC * 64; };
  // ---- up(0), then up(1) with chunk 0's activation between its MFMAs (one chunk only: the activation alone)
  ml_phase<DT, false, true>(accu, ring, w1_block(0), NCH > 1 ? w1_block(1) : w2_chunk(0), xq0, xq1, pk, lane, tab);
  ML_STAMP(3);
  pack(0);
  if constexpr (NCH > 1) {
    ml_phase<DT, true, true>(accu, ring, w1_block(1), w2_chunk(0), xq0, xq1, pk, lane, tab);
  } else {
#pragma unroll
    for (int hq = 0; hq < ML_QH; ++hq)
#pragma unroll
      for (int k = 0; k < 8; ++k) pk[hq][k] = ml_gelu2<DT>(pk[hq][k], tab);
  }
  ML_STAMP(4);
  write_chunk();                                             // (nobody has read the chunk image yet: no barrier before it)
  fs_barrier();                                              // (B) chunk 0's image is complete
  ML_STAMP(5);
#pragma unroll
  for (int c = 1; c < NCH; ++c) {
    pack(c);                                                 // up(c) ran before the previous chunk's image was written
    // ---- down(c - 1) with the activation of up(c)'s output between its MFMAs
    const u32x4* nxt = c + 1 < NCH ? w1_block(c + 1) : w2_chunk(c);
    if (c == 1) ml_phase<DT, true, true>(accd, ring, w2_chunk(c - 1), nxt, dq0, dq1, pk, lane, tab);
    else ml_phase<DT, true, false>(accd, ring, w2_chunk(c - 1), nxt, dq0, dq1, pk, lane, tab);
    fs_barrier();                                            // (A) every wave is done reading the previous chunk's image
    write_chunk();
    fs_barrier();                                            // (B) the chunk image is complete
    if (c + 1 < NCH) ml_phase<DT, false, true>(accu, ring, w1_block(c + 1), w2_chunk(c), xq0, xq1, pk, lane, tab);   // ---- up(c + 1)
  }
  // ---- down(last); its bias is requested first
  const MlBias b2 = ml_load_bias(p.b2, wave);
  ml_phase<DT, false, NCH == 1>(accd, ring, w2_chunk(NCH - 1), nullptr, dq0, dq1, pk, lane, tab);
  // (what the LayerNorm rows need from memory is requested HERE, behind the last phase's loads and before the barrier and
  // the staging of the output: its latency passes under both)
  // ---- LayerNorm rows: half a wave per row, lane lir of the half owns chunks c * 32 + lir (as add_layernorm_kernel<.., 3, 32>)
  const int lir = j, sub = h;
  constexpr int H = ML_H;
  f32x4 gm[3], bt[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int e = (c * 32 + lir) * 4;
    gm[c] = *reinterpret_cast<const f32x4*>(p.gamma + e);
    bt[c] = p.beta ? *reinterpret_cast<const f32x4*>(p.beta + e) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // All of this wave's residual rows are requested at once, before the barrier: 4 iterations x 3 quads = 48 registers that
  // are free here, and the rows' HBM latency passes once instead of once per iteration.
  constexpr int ITERS = ML_ROWS / 2 / ML_WAVES;    // 4 row pairs per wave
  static_assert(ITERS * ML_WAVES * 2 == ML_ROWS, "row pairs divide among the waves");
  f32x4 rs[ITERS][3];
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int64_t row = m0 + 2 * (wave + it * ML_WAVES) + sub;
    const int64_t base = (row < p.M ? row : p.M - 1) * H;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      rs[it][c] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.res) rs[it][c] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.res + base + (c * 32 + lir) * 4));   // read once
    }
  }
  fs_barrier();                                              // both images are dead
  // ---- projection output (+ bias, rounded) -> staging tile [96][H] where the images were
  unsigned char* stg = smem;
#pragma unroll
  for (int pr = 0; pr < 2; ++pr)
#pragma unroll
    for (int hq = 0; hq < ML_QH; ++hq) {
      uint32_t w[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        uint16_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fs_from_f32<DT>(accd[hq][4 * (2 * pr + t) + e] + ml_bias<DT>(b2, 4 * (2 * pr + t) + e, h));
        w[t][0] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
        w[t][1] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
      }
      const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
      *reinterpret_cast<u32x4*>(stg + (size_t)(32 * hq + j) * ML_STAGE_STRIDE + 2 * (wave * 32 + 16 * pr + 8 * h)) = u32x4{s0[0], s1[0], s0[1], s1[1]};
    }
  ML_STAMP(6);
  fs_barrier();                                    // the staging tile is complete
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int lrow = 2 * (wave + it * ML_WAVES) + sub;
    const int64_t row = m0 + lrow;
    f32x4 v[3], y[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      v[c] = ln_load4<DT>(stg + (size_t)lrow * ML_STAGE_STRIDE, (c * 32 + lir) * 4);
      if (p.res) v[c] += rs[it][c];
    }
    ln_row<3, 32>(v, gm, bt, H, lir, p.eps, y);
    if (row < p.M) {
      const int64_t base = row * H;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int e = (c * 32 + lir) * 4;
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
  ML_STAMP(7);
}

template <int DT, int NCH>
static int ml_launch_n(const MlParams& p, hipStream_t s) {
  auto kern = mlp_ln_kernel<DT, NCH>;
  static TsDeviceOnce attr;
  TS_CHECK(ts_allow_max_lds(attr, reinterpret_cast<const void*>(kern)));
  const size_t lds = 2 * (size_t)ML_IMG_UNITS * 16 + 2 * 2 * ML_TAB_N;   // (the staging tile, 96 x 784 bytes, fits inside the images; + the bf16 activation table)
  const int64_t grid = (p.M + ML_ROWS - 1) / ML_ROWS;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * ML_WAVES), lds, s, p);
  TS_HIP(hipGetLastError());
  return TS_OK;
}
template <int DT>
static int ml_launch(const MlParams& p, hipStream_t s) {
  switch (p.nchunk) {
    case 1: return ml_launch_n<DT, 1>(p, s);
    case 2: return ml_launch_n<DT, 2>(p, s);
    case 3: return ml_launch_n<DT, 3>(p, s);
    case 4: return ml_launch_n<DT, 4>(p, s);
  }
  ts_set_error("mlp_add_layernorm: I = %d not supported (384, 768, 1152 or 1536)", p.I);
  return TS_ERR_UNSUPPORTED;
}

extern "C" int ts_mlp_add_layernorm(const void* w1_tiled, const void* b1, const void* w2_tiled, const void* b2, const void* x,
                                    const float* residual, const float* gamma, const float* beta, float eps, int32_t dtype,
                                    int64_t M, int32_t H, int32_t I, float* out_f32, void* out_lp, int32_t device, void* stream) {
  if (M == 0) return TS_OK;
  if (!w1_tiled || !w2_tiled || !x || !gamma || (!out_f32 && !out_lp) || M < 0 || H <= 0 || I <= 0 ||
      (dtype != TS_F16 && dtype != TS_BF16)) {
    ts_set_error("bad arguments to mlp_add_layernorm");
    return TS_ERR_INVALID;
  }
  const uintptr_t al = reinterpret_cast<uintptr_t>(w1_tiled) | reinterpret_cast<uintptr_t>(w2_tiled) | reinterpret_cast<uintptr_t>(x) |
                       reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) |
                       reinterpret_cast<uintptr_t>(out_f32) | reinterpret_cast<uintptr_t>(out_lp);
  if (H != ML_H || (I % (16 * ML_KGC)) || (al & 15) || ((reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(b2)) & 7) ||
      (M + ML_ROWS - 1) / ML_ROWS > 0x7fffffff) {
    ts_set_error("mlp_add_layernorm: H = %d (must be %d), I = %d (multiple of %d) or alignment not supported", H, ML_H, I, 16 * ML_KGC);
    return TS_ERR_UNSUPPORTED;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != device) TS_HIP(hipSetDevice(device));
  MlParams p;
  p.w1_tiled = (const u32x4*)w1_tiled; p.w2_tiled = (const u32x4*)w2_tiled; p.b1 = (const uint16_t*)b1; p.b2 = (const uint16_t*)b2;
  p.x = (const uint16_t*)x; p.res = residual; p.gamma = gamma; p.beta = beta; p.eps = eps;
  p.out_f32 = out_f32; p.out_lp = (uint16_t*)out_lp; p.M = M; p.I = I; p.nchunk = I / (16 * ML_KGC);
  hipStream_t s = (hipStream_t)stream;
  const int st = dtype == TS_BF16 ? ml_launch<TS_BF16>(p, s) : ml_launch<TS_F16>(p, s);
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  return st;
}
