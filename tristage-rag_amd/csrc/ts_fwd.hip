// HIP kernels between the GEMMs of the encoder forwards (gfx950): residual add + LayerNorm (post-LN and pre-LN), embedding
// gather + LayerNorm, self-attention of a right-padded batch, rotary embedding, gated GELU.  The GEMMs themselves stay
// with PyTorch-ROCm (BASELINE north_star); the host side that strings them together is tristage_rag_amd/encoders.py
// (LeanBertEncoder / LeanBertClassifier / LeanModernBertEncoder).
//
// ---- fused residual add + LayerNorm
// What sits BETWEEN the GEMMs of a post-LN encoder layer (BERT / RoBERTa / XLM-R: the cross-encoder the reference reaches
// through CrossEncoder.predict, reference src/stage3_reranker.py:127-131) is three elementwise kernels under
// autocast — residual add (bf16 + fp32 -> fp32), LayerNorm (fp32 -> fp32), cast for the next GEMM (fp32 -> bf16):
// 24 bytes per element of HBM traffic.  At the batch sizes of search_many (1024 pairs x ~110 tokens x H = 384)
// the activations are 10^8 elements and these passes are a fifth of the forward.  Here they are ONE pass:
//     y = LayerNorm(x + residual) * gamma + beta      (statistics and arithmetic in fp32, like torch.layer_norm)
// read x (16-bit or fp32) and the fp32 residual once, write y in fp32 (the next residual) and in the 16-bit type
// (the next GEMM's input): 12 bytes per element.  Persistent waves, one row per wave (two when a row is at most 96
// four-element chunks), the row in registers, two-pass mean / variance, the next row's loads issued before the reductions.
#include "ts_common.h"
#include <initializer_list>

#include "ts_ln_dev.h"   // ln_load4 / ln_pack2 / ln_row: the row arithmetic, shared with ts_linear.hip
#define LN_MAX_CHUNKS 8   // 4-element chunks per lane: H <= 2048

struct LnParams {
  const void* x;            // [rows, H] of XDT — or, for the embedding variant, the fp32 word table [V, H]
  const float* res;         // fp32 [rows, H] or null
  const float *gamma, *beta;
  float eps;
  int64_t rows;
  int H;
  float* out_f32;
  void* out_lp;
  int lp_dt;
  int prenorm;              // out_f32 receives x + residual (the residual stream of a pre-LN model) instead of y
  // embedding variant: row r = (word[ids[r]] + type[type_ids[r] or 0]) + position[pos_ids[r]]
  const int64_t *ids, *pos_ids, *type_ids;
  const float *pos_tab, *typ_tab;
};

// One row into registers: LPR lanes per row (64, or 32 = two rows per wave), NCH 4-element chunks per lane.
template <int XDT, int NCH, int LPR, bool EMB>
__device__ __forceinline__ void ln_fetch(const LnParams& p, int64_t row, int lir, f32x4 (&v)[NCH]) {
  if constexpr (EMB) {
    const int64_t w = p.ids[row] * p.H, ps = p.pos_ids[row] * p.H, t = p.type_ids ? p.type_ids[row] * p.H : 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int e = (c * LPR + lir) * 4;
      v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (e < p.H)
        v[c] = (ln_load4<TS_F32>(p.x, w + e) + *reinterpret_cast<const f32x4*>(p.typ_tab + t + e)) +
               *reinterpret_cast<const f32x4*>(p.pos_tab + ps + e);
    }
  } else {
    const int64_t base = row * p.H;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int e = (c * LPR + lir) * 4;
      v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (e < p.H) {
        v[c] = ln_load4<XDT>(p.x, base + e);
        if (p.res) v[c] += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.res + base + e));   // read once
      }
    }
  }
}

// Persistent waves, 64 / LPR rows each per iteration; the NEXT rows' loads are issued before these rows' reductions so
// that the two dependent shuffle trees do not leave the memory pipe empty.  LPR = 32 for H <= 384 keeps every lane busy
// (H = 384 is 96 chunks of 4: three per lane of a half wave).
template <int XDT, int NCH, int LPR, bool EMB>
__global__ __launch_bounds__(256) void add_layernorm_kernel(LnParams p) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, lir = lane % LPR, sub = lane / LPR;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6) * RPW;
  const int H = p.H;
  f32x4 g[NCH], bt[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * LPR + lir) * 4;
    g[c] = bt[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (e < H) {
      g[c] = *reinterpret_cast<const f32x4*>(p.gamma + e);
      if (p.beta) bt[c] = *reinterpret_cast<const f32x4*>(p.beta + e);
    }
  }
  f32x4 v[NCH], nx[NCH];
  int64_t row = wave * RPW + sub;
  if (row < p.rows) ln_fetch<XDT, NCH, LPR, EMB>(p, row, lir, v);
  while (row < p.rows) {
    const int64_t next = row + stride;
    if (next < p.rows) ln_fetch<XDT, NCH, LPR, EMB>(p, next, lir, nx);
    f32x4 yv[NCH];
    ln_row<NCH, LPR>(v, g, bt, H, lir, p.eps, yv);
    const int64_t base = row * H;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int e = (c * LPR + lir) * 4;
      if (e < H) {
        const f32x4 y = yv[c];
        // (the fp32 stream is next read two GEMMs later: it need not displace the 16-bit copy the next GEMM reads at once)
        if (p.out_f32) __builtin_nontemporal_store(p.prenorm ? v[c] : y, reinterpret_cast<f32x4*>(p.out_f32 + base + e));
        if (p.out_lp) {
          uint2 pk;
          pk.x = ln_pack2(y[0], y[1], p.lp_dt);
          pk.y = ln_pack2(y[2], y[3], p.lp_dt);
          *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out_lp) + base + e) = pk;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) v[c] = nx[c];
    row = next;
  }
}

template <int XDT, bool EMB>
static int ln_launch(const LnParams& p, hipStream_t s) {
  const int q = p.H / 4;                                   // 4-element chunks per row
  const bool half = q <= 96;                               // two rows per wave
  const int nch = half ? (q + 31) / 32 : (q + 63) / 64;
  const int64_t want = half ? (p.rows + 7) / 8 : (p.rows + 3) / 4;
  const int grid = (int)(want < 256 * 8 ? want : 256 * 8);
#define LN_GO(NCH, LPR) hipLaunchKernelGGL((add_layernorm_kernel<XDT, NCH, LPR, EMB>), dim3(grid), dim3(256), 0, s, p)
  if (half) {
    if (nch <= 1) LN_GO(1, 32); else if (nch <= 2) LN_GO(2, 32); else LN_GO(3, 32);
  } else {
    if (nch <= 2) LN_GO(2, 64); else if (nch <= 3) LN_GO(3, 64); else if (nch <= 4) LN_GO(4, 64); else LN_GO(LN_MAX_CHUNKS, 64);
  }
#undef LN_GO
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { ts_set_error("layernorm launch failed: %s", hipGetErrorString(e)); return TS_ERR_HIP; }
  return TS_OK;
}

static bool ln_aligned(std::initializer_list<const void*> ps) {
  uintptr_t al = 0;
  for (const void* q : ps) al |= reinterpret_cast<uintptr_t>(q);
  return (al & 15) == 0;
}

static int add_layernorm_impl(const void* x, int32_t x_dtype, const float* residual, const float* gamma, const float* beta,
                              float eps, int64_t rows, int32_t H, float* out_f32, void* out_lp, int32_t lp_dtype, int prenorm,
                              int32_t device, void* stream) {
  if (rows == 0) return TS_OK;
  if (!x || !gamma || rows < 0 || H <= 0 || (!out_f32 && !out_lp) ||
      (x_dtype != TS_F32 && x_dtype != TS_F16 && x_dtype != TS_BF16) || (out_lp && lp_dtype != TS_F16 && lp_dtype != TS_BF16)) {
    ts_set_error("bad arguments to add_layernorm");
    return TS_ERR_INVALID;
  }
  if ((H % 4) != 0 || H > LN_MAX_CHUNKS * 256) {
    ts_set_error("add_layernorm: H = %d not supported (multiple of 4, <= %d)", H, LN_MAX_CHUNKS * 256);
    return TS_ERR_UNSUPPORTED;
  }
  if (!ln_aligned({x, residual, gamma, beta, out_f32, out_lp})) {
    ts_set_error("add_layernorm: pointers must be 16-byte aligned");
    return TS_ERR_UNSUPPORTED;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != device) TS_HIP(hipSetDevice(device));
  LnParams p = {};
  p.x = x; p.res = residual; p.gamma = gamma; p.beta = beta; p.eps = eps; p.rows = rows; p.H = H;
  p.out_f32 = out_f32; p.out_lp = out_lp; p.lp_dt = lp_dtype; p.prenorm = prenorm;
  hipStream_t s = (hipStream_t)stream;
  const int st = x_dtype == TS_F32 ? ln_launch<TS_F32, false>(p, s) : x_dtype == TS_F16 ? ln_launch<TS_F16, false>(p, s)
                                                                                        : ln_launch<TS_BF16, false>(p, s);
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  return st;
}

extern "C" int ts_add_layernorm(const void* x, int32_t x_dtype, const float* residual, const float* gamma,
                                const float* beta, float eps, int64_t rows, int32_t H, float* out_f32, void* out_lp,
                                int32_t lp_dtype, int32_t device, void* stream) {
  return add_layernorm_impl(x, x_dtype, residual, gamma, beta, eps, rows, H, out_f32, out_lp, lp_dtype, 0, device, stream);
}

extern "C" int ts_add_prenorm(const void* x, int32_t x_dtype, const float* residual, const float* gamma, const float* beta,
                              float eps, int64_t rows, int32_t H, float* out_sum, void* out_lp, int32_t lp_dtype,
                              int32_t device, void* stream) {
  return add_layernorm_impl(x, x_dtype, residual, gamma, beta, eps, rows, H, out_sum, out_lp, lp_dtype, 1, device, stream);
}

extern "C" int ts_embed_layernorm(const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids, const float* word_tab,
                                  const float* pos_tab, const float* typ_tab, const float* gamma, const float* beta, float eps,
                                  int64_t rows, int32_t H, float* out_f32, void* out_lp, int32_t lp_dtype, int32_t device,
                                  void* stream) {
  if (rows == 0) return TS_OK;
  if (!ids || !pos_ids || !word_tab || !pos_tab || !typ_tab || !gamma || rows < 0 || H <= 0 || (!out_f32 && !out_lp) ||
      (out_lp && lp_dtype != TS_F16 && lp_dtype != TS_BF16)) {
    ts_set_error("bad arguments to embed_layernorm");
    return TS_ERR_INVALID;
  }
  if ((H % 4) != 0 || H > LN_MAX_CHUNKS * 256 || !ln_aligned({word_tab, pos_tab, typ_tab, gamma, beta, out_f32, out_lp})) {
    ts_set_error("embed_layernorm: H = %d (multiple of 4, <= %d) or pointer alignment (16 bytes) not supported", H, LN_MAX_CHUNKS * 256);
    return TS_ERR_UNSUPPORTED;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != device) TS_HIP(hipSetDevice(device));
  LnParams p = {};
  p.x = word_tab; p.gamma = gamma; p.beta = beta; p.eps = eps; p.rows = rows; p.H = H;
  p.out_f32 = out_f32; p.out_lp = out_lp; p.lp_dt = lp_dtype;
  p.ids = ids; p.pos_ids = pos_ids; p.type_ids = type_ids; p.pos_tab = pos_tab; p.typ_tab = typ_tab;
  const int st = ln_launch<TS_F32, true>(p, (hipStream_t)stream);
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  return st;
}

// ---------------------------------------------------------------------------------------------------------------
// Self-attention of a right-padded batch without a mask tensor (gfx950).
//
// torch's scaled_dot_product_attention with a padding mask costs 0.8 ms per layer at 1024 x 168 tokens x 12 heads
// x 32 (0.35 ms without a mask, which padded batches cannot use; the jagged-tensor path: 1.07 ms;
// tools/sdpa_probe.py, tools/njt_probe.py) — 40 % of the cross-encoder forward of search_many.  Each sequence only
// needs its own `len` tokens: one workgroup per (head, sequence), K and V^T of the valid tokens staged in LDS,
// each wave takes query tiles of 32 and runs an online-softmax pass over the key tiles on the matrix cores:
//     S^T = K Q^T         v_mfma_f32_32x32x16_bf16, A = K rows straight from LDS, B = Q rows straight from global
//     P^T = exp(S^T - m)  in the accumulator layout (column = query = lane: the running max / sum are per lane)
//     O^T += V^T P^T      A = V^T rows from LDS, B = P^T: the accumulator registers of two k-halves exchanged
//                         between lane l and lane l^32, packed to 16 bit
// The qkv tensor is read in place ([B, L, 3, heads, dh], the output of the fused QKV GEMM); padded query positions
// are not computed (their rows of the output stay as the caller initialised them).
typedef __bf16 fw_bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 fw_h8 __attribute__((ext_vector_type(8)));
typedef float fw_f16v __attribute__((ext_vector_type(16)));
typedef uint32_t fw_u4 __attribute__((ext_vector_type(4)));

template <int DT>
__device__ __forceinline__ fw_f16v fw_mma(const fw_u4& a, const fw_u4& b, fw_f16v c) {
  if constexpr (DT == TS_F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(fw_h8, a), __builtin_bit_cast(fw_h8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(fw_bf8, a), __builtin_bit_cast(fw_bf8, b), c, 0, 0, 0);
}
template <int DT> __device__ __forceinline__ uint32_t fw_pack2(float a, float b) { return ln_pack2(a, b, DT); }
typedef float fw_f2 __attribute__((ext_vector_type(2)));
template <int DT> __device__ __forceinline__ uint32_t fw_pack2v(fw_f2 v) {     // one v_cvt_pk_{bf16,f16}_f32
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  if constexpr (DT == TS_F16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h2));
  else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2));
}


template <int DT> __device__ __forceinline__ float fw_to_f32(uint16_t v) {
  if constexpr (DT == TS_F16) return (float)__builtin_bit_cast(_Float16, v);
  else return __uint_as_float((uint32_t)v << 16);
}
template <int DT> __device__ __forceinline__ uint16_t fw_from_f32(float v) {
  if constexpr (DT == TS_F16) return __builtin_bit_cast(uint16_t, (_Float16)v);
  else return __builtin_bit_cast(uint16_t, (__bf16)v);
}

// Rotary embedding of one 8-element chunk (transformers' apply_rotary_pos_emb, exactly the arithmetic of rope_kernel
// below: fp32 products and sum each rounded, no fused multiply-add, one rounding to the 16-bit type): `own` holds
// x[d0 .. d0+8), `par` the partner chunk x[d0 +- DH/2 ...], cs / sn the table entries of the FIRST-half index (the tables
// repeat: cos[d + DH/2] == cos[d]); second = the chunk lies in the second half.
template <int DT>
__device__ __forceinline__ fw_u4 fw_rope8(const fw_u4& own, const fw_u4& par, const float* cs, const float* sn, bool second) {
#pragma clang fp contract(off)
  const float4 c0 = *reinterpret_cast<const float4*>(cs), c1 = *reinterpret_cast<const float4*>(cs + 4);
  const float4 s0 = *reinterpret_cast<const float4*>(sn), s1 = *reinterpret_cast<const float4*>(sn + 4);
  const float cf[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  const float sf[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  fw_u4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float x = fw_to_f32<DT>((uint16_t)(own[j] >> (16 * k))), xp = fw_to_f32<DT>((uint16_t)(par[j] >> (16 * k)));
      const float other = second ? xp : -xp;
      const float rot = (x * cf[2 * j + k]) + (other * sf[2 * j + k]);
      w |= (uint32_t)fw_from_f32<DT>(rot) << (16 * k);
    }
    o[j] = w;
  }
  return o;
}

struct AttnParams {
  const uint16_t* qkv;   // [B, L, 3, heads, DH] 16-bit
  const int32_t* lens;   // [B]
  uint16_t* out;         // [B, L, heads*DH]
  int L, heads;
  float scale;
  int window;            // > 0: query q sees key k only if |q - k| <= window (ModernBERT's local layers); 0 = all keys
  const float *rope_cos, *rope_sin;   // [L, DH] fp32 or null: rotary embedding applied to q and k as they are loaded
  const int32_t* offs;   // [B] or null: sequence b starts at token offs[b] of a PACKED batch (qkv [T, 3, heads, DH], out [T, H]);
                         // null: the padded layout, sequence b starts at token b * L
};

// Lanes l and l^32 exchange through v_permlane32_swap: swap(a, b) leaves a = [a.lo32, b.lo32], b = [a.hi32, b.hi32].
__device__ __forceinline__ float fw_max_halves(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float fw_sum_halves(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

#define ATTN_KPAD 8   // K rows in LDS are DH + 8 elements apart: the 16-byte fragment reads of 16 lanes hit 64 distinct banks

template <int DT, int DH, bool ROPE>
__global__ __launch_bounds__(256) void attn_varlen_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int head = blockIdx.x, b = blockIdx.y;
  const int len = min(p.lens[b], p.L);
  if (len <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwave = nthr >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int H = p.heads * DH;
  constexpr int KS = DH + ATTN_KPAD;
  const int64_t tstride = 3 * (int64_t)H;                          // elements between consecutive tokens
  const int64_t tok0 = p.offs ? (int64_t)p.offs[b] : (int64_t)b * p.L;     // first token of this sequence
  const uint16_t* base = p.qkv + tok0 * tstride + head * DH;
  const int ntile = (len + 31) >> 5, lp = ntile * 32;
  const int vts = lp + 8;                                          // row stride of V^T (elements): +8 against bank conflicts
  uint16_t* Ks = reinterpret_cast<uint16_t*>(smem);                // [lp][KS]
  uint16_t* Vt = Ks + (size_t)lp * KS;                             // [DH][vts]
  // ---- the first query tile's loads go out before everything else (see below)
  fw_u4 qnext[DH / 16];
  if constexpr (!ROPE) {
    const int qrow = min(wave * 32 + r, len - 1);
#pragma unroll
    for (int s = 0; s < DH / 16; ++s)
      qnext[s] = *reinterpret_cast<const fw_u4*>(base + (int64_t)qrow * tstride + 16 * s + 8 * h);
  }
  // ---- stage K (row-major) and V (transposed); rows beyond len are zeros
  if constexpr (ROPE) {
    // a thread takes a chunk of the first half of a row TOGETHER with its partner in the second half: each is the other's
    // rotate_half operand.  Two such pairs in flight
    constexpr int HC = DH / 16;                                    // chunks per half row
    const int npair = lp * HC;
    for (int i0 = tid; i0 < npair; i0 += 2 * nthr) {
      fw_u4 ka[2], kb[2], va[2], vb[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = i0 + u * nthr, t = i / HC, c = i % HC;
        ka[u] = kb[u] = va[u] = vb[u] = fw_u4{0u, 0u, 0u, 0u};
        if (i < npair && t < len) {
          const uint16_t* row = base + (int64_t)t * tstride;
          ka[u] = *reinterpret_cast<const fw_u4*>(row + H + 8 * c);
          kb[u] = *reinterpret_cast<const fw_u4*>(row + H + DH / 2 + 8 * c);
          va[u] = *reinterpret_cast<const fw_u4*>(row + 2 * H + 8 * c);
          vb[u] = *reinterpret_cast<const fw_u4*>(row + 2 * H + DH / 2 + 8 * c);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = i0 + u * nthr, t = i / HC, c = i % HC;
        if (i < npair) {
          fw_u4 ra = ka[u], rb = kb[u];
          if (t < len) {
            const float* cs = p.rope_cos + (int64_t)t * DH + 8 * c;
            const float* sn = p.rope_sin + (int64_t)t * DH + 8 * c;
            ra = fw_rope8<DT>(ka[u], kb[u], cs, sn, false);
            rb = fw_rope8<DT>(kb[u], ka[u], cs, sn, true);
          }
          *reinterpret_cast<fw_u4*>(Ks + (size_t)t * KS + 8 * c) = ra;
          *reinterpret_cast<fw_u4*>(Ks + (size_t)t * KS + DH / 2 + 8 * c) = rb;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            Vt[(size_t)(8 * c + 2 * j) * vts + t] = (uint16_t)(va[u][j] & 0xffffu);
            Vt[(size_t)(8 * c + 2 * j + 1) * vts + t] = (uint16_t)(va[u][j] >> 16);
            Vt[(size_t)(DH / 2 + 8 * c + 2 * j) * vts + t] = (uint16_t)(vb[u][j] & 0xffffu);
            Vt[(size_t)(DH / 2 + 8 * c + 2 * j + 1) * vts + t] = (uint16_t)(vb[u][j] >> 16);
          }
        }
      }
    }
  } else {
  // Four chunks per thread in flight
  const int nchunk = lp * (DH / 8);
  for (int i0 = tid; i0 < nchunk; i0 += 4 * nthr) {
    fw_u4 kv[4], vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * nthr, t = i / (DH / 8), c = i % (DH / 8);
      kv[u] = vv[u] = fw_u4{0u, 0u, 0u, 0u};
      if (i < nchunk && t < len) {
        kv[u] = *reinterpret_cast<const fw_u4*>(base + (int64_t)t * tstride + H + 8 * c);
        vv[u] = *reinterpret_cast<const fw_u4*>(base + (int64_t)t * tstride + 2 * H + 8 * c);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * nthr, t = i / (DH / 8), c = i % (DH / 8);
      if (i < nchunk) {
        *reinterpret_cast<fw_u4*>(Ks + (size_t)t * KS + 8 * c) = kv[u];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          Vt[(size_t)(8 * c + 2 * j) * vts + t] = (uint16_t)(vv[u][j] & 0xffffu);
          Vt[(size_t)(8 * c + 2 * j + 1) * vts + t] = (uint16_t)(vv[u][j] >> 16);
        }
      }
    }
  }
  }
  __syncthreads();
  const float c2 = p.scale * 1.44269504088896341f;                 // softmax in base 2: exp(x*scale - m) = exp2(x*c2 - m2)

  for (int qt = wave; qt < ntile; qt += nwave) {
    // ---- this tile's 32 queries as the B operand of S^T = K Q^T: lane (query r, half h) holds Q[query][16 s + 8h ..+8]
    // (rows beyond len repeat the last one; never stored.)  The NEXT tile's rows are requested now and used a whole key
    // loop later
    fw_u4 qf[DH / 16];
    if constexpr (ROPE) {
      // rotated as they are loaded (own chunk + its partner half a head away); no prefetch of the next tile here
      const int qrow = min(qt * 32 + r, len - 1);
      const uint16_t* row = base + (int64_t)qrow * tstride;
#pragma unroll
      for (int s = 0; s < DH / 16; ++s) {
        const int d0 = 16 * s + 8 * h, dp = d0 ^ (DH / 2), dt = d0 & (DH / 2 - 1);
        const fw_u4 own = *reinterpret_cast<const fw_u4*>(row + d0), par = *reinterpret_cast<const fw_u4*>(row + dp);
        qf[s] = fw_rope8<DT>(own, par, p.rope_cos + (int64_t)qrow * DH + dt, p.rope_sin + (int64_t)qrow * DH + dt, d0 >= DH / 2);
      }
    } else {
#pragma unroll
    for (int s = 0; s < DH / 16; ++s) qf[s] = qnext[s];
    if (qt + nwave < ntile) {
      const int qrow = min((qt + nwave) * 32 + r, len - 1);
#pragma unroll
      for (int s = 0; s < DH / 16; ++s)
        qnext[s] = *reinterpret_cast<const fw_u4*>(base + (int64_t)qrow * tstride + 16 * s + 8 * h);
    }
    }
    float m = -3.0e38f, l = 0.f;                                   // running max (base-2 scaled) and sum of this lane's query
    fw_f16v oacc[DH / 32];
#pragma unroll
    for (int d = 0; d < DH / 32; ++d)
#pragma unroll
      for (int x = 0; x < 16; ++x) oacc[d][x] = 0.f;

    // key tiles this query tile can see at all (a window leaves 2 * window / 32 + 3 of them at most)
    const int kt_lo = p.window > 0 ? max(0, (qt * 32 - p.window) >> 5) : 0;
    const int kt_hi = p.window > 0 ? min(ntile - 1, (qt * 32 + 31 + p.window) >> 5) : ntile - 1;
    for (int kt = kt_lo; kt <= kt_hi; ++kt) {
      fw_f16v s;
#pragma unroll
      for (int x = 0; x < 16; ++x) s[x] = 0.f;
#pragma unroll
      for (int st = 0; st < DH / 16; ++st) {
        const fw_u4 kf = *reinterpret_cast<const fw_u4*>(Ks + (size_t)(kt * 32 + r) * KS + 16 * st + 8 * h);
        s = fw_mma<DT>(kf, qf[st], s);                             // rows = keys, column = this lane's query
      }
      if (kt == ntile - 1 && (len & 31)) {                         // keys beyond len (last tile only)
#pragma unroll
        for (int x = 0; x < 16; ++x)
          if (kt * 32 + (x & 3) + 8 * (x >> 2) + 4 * h >= len) s[x] = -1.0e30f;   // (times c2 below: stays finite)
      }
      if (p.window > 0 && (kt * 32 + 31 - qt * 32 > p.window || qt * 32 + 31 - kt * 32 > p.window)) {   // tile straddles the window edge
        const int q = qt * 32 + r;
#pragma unroll
        for (int x = 0; x < 16; ++x) {
          const int dist = kt * 32 + (x & 3) + 8 * (x >> 2) + 4 * h - q;
          if (dist > p.window || -dist > p.window) s[x] = -1.0e30f;
        }
      }
      // ---- online softmax (per lane: one query; the other half of its keys sits in lane ^ 32).  Packed fp32 arithmetic
      // (v_pk_fma / v_pk_add), v_max3, v_cvt_pk: this loop is bound by the vector ALU, not by the matrix cores
      // (the scaled scores first: products are canonical numbers, so the maxima below fold into v_max3 without a
      // quieting v_max per operand — and no hand-written instruction reads a matrix-core result, whose wait states
      // only the compiler's own instructions get)
#pragma unroll
      for (int x = 0; x < 16; x += 2) {
        const fw_f2 t = fw_f2{s[x], s[x + 1]} * c2;
        s[x] = t[0];
        s[x + 1] = t[1];
      }
      float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
      for (int x = 3; x < 15; x += 2) mx = fmaxf(fmaxf(mx, s[x]), s[x + 1]);
      mx = fw_max_halves(fmaxf(mx, s[15]));
      if (__any(mx > m)) {                                         // some query's maximum moved: rescale what is accumulated
        const float mn = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        l *= alpha;
        m = mn;
#pragma unroll
        for (int d = 0; d < DH / 32; ++d)
#pragma unroll
          for (int x = 0; x < 16; ++x) oacc[d][x] *= alpha;
      }
      fw_f2 rs2 = {0.f, 0.f};
      uint32_t pk[8];
#pragma unroll
      for (int x = 0; x < 16; x += 2) {
        const fw_f2 t = fw_f2{s[x], s[x + 1]} - m;                 // (masked keys: exp2(-huge) = 0)
        const fw_f2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
        rs2 += e;
        pk[x >> 1] = fw_pack2v<DT>(e);
      }
      l += fw_sum_halves(rs2[0] + rs2[1]);
      // ---- O^T += V^T P^T, two k steps of 16 keys.  B operand: lane (query, h) needs P[query][16 s2 + 8h + j]; the
      // accumulator holds keys 16 s2 + {0..3 | 8..11} (+4h): packed to 16 bit, then lanes l and l^32 swap halves
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const auto e0 = __builtin_amdgcn_permlane32_swap(pk[4 * s2], pk[4 * s2 + 2], false, false);       // [0]: keys 0..1 | 8..9   [1]: 4..5 | 12..13
        const auto e1 = __builtin_amdgcn_permlane32_swap(pk[4 * s2 + 1], pk[4 * s2 + 3], false, false);   // [0]: keys 2..3 | 10..11 [1]: 6..7 | 14..15
        const fw_u4 pf = {e0[0], e1[0], e0[1], e1[1]};
#pragma unroll
        for (int d = 0; d < DH / 32; ++d) {
          const fw_u4 vf = *reinterpret_cast<const fw_u4*>(Vt + (size_t)(32 * d + r) * vts + kt * 32 + 16 * s2 + 8 * h);
          oacc[d] = fw_mma<DT>(vf, pf, oacc[d]);                   // rows = d, column = this lane's query
        }
      }
    }
    // ---- O = O^T / l, written as 4 consecutive d (8 bytes) per register group
    const int q = qt * 32 + r;
    if (q < len) {
      const float inv = 1.0f / l;
      uint16_t* orow = p.out + (tok0 + q) * H + head * DH;
#pragma unroll
      for (int d = 0; d < DH / 32; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 pk;
          pk.x = fw_pack2<DT>(oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv);
          pk.y = fw_pack2<DT>(oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv);
          *reinterpret_cast<uint2*>(orow + 32 * d + 8 * g + 4 * h) = pk;
        }
    }
  }
}

static size_t attn_lds_bytes(int L, int dh) {
  const int lp = (L + 31) / 32 * 32;
  return ((size_t)lp * (dh + ATTN_KPAD) + (size_t)dh * (lp + 8)) * 2;
}

template <int DT, int DH, bool ROPE>
static int launch_attn_t(const AttnParams& p, int B, size_t lds, hipStream_t s) {
  auto kern = attn_varlen_kernel<DT, DH, ROPE>;
  static TsDeviceOnce lds_attr;
  TS_CHECK(ts_allow_max_lds(lds_attr, reinterpret_cast<const void*>(kern)));
  // waves per workgroup: each takes query tiles w, w + waves, ...; the fewest waves that keep the number of rounds of
  // four (6 tiles: 3 waves x 2 rounds, not 4 waves of which two idle in the second round)
  const int ntile = (p.L + 31) / 32, rounds = (ntile + 3) / 4;
  int waves = 4;
  while (waves > 1 && (ntile + waves - 2) / (waves - 1) == rounds) --waves;
  hipLaunchKernelGGL(kern, dim3(p.heads, B), dim3(64 * waves), lds, s, p);
  TS_HIP(hipGetLastError());
  return TS_OK;
}
template <int DT, int DH>
static int launch_attn(const AttnParams& p, int B, size_t lds, hipStream_t s) {
  return p.rope_cos ? launch_attn_t<DT, DH, true>(p, B, lds, s) : launch_attn_t<DT, DH, false>(p, B, lds, s);
}

extern "C" int ts_attention_varlen(const void* qkv, const int32_t* lens, int32_t B, int32_t L, int32_t heads, int32_t dh,
                                   int32_t dtype, float scale, int32_t window, const float* rope_cos, const float* rope_sin,
                                   const int32_t* offs, void* out, int32_t device, void* stream) {
  if (B == 0 || L == 0) return TS_OK;
  if (!qkv || !lens || !out || B < 0 || L < 0 || heads <= 0 || window < 0 || (dtype != TS_F16 && dtype != TS_BF16) ||
      ((rope_cos == nullptr) != (rope_sin == nullptr))) {
    ts_set_error("bad arguments to attention_varlen");
    return TS_ERR_INVALID;
  }
  const size_t lds = attn_lds_bytes(L, dh);
  if ((dh != 32 && dh != 64) || lds > 160 * 1024 || B > 65535 ||
      ((reinterpret_cast<uintptr_t>(qkv) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(rope_cos) |
        reinterpret_cast<uintptr_t>(rope_sin)) & 15)) {
    ts_set_error("attention_varlen: head dimension %d / length %d / alignment not supported", dh, L);
    return TS_ERR_UNSUPPORTED;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != device) TS_HIP(hipSetDevice(device));
  AttnParams p;
  p.qkv = (const uint16_t*)qkv; p.lens = lens; p.out = (uint16_t*)out; p.L = L; p.heads = heads; p.scale = scale; p.window = window;
  p.rope_cos = rope_cos; p.rope_sin = rope_sin; p.offs = offs;
  int st;
  if (dtype == TS_F16) st = dh == 32 ? launch_attn<TS_F16, 32>(p, B, lds, (hipStream_t)stream) : launch_attn<TS_F16, 64>(p, B, lds, (hipStream_t)stream);
  else st = dh == 32 ? launch_attn<TS_BF16, 32>(p, B, lds, (hipStream_t)stream) : launch_attn<TS_BF16, 64>(p, B, lds, (hipStream_t)stream);
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  return st;
}


// ---------------------------------------------------------------------------------------------------------------
// Rotary position embedding on the q and k thirds of a fused projection, in place, and the gated GELU of the MLP
// (ModernBERT: the default stage-2 token encoder of the reference, src/stage2_rescorer.py:30, through AutoModel).
// Both restate transformers' arithmetic: rotation in fp32 from fp32 cos / sin tables — two products and a sum, each
// rounded, no fused multiply-add — then one rounding to the 16-bit type; gelu(x) = (x * 0.5) * (1 + erf(x / sqrt 2))
// in fp32 rounded to 16 bit, times the gate rounded again.
#pragma clang fp contract(off)
template <int DT>
__global__ __launch_bounds__(256) void rope_kernel(uint16_t* qkv, const float* cosv, const float* sinv, int64_t tokens, int L,
                                                   int heads, int dh) {
  // one thread: 4 consecutive d of the first half and their partners in the second half, for q and for k
  const int per_head = dh / 8;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= tokens * heads * per_head) return;
  const int c = (int)(i % per_head), head = (int)((i / per_head) % heads);
  const int64_t tok = i / ((int64_t)per_head * heads);
  const int t = (int)(tok % L), d0 = 4 * c, half = dh / 2;
  const float4 cs = *reinterpret_cast<const float4*>(cosv + (int64_t)t * dh + d0);
  const float4 sn = *reinterpret_cast<const float4*>(sinv + (int64_t)t * dh + d0);
  const float4 cs2 = *reinterpret_cast<const float4*>(cosv + (int64_t)t * dh + half + d0);
  const float4 sn2 = *reinterpret_cast<const float4*>(sinv + (int64_t)t * dh + half + d0);
  const float cf[4] = {cs.x, cs.y, cs.z, cs.w}, sf[4] = {sn.x, sn.y, sn.z, sn.w};
  const float cg[4] = {cs2.x, cs2.y, cs2.z, cs2.w}, sg[4] = {sn2.x, sn2.y, sn2.z, sn2.w};
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    uint16_t* row = qkv + (tok * 3 + which) * (int64_t)heads * dh + (int64_t)head * dh;
    uint2 a = *reinterpret_cast<const uint2*>(row + d0), b = *reinterpret_cast<const uint2*>(row + half + d0);
    uint16_t av[4] = {(uint16_t)a.x, (uint16_t)(a.x >> 16), (uint16_t)a.y, (uint16_t)(a.y >> 16)};
    uint16_t bv[4] = {(uint16_t)b.x, (uint16_t)(b.x >> 16), (uint16_t)b.y, (uint16_t)(b.y >> 16)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x1 = fw_to_f32<DT>(av[j]), x2 = fw_to_f32<DT>(bv[j]);
      const float lo = (x1 * cf[j]) + ((-x2) * sf[j]);      // q * cos + rotate_half(q) * sin, first half
      const float hi = (x2 * cg[j]) + (x1 * sg[j]);         // ... second half
      av[j] = fw_from_f32<DT>(lo);
      bv[j] = fw_from_f32<DT>(hi);
    }
    a.x = av[0] | ((uint32_t)av[1] << 16); a.y = av[2] | ((uint32_t)av[3] << 16);
    b.x = bv[0] | ((uint32_t)bv[1] << 16); b.y = bv[2] | ((uint32_t)bv[3] << 16);
    *reinterpret_cast<uint2*>(row + d0) = a;
    *reinterpret_cast<uint2*>(row + half + d0) = b;
  }
}

extern "C" int ts_rope_inplace(void* qkv, int32_t dtype, const float* cos_tab, const float* sin_tab, int64_t B, int32_t L,
                               int32_t heads, int32_t dh, int32_t device, void* stream) {
  if (B == 0 || L == 0) return TS_OK;
  if (!qkv || !cos_tab || !sin_tab || B < 0 || L < 0 || heads <= 0 || dh <= 0 || (dtype != TS_F16 && dtype != TS_BF16)) {
    ts_set_error("bad arguments to rope_inplace");
    return TS_ERR_INVALID;
  }
  if ((dh % 8) != 0 || !ln_aligned({qkv, cos_tab, sin_tab})) {
    ts_set_error("rope_inplace: head dimension %d (multiple of 8) or pointer alignment (16 bytes) not supported", dh);
    return TS_ERR_UNSUPPORTED;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != device) TS_HIP(hipSetDevice(device));
  const int64_t n = B * L * heads * (dh / 8);
  const int64_t grid = (n + 255) / 256;
  if (dtype == TS_F16) hipLaunchKernelGGL(rope_kernel<TS_F16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (uint16_t*)qkv, cos_tab, sin_tab, B * L, L, heads, dh);
  else hipLaunchKernelGGL(rope_kernel<TS_BF16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (uint16_t*)qkv, cos_tab, sin_tab, B * L, L, heads, dh);
  const hipError_t e = hipGetLastError();
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  if (e != hipSuccess) { ts_set_error("rope launch failed: %s", hipGetErrorString(e)); return TS_ERR_HIP; }
  return TS_OK;
}

template <int DT>
__global__ __launch_bounds__(256) void geglu_kernel(const uint16_t* u, uint16_t* out, int64_t rows, int I) {
  const int per_row = I / 8;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * per_row) return;
  const int64_t row = i / per_row;
  const int c = (int)(i % per_row);
  const fw_u4 a = *reinterpret_cast<const fw_u4*>(u + row * 2 * I + 8 * c);
  const fw_u4 g = *reinterpret_cast<const fw_u4*>(u + row * 2 * I + I + 8 * c);
  fw_u4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float x = fw_to_f32<DT>((uint16_t)(a[j] >> (16 * k))), gt = fw_to_f32<DT>((uint16_t)(g[j] >> (16 * k)));
      const float act = fw_to_f32<DT>(fw_from_f32<DT>((x * 0.5f) * (1.0f + erff(x * 0.70710678118654752440f))));
      w |= (uint32_t)fw_from_f32<DT>(act * gt) << (16 * k);
    }
    o[j] = w;
  }
  *reinterpret_cast<fw_u4*>(out + row * I + 8 * c) = o;
}

extern "C" int ts_geglu(const void* u, int32_t dtype, int64_t rows, int32_t I, void* out, int32_t device, void* stream) {
  if (rows == 0 || I == 0) return TS_OK;
  if (!u || !out || rows < 0 || I < 0 || (dtype != TS_F16 && dtype != TS_BF16)) {
    ts_set_error("bad arguments to geglu");
    return TS_ERR_INVALID;
  }
  if ((I % 8) != 0 || !ln_aligned({u, out})) {
    ts_set_error("geglu: width %d (multiple of 8) or pointer alignment (16 bytes) not supported", I);
    return TS_ERR_UNSUPPORTED;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != device) TS_HIP(hipSetDevice(device));
  const int64_t grid = (rows * (I / 8) + 255) / 256;
  if (dtype == TS_F16) hipLaunchKernelGGL(geglu_kernel<TS_F16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)u, (uint16_t*)out, rows, I);
  else hipLaunchKernelGGL(geglu_kernel<TS_BF16>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)u, (uint16_t*)out, rows, I);
  const hipError_t e = hipGetLastError();
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  if (e != hipSuccess) { ts_set_error("geglu launch failed: %s", hipGetErrorString(e)); return TS_ERR_HIP; }
  return TS_OK;
}
