// Feed-forward "up" projection with its bias and erf GELU in ONE kernel (gfx950):
//     C[M, N] = gelu(A[M, K] W[N, K]^T + bias[N])            A, W, bias, C 16-bit (bf16 / fp16), fp32 accumulation
// — what a BERT-family encoder layer (the cross-encoder the reference reaches through CrossEncoder.predict,
// reference src/stage3_reranker.py:127-131; BertIntermediate) computes between its two GEMMs.  As two library calls it is a
// GEMM that writes M x N 16-bit values (hipBLASLt, 277 us at 172 032 x 1536 x 384) and an elementwise pass that reads
// and rewrites them (179 us at 5.9 TB/s): the 4H-wide activation crosses HBM three times.  hipBLASLt's own GELU
// epilogue is the tanh approximation, not the erf GELU of these checkpoints, so the fusion is written out here.
//
// 128 x 128 output tile per workgroup (4 waves, 64 x 64 each = 2 x 2 blocks of v_mfma_f32_32x32x16), K walked in
// steps of 64 through double-buffered LDS tiles (rows padded by 16 bytes: conflict-free 16-byte fragment reads), the
// next step's global loads in flight under the MFMAs.  The rounding points are the unfused path's: the sum plus bias is
// rounded to the 16-bit type (the linear's output), GELU is evaluated in fp32 on that value and rounded again.  The
// finished tile goes through LDS so that it leaves in whole 256-byte rows.  Consecutive workgroups are the N tiles of one
// row block: the A rows are fetched from HBM once, W (N x K, ~1 MB) stays in L2.
#include "ts_common.h"
#include <algorithm>

typedef uint32_t ff_u4 __attribute__((ext_vector_type(4)));
typedef float ff_f16v __attribute__((ext_vector_type(16)));
typedef __bf16 ff_bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 ff_h8 __attribute__((ext_vector_type(8)));

#define FF_BM 128
#define FF_BN 128
#define FF_BK 32
#define FF_LDK (FF_BK + 8)      // LDS row stride of the operand tiles (elements)
#define FF_LDC (FF_BN + 8)      // ... of the staged output tile

template <int DT> __device__ __forceinline__ ff_f16v ff_mma(const ff_u4& a, const ff_u4& b, ff_f16v c) {
  if constexpr (DT == TS_F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ff_h8, a), __builtin_bit_cast(ff_h8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ff_bf8, a), __builtin_bit_cast(ff_bf8, b), c, 0, 0, 0);
}
template <int DT> __device__ __forceinline__ float ff_to_f32(uint16_t v) {
  if constexpr (DT == TS_F16) return (float)__builtin_bit_cast(_Float16, v);
  else return __uint_as_float((uint32_t)v << 16);
}
template <int DT> __device__ __forceinline__ uint16_t ff_from_f32(float v) {
  if constexpr (DT == TS_F16) return __builtin_bit_cast(uint16_t, (_Float16)v);
  else return __builtin_bit_cast(uint16_t, (__bf16)v);
}

// erf to 1.5e-7 absolute (Abramowitz & Stegun 7.1.26: a rational in t = 1 / (1 + p |x|) times exp(-x^2)) — the library erff
// is ~40 vector instructions with branches and made this epilogue three times as long as the K loop; after the
// rounding of GELU's result to 8 (bf16) or 11 (fp16) mantissa bits the two agree except on rare near-ties.
__device__ __forceinline__ float ff_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * ax * ax);
  const float r = fmaf(-poly, e, 1.0f);
  return copysignf(r, x);
}

struct FfnParams {
  const uint16_t *A, *W, *bias;   // [M, K], [N, K], [N] or null
  uint16_t* C;                    // [M, N]
  int64_t M;
  int N, K;
};

template <int DT>
__global__ __launch_bounds__(256) void ffn_up_gelu_kernel(FfnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* As = reinterpret_cast<uint16_t*>(smem);                 // [2][BM][LDK]
  uint16_t* Ws = As + 2 * FF_BM * FF_LDK;                            // [2][BN][LDK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * FF_BN;
  const int64_t m0 = (int64_t)blockIdx.y * FF_BM;
  const int K = p.K;
  // global -> LDS mapping: 4 chunks of 16 bytes of each operand tile per thread and K step
  constexpr int NLD = FF_BM * (FF_BK / 8) / 256;                    // 16-byte chunks of one operand tile per thread
  constexpr int CPR = FF_BK / 8;                                    // chunks per tile row
  int lrow[NLD], lcol[NLD];
  const uint16_t *ga[NLD], *gw[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int q = tid + 256 * i;
    lrow[i] = q / CPR;
    lcol[i] = (q % CPR) * 8;
    const int64_t am = m0 + lrow[i] < p.M ? m0 + lrow[i] : p.M - 1;      // (rows beyond M repeat the last one; never stored)
    ga[i] = p.A + am * K + lcol[i];
    gw[i] = p.W + (int64_t)(n0 + lrow[i]) * K + lcol[i];
  }
  ff_u4 ra[NLD], rw[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    ra[i] = *reinterpret_cast<const ff_u4*>(ga[i]);
    rw[i] = *reinterpret_cast<const ff_u4*>(gw[i]);
  }
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    *reinterpret_cast<ff_u4*>(As + lrow[i] * FF_LDK + lcol[i]) = ra[i];
    *reinterpret_cast<ff_u4*>(Ws + lrow[i] * FF_LDK + lcol[i]) = rw[i];
  }
  __syncthreads();

  ff_f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[i][j][x] = 0.f;

#ifdef FF_NO_LOOP    // ablation builds only
  const int nk = 1;
#else
  const int nk = K / FF_BK;
#endif
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        ra[i] = *reinterpret_cast<const ff_u4*>(ga[i] + (kt + 1) * FF_BK);
        rw[i] = *reinterpret_cast<const ff_u4*>(gw[i] + (kt + 1) * FF_BK);
      }
    }
    const uint16_t* at = As + buf * FF_BM * FF_LDK + (wm * 64 + r) * FF_LDK + 8 * h;
    const uint16_t* wt = Ws + buf * FF_BN * FF_LDK + (wn * 64 + r) * FF_LDK + 8 * h;
#pragma unroll
    for (int ks = 0; ks < FF_BK / 16; ++ks) {
      const ff_u4 a0 = *reinterpret_cast<const ff_u4*>(at + 16 * ks);
      const ff_u4 a1 = *reinterpret_cast<const ff_u4*>(at + 32 * FF_LDK + 16 * ks);
      const ff_u4 b0 = *reinterpret_cast<const ff_u4*>(wt + 16 * ks);
      const ff_u4 b1 = *reinterpret_cast<const ff_u4*>(wt + 32 * FF_LDK + 16 * ks);
      acc[0][0] = ff_mma<DT>(a0, b0, acc[0][0]);       // rows = m (A operand), columns = n (lane = n)
      acc[0][1] = ff_mma<DT>(a0, b1, acc[0][1]);
      acc[1][0] = ff_mma<DT>(a1, b0, acc[1][0]);
      acc[1][1] = ff_mma<DT>(a1, b1, acc[1][1]);
    }
    if (more) {
      uint16_t* an = As + (buf ^ 1) * FF_BM * FF_LDK;
      uint16_t* wnx = Ws + (buf ^ 1) * FF_BN * FF_LDK;
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        *reinterpret_cast<ff_u4*>(an + lrow[i] * FF_LDK + lcol[i]) = ra[i];
        *reinterpret_cast<ff_u4*>(wnx + lrow[i] * FF_LDK + lcol[i]) = rw[i];
      }
    }
    __syncthreads();
  }

  // ---- epilogue: + bias, round (the linear's output), erf GELU, round; staged in LDS, stored in whole rows
  uint16_t* Cs = reinterpret_cast<uint16_t*>(smem);                 // [BM][LDC]  (the operand tiles are done: barrier above)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int nl = wn * 64 + j * 32 + r;
    const float b = p.bias ? ff_to_f32<DT>(p.bias[n0 + nl]) : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        const int ml = wm * 64 + i * 32 + (x & 3) + 8 * (x >> 2) + 4 * h;
        const float u = ff_to_f32<DT>(ff_from_f32<DT>(acc[i][j][x] + b));
#ifdef FF_NO_GELU   // ablation builds only
        Cs[ml * FF_LDC + nl] = ff_from_f32<DT>(u);
#else
        Cs[ml * FF_LDC + nl] = ff_from_f32<DT>((u * 0.5f) * (1.0f + ff_erf(u * 0.70710678118654752440f)));
#endif
      }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = tid + 256 * i, row = q >> 4, c = (q & 15) * 8;
    if (m0 + row < p.M)
      *reinterpret_cast<ff_u4*>(p.C + (m0 + row) * p.N + n0 + c) = *reinterpret_cast<const ff_u4*>(Cs + row * FF_LDC + c);
  }
}

extern "C" int ts_ffn_up_gelu(const void* a, const void* w, const void* bias, int32_t dtype, int64_t M, int32_t N, int32_t K,
                              void* out, int32_t device, void* stream) {
  if (M == 0 || N == 0) return TS_OK;
  if (!a || !w || !out || M < 0 || N < 0 || K <= 0 || (dtype != TS_F16 && dtype != TS_BF16)) {
    ts_set_error("bad arguments to ffn_up_gelu");
    return TS_ERR_INVALID;
  }
  const uintptr_t al = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(out);
  if ((N % FF_BN) != 0 || (K % 64) != 0 || (al & 15) || (reinterpret_cast<uintptr_t>(bias) & 1) || (M + FF_BM - 1) / FF_BM > 65535) {
    ts_set_error("ffn_up_gelu: N = %d (multiple of %d), K = %d (multiple of %d), M or pointer alignment not supported", N, FF_BN, K, 64);
    return TS_ERR_UNSUPPORTED;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != device) TS_HIP(hipSetDevice(device));
  FfnParams p;
  p.A = (const uint16_t*)a; p.W = (const uint16_t*)w; p.bias = (const uint16_t*)bias; p.C = (uint16_t*)out;
  p.M = M; p.N = N; p.K = K;
  const size_t lds = std::max((size_t)2 * (FF_BM + FF_BN) * FF_LDK * 2, (size_t)FF_BM * FF_LDC * 2);   // operand double buffers / staged output tile
  const dim3 grid((unsigned)(N / FF_BN), (unsigned)((M + FF_BM - 1) / FF_BM));
  int st = TS_OK;
  if (dtype == TS_F16) {
    static TsDeviceOnce attr;
    st = ts_allow_max_lds(attr, reinterpret_cast<const void*>(ffn_up_gelu_kernel<TS_F16>));
    if (st == TS_OK) hipLaunchKernelGGL(ffn_up_gelu_kernel<TS_F16>, grid, dim3(256), lds, (hipStream_t)stream, p);
  } else {
    static TsDeviceOnce attr;
    st = ts_allow_max_lds(attr, reinterpret_cast<const void*>(ffn_up_gelu_kernel<TS_BF16>));
    if (st == TS_OK) hipLaunchKernelGGL(ffn_up_gelu_kernel<TS_BF16>, grid, dim3(256), lds, (hipStream_t)stream, p);
  }
  const hipError_t e = hipGetLastError();
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  if (st != TS_OK) return st;
  if (e != hipSuccess) { ts_set_error("ffn_up_gelu launch failed: %s", hipGetErrorString(e)); return TS_ERR_HIP; }
  return TS_OK;
}
