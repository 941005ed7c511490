// Row arithmetic of the fused residual add + LayerNorm, shared by add_layernorm_kernel (ts_fwd.hip) and the projection
// with a LayerNorm epilogue (ts_linear.hip: proj_ln_kernel) so that the two produce the same bits from the same row.
#pragma once
#include "ts_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int XDT> __device__ __forceinline__ f32x4 ln_load4(const void* p, int64_t idx);
template <> __device__ __forceinline__ f32x4 ln_load4<TS_F32>(const void* p, int64_t idx) {
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + idx);
}
template <> __device__ __forceinline__ f32x4 ln_load4<TS_BF16>(const void* p, int64_t idx) {
  const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p) + idx);
  return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
               __uint_as_float(u.y & 0xffff0000u)};
}
template <> __device__ __forceinline__ f32x4 ln_load4<TS_F16>(const void* p, int64_t idx) {
  typedef _Float16 h4 __attribute__((ext_vector_type(4)));
  const h4 v = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(p) + idx);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ uint32_t ln_pack2(float a, float b, int dt) {
  if (dt == TS_F16) {
    const _Float16 x = (_Float16)a, y = (_Float16)b;
    return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
  }
  const __bf16 x = (__bf16)a, y = (__bf16)b;   // round to nearest even, like tensor.to(torch.bfloat16)
  return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
}

// One row held by LPR lanes (lane `lir` of them owns the 4-element chunks c * LPR + lir, zeros beyond H): two-pass
// mean / variance over the row, then y = (v - mean) * rstd * gamma + beta.  Every lane of the LPR must call it.
template <int NCH, int LPR>
__device__ __forceinline__ void ln_row(const f32x4 (&v)[NCH], const f32x4 (&g)[NCH], const f32x4 (&bt)[NCH], int H, int lir,
                                       float eps, f32x4 (&y)[NCH]) {
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) sum += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);   // (chunks beyond H are zeros)
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = (c * LPR + lir) * 4;
    if (e < H) {
      const f32x4 d = v[c] - mean;
      sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = 1.0f / sqrtf(sq / (float)H + eps);
#pragma unroll
  for (int c = 0; c < NCH; ++c) y[c] = (v[c] - mean) * rstd * g[c] + bt[c];
}
