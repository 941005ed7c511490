// Device helpers shared by the projection kernels (ts_linear.hip: ffn_stream_kernel, proj_ln_kernel; ts_mlp.hip: mlp_ln_kernel).
#pragma once
#include "ts_scan_dev.h"

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


// lgkmcnt(0) + s_barrier: what a wave owes the others at these barriers is its LDS traffic; its global loads (the compute
// waves' ring, the loaders' next chunk / next image) stay in flight — __syncthreads() would add vmcnt(0)
__device__ __forceinline__ void fs_barrier() {
  __asm__ volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0); vmcnt / expcnt untouched
  __builtin_amdgcn_s_barrier();
  __asm__ volatile("" ::: "memory");
}

