// Fused residual-add + LayerNorm for the cross-encoder / encoder forwards (gfx950).
//
// The transformer forwards of the path are PyTorch-ROCm GEMMs and attention (BASELINE north_star); what sits
// BETWEEN the GEMMs of a post-LN encoder layer (BERT / RoBERTa / XLM-R: the cross-encoder the reference reaches
// through CrossEncoder.predict, reference src/stage3_reranker.py:127-131) is three elementwise kernels under
// autocast — residual add (bf16 + fp32 -> fp32), LayerNorm (fp32 -> fp32), cast for the next GEMM (fp32 -> bf16):
// 24 bytes per element of HBM traffic.  At the batch sizes of search_many (1024 pairs x ~110 tokens x H = 384)
// the activations are 10^8 elements and these passes are a fifth of the forward.  Here they are ONE pass:
//     y = LayerNorm(x + residual) * gamma + beta      (statistics and arithmetic in fp32, like torch.layer_norm)
// read x (16-bit or fp32) and the fp32 residual once, write y in fp32 (the next residual) and in the 16-bit type
// (the next GEMM's input): 12 bytes per element.  One wave per row, the row in registers, two-pass mean / variance.
#include "ts_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LN_MAX_CHUNKS 8   // 4-element chunks per lane: H <= 2048

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

template <int XDT>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const void* x, const float* res, const float* gamma,
                                                            const float* beta, float eps, int64_t rows, int H,
                                                            float* out_f32, void* out_lp, int lp_dt) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int nch = (H / 4 + 63) / 64;   // chunks per lane (host: <= LN_MAX_CHUNKS)
  for (int64_t row = wave; row < rows; row += nwaves) {
    const int64_t base = row * H;
    f32x4 v[LN_MAX_CHUNKS];
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
      const int e = (c * 64 + lane) * 4;
      v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < nch && e < H) {
        v[c] = ln_load4<XDT>(x, base + e);
        if (res) v[c] += *reinterpret_cast<const f32x4*>(res + base + e);
        sum += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
      const int e = (c * 64 + lane) * 4;
      if (c < nch && e < H) {
        const f32x4 d = v[c] - mean;
        sq += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    const float rstd = 1.0f / sqrtf(sq / (float)H + eps);
#pragma unroll
    for (int c = 0; c < LN_MAX_CHUNKS; ++c) {
      const int e = (c * 64 + lane) * 4;
      if (c < nch && e < H) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + e), b = *reinterpret_cast<const f32x4*>(beta + e);
        const f32x4 y = (v[c] - mean) * rstd * g + b;
        if (out_f32) *reinterpret_cast<f32x4*>(out_f32 + base + e) = y;
        if (out_lp) {
          uint2 pk;
          pk.x = ln_pack2(y[0], y[1], lp_dt);
          pk.y = ln_pack2(y[2], y[3], lp_dt);
          *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out_lp) + base + e) = pk;
        }
      }
    }
  }
}

extern "C" int ts_add_layernorm(const void* x, int32_t x_dtype, const float* residual, const float* gamma,
                                const float* beta, float eps, int64_t rows, int32_t H, float* out_f32, void* out_lp,
                                int32_t lp_dtype, int32_t device, void* stream) {
  if (rows == 0) return TS_OK;
  if (!x || !gamma || !beta || rows < 0 || H <= 0 || (!out_f32 && !out_lp) ||
      (x_dtype != TS_F32 && x_dtype != TS_F16 && x_dtype != TS_BF16) || (out_lp && lp_dtype != TS_F16 && lp_dtype != TS_BF16)) {
    ts_set_error("bad arguments to add_layernorm");
    return TS_ERR_INVALID;
  }
  if ((H % 4) != 0 || H > LN_MAX_CHUNKS * 256) {
    ts_set_error("add_layernorm: H = %d not supported (multiple of 4, <= %d)", H, LN_MAX_CHUNKS * 256);
    return TS_ERR_UNSUPPORTED;
  }
  const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(gamma) |
                       reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(out_f32) | reinterpret_cast<uintptr_t>(out_lp);
  if (al & 15) {
    ts_set_error("add_layernorm: pointers must be 16-byte aligned");
    return TS_ERR_UNSUPPORTED;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != device) TS_HIP(hipSetDevice(device));
  const int64_t want = (rows + 3) / 4;
  const int grid = (int)(want < 256 * 16 ? want : 256 * 16);
  hipStream_t s = (hipStream_t)stream;
  switch (x_dtype) {
    case TS_F32: hipLaunchKernelGGL(add_layernorm_kernel<TS_F32>, dim3(grid), dim3(256), 0, s, x, residual, gamma, beta, eps, rows, H, out_f32, out_lp, lp_dtype); break;
    case TS_F16: hipLaunchKernelGGL(add_layernorm_kernel<TS_F16>, dim3(grid), dim3(256), 0, s, x, residual, gamma, beta, eps, rows, H, out_f32, out_lp, lp_dtype); break;
    default: hipLaunchKernelGGL(add_layernorm_kernel<TS_BF16>, dim3(grid), dim3(256), 0, s, x, residual, gamma, beta, eps, rows, H, out_f32, out_lp, lp_dtype); break;
  }
  const hipError_t e = hipGetLastError();
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  if (e != hipSuccess) { ts_set_error("add_layernorm launch failed: %s", hipGetErrorString(e)); return TS_ERR_HIP; }
  return TS_OK;
}
