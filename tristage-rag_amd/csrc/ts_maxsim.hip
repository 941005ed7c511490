// Stage-2 MaxSim for gfx950: all candidates of one query in one launch.
//
// Replaces the per-candidate loop of ColBERTScorer.rescore_candidates
// (reference src/stage2_rescorer.py:268-276) around _maxsim_score /
// _colbert_score (:167-201):  for query tokens Q[Lq,H] and document tokens
// D[Ld,H]:  S = normalize(Q) . normalize(D)^T,  m_i = max_j S_ij,
//   maxsim  = mean_i m_i          colbert = sum_i softmax(m)_i * m_i
// normalize = x / max(|x|_2, 1e-12) (torch.nn.functional.normalize).
//
// One 256-thread workgroup per document.  The four waves split the document's
// 32-token tiles; each wave computes 32x32 tiles of Q.D^T with the exact-f32
// MFMA (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, so the products match a
// CPU fp32 matmul to rounding) straight from global memory — both operands are
// "row per lane", so no LDS staging is needed — and gets the squared norms of
// its rows for free from the operand values it loads.  Row maxima are reduced
// with wave shuffles; only the per-row maxima cross LDS.
#include "ts_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MS_THREADS 256
#define MS_WAVES 4

template <typename T> struct Ld4;
template <> struct Ld4<float> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
    const float4 x = *reinterpret_cast<const float4*>(p);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  }
  static __device__ __forceinline__ float ld1(const float* p) { return *p; }
};
template <> struct Ld4<_Float16> {
  static __device__ __forceinline__ void ld(const _Float16* p, float (&v)[4]) {
    const uint2 x = *reinterpret_cast<const uint2*>(p);
    v[0] = (float)__builtin_bit_cast(_Float16, (uint16_t)(x.x & 0xffff));
    v[1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(x.x >> 16));
    v[2] = (float)__builtin_bit_cast(_Float16, (uint16_t)(x.y & 0xffff));
    v[3] = (float)__builtin_bit_cast(_Float16, (uint16_t)(x.y >> 16));
  }
  static __device__ __forceinline__ float ld1(const _Float16* p) { return (float)*p; }
};
template <> struct Ld4<__bf16> {
  static __device__ __forceinline__ void ld(const __bf16* p, float (&v)[4]) {
    const uint2 x = *reinterpret_cast<const uint2*>(p);
    v[0] = __builtin_bit_cast(float, x.x << 16);
    v[1] = __builtin_bit_cast(float, x.x & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, x.y << 16);
    v[3] = __builtin_bit_cast(float, x.y & 0xffff0000u);
  }
  static __device__ __forceinline__ float ld1(const __bf16* p) {
    return __builtin_bit_cast(float, (uint32_t)(*reinterpret_cast<const uint16_t*>(p)) << 16);
  }
};

__device__ __forceinline__ int ms_acc_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

template <typename T>
__global__ __launch_bounds__(MS_THREADS) void maxsim_kernel(
    const T* __restrict__ q, int Lq, const T* __restrict__ docs,
    const int32_t* __restrict__ doc_off, const int64_t* __restrict__ starts,
    const int32_t* __restrict__ lens, int H, int mode, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* msc = reinterpret_cast<float*>(smem);     // [nqt*32] final m_i
  float* pm = msc + ((Lq + 31) / 32) * 32;         // [MS_WAVES][32] per-wave row maxima
  float* qs = pm + MS_WAVES * 32;                  // [32] squared norms of the q tile
  float* red = qs + 32;                            // [MS_THREADS] reduction scratch

  const int doc = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // packed candidates (doc_off prefix sums) or rows [start, start+len) of a resident
  // token store (the stage-2 cache: no gather copy, the kernel reads in place)
  const int64_t d0 = starts ? starts[doc] : (int64_t)doc_off[doc];
  const int Ld = starts ? lens[doc] : (doc_off[doc + 1] - doc_off[doc]);
  if (Ld <= 0 || Lq <= 0) {
    // reference: a candidate that cannot be scored keeps 0.0
    // (src/stage2_rescorer.py:285-291)
    if (tid == 0) out[doc] = 0.f;
    return;
  }
  const int nqt = (Lq + 31) / 32;
  const int ndt = (Ld + 31) / 32;
  const int r = lane & 31, h = lane >> 5;
  const bool vec = (H % 8) == 0;

  for (int qt = 0; qt < nqt; ++qt) {
    const int qi = qt * 32 + r;
    const T* qrow = q + (size_t)(qi < Lq ? qi : Lq - 1) * H;
    float rowmax[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) rowmax[x] = -3.402823466e38f;
    float qsq = 0.f;

    for (int dt = wave; dt < ndt; dt += MS_WAVES) {
      const int dj = dt * 32 + r;
      const T* drow = docs + (size_t)(d0 + (dj < Ld ? dj : Ld - 1)) * H;
      f32x16 acc;
#pragma unroll
      for (int x = 0; x < 16; ++x) acc[x] = 0.f;
      float dsq = 0.f;
      const bool first = (dt == 0);  // wave 0, first tile: also norms of the q rows
      if (vec) {
        for (int k0 = 0; k0 < H; k0 += 8) {
          float a[4], b[4];
          Ld4<T>::ld(qrow + k0 + 4 * h, a);
          Ld4<T>::ld(drow + k0 + 4 * h, b);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
            dsq = fmaf(b[t], b[t], dsq);
            if (first) qsq = fmaf(a[t], a[t], qsq);
          }
        }
      } else {
        for (int k0 = 0; k0 < H; k0 += 8) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int k = k0 + 4 * h + t;
            const float a = (k < H) ? Ld4<T>::ld1(qrow + k) : 0.f;
            const float b = (k < H) ? Ld4<T>::ld1(drow + k) : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            dsq = fmaf(b, b, dsq);
            if (first) qsq = fmaf(a, a, qsq);
          }
        }
      }
      dsq += __shfl_xor(dsq, 32, 64);
      const float invd = 1.0f / fmaxf(sqrtf(dsq), 1e-12f);
      const bool valid = dj < Ld;
#pragma unroll
      for (int x = 0; x < 16; ++x) {
        float v = valid ? acc[x] * invd : -3.402823466e38f;
        v = fmaxf(v, __shfl_xor(v, 16, 64));
        v = fmaxf(v, __shfl_xor(v, 8, 64));
        v = fmaxf(v, __shfl_xor(v, 4, 64));
        v = fmaxf(v, __shfl_xor(v, 2, 64));
        v = fmaxf(v, __shfl_xor(v, 1, 64));
        rowmax[x] = fmaxf(rowmax[x], v);
      }
    }
    if (r == 0) {
#pragma unroll
      for (int x = 0; x < 16; ++x) pm[wave * 32 + ms_acc_row(x, lane)] = rowmax[x];
    }
    if (wave == 0) {
      qsq += __shfl_xor(qsq, 32, 64);
      if (lane < 32) qs[lane] = qsq;
    }
    __syncthreads();
    if (tid < 32) {
      const int i = qt * 32 + tid;
      float m = pm[tid];
#pragma unroll
      for (int wv = 1; wv < MS_WAVES; ++wv) m = fmaxf(m, pm[wv * 32 + tid]);
      const float invq = 1.0f / fmaxf(sqrtf(qs[tid]), 1e-12f);
      if (i < Lq) msc[i] = m * invq;
    }
    __syncthreads();
  }

  // ---- reduce over the Lq per-token maxima
  if (mode == 0) {
    float s = 0.f;
    for (int i = tid; i < Lq; i += MS_THREADS) s += msc[i];
    red[tid] = s;
    __syncthreads();
    for (int off = MS_THREADS / 2; off > 0; off >>= 1) {
      if (tid < off) red[tid] += red[tid + off];
      __syncthreads();
    }
    if (tid == 0) out[doc] = red[0] / (float)Lq;
  } else {
    float mx = -3.402823466e38f;
    for (int i = tid; i < Lq; i += MS_THREADS) mx = fmaxf(mx, msc[i]);
    red[tid] = mx;
    __syncthreads();
    for (int off = MS_THREADS / 2; off > 0; off >>= 1) {
      if (tid < off) red[tid] = fmaxf(red[tid], red[tid + off]);
      __syncthreads();
    }
    mx = red[0];
    __syncthreads();
    float num = 0.f, den = 0.f;
    for (int i = tid; i < Lq; i += MS_THREADS) {
      const float e = expf(msc[i] - mx);
      den += e;
      num += e * msc[i];
    }
    red[tid] = den;
    __syncthreads();
    for (int off = MS_THREADS / 2; off > 0; off >>= 1) {
      if (tid < off) red[tid] += red[tid + off];
      __syncthreads();
    }
    den = red[0];
    __syncthreads();
    red[tid] = num;
    __syncthreads();
    for (int off = MS_THREADS / 2; off > 0; off >>= 1) {
      if (tid < off) red[tid] += red[tid + off];
      __syncthreads();
    }
    if (tid == 0) out[doc] = red[0] / den;
  }
}

template <typename T>
static int launch_maxsim_t(const T* q, int Lq, const T* docs, const int32_t* off,
                           const int64_t* starts, const int32_t* lens,
                           int n_docs, int H, int mode, float* out, hipStream_t s) {
  const size_t lds = (size_t)(((Lq + 31) / 32) * 32 + MS_WAVES * 32 + 32 + MS_THREADS) * 4;
  if (lds > 60 * 1024) {
    ts_set_error("maxsim: query of %d tokens is too long", Lq);
    return TS_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(maxsim_kernel<T>, dim3(n_docs), dim3(MS_THREADS), lds, s, q, Lq,
                     docs, off, starts, lens, H, mode, out);
  TS_HIP(hipGetLastError());
  return TS_OK;
}

int ts_launch_maxsim(const void* q, int Lq, const void* docs, const int32_t* doc_off,
                     const int64_t* starts, const int32_t* lens,
                     int n_docs, int H, int dtype, int mode, float* out,
                     hipStream_t stream) {
  if (n_docs <= 0) return TS_OK;
  switch (dtype) {
    case TS_F32: return launch_maxsim_t<float>((const float*)q, Lq, (const float*)docs, doc_off, starts, lens, n_docs, H, mode, out, stream);
    case TS_F16: return launch_maxsim_t<_Float16>((const _Float16*)q, Lq, (const _Float16*)docs, doc_off, starts, lens, n_docs, H, mode, out, stream);
    case TS_BF16: return launch_maxsim_t<__bf16>((const __bf16*)q, Lq, (const __bf16*)docs, doc_off, starts, lens, n_docs, H, mode, out, stream);
  }
  ts_set_error("bad dtype %d", dtype);
  return TS_ERR_INVALID;
}
