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

#define FS_THREADS 512   // 8 waves: each takes weight blocks w, w + 8, ...
#define FS_WAVES 8

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

struct FsParams {
  const u32x4* w_tiled;    // [N/32][kg][64]
  const uint16_t* x;       // [M, K]
  const uint16_t* bias;    // [N] or null
  uint16_t* out;           // [M, N]
  int64_t M;
  int N, K, kg, gelu;
};

template <int DT, int QH>
__global__ __launch_bounds__(FS_THREADS) void ffn_stream_kernel(FsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* qlds = reinterpret_cast<u32x4*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kg = p.kg, K = p.K;
  const int64_t m0 = (int64_t)blockIdx.x * (32 * QH);
  const int nblk = p.N / 32;
  // ---- the first weight loads go out before anything else
  const u32x4* base = p.w_tiled + lane;
  const size_t blk_units = (size_t)kg * 64;
  int blk = wave;
  const bool active = blk < nblk;
  const u32x4* cur = base + (size_t)(active ? blk : 0) * blk_units;
  u32x4 ring[TS_RING];
  if (active) {
#pragma unroll
    for (int i = 0; i < TS_RING; ++i) ring[i] = cur[(size_t)i * 64];
  }
  // ---- the activation image: row-major rows -> [g][hq][lane] 16-byte units (coalesced reads, scattered LDS writes)
  {
    const int cpr = K / 8;                              // 16-byte chunks per row
    const int nchunk = 32 * QH * cpr;
    for (int q0 = tid; q0 < nchunk; q0 += 8 * FS_THREADS) {
      u32x4 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int q = q0 + j * FS_THREADS;
        const int row = q / cpr, c = q % cpr;
        const int64_t m = m0 + row < p.M ? m0 + row : p.M - 1;
        t[j] = (q < nchunk) ? *reinterpret_cast<const u32x4*>(p.x + m * K + 8 * c) : u32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int q = q0 + j * FS_THREADS;
        if (q < nchunk) {
          const int row = q / cpr, c = q % cpr;
          qlds[(size_t)((c >> 1) * QH + (row >> 5)) * 64 + 32 * (c & 1) + (row & 31)] = t[j];
        }
      }
    }
  }
  __syncthreads();
  if (!active) return;
  const u32x4* ql = qlds + lane;
  const int j = lane & 31, h = lane >> 5;
  while (true) {
    const int blkn = blk + FS_WAVES;
    const bool has_next = blkn < nblk;
    const u32x4* nxt = base + (size_t)(has_next ? blkn : blk) * blk_units;
    f32x16 acc[QH];
#pragma unroll
    for (int hq = 0; hq < QH; ++hq)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[hq][r] = 0.f;
    int g0 = 0;
    for (; g0 < kg - TS_RING; g0 += TS_RING) {
#pragma unroll
      for (int i = 0; i < TS_RING; ++i) {
#pragma unroll
        for (int hq = 0; hq < QH; ++hq) mma_group<DT>(acc[hq], ring[i], ql[(size_t)((g0 + i) * QH + hq) * 64]);
        ring[i] = cur[(size_t)(g0 + i + TS_RING) * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < TS_RING; ++i) {
#pragma unroll
      for (int hq = 0; hq < QH; ++hq) mma_group<DT>(acc[hq], ring[i], ql[(size_t)((g0 + i) * QH + hq) * 64]);
      ring[i] = nxt[(size_t)i * 64];
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: rows of acc = 32 output features of this block, column = activation row j of quarter hq.  A lane holds 4
    // consecutive features per register group; groups 2p and 2p+1 are exchanged with lane ^ 32 so that each lane ends up
    // with 8 consecutive features (16 bytes) per pair: half 0 gets features 16p + 0..7, half 1 features 16p + 8..15
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      float b[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int n = blk * 32 + 8 * (2 * pr + t) + 4 * h;
        b[t][0] = b[t][1] = b[t][2] = b[t][3] = 0.f;
        if (p.bias) {
          const uint2 bb = *reinterpret_cast<const uint2*>(p.bias + n);
          b[t][0] = fs_to_f32<DT>((uint16_t)bb.x); b[t][1] = fs_to_f32<DT>((uint16_t)(bb.x >> 16));
          b[t][2] = fs_to_f32<DT>((uint16_t)bb.y); b[t][3] = fs_to_f32<DT>((uint16_t)(bb.y >> 16));
        }
      }
#pragma unroll
      for (int hq = 0; hq < QH; ++hq) {
        uint32_t w[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          uint16_t o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float u = fs_to_f32<DT>(fs_from_f32<DT>(acc[hq][4 * (2 * pr + t) + e] + b[t][e]));
            o[e] = p.gelu ? fs_from_f32<DT>((u * 0.5f) * (1.0f + fs_erf(u * 0.70710678118654752440f))) : fs_from_f32<DT>(u);
          }
          w[t][0] = (uint32_t)o[0] | ((uint32_t)o[1] << 16);
          w[t][1] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
        }
        const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
        const int64_t m = m0 + 32 * hq + j;
        if (m < p.M) {
          const u32x4 pk = {s0[0], s1[0], s0[1], s1[1]};
          *reinterpret_cast<u32x4*>(p.out + m * p.N + blk * 32 + 16 * pr + 8 * h) = pk;
        }
      }
    }
    if (!has_next) break;
    blk = blkn;
    cur = nxt;
  }
}

template <int DT, int QH>
static int fs_launch(const FsParams& p, hipStream_t s) {
  auto kern = ffn_stream_kernel<DT, QH>;
  static TsDeviceOnce attr;
  TS_CHECK(ts_allow_max_lds(attr, reinterpret_cast<const void*>(kern)));
  const size_t lds = (size_t)p.kg * QH * 1024;
  const int64_t grid = (p.M + 32 * QH - 1) / (32 * QH);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(FS_THREADS), lds, s, p);
  TS_HIP(hipGetLastError());
  return TS_OK;
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
  const int qh = (size_t)(K / 16) * 3 * 1024 <= 160 * 1024 ? 3 : (size_t)(K / 16) * 2 * 1024 <= 160 * 1024 ? 2 : 1;
  const int gelu = act;
  const uintptr_t al = reinterpret_cast<uintptr_t>(w_tiled) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out);
  if ((N % 32) || (K % 128) || (size_t)(K / 16) * qh * 1024 > 160 * 1024 || (al & 15) || (reinterpret_cast<uintptr_t>(bias) & 7) ||
      (M + 32 * qh - 1) / (32 * qh) > 0x7fffffff) {
    ts_set_error("linear_act: N = %d (multiple of 32), K = %d (multiple of 128, at most 2560) or alignment not supported", N, K);
    return TS_ERR_UNSUPPORTED;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != device) TS_HIP(hipSetDevice(device));
  FsParams p;
  p.w_tiled = (const u32x4*)w_tiled; p.x = (const uint16_t*)x; p.bias = (const uint16_t*)bias; p.out = (uint16_t*)out;
  p.M = M; p.N = N; p.K = K; p.kg = K / 16; p.gelu = gelu;
  hipStream_t s = (hipStream_t)stream;
  int st = TS_ERR_INVALID;
  if (dtype == TS_BF16) st = qh == 3 ? fs_launch<TS_BF16, 3>(p, s) : qh == 2 ? fs_launch<TS_BF16, 2>(p, s) : fs_launch<TS_BF16, 1>(p, s);
  else st = qh == 3 ? fs_launch<TS_F16, 3>(p, s) : qh == 2 ? fs_launch<TS_F16, 2>(p, s) : fs_launch<TS_F16, 1>(p, s);
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  return st;
}
