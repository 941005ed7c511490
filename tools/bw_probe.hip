// HBM read-bandwidth probe for MI355X: what a pure streaming read reaches with
// the access patterns the scan kernel could use.  Standalone (no torch).
//   hipcc -O3 --offload-arch=gfx950 tools/bw_probe.hip -o tools/bw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;} } while(0)

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}
__device__ __forceinline__ void fold(u32x4& a, const u32x4& v) { a[0]^=v[0]; a[1]^=v[1]; a[2]^=v[2]; a[3]^=v[3]; }

// pattern 0: linear front — unit u (1 KiB) is read by wave (u % W) at step u / W
template <bool NT, int UNROLL>
__global__ __launch_bounds__(512) void k_linear(const u32x4* p, size_t units, uint32_t* out) {
  const int lane = threadIdx.x & 63;
  const size_t W = (size_t)gridDim.x * (blockDim.x >> 6);
  size_t u = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  u32x4 acc = {0,0,0,0};
  for (; u + (UNROLL-1) * W < units; u += UNROLL * W) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) v[i] = ld<NT>(p + (u + i * W) * 64 + lane);
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) fold(acc, v[i]);
  }
  for (; u < units; u += W) fold(acc, ld<NT>(p + u * 64 + lane));
  if ((acc[0]^acc[1]^acc[2]^acc[3]) == 0x12345678u) out[0] = 1;
}
// pattern 1: each wave reads whole `blk` KiB blocks sequentially (the scan kernel's pattern)
template <bool NT, int UNROLL>
__global__ __launch_bounds__(512) void k_blocked(const u32x4* p, size_t nblocks, int blk, uint32_t* out) {
  const int lane = threadIdx.x & 63;
  const size_t W = (size_t)gridDim.x * (blockDim.x >> 6);
  size_t b = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  u32x4 acc = {0,0,0,0};
  for (; b < nblocks; b += W) {
    const u32x4* q = p + b * blk * 64 + lane;
    for (int g = 0; g < blk; g += UNROLL) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int i = 0; i < UNROLL; ++i) v[i] = ld<NT>(q + (size_t)(g + i) * 64);
#pragma unroll
      for (int i = 0; i < UNROLL; ++i) fold(acc, v[i]);
    }
  }
  if ((acc[0]^acc[1]^acc[2]^acc[3]) == 0x12345678u) out[0] = 1;
}
// pattern 2: a workgroup's 8 waves read adjacent blocks, and workgroup w of G takes the
// contiguous range [w*chunk, (w+1)*chunk) of blocks (per-CU contiguous regions)
template <bool NT, int UNROLL>
__global__ __launch_bounds__(512) void k_region(const u32x4* p, size_t nblocks, int blk, uint32_t* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const size_t per = (nblocks + gridDim.x - 1) / gridDim.x;
  const size_t lo = per * blockIdx.x, hi = (lo + per < nblocks) ? lo + per : nblocks;
  u32x4 acc = {0,0,0,0};
  for (size_t b = lo + wave; b < hi; b += nw) {
    const u32x4* q = p + b * blk * 64 + lane;
    for (int g = 0; g < blk; g += UNROLL) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int i = 0; i < UNROLL; ++i) v[i] = ld<NT>(q + (size_t)(g + i) * 64);
#pragma unroll
      for (int i = 0; i < UNROLL; ++i) fold(acc, v[i]);
    }
  }
  if ((acc[0]^acc[1]^acc[2]^acc[3]) == 0x12345678u) out[0] = 1;
}

// pattern 3: ROW-MAJOR rows read in MFMA A-fragment order (what a MaxSim kernel over a
// row-major token store must do): a wave owns a 32-row tile; v_mfma_32x32x16 layout:
// lane (r = l%32, h = l/32) reads 16 B at row r, byte 32g + 16h  -> 32 rows x 32 B per instruction.
// RPI = 16: the 16x16x32 layout, lane (r = l%16, q = l/16) reads row r, byte 64g + 16q (16 rows x 64 B).
template <bool NT, int UNROLL, int RPI>
__global__ __launch_bounds__(512) void k_rowfrag(const u32x4* p, size_t ntiles, int rowbytes, uint32_t* out) {
  const int lane = threadIdx.x & 63;
  const size_t W = (size_t)gridDim.x * (blockDim.x >> 6);
  size_t t = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  constexpr int PER = 64 / RPI;              // lanes per row per instruction
  const int r = lane % RPI, c = lane / RPI;
  const int steps = rowbytes / (16 * PER);
  u32x4 acc = {0,0,0,0};
  for (; t < ntiles; t += W) {
    for (int sub = 0; sub < 32 / RPI; ++sub) {
      const char* base = (const char*)p + (t * 32 + sub * RPI + r) * (size_t)rowbytes + 16 * c;
      for (int g = 0; g < steps; g += UNROLL) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) v[i] = ld<NT>((const u32x4*)(base + (size_t)(g + i) * 16 * PER));
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) fold(acc, v[i]);
      }
    }
  }
  if ((acc[0]^acc[1]^acc[2]^acc[3]) == 0x12345678u) out[0] = 1;
}

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? atof(argv[1]) : 15.36) * 1e9;
  const int blk = 48;  // KiB per block (768 x fp16 x 32 rows)
  const size_t nblocks = bytes / (blk * 1024);
  const size_t units = nblocks * blk;
  u32x4* p; uint32_t* out;
  CK(hipMalloc(&p, units * 1024)); CK(hipMalloc(&out, 4));
  CK(hipMemset(p, 1, units * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch) {
    float best = 1e9, sum = 0; const int reps = 6;
    for (int r = 0; r < reps + 1; ++r) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (r) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-44s avg %.3f ms  %.1f GB/s   best %.1f GB/s\n", name, sum / reps, units * 1024.0 / (sum / reps) / 1e6, units * 1024.0 / best / 1e6);
  };
  if (argc > 2) {   // row-major fragment-order reads only
    const int rowbytes = 1536; const size_t ntiles = units * 1024 / (32 * (size_t)rowbytes);
    for (int grid : {256, 512, 1024}) {
      char nm[128];
      snprintf(nm, sizeof nm, "rowfrag 32x32B     grid=%d x512 unroll8", grid);
      run(nm, [&]{ k_rowfrag<false, 8, 32><<<grid, 512>>>(p, ntiles, rowbytes, out); });
      snprintf(nm, sizeof nm, "rowfrag 32x32B nt  grid=%d x512 unroll8", grid);
      run(nm, [&]{ k_rowfrag<true, 8, 32><<<grid, 512>>>(p, ntiles, rowbytes, out); });
      snprintf(nm, sizeof nm, "rowfrag 16x64B     grid=%d x512 unroll8", grid);
      run(nm, [&]{ k_rowfrag<false, 8, 16><<<grid, 512>>>(p, ntiles, rowbytes, out); });
      snprintf(nm, sizeof nm, "rowfrag 16x64B nt  grid=%d x512 unroll8", grid);
      run(nm, [&]{ k_rowfrag<true, 8, 16><<<grid, 512>>>(p, ntiles, rowbytes, out); });
      snprintf(nm, sizeof nm, "rowfrag 32x32B nt  grid=%d x256 unroll12", grid * 2);
      run(nm, [&]{ k_rowfrag<true, 12, 32><<<grid * 2, 256>>>(p, ntiles, rowbytes, out); });
      snprintf(nm, sizeof nm, "blocked nt (contig) grid=%d x512 unroll8", grid);
      run(nm, [&]{ k_blocked<true, 8><<<grid, 512>>>(p, nblocks, blk, out); });
    }
    return 0;
  }
  for (int grid : {256, 512, 1024, 2048}) {
    char nm[128];
    snprintf(nm, sizeof nm, "linear      grid=%d x512 unroll8", grid);
    run(nm, [&]{ k_linear<false, 8><<<grid, 512>>>(p, units, out); });
    snprintf(nm, sizeof nm, "linear  nt  grid=%d x512 unroll8", grid);
    run(nm, [&]{ k_linear<true, 8><<<grid, 512>>>(p, units, out); });
    snprintf(nm, sizeof nm, "blocked     grid=%d x512 unroll8", grid);
    run(nm, [&]{ k_blocked<false, 8><<<grid, 512>>>(p, nblocks, blk, out); });
    snprintf(nm, sizeof nm, "blocked nt  grid=%d x512 unroll8", grid);
    run(nm, [&]{ k_blocked<true, 8><<<grid, 512>>>(p, nblocks, blk, out); });
    snprintf(nm, sizeof nm, "region  nt  grid=%d x512 unroll8", grid);
    run(nm, [&]{ k_region<true, 8><<<grid, 512>>>(p, nblocks, blk, out); });
  }
  run("linear  nt  grid=2048 x256 unroll4", [&]{ k_linear<true, 4><<<2048, 256>>>(p, units, out); });
  run("linear  nt  grid=4096 x256 unroll8", [&]{ k_linear<true, 8><<<4096, 256>>>(p, units, out); });
  run("linear  nt  grid=8192 x256 unroll2", [&]{ k_linear<true, 2><<<8192, 256>>>(p, units, out); });
  return 0;
}
