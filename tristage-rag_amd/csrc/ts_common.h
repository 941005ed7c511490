// Internal declarations shared by the HIP translation units of libtristage.so.
// Not part of the public ABI (that is include/tristage.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/tristage.h"

// ---------------------------------------------------------------- errors
void ts_set_error(const char* fmt, ...);

#define TS_HIP(call)                                                        \
  do {                                                                      \
    hipError_t e_ = (call);                                                 \
    if (e_ != hipSuccess) {                                                 \
      ts_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),   \
                   __FILE__, __LINE__);                                     \
      return (e_ == hipErrorOutOfMemory) ? TS_ERR_OOM : TS_ERR_HIP;         \
    }                                                                       \
  } while (0)

#define TS_CHECK(call)            \
  do {                            \
    int s_ = (call);              \
    if (s_ != TS_OK) return s_;   \
  } while (0)

// ---------------------------------------------------------------- once per device
// hipFuncSetAttribute (MaxDynamicSharedMemorySize) is a PER-DEVICE property of a kernel: a
// process that drives several GPUs (ts_index_create / ts_maxsim / ts_merge_topk all take a
// `device`) has to set it on each of them, and two host threads may reach the first launch at
// the same time.  One TsDeviceOnce per kernel instantiation: a bit per device, set under a
// mutex after the action succeeded (a failed action is retried by the next caller).
#include <atomic>
#include <mutex>
#define TS_MAX_DEVICES 64
struct TsDeviceOnce {
  std::mutex mu;
  std::atomic<uint64_t> done{0};
};
template <class F>
static inline int ts_once_per_device(TsDeviceOnce& o, int dev, F&& action) {
  if (dev < 0 || dev >= TS_MAX_DEVICES) return action();  // untracked device: the action is idempotent
  const uint64_t bit = 1ull << dev;
  if (o.done.load(std::memory_order_acquire) & bit) return TS_OK;
  std::lock_guard<std::mutex> lk(o.mu);
  if (o.done.load(std::memory_order_relaxed) & bit) return TS_OK;
  const int st = action();
  if (st == TS_OK) o.done.fetch_or(bit, std::memory_order_release);
  return st;
}
// the 96-160 KiB LDS kernels: allow the full 160 KiB of dynamic LDS on the CURRENT device
static inline int ts_allow_max_lds(TsDeviceOnce& o, const void* kernel) {
  int dev = -1;
  TS_HIP(hipGetDevice(&dev));
  return ts_once_per_device(o, dev, [&]() -> int {
    TS_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return TS_OK;
  });
}

// ---------------------------------------------------------------- layout
//
// The corpus is NOT kept row-major.  It is stored in the order the MFMA A
// operand wants it, so that every wave-level load of the scan kernel is one
// fully contiguous 1 KiB read (64 lanes x 16 B):
//
//   row block  b = row / 32            (32 rows share one MFMA tile)
//   k group    g = 0 .. KG-1           (one 16-byte fragment per lane)
//   lane       l = h*32 + r,  r = row % 32,  h = 0/1
//
//   16-byte unit index = (b*KG + g)*64 + l
//
//   f16/bf16: the unit holds 8 elements, k = 16*g + 8*h + j   (j = 0..7)
//             -> exactly the A fragment of v_mfma_f32_32x32x16_{f16,bf16}
//   f32:      the unit holds 4 elements, k = 8*g + 2*t + h    (t = 0..3)
//             -> element t is the A operand of the t-th v_mfma_f32_32x32x2_f32
//
// The query batch is laid out the same way (the "Q image"): unit index
// (g*QH + hq)*64 + l holds, for query 32*hq + (l & 31), the same k's.
// dim is zero-padded to a multiple of TS_RING groups so the scan kernel's
// register ring never straddles a partial group.
#ifndef TS_RING
#define TS_RING 8          // k groups kept in flight per wave (1 KiB each)
#endif
#define TS_ROWS_PER_BLOCK 32
#define TS_MAX_Q 64        // queries per scan pass (2 MFMA column halves)

struct TsLayout {
  int dtype;   // ts_dtype of the stored corpus
  int esize;   // bytes per element
  int epl;     // elements per lane per group (8 or 4)
  int gk;      // k values covered by one group (16 or 8)
  int dim;     // logical dimension
  int dpad;    // padded dimension (multiple of gk*TS_RING)
  int kg;      // groups per row block = dpad / gk
};

static inline TsLayout ts_make_layout(int dim, int dtype) {
  TsLayout L;
  L.dtype = dtype;
  L.esize = (dtype == TS_F32) ? 4 : 2;
  L.epl = 16 / L.esize;
  L.gk = 2 * L.epl;
  L.dim = dim;
  int q = L.gk * TS_RING;
  L.dpad = ((dim + q - 1) / q) * q;
  L.kg = L.dpad / L.gk;
  return L;
}

static inline size_t ts_block_bytes(const TsLayout& L) {
  return (size_t)L.kg * 1024;  // 32 rows * dpad * esize
}

// ---------------------------------------------------------------- scan
struct ScanParams {
  const uint4* corpus;   // tiled corpus
  const uint4* qimg;     // Q image (KG*QH KiB)
  int kg;
  int nq;                // valid queries in this pass
  int64_t nwork;         // row blocks to process
  int64_t blk0;          // block index of work item w = blk0 + w*blk_stride
  int64_t blk_stride;
  int64_t ntotal;        // valid rows in the index
  // dense mode: dense[q*dense_ld + w*32 + i]
  float* dense;
  int64_t dense_ld;
  // filter mode
  const float* tau;       // [64] per-query lower bound (score >= tau passes)
  uint32_t* cand_cnt;     // [64]
  float* cand_score;      // [64][cand_cap]
  int32_t* cand_id;       // [64][cand_cap] local row ids
  uint32_t cand_cap;
};

enum { SCAN_DENSE = 0, SCAN_FILTER = 1 };

int ts_launch_scan(const TsLayout& L, int mode, int qh, const ScanParams& p,
                   int num_cus, hipStream_t stream);
// LDS bytes the scan kernel needs for qh*32 queries (Q image + candidate staging)
size_t ts_scan_lds_bytes(const TsLayout& L, int qh);
// fp32 storage, 32-query passes, 512 < d <= 768: the bf16x3 split scan (ts_scan_f32s.hip) instead of the exact-f32 MFMA
bool ts_use_f32_split(const TsLayout& L, int qh);
size_t ts_scan_f32s_lds_bytes(const TsLayout& L);
int ts_launch_scan_f32s(const TsLayout& L, int mode, const ScanParams& p, int num_cus, hipStream_t stream);
int ts_launch_qprep_f32s(const TsLayout& L, const void* q, int q_dtype, int nq, uint4* qimg, uint32_t* cand_cnt,
                         uint32_t* status, hipStream_t stream);

// rows [n, dim] (row-major, in_dtype) -> tiled storage at rows [row0, row0+n)
int ts_launch_relayout(const TsLayout& L, const void* rows, int in_dtype,
                       int64_t n, int64_t row0, uint4* tiled, bool normalize,
                       float* den_scratch, hipStream_t stream);
// tiled -> row-major float32
int ts_launch_reconstruct(const TsLayout& L, const uint4* tiled, int64_t row0,
                          int64_t n, float* out, hipStream_t stream);
// queries [nq, dim] (q_dtype) -> Q image in the storage dtype; also clears
// cand_cnt[64] and status[0] when they are non-null
int ts_launch_qprep(const TsLayout& L, const void* q, int q_dtype, int nq,
                    int qh, uint4* qimg, uint32_t* cand_cnt, uint32_t* status,
                    hipStream_t stream);

// ---------------------------------------------------------------- one-launch search (ts_fused.hip)
// query preparation + threshold estimation + fused scan/filter in one kernel; see ts_fused.hip
struct TsFusedArgs {
  const uint4* corpus;
  const void* queries;      // [nq, dim] rows of q_dtype (device)
  int q_dtype;
  int nq;
  int64_t nblk, ntotal;
  int scan_wgs, tau_wgs;    // workgroups streaming the corpus / estimating the thresholds
  int64_t n_sample;         // row blocks that feed the threshold sample: sample item s is block s*sample_stride
  int64_t sample_stride;    // sample item s is row block s*sample_stride
  uint32_t m;               // wanted rank among the sample's 16-row group maxima
  uint32_t expect;          // sample slots per query (two per sample block; a multiple of 4, <= TS_FUSED_MAX_KEYS)
  uint32_t gen;             // generation tag of the launch (non-zero, unique per workspace set)
  uint32_t arrive_goal;     // value of *arrive once every sample wave has reported
  uint32_t wait_iters;      // bound of every in-kernel spin
  uint32_t* skeys;          // ts_fused_keys_bytes(): [64][TS_FUSED_MAX_KEYS] sample keys, all-zero between launches
  uint32_t* arrive;
  unsigned long long* tau64;   // [64]
  uint32_t* cand_cnt;       // [64], zero at launch
  float* cand_score;
  int32_t* cand_id;
  uint32_t cand_cap;
};
#define TS_FUSED_MAX_KEYS 12288   // per query; the threshold role selects among two queries' keys in LDS (96 KiB)
size_t ts_fused_keys_bytes();
int ts_launch_fused(const TsLayout& L, int qh, const TsFusedArgs& a, hipStream_t stream);

// ---------------------------------------------------------------- select
enum { SEL_DENSE = 0, SEL_PAIRS32 = 1, SEL_MERGE64 = 2 };

#define TS_SEL_LDS_KEYS 16384   // 64-bit keys held in LDS by the select kernel
#define TS_STATUS_OVERFLOW 1u
#define TS_STATUS_SHORT 2u

struct SelParams {
  int mode;
  const float* scores;     // [nq][stride]
  const int32_t* ids32;    // PAIRS32: [nq][stride]
  const int64_t* ids64;    // MERGE64: [nq][stride] (entries with id<0 ignored)
  int64_t stride;          // elements between consecutive queries
  uint32_t seg_len;        // 0, or length of each concatenated list
  int64_t seg_stride;      // distance between consecutive lists of a query (scores)
  int64_t seg_stride_ids;  // same for the int64 ids (MERGE64; 0 = same as seg_stride)
  uint32_t n;              // entries per query (if n_per_q == null)
  const uint32_t* n_per_q; // optional device counts, clamped to n_cap
  uint32_t n_cap;
  int32_t id_base;         // DENSE: id = index + id_base
  int k;                   // entries to output per query
  uint32_t need;           // status SHORT if available < need (0 = no check)
  float* out_scores;       // [nq][out_stride]
  int64_t* out_ids64;      // final output (id + id_offset), or null
  int32_t* out_ids32;      // intermediate output (local ids), or null
  int64_t out_stride;
  int64_t id_offset;
  uint32_t* status;        // optional device status word (TS_STATUS_* bits are OR-ed in)
  uint32_t* host_report;   // optional, device view of pinned host memory:
                           // [q] = candidate count of query q, [64] |= status bits
  uint32_t* clear_counts;  // optional: n_per_q is given back as zeros (the one-launch search has no
                           // preparation kernel that would clear it)
};

int ts_launch_select(const SelParams& p, int nq, hipStream_t stream);

// tau[q] = (approximately) the m-th largest of sample[q][0..n) for q < nq,
// +FLT_MAX for nq <= q < 64
int ts_launch_tau(const float* sample, int64_t ld, uint32_t n, uint32_t m,
                  int nq, float* tau, hipStream_t stream);

// ---------------------------------------------------------------- maxsim
int ts_launch_maxsim(const void* q, int Lq, const void* docs,
                     const int32_t* doc_off, const int64_t* starts, const int32_t* lens,
                     int n_docs, int H, int dtype, int mode, float* out, hipStream_t stream);
// HBM-bound streaming form for f16/bf16 token matrices (ts_maxsim16.hip).  Returns
// TS_ERR_UNSUPPORTED without an error string for shapes it does not take.
int ts_launch_maxsim16(const void* q, int Lq, const void* docs, const int32_t* doc_off,
                       const int64_t* starts, const int32_t* lens, int n_docs, int H, int dtype,
                       int mode, float* out, int device, hipStream_t stream);
int ts_launch_maxsim16_batch(const void* q, const int32_t* q_off, int nq, const void* store,
                             const int64_t* starts, const int32_t* lens, const int32_t* cand_off,
                             int H, int dtype, int mode, float* out, int device, hipStream_t stream);
