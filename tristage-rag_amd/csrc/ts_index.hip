// libtristage.so — index handle, search orchestration and the C ABI
// (include/tristage.h).  The handle plays the role of the FAISS index object
// the reference keeps in Stage1Retriever.faiss_index
// (reference src/stage1_retriever.py:126, 256-283, 313, 380).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "ts_common.h"

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

void ts_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ts_last_error(void) { return g_err; }
extern "C" int ts_abi_version(void) { return TS_ABI_VERSION; }

namespace {

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

int ensure(DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes && b.p) return TS_OK;
  if (b.p) {
    TS_HIP(hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
  }
  // round up so slowly growing requests do not reallocate every time
  size_t want = (bytes + 0xFFFFF) & ~(size_t)0xFFFFF;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    b.p = nullptr;
    ts_set_error("hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
    return TS_ERR_OOM;
  }
  b.bytes = want;
  return TS_OK;
}

void release(DevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
}

}  // namespace
#define TS_NPHASE 8
#define TS_NSETS 4         // workspace sets = synchronous searches that may run at once on one handle
#define TS_ASYNC_SLOTS 1024   // report slots; at most a quarter of them (256 passes) may belong to unfinished asynchronous searches
#define TS_SLOT_WORDS 80   // 64 counts + status word, padded
namespace {
// tuning constants of the filter path (DESIGN.md "threshold sampling")
constexpr int64_t kMinFilterRows = 32768;   // below this the dense path is used
constexpr int kMaxFilterK = 2048;
constexpr uint32_t kCandCap = 16384;        // candidate slots per query
constexpr int64_t kMinSampleRows = 8192;
constexpr int64_t kSampleDiv = 128;         // sample >= N/128 rows
constexpr uint32_t kMinSampleRank = 24;
constexpr uint32_t kOversample = 4;         // expected candidates ~ 4k per query
constexpr int64_t kDenseChunkRows = 1 << 20;
constexpr int64_t kOneLaunchMaxRows = 4 << 20;   // default use of the one-launch search (search_pass_on)
constexpr int kHeadStartUs = 12;           // pipelined mode: delay of the select behind the next scan (search_pass)

}  // namespace

struct ts_index {
  int device = 0;
  int num_cus = 256;
  TsLayout L{};
  int64_t ntotal = 0;
  int64_t cap_blocks = 0;
  uint4* corpus = nullptr;
  int64_t id_offset = 0;
  int64_t info[4] = {0, 0, 0, 0};
  // Per-search workspace, double-buffered so that consecutive pipelined searches never
  // share a buffer that is still being read (see search_pass).
  struct WSet {
    DevBuf qimg, small, cand_score, cand_id, sample;
    DevBuf dense, list_score, list_id;   // the dense path's score chunk and per-chunk lists
    DevBuf hist, spill;                  // one-launch search (ts_fused.hip): threshold histogram, parked score tiles
    uint32_t gen = 0;                    // generation tag of the last one-launch search on this set
    uint32_t arrive_total = 0;           // running goal of the set's arrival hint counter
    bool hist_dirty = false;             // a launch may have left entries behind: cleared before the next one
    unsigned long long* tau64() { return (unsigned long long*)((char*)small.p + 1024); }
    uint32_t* arrive() { return (uint32_t*)((char*)small.p + 2048); }
    hipEvent_t ev_pro = nullptr, ev_scan = nullptr, ev_sel = nullptr, ev_in = nullptr;  // timing disabled
    bool used = false;  // ev_sel has been recorded at least once
    bool busy = false;  // held by a search that is being enqueued / a synchronous search in flight (h->mu)
    float* tau() { return (float*)small.p; }
    uint32_t* cand_cnt() { return (uint32_t*)small.p + 64; }
    uint32_t* status() { return (uint32_t*)small.p + 128; }
  };
  WSet ws[TS_NSETS];
  uint64_t set_next = 0;
  // Concurrency (include/tristage.h "threading"): searches from several host threads on ONE
  // handle are safe.  `mu` guards the set pool, the slot ring, the pending list, tickets and
  // `info`; a search owns its WSet from acquire_set() to release_set() — a synchronous one until
  // its result is verified (its exact fallback re-reads the set's query image), an asynchronous
  // one only while it is being enqueued (later users are ordered behind it by ev_sel).
  std::mutex mu;
  std::condition_variable cv;
  std::mutex host_mu;   // host-pointer searches share one staging area: serialised
  std::mutex prof_mu;   // per-phase timing uses one set of events: profiled searches are serialised
  bool slot_busy[TS_ASYNC_SLOTS] = {};
  hipStream_t s_pro = nullptr, s_scan = nullptr, s_sel = nullptr;  // TS_FLAG_PIPELINE only
  // staging of host-pointer calls (add / reconstruct need exclusive access anyway; searches: host_mu)
  DevBuf stage, den, qstage, out_s, out_i;
  uint32_t* host_status = nullptr;  // pinned + mapped: written by the select kernel
  uint32_t* host_status_dev = nullptr;  // device view of host_status
  // asynchronous searches (TS_FLAG_ASYNC): each pass reports into its own slot of
  // the mapped host ring; ts_index_finish() syncs once and inspects them all
  struct Pending { int64_t ticket; int slot; int nq; uint32_t S; uint32_t m; hipEvent_t e0, e1; int set; };
  Pending pending[TS_ASYNC_SLOTS];
  int npending = 0;
  uint64_t slot_next = 0;
  int64_t next_ticket = 0;
  hipEvent_t async_ev[2 * TS_ASYNC_SLOTS] = {};
  // optional per-phase timing with HIP events on the caller's stream
  // `profiling` / `prof_every` change only under exclusive access (ts_index_set_profiling); a profiled search holds
  // prof_mu for its whole duration, which is what makes the ONE set of events `ev` safe; everything that varies
  // per search lives in the search's own ProfCtx, and the accumulators are updated under `mu`.
  bool profiling = false;
  int prof_every = 1;        // time every prof_every-th search (timing events cost ~5 us each in-stream)
  std::atomic<uint64_t> prof_seq{0};
  hipEvent_t ev[TS_NPHASE + 1] = {};
  double phase_ms[TS_NPHASE] = {};
  int64_t phase_cnt[TS_NPHASE] = {};
};

// Per-search profiling state (round 2 kept these three in the handle, written by every search: a data race
// between concurrent callers of one handle even with profiling off).
struct ProfCtx {
  bool now = false;         // this search is being timed (implies h->profiling, i.e. prof_mu is held)
  bool scan_only = false;   // asynchronous searches: only the scan+filter interval
  int nev = 0;
  int ev_phase[TS_NPHASE + 1] = {};
};

// profiling: mark(h, pc, phase, s) records an event; the time between two marks is
// charged to the phase of the FIRST mark.  Collected after the search's sync.
static void prof_mark(ts_index* h, ProfCtx& pc, int phase, hipStream_t s) {
  if (!pc.now || pc.nev > TS_NPHASE) return;
  if (pc.scan_only && phase != 3 && phase != 4) return;
  if (hipEventRecord(h->ev[pc.nev], s) != hipSuccess) return;
  pc.ev_phase[pc.nev] = phase;
  ++pc.nev;
}
static void prof_collect(ts_index* h, ProfCtx& pc) {
  if (!pc.now) return;
  std::lock_guard<std::mutex> lk(h->mu);
  for (int i = 0; i + 1 < pc.nev; ++i) {
    float ms = 0.f;
    const int ph = pc.ev_phase[i];
    if (ph >= 0 && ph < TS_NPHASE && hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]) == hipSuccess) {
      h->phase_ms[ph] += ms;
      h->phase_cnt[ph] += 1;
    }
  }
  pc.nev = 0;
}

static int grow_corpus(ts_index* h, int64_t need_blocks, bool exact, hipStream_t s) {
  if (need_blocks <= h->cap_blocks) return TS_OK;
  int64_t new_cap = need_blocks;
  if (!exact && h->cap_blocks > 0) new_cap = std::max(need_blocks, h->cap_blocks + h->cap_blocks / 2);
  const size_t bb = ts_block_bytes(h->L);
  void* np = nullptr;
  hipError_t e = hipMalloc(&np, (size_t)new_cap * bb);
  if (e != hipSuccess) {
    ts_set_error("cannot allocate %zu bytes for the corpus: %s", (size_t)new_cap * bb,
                 hipGetErrorString(e));
    return TS_ERR_OOM;
  }
  const int64_t used_blocks = (h->ntotal + TS_ROWS_PER_BLOCK - 1) / TS_ROWS_PER_BLOCK;
  if (used_blocks > 0)
    TS_HIP(hipMemcpyAsync(np, h->corpus, (size_t)used_blocks * bb, hipMemcpyDeviceToDevice, s));
  TS_HIP(hipMemsetAsync((char*)np + (size_t)used_blocks * bb, 0,
                        (size_t)(new_cap - used_blocks) * bb, s));
  TS_HIP(hipStreamSynchronize(s));
  if (h->corpus) TS_HIP(hipFree(h->corpus));
  h->corpus = (uint4*)np;
  h->cap_blocks = new_cap;
  return TS_OK;
}

extern "C" int ts_index_create(int32_t dim, int32_t storage_dtype, int32_t metric,
                               int32_t device, ts_index** out) {
  if (!out) { ts_set_error("out is null"); return TS_ERR_INVALID; }
  *out = nullptr;
  if (dim <= 0 || dim > 65536) { ts_set_error("bad dim %d", dim); return TS_ERR_INVALID; }
  if (storage_dtype != TS_F32 && storage_dtype != TS_F16 && storage_dtype != TS_BF16) {
    ts_set_error("bad storage dtype %d", storage_dtype);
    return TS_ERR_INVALID;
  }
  if (metric != TS_METRIC_INNER_PRODUCT) {
    ts_set_error("only the inner-product metric is supported (reference uses IndexFlatIP)");
    return TS_ERR_UNSUPPORTED;
  }
  int ndev = 0;
  TS_HIP(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) {
    ts_set_error("device %d not present (%d HIP devices)", device, ndev);
    return TS_ERR_INVALID;
  }
  TsLayout L = ts_make_layout(dim, storage_dtype);
  if (ts_scan_lds_bytes(L, 1) > 160 * 1024) {
    ts_set_error("dim %d with dtype %d needs %zu bytes of LDS for 32 queries (max 163840)", dim,
                 storage_dtype, ts_scan_lds_bytes(L, 1));
    return TS_ERR_UNSUPPORTED;
  }
  DeviceGuard g(device);
  if (!g.ok) { ts_set_error("hipSetDevice(%d) failed", device); return TS_ERR_HIP; }
  ts_index* h = new (std::nothrow) ts_index();
  if (!h) { ts_set_error("out of host memory"); return TS_ERR_OOM; }
  h->device = device;
  h->L = L;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
    h->num_cus = prop.multiProcessorCount;
  int st = TS_OK;
  for (ts_index::WSet& w : h->ws) {
    if (st == TS_OK) st = ensure(w.small, 4096);
    if (st == TS_OK && hipMemset(w.small.p, 0, 4096) != hipSuccess) { ts_set_error("hipMemset failed"); st = TS_ERR_HIP; }
    if (st == TS_OK) st = ensure(w.qimg, (size_t)L.kg * 2 * 1024);
    if (st == TS_OK && (hipEventCreateWithFlags(&w.ev_pro, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&w.ev_scan, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&w.ev_sel, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&w.ev_in, hipEventDisableTiming) != hipSuccess)) {
      ts_set_error("hipEventCreate failed");
      st = TS_ERR_HIP;
    }
  }
  if (st == TS_OK &&
      (hipHostMalloc((void**)&h->host_status, (size_t)(TS_ASYNC_SLOTS + 1) * TS_SLOT_WORDS * 4, hipHostMallocMapped) != hipSuccess ||
       hipHostGetDevicePointer((void**)&h->host_status_dev, h->host_status, 0) != hipSuccess)) {
    ts_set_error("hipHostMalloc(mapped) failed");
    st = TS_ERR_HIP;
  }
  if (st != TS_OK) {
    ts_index_destroy(h);
    return st;
  }
  *out = h;
  return TS_OK;
}

extern "C" int ts_index_destroy(ts_index* h) {
  if (!h) return TS_OK;
  DeviceGuard g(h->device);
  (void)hipDeviceSynchronize();
  if (h->corpus) (void)hipFree(h->corpus);
  DevBuf* bufs[] = {&h->stage, &h->den, &h->qstage, &h->out_s, &h->out_i};
  for (DevBuf* b : bufs) release(*b);
  for (ts_index::WSet& w : h->ws) {
    DevBuf* wb[] = {&w.qimg, &w.small, &w.cand_score, &w.cand_id, &w.sample, &w.dense, &w.list_score, &w.list_id,
                    &w.hist, &w.spill};
    for (DevBuf* b : wb) release(*b);
    hipEvent_t evs[] = {w.ev_pro, w.ev_scan, w.ev_sel, w.ev_in};
    for (hipEvent_t e : evs)
      if (e) (void)hipEventDestroy(e);
  }
  hipStream_t strs[] = {h->s_pro, h->s_scan, h->s_sel};
  for (hipStream_t st_ : strs)
    if (st_) (void)hipStreamDestroy(st_);
  if (h->host_status) (void)hipHostFree(h->host_status);
  for (hipEvent_t e : h->ev)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->async_ev)
    if (e) (void)hipEventDestroy(e);
  delete h;
  return TS_OK;
}

extern "C" int ts_index_reset(ts_index* h) {
  if (!h) { ts_set_error("null handle"); return TS_ERR_INVALID; }
  h->ntotal = 0;
  return TS_OK;
}

extern "C" int ts_index_reserve(ts_index* h, int64_t nrows) {
  if (!h || nrows < 0) { ts_set_error("bad arguments"); return TS_ERR_INVALID; }
  if (nrows >= (1LL << 31)) { ts_set_error("at most 2^31-1 rows per index"); return TS_ERR_UNSUPPORTED; }
  DeviceGuard g(h->device);
  return grow_corpus(h, (nrows + TS_ROWS_PER_BLOCK - 1) / TS_ROWS_PER_BLOCK, true, nullptr);
}

extern "C" int64_t ts_index_ntotal(const ts_index* h) { return h ? h->ntotal : -1; }
extern "C" int32_t ts_index_dim(const ts_index* h) { return h ? h->L.dim : -1; }
extern "C" int32_t ts_index_dtype(const ts_index* h) { return h ? h->L.dtype : -1; }

extern "C" int ts_index_set_id_offset(ts_index* h, int64_t offset) {
  if (!h) { ts_set_error("null handle"); return TS_ERR_INVALID; }
  h->id_offset = offset;
  return TS_OK;
}

extern "C" int ts_index_last_search_info(const ts_index* h, int64_t info[4]) {
  if (!h || !info) { ts_set_error("bad arguments"); return TS_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(const_cast<ts_index*>(h)->mu);
  for (int i = 0; i < 4; ++i) info[i] = h->info[i];
  return TS_OK;
}

static size_t dtype_size(int dt) { return dt == TS_F32 ? 4 : 2; }
static bool dtype_ok(int dt) { return dt == TS_F32 || dt == TS_F16 || dt == TS_BF16; }

extern "C" int ts_index_add(ts_index* h, const void* rows, int64_t n, int32_t rows_dtype,
                            uint32_t flags, void* stream) {
  if (!h) { ts_set_error("null handle"); return TS_ERR_INVALID; }
  if (n == 0) return TS_OK;
  if (!rows || n < 0 || !dtype_ok(rows_dtype)) { ts_set_error("bad arguments to add"); return TS_ERR_INVALID; }
  if (h->ntotal + n >= (1LL << 31)) { ts_set_error("at most 2^31-1 rows per index"); return TS_ERR_UNSUPPORTED; }
  DeviceGuard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  const int64_t need_blocks = (h->ntotal + n + TS_ROWS_PER_BLOCK - 1) / TS_ROWS_PER_BLOCK;
  TS_CHECK(grow_corpus(h, need_blocks, false, s));
  const bool norm = (flags & TS_FLAG_NORMALIZE) != 0;
  const size_t row_bytes = (size_t)h->L.dim * dtype_size(rows_dtype);
  if (flags & TS_FLAG_HOST_PTR) {
    // chunked upload through a device staging buffer
    int64_t chunk = std::max<int64_t>(1, (int64_t)((64u << 20) / row_bytes));
    chunk = std::min(chunk, n);
    TS_CHECK(ensure(h->stage, (size_t)chunk * row_bytes));
    if (norm) TS_CHECK(ensure(h->den, (size_t)chunk * 4));
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
      const int64_t c = std::min(chunk, n - r0);
      TS_HIP(hipMemcpyAsync(h->stage.p, (const char*)rows + (size_t)r0 * row_bytes,
                            (size_t)c * row_bytes, hipMemcpyHostToDevice, s));
      TS_CHECK(ts_launch_relayout(h->L, h->stage.p, rows_dtype, c, h->ntotal + r0, h->corpus,
                                  norm, (float*)h->den.p, s));
      // the staging buffer is reused by the next chunk
      TS_HIP(hipStreamSynchronize(s));
    }
  } else {
    if (norm) TS_CHECK(ensure(h->den, (size_t)n * 4));
    TS_CHECK(ts_launch_relayout(h->L, rows, rows_dtype, n, h->ntotal, h->corpus, norm,
                                (float*)h->den.p, s));
    TS_HIP(hipStreamSynchronize(s));
  }
  h->ntotal += n;
  return TS_OK;
}

extern "C" int ts_index_reconstruct(ts_index* h, int64_t row0, int64_t n, float* out,
                                    uint32_t flags, void* stream) {
  if (!h || !out || row0 < 0 || n < 0 || row0 + n > h->ntotal) {
    ts_set_error("bad arguments to reconstruct");
    return TS_ERR_INVALID;
  }
  if (n == 0) return TS_OK;
  DeviceGuard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  if (flags & TS_FLAG_HOST_PTR) {
    const size_t row_bytes = (size_t)h->L.dim * 4;
    int64_t chunk = std::max<int64_t>(1, (int64_t)((64u << 20) / row_bytes));
    chunk = std::min(chunk, n);
    TS_CHECK(ensure(h->stage, (size_t)chunk * row_bytes));
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
      const int64_t c = std::min(chunk, n - r0);
      TS_CHECK(ts_launch_reconstruct(h->L, h->corpus, row0 + r0, c, (float*)h->stage.p, s));
      TS_HIP(hipMemcpyAsync((char*)out + (size_t)r0 * row_bytes, h->stage.p,
                            (size_t)c * row_bytes, hipMemcpyDeviceToHost, s));
      TS_HIP(hipStreamSynchronize(s));
    }
  } else {
    TS_CHECK(ts_launch_reconstruct(h->L, h->corpus, row0, n, out, s));
    TS_HIP(hipStreamSynchronize(s));
  }
  return TS_OK;
}

// ------------------------------------------------------------------ search
static int dense_path(ts_index* h, ts_index::WSet& W, int nq, int qh, int k, float* out_s,
                      int64_t* out_i, hipStream_t s) {
  const int64_t N = h->ntotal;
  const int64_t nblk = (N + TS_ROWS_PER_BLOCK - 1) / TS_ROWS_PER_BLOCK;
  const int64_t chunk_rows = std::min<int64_t>(kDenseChunkRows, nblk * TS_ROWS_PER_BLOCK);
  const int64_t nch = (nblk * TS_ROWS_PER_BLOCK + chunk_rows - 1) / chunk_rows;
  TS_CHECK(ensure(W.dense, (size_t)nq * chunk_rows * 4));
  if (nch > 1) {
    TS_CHECK(ensure(W.list_score, (size_t)nq * nch * k * 4));
    TS_CHECK(ensure(W.list_id, (size_t)nq * nch * k * 4));
  }
  for (int64_t c = 0; c < nch; ++c) {
    const int64_t row0 = c * chunk_rows;
    const int64_t rows = std::min(chunk_rows, N - row0);
    ScanParams sp{};
    sp.corpus = h->corpus;
    sp.qimg = (const uint4*)W.qimg.p;
    sp.kg = h->L.kg;
    sp.nq = nq;
    sp.nwork = (rows + TS_ROWS_PER_BLOCK - 1) / TS_ROWS_PER_BLOCK;
    sp.blk0 = row0 / TS_ROWS_PER_BLOCK;
    sp.blk_stride = 1;
    sp.ntotal = N;
    sp.dense = (float*)W.dense.p;
    sp.dense_ld = chunk_rows;
    TS_CHECK(ts_launch_scan(h->L, SCAN_DENSE, qh, sp, h->num_cus, s));
    SelParams p{};
    p.mode = SEL_DENSE;
    p.scores = (const float*)W.dense.p;
    p.stride = chunk_rows;
    p.n = (uint32_t)rows;
    p.id_base = (int32_t)row0;
    p.k = k;
    if (nch == 1) {
      p.out_scores = out_s;
      p.out_ids64 = out_i;
      p.out_stride = k;
      p.id_offset = h->id_offset;
    } else {
      p.out_scores = (float*)W.list_score.p + c * k;
      p.out_ids32 = (int32_t*)W.list_id.p + c * k;
      p.out_stride = nch * k;
    }
    TS_CHECK(ts_launch_select(p, nq, s));
  }
  if (nch > 1) {
    SelParams p{};
    p.mode = SEL_PAIRS32;
    p.scores = (const float*)W.list_score.p;
    p.ids32 = (const int32_t*)W.list_id.p;
    p.stride = nch * k;
    p.n = (uint32_t)(nch * k);
    p.k = k;
    p.out_scores = out_s;
    p.out_ids64 = out_i;
    p.out_stride = k;
    p.id_offset = h->id_offset;
    TS_CHECK(ts_launch_select(p, nq, s));
  }
  return TS_OK;
}

// One wave that does nothing for ~`us` microseconds (bounded: at most `max_iter` short sleeps).
// Put in front of a helper kernel that becomes runnable at the same instant as the next scan, it
// lets the scan's workgroups be placed first (see search_pass).
__global__ void head_start_kernel(int us, int max_iter) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
  for (int i = 0; i < max_iter; ++i) {
    if (__builtin_amdgcn_s_memrealtime() - t0 >= (unsigned long long)us * 100ull) break;
    __builtin_amdgcn_s_sleep(32);
  }
}

static int ensure_streams(ts_index* h) {
  std::lock_guard<std::mutex> lk(h->mu);
  if (h->s_pro) return TS_OK;
  // The scan stream outranks the two helper streams: when a scan and the previous
  // search's select become runnable at the same instant (both wait for the same scan
  // to end), the scan's workgroups must be placed first — a scan workgroup that starts
  // late finishes late, because every workgroup owns a fixed share of the rows.
  int lo = 0, hi = 0;  // numerically lower = higher priority
  TS_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  TS_HIP(hipStreamCreateWithPriority(&h->s_scan, hipStreamNonBlocking, hi));
  TS_HIP(hipStreamCreateWithPriority(&h->s_sel, hipStreamNonBlocking, lo));
  TS_HIP(hipStreamCreateWithPriority(&h->s_pro, hipStreamNonBlocking, lo));   // (last: it is the "created" flag)
  return TS_OK;
}

// One pass of <= 32*qh queries (device pointers).
//
// A pass has three phases, each touching one workspace set W:
//   P   query prep, sample scan, thresholds      writes W.qimg, W.sample, W.tau; clears W.cand_cnt
//   S   fused scan+filter over the whole shard   reads W.qimg, W.tau; writes W.cand_*
//   Sel exact top-k of the candidates            reads W.cand_*; writes the caller's outputs + host slot
// Normally all three are enqueued on the caller's stream.  With TS_FLAG_PIPELINE they go to
// three internal streams chained by events (P -> S -> Sel), and the caller's stream only waits
// for Sel: P of the NEXT search and Sel of the PREVIOUS one then run beside the current S, which
// leaves an eighth of the CUs free and is HBM-bound anyway.  Passes alternate between two
// workspace sets; a set is reused only after the Sel that last read it (its ev_sel).
// ---- geometry of the one-launch search
struct FusedPlan {
  int scan_wgs, tau_wgs;
  int64_t n_sample, sample_stride, sample_rows;
  uint32_t m, expect, sample_waves;
};
static bool plan_fused(const ts_index* h, int64_t N, int64_t nblk, int k, bool pipe, FusedPlan* fp) {
  // the scan takes 7/8 of the CUs (the kernel is HBM-bound: 224 CUs stream as fast as 256, tools/cus.sh); the
  // threshold workgroups, and in pipelined mode the neighbours' select and the exchange, use the rest.  The next
  // search's kernel follows on the SAME stream, so its workgroups are dispatched the moment this one ends, ahead
  // of the select that waits for this kernel through an event on another stream: no placement race.
  (void)pipe;
  int scan_wgs = h->num_cus - h->num_cus / 8;
#ifdef TS_TUNING
  static const int dbg_cus = getenv("TS_SCAN_CUS") ? atoi(getenv("TS_SCAN_CUS")) : 0;
  if (dbg_cus > 0) scan_wgs = dbg_cus;
#endif
  if (scan_wgs < 1) return false;
  const int tau_wgs = std::max(1, std::min(64, h->num_cus - scan_wgs));
  const int64_t nwaves = (int64_t)scan_wgs * 8;
  if (nblk < nwaves) return false;                         // fewer row blocks than waves: a latency-bound corpus
  const int64_t want_rows = std::max(kMinSampleRows, N / kSampleDiv);
  int64_t R = (want_rows + nwaves * TS_ROWS_PER_BLOCK - 1) / (nwaves * TS_ROWS_PER_BLOCK);
  R = std::max<int64_t>(1, std::min<int64_t>(R, 16));
  int64_t n_sample = std::min(nblk, R * nwaves);
  // A scan wave can deliver at most THREE sample rounds before it needs the thresholds: it parks two score tiles
  // and blocks in tau_wait() behind the third block's key store (ts_fused.hip).  A fourth round's keys would never
  // be written, the threshold waves and the scan waves would wait for each other until both time out, and the
  // batch would fall back to the dense path — correct but 40-60 ms (round 2: any corpus above ~22 M rows on 256 CUs).
  n_sample = std::min<int64_t>(n_sample, 3 * nwaves);
  n_sample = std::min<int64_t>(n_sample, TS_FUSED_MAX_KEYS / 2) & ~(int64_t)255;   // two slots per block; groups of 8 blocks;
  if (n_sample < 256) return false;                                                 // slots a multiple of 512
  int64_t stride = nblk / (n_sample / 8);                   // between the first blocks of consecutive 8-block groups (>= 8)
#ifdef TS_TUNING   // experiment: the sample is the first n_sample blocks of the corpus (biased for grouped corpora)
  static const bool dbg_prefix = getenv("TS_FUSED_PREFIX_SAMPLE") != nullptr;
  if (dbg_prefix) stride = 8;
#endif
  const int64_t rows = n_sample * TS_ROWS_PER_BLOCK;
  const int64_t oversample = k > 1024 ? 3 : kOversample;
  uint32_t m = (uint32_t)((oversample * (int64_t)k * rows + N - 1) / N);
  m = std::max(m, kMinSampleRank);
  const uint32_t expect = (uint32_t)(2 * n_sample);        // one group maximum per (wave, round, lane half)
  if ((uint64_t)m * 4 > expect) return false;              // group maxima would no longer stand in for scores
  fp->scan_wgs = scan_wgs;
  fp->tau_wgs = tau_wgs;
  fp->n_sample = n_sample;
  fp->sample_stride = stride;
  fp->sample_rows = rows;
  fp->m = m;
  fp->expect = expect;
  fp->sample_waves = (uint32_t)std::min<int64_t>(scan_wgs, n_sample / 8);   // WORKGROUPS that report: one arrival each
  return true;
}

// ---- the set pool and the report-slot ring (both under h->mu)
static ts_index::WSet* acquire_set(ts_index* h) {
  std::unique_lock<std::mutex> lk(h->mu);
  for (;;) {
    for (int t = 0; t < TS_NSETS; ++t) {
      ts_index::WSet& w = h->ws[(h->set_next + t) % TS_NSETS];
      if (!w.busy) {
        h->set_next = (h->set_next + t + 1) % TS_NSETS;   // round robin: pipelined searches alternate sets
        w.busy = true;
        return &w;
      }
    }
    h->cv.wait(lk);   // TS_NSETS synchronous searches in flight: wait for one to return
  }
}
static void release_set(ts_index* h, ts_index::WSet* w) {
  {
    std::lock_guard<std::mutex> lk(h->mu);
    w->busy = false;
  }
  h->cv.notify_one();
}
// a free slot of the mapped host ring (at most 64 asynchronous passes + TS_NSETS synchronous ones hold one)
static int alloc_slot(ts_index* h) {
  std::lock_guard<std::mutex> lk(h->mu);
  for (int t = 0; t < TS_ASYNC_SLOTS; ++t) {
    const int sl = (int)((h->slot_next + t) % TS_ASYNC_SLOTS);
    if (!h->slot_busy[sl]) {
      h->slot_busy[sl] = true;
      h->slot_next = (uint64_t)sl + 1;
      return sl;
    }
  }
  return -1;
}
static void free_slot(ts_index* h, int sl) {
  std::lock_guard<std::mutex> lk(h->mu);
  if (sl >= 0 && sl < TS_ASYNC_SLOTS) h->slot_busy[sl] = false;
}
// a report slot is given back on every exit of a search unless its ownership has moved to pending[]
struct SlotGuard {
  ts_index* h;
  int slot;
  SlotGuard(ts_index* h_, int slot_) : h(h_), slot(slot_) {}
  ~SlotGuard() { if (slot >= 0) free_slot(h, slot); }
  void handed_over() { slot = -1; }
  SlotGuard(const SlotGuard&) = delete;
  SlotGuard& operator=(const SlotGuard&) = delete;
};
static void set_info(ts_index* h, int64_t a, int64_t b, int64_t c, int64_t d) {
  std::lock_guard<std::mutex> lk(h->mu);
  h->info[0] = a; h->info[1] = b; h->info[2] = c; h->info[3] = d;
}

static int search_pass_on(ts_index* h, ts_index::WSet& W, const void* dq, int nq, int q_dtype, int k,
                          float* out_s, int64_t* out_i, uint32_t flags, hipStream_t s);

static int search_pass(ts_index* h, const void* dq, int nq, int q_dtype, int k, float* out_s,
                       int64_t* out_i, uint32_t flags, hipStream_t s) {
  // per-phase timing shares one set of events per handle: profiled searches run one at a time
  std::unique_lock<std::mutex> plk(h->prof_mu, std::defer_lock);
  if (h->profiling) plk.lock();
  ts_index::WSet* W = acquire_set(h);
  const int st = search_pass_on(h, *W, dq, nq, q_dtype, k, out_s, out_i, flags, s);
  release_set(h, W);
  return st;
}

static int search_pass_on(ts_index* h, ts_index::WSet& W, const void* dq, int nq, int q_dtype, int k,
                          float* out_s, int64_t* out_i, uint32_t flags, hipStream_t s) {
  const int64_t N = h->ntotal;
  const int64_t nblk = (N + TS_ROWS_PER_BLOCK - 1) / TS_ROWS_PER_BLOCK;
  const int qh = nq > 32 ? 2 : 1;
  const bool filter = !(flags & TS_FLAG_NO_FILTER) && k <= kMaxFilterK && N >= kMinFilterRows &&
                      N >= 32 * (int64_t)k;
  const bool async = (flags & TS_FLAG_ASYNC) != 0;
  const bool pipe = filter && async && (flags & TS_FLAG_PIPELINE) != 0;
  if (pipe) TS_CHECK(ensure_streams(h));
  // ---- one-launch search (ts_fused.hip) whenever its threshold estimate is valid: the sample's 16-row
  // group maxima stand in for the scores, which holds while the wanted rank is far below the group count
  FusedPlan fp{};
  // Where it is taken by default: unpipelined searches of up to kOneLaunchMaxRows rows, the regime in which the three
  // launches it removes are a visible share of the batch (1.25 M x 768: 0.343 vs 0.359 ms per batch back to back,
  // 0.376 vs 0.391 synchronous).  At 10 M rows both paths take 2.27 ms and the plain scan kernel is the cleaner
  // roofline object; pipelined (TS_FLAG_PIPELINE) the five-launch path already hides its preparation under the
  // previous scan and is 3 % ahead (0.335 vs 0.345 ms with the exchange).  TS_FLAG_ONE_LAUNCH forces it anywhere.
  const bool want_fused = (flags & TS_FLAG_ONE_LAUNCH) || (!pipe && N <= kOneLaunchMaxRows);
  const bool fused = filter && !(flags & TS_FLAG_CLASSIC) && want_fused && !ts_use_f32_split(h->L, qh) &&
                     plan_fused(h, N, nblk, k, pipe, &fp);   // (fp32 storage at 32 queries per pass: the split scan, five launches)
  // (the one-launch search has no preparation phase: what remains of "P" rides on the scan stream)
  hipStream_t sP = pipe ? (fused ? h->s_scan : h->s_pro) : s, sS = pipe ? h->s_scan : s, sL = pipe ? h->s_sel : s;
  // the set may still be in use on the GPU by the (asynchronous) search that had it last, possibly
  // on another stream
  // (asked first whether that search is already over — with four sets in rotation it normally is: a wait on
  // another stream's event in front of every scan, even a satisfied one, costs the scan stream 10-20 us)
  if (W.used && hipEventQuery(W.ev_sel) != hipSuccess) TS_HIP(hipStreamWaitEvent(sP, W.ev_sel, 0));
  // the output buffers may be memory the caller's stream is still reading (a recycled
  // allocation): the final phase must not start before the stream's work issued so far
  if (pipe) TS_HIP(hipEventRecord(W.ev_in, s));

  ProfCtx pc;
  pc.now = h->profiling && (h->prof_seq.fetch_add(1, std::memory_order_relaxed) % (uint64_t)h->prof_every == 0);
  pc.scan_only = async;
  prof_mark(h, pc, 0, sP);
  if (!fused) TS_CHECK(ts_launch_qprep(h->L, dq, q_dtype, nq, qh, (uint4*)W.qimg.p, W.cand_cnt(), W.status(), sP));
  if (!filter) {
    set_info(h, 0, 0, 0, 0);
    prof_mark(h, pc, 5, s);
    TS_CHECK(dense_path(h, W, nq, qh, k, out_s, out_i, s));
    prof_mark(h, pc, -1, s);
    TS_HIP(hipEventRecord(W.ev_sel, s));
    W.used = true;
    if (async) return TS_OK;  // exact by construction: nothing to verify
    TS_HIP(hipStreamSynchronize(s));
    prof_collect(h, pc);
    return TS_OK;
  }
  int64_t S = 0;
  uint32_t m = 0;
  TS_CHECK(ensure(W.cand_score, (size_t)TS_MAX_Q * kCandCap * 4));
  TS_CHECK(ensure(W.cand_id, (size_t)TS_MAX_Q * kCandCap * 4));
  if (fused) {
    // ---- ONE launch: query image, thresholds and scan+filter (see ts_fused.hip)
    if (!W.hist.p) {
      TS_CHECK(ensure(W.hist, ts_fused_keys_bytes()));
      W.hist_dirty = true;
    }
    if (W.hist_dirty) {
      // a launch on this set gave up somewhere: slots may hold stale keys, and sample workgroups that left before
      // reporting have put the arrival counter behind its running goal for good — start all three from zero
      TS_HIP(hipMemsetAsync(W.hist.p, 0, ts_fused_keys_bytes(), sS));
      TS_HIP(hipMemsetAsync(W.cand_cnt(), 0, 256, sS));
      TS_HIP(hipMemsetAsync(W.arrive(), 0, 4, sS));
      W.arrive_total = 0;
      W.hist_dirty = false;
    }
    if (++W.gen == 0) W.gen = 1;
    W.arrive_total += fp.sample_waves;
    TsFusedArgs a{};
    a.corpus = h->corpus;
    a.queries = dq;
    a.q_dtype = q_dtype;
    a.nq = nq;
    a.nblk = nblk;
    a.ntotal = N;
    a.scan_wgs = fp.scan_wgs;
    a.tau_wgs = fp.tau_wgs;
    a.n_sample = fp.n_sample;
    a.sample_stride = fp.sample_stride;
    a.m = fp.m;
    a.expect = fp.expect;
    a.gen = W.gen;
    a.arrive_goal = W.arrive_total;
    a.wait_iters = 40000;
    a.skeys = (uint32_t*)W.hist.p;
    a.arrive = W.arrive();
    a.tau64 = W.tau64();
    a.cand_cnt = W.cand_cnt();
    a.cand_score = (float*)W.cand_score.p;
    a.cand_id = (int32_t*)W.cand_id.p;
    a.cand_cap = kCandCap;
    prof_mark(h, pc, 3, sS);
    TS_CHECK(ts_launch_fused(h->L, qh, a, sS));
    prof_mark(h, pc, 4, sS);
    S = fp.sample_rows;
    m = fp.m;
  } else {
  // ---- five launches: sample -> thresholds -> scan+filter (-> select below)
  int64_t nsb = std::max(kMinSampleRows, N / kSampleDiv) / TS_ROWS_PER_BLOCK;
  // one sample block per scan wave (8 waves per CU): a ragged last round would
  // make the short sample scan ~1.7x longer than it has to be
  const int64_t round = (int64_t)h->num_cus * 8;
  if (nsb > round) nsb -= nsb % round;
  nsb = std::min(nsb, nblk);
  const int64_t sstride = nblk / nsb;
  S = nsb * TS_ROWS_PER_BLOCK;
  // expected survivors per query ~ oversample * k; 3x for large k keeps the worst query of a
  // batch (about 1.5x the mean) well inside the 16384 candidate slots
  const int64_t oversample = k > 1024 ? 3 : kOversample;
  m = (uint32_t)((oversample * (int64_t)k * S + N - 1) / N);
  m = std::max(m, kMinSampleRank);
  TS_CHECK(ensure(W.sample, (size_t)nq * S * 4));

  ScanParams sp{};
  sp.corpus = h->corpus;
  sp.qimg = (const uint4*)W.qimg.p;
  sp.kg = h->L.kg;
  sp.nq = nq;
  sp.ntotal = N;
  // (1) dense scores of a strided sample of row blocks
  sp.nwork = nsb;
  sp.blk0 = 0;
  sp.blk_stride = sstride;
  sp.dense = (float*)W.sample.p;
  sp.dense_ld = S;
  prof_mark(h, pc, 1, sP);
  TS_CHECK(ts_launch_scan(h->L, SCAN_DENSE, qh, sp, h->num_cus, sP));
  prof_mark(h, pc, 2, sP);
  // (2) per-query threshold = ~m-th best sample score
#ifdef TS_TUNING  // ablation builds only (tools/variants.sh): thresholds = +inf, nothing survives
  static const bool dbg_tau_inf = getenv("TS_DEBUG_TAU_INF") != nullptr;
#else
  constexpr bool dbg_tau_inf = false;
#endif
  TS_CHECK(ts_launch_tau((const float*)W.sample.p, S, dbg_tau_inf ? 0xFFFFFFFFu : (uint32_t)S, m, nq,
                         W.tau(), sP));
  if (pipe) {
    TS_HIP(hipEventRecord(W.ev_pro, sP));
    TS_HIP(hipStreamWaitEvent(sS, W.ev_pro, 0));
  }
  // (3) the full scan; only scores >= tau leave the registers.  It occupies 7/8 of the CUs:
  // the kernel is HBM-bound (measured: 224 CUs stream 1.5 % FASTER than 256, tools/cus.sh)
  // and the free CUs are where the neighbouring searches' small kernels run.
  sp.nwork = nblk;
  sp.blk_stride = 1;
  sp.dense = nullptr;
  sp.tau = W.tau();
  sp.cand_cnt = W.cand_cnt();
  sp.cand_score = (float*)W.cand_score.p;
  sp.cand_id = (int32_t*)W.cand_id.p;
  sp.cand_cap = kCandCap;
  int scan_cus = h->num_cus - h->num_cus / 8;
#ifdef TS_TUNING  // ablation builds only: how many CUs the fused scan may occupy
  static const int dbg_cus = getenv("TS_SCAN_CUS") ? atoi(getenv("TS_SCAN_CUS")) : 0;
  if (dbg_cus > 0) scan_cus = dbg_cus;
#endif
  prof_mark(h, pc, 3, sS);
  TS_CHECK(ts_launch_scan(h->L, SCAN_FILTER, qh, sp, scan_cus, sS));
  prof_mark(h, pc, 4, sS);
  }  // five launches
  if (pipe) {
    TS_HIP(hipEventRecord(W.ev_scan, sS));
    TS_HIP(hipStreamWaitEvent(sL, W.ev_scan, 0));
    TS_HIP(hipStreamWaitEvent(sL, W.ev_in, 0));
    // This select and the NEXT search's scan wait for the same event.  If the select's 64
    // workgroups (128 KiB of LDS each: they cannot share a CU with a scan workgroup) are placed
    // first they take 64 CUs, 32 of the scan's 224 persistent workgroups have to wait for them,
    // and the whole scan ends that much later — stream priority does not prevent it (measured:
    // overlapped scans of 0.305-0.326 ms instead of 0.289 at 1.25 M rows, 2.263 instead of 2.207
    // at 10 M).  A head start of a few microseconds for the scan does: the select then finds the
    // 32 CUs the scan grid leaves free.
#ifdef TS_TUNING
    static const int head_us = getenv("TS_HEAD_START_US") ? atoi(getenv("TS_HEAD_START_US")) : kHeadStartUs;
#else
    constexpr int head_us = kHeadStartUs;
#endif
    // (the one-launch scan occupies 3/4 of the CUs: the select's 64 workgroups always fit beside it)
    if (head_us > 0 && !fused) hipLaunchKernelGGL(head_start_kernel, dim3(1), dim3(64), 0, sL, head_us, 4096);
  }
  // (4) exact top-k of the candidates; verifies that >= k of them exist
  SelParams p{};
  p.mode = SEL_PAIRS32;
  p.scores = (const float*)W.cand_score.p;
  p.ids32 = (const int32_t*)W.cand_id.p;
  p.stride = kCandCap;
  p.n_per_q = W.cand_cnt();
  p.n_cap = kCandCap;
  p.need = (uint32_t)std::min<int64_t>(k, N);
  p.k = k;
  p.out_scores = out_s;
  p.out_ids64 = out_i;
  p.out_stride = k;
  p.id_offset = h->id_offset;
  p.status = W.status();
  p.clear_counts = W.cand_cnt();   // counts go back to zero: the one-launch search has no preparation kernel to do it
  // the select kernel reports the candidate counts and the status word straight
  // into mapped host memory: no copy kernel between it and the sync; every search in flight
  // has its own slot of the ring
  const int slot = alloc_slot(h);
  if (slot < 0) { ts_set_error("no free report slot"); return TS_ERR_INVALID; }
  SlotGuard slot_guard(h, slot);   // every early return below gives the slot back
  uint32_t* rep = h->host_status + (size_t)slot * TS_SLOT_WORDS;
  p.host_report = h->host_status_dev + (size_t)slot * TS_SLOT_WORDS;
  for (int i = 0; i < 65; ++i) rep[i] = 0;
  TS_CHECK(ts_launch_select(p, nq, sL));
  TS_HIP(hipEventRecord(W.ev_sel, sL));
  W.used = true;
  if (pipe) TS_HIP(hipStreamWaitEvent(s, W.ev_sel, 0));  // later work on the caller's stream sees the result
  if (async) {
    // verified later, by ts_index_finish(); nothing here waits for the GPU
    std::lock_guard<std::mutex> lk(h->mu);
    ts_index::Pending& pe = h->pending[h->npending];
    pe.ticket = h->next_ticket; pe.slot = slot; pe.nq = nq; pe.S = (uint32_t)S; pe.m = m;
    pe.e0 = pe.e1 = nullptr;
    pe.set = fused ? (int)(&W - h->ws) : -1;
    if (pc.now && pc.nev == 2) {  // the scan+filter interval of this pass (prof_mu is held: h->ev is ours)
      pe.e0 = h->ev[0]; pe.e1 = h->ev[1];
      // hand the two events over and give the handle fresh ones
      hipEvent_t n0 = nullptr, n1 = nullptr;
      if (hipEventCreate(&n0) == hipSuccess && hipEventCreate(&n1) == hipSuccess) { h->ev[0] = n0; h->ev[1] = n1; }
      else { pe.e0 = pe.e1 = nullptr; }
    }
    ++h->npending;
    slot_guard.handed_over();   // ts_index_finish() frees it
    return TS_OK;
  }
  prof_mark(h, pc, -1, s);
  TS_HIP(hipStreamSynchronize(s));
  prof_collect(h, pc);
  uint32_t maxc = 0;
  for (int i = 0; i < nq; ++i) maxc = std::max(maxc, rep[i]);
  const bool redo = rep[64] != 0;
  set_info(h, (redo ? 2 : 1) | (fused ? 16 : 0), maxc, S, m);
  if (redo) {
    // a threshold was too high (fewer than k survivors) or too low (candidate
    // list overflowed, e.g. massive score ties): redo this pass exactly.
    if (fused) {   // (no prepared query image yet; a threshold workgroup that gave up may have left entries behind)
      W.hist_dirty = true;
      TS_CHECK(ts_launch_qprep(h->L, dq, q_dtype, nq, qh, (uint4*)W.qimg.p, W.cand_cnt(), W.status(), s));
    }
    prof_mark(h, pc, 5, s);
    TS_CHECK(dense_path(h, W, nq, qh, k, out_s, out_i, s));
    prof_mark(h, pc, -1, s);
    TS_HIP(hipStreamSynchronize(s));
    prof_collect(h, pc);
  }
  return TS_OK;
}

extern "C" int ts_index_search(ts_index* h, const void* queries, int32_t nq, int32_t q_dtype,
                               int32_t k, float* out_scores, int64_t* out_ids, uint32_t flags,
                               void* stream) {
  if (!h) { ts_set_error("null handle"); return TS_ERR_INVALID; }
  if (nq == 0) return TS_OK;
  if (!queries || !out_scores || !out_ids || nq < 0 || k <= 0 || !dtype_ok(q_dtype)) {
    ts_set_error("bad arguments to search");
    return TS_ERR_INVALID;
  }
  if (h->ntotal == 0) {
    ts_set_error("No documents indexed. Call add_documents() first.");
    return TS_ERR_EMPTY;
  }
  DeviceGuard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  const int qp = (ts_scan_lds_bytes(h->L, 2) <= 160 * 1024) ? 64 : 32;  // queries per pass
  if ((flags & TS_FLAG_PIPELINE) && !(flags & TS_FLAG_ASYNC)) {
    ts_set_error("TS_FLAG_PIPELINE needs TS_FLAG_ASYNC");
    return TS_ERR_INVALID;
  }
  if (flags & TS_FLAG_ASYNC) {
    if (flags & TS_FLAG_HOST_PTR) { ts_set_error("TS_FLAG_ASYNC needs device pointers"); return TS_ERR_INVALID; }
    const int passes = (nq + qp - 1) / qp;
    if (passes > 4) { ts_set_error("TS_FLAG_ASYNC: at most %d queries per call", 4 * qp); return TS_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->npending + passes > TS_ASYNC_SLOTS / 4) {
      ts_set_error("too many unfinished asynchronous searches; call ts_index_finish()");
      return TS_ERR_INVALID;
    }
  }
  const size_t qrow = (size_t)h->L.dim * dtype_size(q_dtype);
  const void* dq = queries;
  float* ds = out_scores;
  int64_t* di = out_ids;
  const bool host = (flags & TS_FLAG_HOST_PTR) != 0;
  std::unique_lock<std::mutex> hlk(h->host_mu, std::defer_lock);
  if (host) {
    hlk.lock();   // one staging area per handle
    TS_CHECK(ensure(h->qstage, (size_t)nq * qrow));
    TS_CHECK(ensure(h->out_s, (size_t)nq * k * 4));
    TS_CHECK(ensure(h->out_i, (size_t)nq * k * 8));
    TS_HIP(hipMemcpyAsync(h->qstage.p, queries, (size_t)nq * qrow, hipMemcpyHostToDevice, s));
    dq = h->qstage.p;
    ds = (float*)h->out_s.p;
    di = (int64_t*)h->out_i.p;
  }
  for (int q0 = 0; q0 < nq; q0 += qp) {
    const int c = std::min(qp, nq - q0);
    TS_CHECK(search_pass(h, (const char*)dq + (size_t)q0 * qrow, c, q_dtype, k,
                         ds + (size_t)q0 * k, di + (size_t)q0 * k, flags, s));
  }
  if (flags & TS_FLAG_ASYNC) {
    std::lock_guard<std::mutex> lk(h->mu);
    ++h->next_ticket;
  }
  if (host) {
    TS_HIP(hipMemcpyAsync(out_scores, ds, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s));
    TS_HIP(hipMemcpyAsync(out_ids, di, (size_t)nq * k * 8, hipMemcpyDeviceToHost, s));
    TS_HIP(hipStreamSynchronize(s));
  }
  return TS_OK;
}

// All inner products (no selection): the dense scan writes straight into the caller's matrix.
extern "C" int ts_index_scores(ts_index* h, const void* queries, int32_t nq, int32_t q_dtype, float* out,
                               int64_t ld, void* stream) {
  if (!h) { ts_set_error("null handle"); return TS_ERR_INVALID; }
  if (nq == 0) return TS_OK;
  const int64_t N = h->ntotal;
  const int64_t nblk = (N + TS_ROWS_PER_BLOCK - 1) / TS_ROWS_PER_BLOCK;
  if (!queries || !out || nq < 0 || !dtype_ok(q_dtype) || ld < nblk * TS_ROWS_PER_BLOCK || (ld & 3) ||
      (reinterpret_cast<uintptr_t>(out) & 15)) {
    ts_set_error("bad arguments to scores (ld must be a multiple of 4 and >= ntotal rounded up to 32, out 16-byte aligned)");
    return TS_ERR_INVALID;
  }
  if (N == 0) { ts_set_error("No documents indexed. Call add_documents() first."); return TS_ERR_EMPTY; }
  DeviceGuard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  const int qp = (ts_scan_lds_bytes(h->L, 2) <= 160 * 1024) ? 64 : 32;
  const size_t qrow = (size_t)h->L.dim * dtype_size(q_dtype);
  for (int q0 = 0; q0 < nq; q0 += qp) {
    const int c = std::min(qp, nq - q0);
    const int qh = c > 32 ? 2 : 1;
    ts_index::WSet* W = acquire_set(h);
    int st = TS_OK;
    if (W->used && hipStreamWaitEvent(s, W->ev_sel, 0) != hipSuccess) st = TS_ERR_HIP;
    if (st == TS_OK)
      st = ts_launch_qprep(h->L, (const char*)queries + (size_t)q0 * qrow, q_dtype, c, qh, (uint4*)W->qimg.p,
                           nullptr, nullptr, s);
    if (st == TS_OK) {
      ScanParams sp{};
      sp.corpus = h->corpus;
      sp.qimg = (const uint4*)W->qimg.p;
      sp.kg = h->L.kg;
      sp.nq = c;
      sp.nwork = nblk;
      sp.blk0 = 0;
      sp.blk_stride = 1;
      sp.ntotal = N;
      sp.dense = out + (size_t)q0 * ld;
      sp.dense_ld = ld;
      st = ts_launch_scan(h->L, SCAN_DENSE, qh, sp, h->num_cus, s);
    }
    if (st == TS_OK && hipEventRecord(W->ev_sel, s) == hipSuccess) W->used = true;   // orders the set's next user behind this scan
    else if (st == TS_OK) st = TS_ERR_HIP;
    release_set(h, W);
    if (st != TS_OK) {
      if (st == TS_ERR_HIP) ts_set_error("HIP call failed in ts_index_scores");
      return st;
    }
  }
  TS_HIP(hipStreamSynchronize(s));
  return TS_OK;
}

// 1 = a search with this k goes through the threshold filter (its result has to be VERIFIED: a synchronous call does
// it, an asynchronous one leaves it to ts_index_finish), 0 = it takes the dense path, which is exact by construction —
// an asynchronous search is then final as soon as the stream reaches it.  Mirrors search_pass_on.
extern "C" int ts_index_filter_path(const ts_index* h, int32_t k) {
  if (!h || k <= 0) { ts_set_error("bad arguments to filter_path"); return TS_ERR_INVALID; }
  const int64_t N = h->ntotal;
  return (k <= kMaxFilterK && N >= kMinFilterRows && N >= 32 * (int64_t)k) ? 1 : 0;
}

extern "C" int64_t ts_index_last_ticket(const ts_index* h) {
  if (!h) return -1;
  std::lock_guard<std::mutex> lk(const_cast<ts_index*>(h)->mu);
  return h->next_ticket - 1;
}

extern "C" int ts_index_finish(ts_index* h, void* stream, int64_t* failed_tickets, int32_t max_failed,
                               int32_t* n_failed) {
  if (!h || !n_failed || max_failed < 0 || (max_failed > 0 && !failed_tickets)) {
    ts_set_error("bad arguments to finish");
    return TS_ERR_INVALID;
  }
  DeviceGuard g(h->device);
  TS_HIP(hipStreamSynchronize((hipStream_t)stream));
  std::lock_guard<std::mutex> lk(h->mu);
  int nf = 0;
  uint32_t maxc = 0;
  for (int i = 0; i < h->npending; ++i) {
    ts_index::Pending& pe = h->pending[i];
    const uint32_t* rep = h->host_status + (size_t)pe.slot * TS_SLOT_WORDS;
    for (int q = 0; q < pe.nq; ++q) maxc = std::max(maxc, rep[q]);
    if (rep[64] != 0) {
      if (pe.set >= 0 && pe.set < TS_NSETS) h->ws[pe.set].hist_dirty = true;
      if (nf == 0 || failed_tickets[nf - 1] != pe.ticket) {
        if (nf < max_failed) failed_tickets[nf] = pe.ticket;
        ++nf;
      }
    }
    if (pe.e0 && pe.e1) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, pe.e0, pe.e1) == hipSuccess) { h->phase_ms[3] += ms; h->phase_cnt[3] += 1; }
      (void)hipEventDestroy(pe.e0);
      (void)hipEventDestroy(pe.e1);
    }
    h->info[0] = 1 | (pe.set >= 0 ? 16 : 0); h->info[2] = pe.S; h->info[3] = pe.m;
    h->slot_busy[pe.slot] = false;
  }
  if (h->npending) h->info[1] = maxc;
  h->npending = 0;
  *n_failed = nf;
  if (nf > max_failed) { ts_set_error("%d searches need a redo but only %d tickets fit", nf, max_failed); return TS_ERR_INVALID; }
  return TS_OK;
}

extern "C" int ts_index_set_profiling(ts_index* h, int32_t on) {
  if (!h) { ts_set_error("null handle"); return TS_ERR_INVALID; }
  DeviceGuard g(h->device);
  if (on && !h->ev[0]) {
    for (hipEvent_t& e : h->ev) TS_HIP(hipEventCreate(&e));
  }
  h->profiling = on != 0;
  h->prof_every = on > 1 ? on : 1;   // on = N: time every N-th search
  h->prof_seq.store(0);
  return TS_OK;
}

extern "C" int ts_index_get_timings(ts_index* h, double ms[8], int64_t counts[8], int32_t reset) {
  if (!h || !ms || !counts) { ts_set_error("bad arguments"); return TS_ERR_INVALID; }
  std::lock_guard<std::mutex> lk(h->mu);
  for (int i = 0; i < TS_NPHASE; ++i) {
    ms[i] = h->phase_ms[i];
    counts[i] = h->phase_cnt[i];
    if (reset) { h->phase_ms[i] = 0.0; h->phase_cnt[i] = 0; }
  }
  return TS_OK;
}

// ------------------------------------------------------------------ read-bandwidth probe
// What a kernel that ONLY reads the index's tiled corpus reaches on this box, with the scan's own access pattern
// (persistent workgroups of 8 waves on 7/8 of the CUs, wave w takes row blocks w, w + W, ...; a block is kg
// contiguous 1 KiB units read 8 at a time with non-temporal 16-byte loads) and nothing else: no LDS, no MFMA, no
// epilogue.  bench.py reports it next to the 8 TB/s specification as the measured ceiling.
typedef uint32_t probe_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void read_probe_kernel(const probe_u32x4* corpus, int64_t nblk, int kg,
                                                         uint32_t* sink) {
  const int lane = threadIdx.x & 63;
  const int64_t W = (int64_t)gridDim.x * (blockDim.x >> 6);
  probe_u32x4 acc = {0u, 0u, 0u, 0u};
  // two groups of 8 loads alternate, so 8-16 KiB per wave are always in flight (the scan keeps 8 KiB in its ring and a
  // few waits apart; a probe that waited for each group before requesting the next read SLOWER than the scan).  No
  // branch around a load: the group index is clamped instead (a wave re-reads at most its last two groups).
  const int64_t b0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int gpb = kg / 8;                                       // groups of 8 units per row block
  const int64_t myblocks = b0 < nblk ? (nblk - b0 + W - 1) / W : 0;
  const int64_t ng = myblocks * gpb;
  if (ng == 0) return;
  auto group = [&](int64_t t) -> const probe_u32x4* {
    t = t < ng ? t : ng - 1;
    return corpus + ((size_t)(b0 + (t / gpb) * W) * kg + (size_t)(t % gpb) * 8) * 64 + lane;
  };
  probe_u32x4 va[8], vb[8];
  {
    const probe_u32x4* q = group(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) va[i] = __builtin_nontemporal_load(q + (size_t)i * 64);
  }
  for (int64_t t = 0; t < ng; t += 2) {
    const probe_u32x4* qb = group(t + 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) vb[i] = __builtin_nontemporal_load(qb + (size_t)i * 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= va[i];
    const probe_u32x4* qa = group(t + 2);
#pragma unroll
    for (int i = 0; i < 8; ++i) va[i] = __builtin_nontemporal_load(qa + (size_t)i * 64);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= vb[i];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u) sink[0] = 1u;   // keeps the loads alive; practically never true
}

extern "C" int ts_index_read_probe(ts_index* h, int32_t reps, double* ms_avg, double* ms_best, int64_t* bytes,
                                   void* stream) {
  if (!h || reps <= 0 || reps > 1000 || !ms_avg || !ms_best || !bytes) {
    ts_set_error("bad arguments to read_probe");
    return TS_ERR_INVALID;
  }
  if (h->ntotal == 0) { ts_set_error("No documents indexed. Call add_documents() first."); return TS_ERR_EMPTY; }
  DeviceGuard g(h->device);
  hipStream_t s = (hipStream_t)stream;
  const int64_t nblk = (h->ntotal + TS_ROWS_PER_BLOCK - 1) / TS_ROWS_PER_BLOCK;
  const int grid = std::max(1, h->num_cus - h->num_cus / 8);
  ts_index::WSet* W = acquire_set(h);          // its 4 KiB scratch page: word 1023 is nobody's
  uint32_t* sink = (uint32_t*)W->small.p + 1023;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int st = TS_OK;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) st = TS_ERR_HIP;
  double sum = 0.0, best = 1e30;
  for (int r = 0; r <= reps && st == TS_OK; ++r) {     // pass 0 is a warm-up
    if (hipEventRecord(e0, s) != hipSuccess) { st = TS_ERR_HIP; break; }
    hipLaunchKernelGGL(read_probe_kernel, dim3(grid), dim3(512), 0, s, (const probe_u32x4*)h->corpus, nblk, h->L.kg, sink);
    float ms = 0.f;
    if (hipGetLastError() != hipSuccess || hipEventRecord(e1, s) != hipSuccess ||
        hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { st = TS_ERR_HIP; break; }
    if (r) { sum += ms; best = std::min(best, (double)ms); }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  release_set(h, W);
  if (st != TS_OK) { ts_set_error("HIP call failed in ts_index_read_probe"); return st; }
  *ms_avg = sum / reps;
  *ms_best = best;
  *bytes = nblk * (int64_t)ts_block_bytes(h->L);
  return TS_OK;
}

// ------------------------------------------------------------------ diagnostics
// The per-device one-time table (ts_common.h TsDeviceOnce) exercised WITHOUT a GPU: n_threads host
// threads race through n_devices device numbers; every device's action must run exactly once, a
// failing action must be retried, and an out-of-range device must run the action every time.
extern "C" int ts_selftest_device_once(int32_t n_threads, int32_t n_devices) {
  if (n_threads <= 0 || n_threads > 256 || n_devices <= 0 || n_devices > TS_MAX_DEVICES) {
    ts_set_error("bad arguments to selftest");
    return TS_ERR_INVALID;
  }
  TsDeviceOnce once;
  std::atomic<int> runs[TS_MAX_DEVICES];
  std::atomic<int> fails[TS_MAX_DEVICES];
  for (int i = 0; i < TS_MAX_DEVICES; ++i) { runs[i] = 0; fails[i] = 0; }
  std::atomic<int> bad{0}, untracked{0};
  auto worker = [&](int t) {
    for (int rep = 0; rep < 200; ++rep) {
      const int dev = (t + rep) % n_devices;
      int st;
      do {
        st = ts_once_per_device(once, dev, [&]() -> int {
          // the first attempt on every odd device fails: it must not be recorded as done
          if ((dev & 1) && fails[dev].fetch_add(1) == 0) return TS_ERR_HIP;
          runs[dev].fetch_add(1);
          return TS_OK;
        });
      } while (st != TS_OK);
      if (!(once.done.load() & (1ull << dev))) bad.fetch_add(1);
    }
    (void)ts_once_per_device(once, TS_MAX_DEVICES + t, [&]() -> int { untracked.fetch_add(1); return TS_OK; });
    (void)ts_once_per_device(once, -1, [&]() -> int { untracked.fetch_add(1); return TS_OK; });
  };
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t) th.emplace_back(worker, t);
  for (std::thread& x : th) x.join();
  for (int d = 0; d < n_devices; ++d)
    if (runs[d].load() != 1) { ts_set_error("device %d: action ran %d times", d, runs[d].load()); return TS_ERR_INVALID; }
  for (int d = n_devices; d < TS_MAX_DEVICES; ++d)
    if (runs[d].load() != 0 || (once.done.load() & (1ull << d))) { ts_set_error("device %d touched", d); return TS_ERR_INVALID; }
  if (bad.load() || untracked.load() != 2 * n_threads) {
    ts_set_error("bit missing after success (%d) or untracked runs %d != %d", bad.load(), untracked.load(), 2 * n_threads);
    return TS_ERR_INVALID;
  }
  return TS_OK;
}

// ------------------------------------------------------------------ merge
static int merge_once(const float* scores, const int64_t* ids, int32_t nlists, int32_t nq, int32_t k,
                      int64_t score_list_stride, int64_t id_list_stride, float* out_scores,
                      int64_t* out_ids, void* stream) {
  SelParams p{};
  p.mode = SEL_MERGE64;
  p.scores = scores;
  p.ids64 = ids;
  p.stride = k;                       // query q of list 0 starts at q*k
  p.seg_len = (uint32_t)k;
  p.seg_stride = score_list_stride;   // next list (scores)
  p.seg_stride_ids = id_list_stride;  // next list (ids)
  p.n = (uint32_t)(nlists * k);
  p.k = k;
  p.out_scores = out_scores;
  p.out_ids64 = out_ids;
  p.out_stride = k;
  return ts_launch_select(p, nq, (hipStream_t)stream);
}

static int merge_impl(const float* scores, const int64_t* ids, int32_t nlists, int32_t nq, int32_t k,
                      int64_t score_list_stride, int64_t id_list_stride, float* out_scores,
                      int64_t* out_ids, int32_t device, void* stream) {
  if (!scores || !ids || !out_scores || !out_ids || nlists <= 0 || nq < 0 || k <= 0) {
    ts_set_error("bad arguments to merge");
    return TS_ERR_INVALID;
  }
  if (nq == 0) return TS_OK;
  if (k > TS_SEL_LDS_KEYS / 2) {
    ts_set_error("merge: k = %d exceeds %d", k, TS_SEL_LDS_KEYS / 2);
    return TS_ERR_UNSUPPORTED;
  }
  DeviceGuard g(device);
  if (!g.ok) { ts_set_error("hipSetDevice(%d) failed", device); return TS_ERR_HIP; }
  if ((int64_t)nlists * k <= TS_SEL_LDS_KEYS)
    return merge_once(scores, ids, nlists, nq, k, score_list_stride, id_list_stride, out_scores, out_ids, stream);
  // More entries per query than one workgroup selects among in LDS (e.g. 16 ranks x k = 2048): merge the lists
  // in groups that fit, then merge the groups' results — the order (score desc, id asc) is total, so merging is
  // associative and the result is the same.  Rare, so the intermediate lists are allocated on the spot.
  const int per = TS_SEL_LDS_KEYS / k;                    // lists per group (>= 2)
  const int ngroups = (nlists + per - 1) / per;
  float* ts = nullptr;
  int64_t* ti = nullptr;
  const size_t cells = (size_t)ngroups * nq * k;
  if (hipMalloc((void**)&ts, cells * 4) != hipSuccess || hipMalloc((void**)&ti, cells * 8) != hipSuccess) {
    if (ts) (void)hipFree(ts);
    ts_set_error("merge: cannot allocate %zu bytes of intermediate lists", cells * 12);
    return TS_ERR_OOM;
  }
  int st = TS_OK;
  for (int gi = 0; gi < ngroups && st == TS_OK; ++gi) {
    const int l0 = gi * per, nl = std::min(per, nlists - l0);
    st = merge_once(scores + (size_t)l0 * score_list_stride, ids + (size_t)l0 * id_list_stride, nl, nq, k,
                    score_list_stride, id_list_stride, ts + (size_t)gi * nq * k, ti + (size_t)gi * nq * k, stream);
  }
  if (st == TS_OK)
    st = merge_impl(ts, ti, ngroups, nq, k, (int64_t)nq * k, (int64_t)nq * k, out_scores, out_ids, device, stream);
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess && st == TS_OK) {
    ts_set_error("merge: stream synchronisation failed");
    st = TS_ERR_HIP;
  }
  (void)hipFree(ts);
  (void)hipFree(ti);
  return st;
}

extern "C" int ts_merge_topk(const float* scores, const int64_t* ids, int32_t nlists,
                             int32_t nq, int32_t k, float* out_scores, int64_t* out_ids,
                             int32_t device, void* stream) {
  return merge_impl(scores, ids, nlists, nq, k, (int64_t)nq * k, (int64_t)nq * k, out_scores, out_ids,
                    device, stream);
}

extern "C" int ts_merge_topk_strided(const float* scores, const int64_t* ids, int32_t nlists,
                                     int32_t nq, int32_t k, int64_t score_list_stride,
                                     int64_t id_list_stride, float* out_scores, int64_t* out_ids,
                                     int32_t device, void* stream) {
  return merge_impl(scores, ids, nlists, nq, k, score_list_stride, id_list_stride, out_scores, out_ids,
                    device, stream);
}

// ------------------------------------------------------------------ maxsim
extern "C" int ts_maxsim(const void* q, int32_t Lq, const void* docs, const int32_t* doc_off,
                         int32_t n_docs, int32_t H, int32_t dtype, int32_t mode, float* out,
                         int32_t device, void* stream) {
  if (n_docs == 0) return TS_OK;
  if (!q || !docs || !doc_off || !out || Lq < 0 || n_docs < 0 || H <= 0 || !dtype_ok(dtype) ||
      (mode != 0 && mode != 1)) {
    ts_set_error("bad arguments to maxsim");
    return TS_ERR_INVALID;
  }
  DeviceGuard g(device);
  if (!g.ok) { ts_set_error("hipSetDevice(%d) failed", device); return TS_ERR_HIP; }
  const int st = ts_launch_maxsim16(q, Lq, docs, doc_off, nullptr, nullptr, n_docs, H, dtype, mode, out,
                                    device, (hipStream_t)stream);
  if (st != TS_ERR_UNSUPPORTED) return st;
  return ts_launch_maxsim(q, Lq, docs, doc_off, nullptr, nullptr, n_docs, H, dtype, mode, out,
                          (hipStream_t)stream);
}

extern "C" int ts_maxsim_indexed(const void* q, int32_t Lq, const void* store, const int64_t* starts,
                                 const int32_t* lens, int32_t n_docs, int32_t H, int32_t dtype,
                                 int32_t mode, float* out, int32_t device, void* stream) {
  if (n_docs == 0) return TS_OK;
  if (!q || !store || !starts || !lens || !out || Lq < 0 || n_docs < 0 || H <= 0 || !dtype_ok(dtype) ||
      (mode != 0 && mode != 1)) {
    ts_set_error("bad arguments to maxsim_indexed");
    return TS_ERR_INVALID;
  }
  DeviceGuard g(device);
  if (!g.ok) { ts_set_error("hipSetDevice(%d) failed", device); return TS_ERR_HIP; }
  const int st = ts_launch_maxsim16(q, Lq, store, nullptr, starts, lens, n_docs, H, dtype, mode, out,
                                    device, (hipStream_t)stream);
  if (st != TS_ERR_UNSUPPORTED) return st;
  return ts_launch_maxsim(q, Lq, store, nullptr, starts, lens, n_docs, H, dtype, mode, out,
                          (hipStream_t)stream);
}

extern "C" int ts_maxsim_indexed_batch(const void* q, const int32_t* q_off, int32_t nq, const void* store,
                                       const int64_t* starts, const int32_t* lens, const int32_t* cand_off,
                                       int32_t H, int32_t dtype, int32_t mode, float* out, int32_t device,
                                       void* stream) {
  if (nq == 0) return TS_OK;
  if (!q || !q_off || !store || !starts || !lens || !cand_off || !out || nq < 0 || H <= 0 ||
      !dtype_ok(dtype) || (mode != 0 && mode != 1)) {
    ts_set_error("bad arguments to maxsim_indexed_batch");
    return TS_ERR_INVALID;
  }
  for (int j = 0; j < nq; ++j)
    if (q_off[j + 1] < q_off[j] || cand_off[j + 1] < cand_off[j]) {
      ts_set_error("maxsim_indexed_batch: offsets must be non-decreasing");
      return TS_ERR_INVALID;
    }
  DeviceGuard g(device);
  if (!g.ok) { ts_set_error("hipSetDevice(%d) failed", device); return TS_ERR_HIP; }
  const int st = ts_launch_maxsim16_batch(q, q_off, nq, store, starts, lens, cand_off, H, dtype, mode, out,
                                          device, (hipStream_t)stream);
  if (st != TS_ERR_UNSUPPORTED) return st;
  // shapes the one-launch form does not take: query by query
  const size_t esize = (dtype == TS_F32) ? 4 : 2;
  for (int j = 0; j < nq; ++j) {
    const int nc = cand_off[j + 1] - cand_off[j];
    if (nc == 0) continue;
    TS_CHECK(ts_maxsim_indexed((const char*)q + (size_t)q_off[j] * H * esize, q_off[j + 1] - q_off[j], store,
                               starts + cand_off[j], lens + cand_off[j], nc, H, dtype, mode,
                               out + cand_off[j], device, stream));
  }
  return TS_OK;
}
