/*
 * tristage.h — C ABI of the MI355X-native retrieval hot path of TriStage-RAG.
 *
 * This is the drop-in boundary.  The reference has no FFI of
 * its own: its native seam is the duck-typed FAISS index object held in
 * Stage1Retriever.faiss_index (reference src/stage1_retriever.py:126) plus
 * the per-candidate torch MaxSim in ColBERTScorer (src/stage2_rescorer.py:167-201).
 * Each entry point below names the reference call site it replaces.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative
 *     ts_status code, and ts_last_error() gives a thread-local message;
 *   - "device pointer" = HIP device memory on the index's device (e.g. a torch
 *     tensor's data_ptr()); `stream` is a hipStream_t passed as void* (NULL =
 *     the default stream);
 *   - the library owns the corpus memory after ts_index_add; the caller owns
 *     every output buffer; a handle is freed only by ts_index_destroy;
 *   - threading: ts_index_search on a built index is safe for
 *     concurrent callers on ONE handle that use distinct streams and output
 *     buffers: each call takes one of 4 internal workspace sets (a 5th
 *     synchronous caller waits for a set), its own report slot, and the exact
 *     fallback of a synchronous call runs on that call's set.  Host-pointer
 *     searches (TS_FLAG_HOST_PTR) share one staging area and are serialised
 *     inside the handle, as are searches while per-phase profiling is on.
 *     Asynchronous submission (TS_FLAG_ASYNC + ts_index_finish) is for ONE
 *     submitting thread per handle at a time; synchronous searches from other
 *     threads may run beside it.  add / reset / reserve / set_id_offset /
 *     destroy need exclusive access.  The stateless entry points (ts_merge_topk*,
 *     ts_maxsim*, and the forward kernels ts_add_layernorm / ts_add_prenorm /
 *     ts_embed_layernorm / ts_attention_varlen / ts_rope_inplace / ts_geglu /
 *     ts_linear_*) are
 *     thread-safe; a ts_bm25 handle serves one caller at a time; per-kernel device attributes are set once per
 *     device under a lock, so one process may drive several GPUs.
 */
#ifndef TRISTAGE_H_
#define TRISTAGE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an existing signature or the meaning of an argument changes, or entry points are added
 * (a caller built against version N may load any library whose ts_abi_version() == N; nothing older, nothing
 * newer).  1 = round 1 (24 entry points).  2 = round 2 changed ts_attention_varlen (offsets, window, rotary
 * tables) and added 15 entry points.  3 = round 3 added ts_index_read_probe, ts_index_filter_path and
 * ts_linear_add_layernorm, ts_mlp_add_layernorm.                                                            */
#define TS_ABI_VERSION 3

typedef struct ts_index ts_index; /* opaque */

enum ts_status {
  TS_OK = 0,
  TS_ERR_INVALID = -1,   /* bad argument */
  TS_ERR_HIP = -2,       /* a HIP runtime call failed */
  TS_ERR_OOM = -3,       /* device allocation failed */
  TS_ERR_EMPTY = -4,     /* search on an index with no rows
                            (reference: ValueError "No documents indexed",
                            src/stage1_retriever.py:370-371) */
  TS_ERR_UNSUPPORTED = -5
};

enum ts_dtype { TS_F32 = 0, TS_F16 = 1, TS_BF16 = 2 };

enum ts_metric { TS_METRIC_INNER_PRODUCT = 0 };

/* ts_index_add / ts_index_search flags */
#define TS_FLAG_HOST_PTR 1u     /* `rows` / `queries` / outputs are host memory */
#define TS_FLAG_NO_FILTER 2u    /* search: force the dense (materialise+select) path */
#define TS_FLAG_ASYNC 8u        /* search: enqueue only (device pointers); results are valid
                                   and verified after ts_index_finish()            */
#define TS_FLAG_PIPELINE 16u    /* with TS_FLAG_ASYNC: overlap this search's query preparation and
                                   final selection with the scans of its neighbours (internal
                                   streams).  The caller guarantees that `queries` is already
                                   complete in memory when the call is made (not the output of
                                   work still pending on `stream`); results are ordered after
                                   the call on `stream` as usual.                             */
#define TS_FLAG_CLASSIC 32u     /* search: take the five-launch filter path (query prep, sample scan,
                                   thresholds, scan+filter, select) even where the one-launch scan
                                   (query image, thresholds and scan+filter in ONE kernel) is the
                                   default: A/B measurements and tests; results are identical    */
#define TS_FLAG_ONE_LAUNCH 64u  /* search: take the one-launch scan wherever its threshold estimate is
                                   valid, also where the five-launch path is the default (pipelined
                                   submission, corpora above 4 M rows); results are identical      */
#define TS_FLAG_NORMALIZE 4u    /* add: L2-normalise rows x/(|x|+1e-8) on device first
                                   (reference src/stage1_retriever.py:285-288) */

/* ---- index lifetime ------------------------------------------------------
 * replaces faiss.IndexFlatIP(d) (reference src/stage1_retriever.py:263,276).
 * storage_dtype: element type the corpus is kept in (TS_F32 = FAISS-exact
 * storage; TS_F16 / TS_BF16 halve the bytes scanned).                        */
int ts_index_create(int32_t dim, int32_t storage_dtype, int32_t metric,
                    int32_t device, ts_index** out);
int ts_index_destroy(ts_index* h);
/* drop all rows, keep the allocation */
int ts_index_reset(ts_index* h);
/* pre-size the corpus allocation for `nrows` rows in total */
int ts_index_reserve(ts_index* h, int64_t nrows);

/* ---- add -----------------------------------------------------------------
 * replaces faiss_index.add(float32[n,d]) (src/stage1_retriever.py:270,277,313).
 * rows: row-major [n, dim] of rows_dtype (device pointer unless
 * TS_FLAG_HOST_PTR).  Rows are appended; ids are assigned consecutively.    */
int ts_index_add(ts_index* h, const void* rows, int64_t n, int32_t rows_dtype,
                 uint32_t flags, void* stream);

/* ---- search --------------------------------------------------------------
 * replaces faiss_index.search(float32[B,d], k) (src/stage1_retriever.py:380):
 * exact inner product of each query against every row; out_scores[B,k]
 * sorted descending, ties by ascending id; out_ids[B,k] int64; when k exceeds
 * the row count the tail is padded with id -1 / score -FLT_MAX (FAISS's
 * convention, which the reference filters at src/stage1_retriever.py:383).
 * Ids are row numbers plus the offset set by ts_index_set_id_offset.
 * Synchronous with respect to `stream` on return.                            */
int ts_index_search(ts_index* h, const void* queries, int32_t nq,
                    int32_t q_dtype, int32_t k, float* out_scores,
                    int64_t* out_ids, uint32_t flags, void* stream);

/* ---- all scores, no selection -----------------------------------------------
 * replaces the numpy product in EmbeddingService.similarity (reference
 * src/embedding_service.py:228-237: cosine of one query against a document
 * matrix, result in document order): out[q*ld + row] = <query q, row> for every
 * row; ld is a multiple of 4 and >= ntotal rounded up to 32 (the entries
 * [ntotal, that bound) of each line are set to -FLT_MAX), `out` device memory,
 * 16-byte aligned.  Synchronous with respect to `stream` on return.            */
int ts_index_scores(ts_index* h, const void* queries, int32_t nq, int32_t q_dtype,
                    float* out, int64_t ld, void* stream);

/* ---- asynchronous searches ------------------------------------------------
 * With TS_FLAG_ASYNC ts_index_search only enqueues work on `stream` and returns;
 * consecutive batches then run back to back on the GPU with no host round trip
 * in between.  Each call gets a ticket (ts_index_last_ticket).  ts_index_finish
 * synchronises the stream once and checks every unfinished search: the tickets
 * whose fused filter could not prove exactness (rare) are
 * returned in failed_tickets[0..*n_failed) and must be repeated by the caller
 * with TS_FLAG_NO_FILTER (synchronously).  At most 256 passes (of <= 64 queries each; <= 32 where the query
 * image of 64 does not fit LDS) may be unfinished (round 2: 64 — a collective finish every 30 batches cost the
 * sharded path 4-7 % of its time at 2.5 M / 1.25 M rows per rank).                                              */
int64_t ts_index_last_ticket(const ts_index* h);
/* 1 if a search for top-k on this index takes the threshold-filter path (exact only once verified: a synchronous
 * call verifies before it returns, an asynchronous one in ts_index_finish), 0 if it takes the dense path, whose result
 * is exact by construction — an asynchronous search is then final when the stream reaches it, so a caller may consume
 * it in stream order without waiting (the per-query path of RetrievalPipeline.search on small corpora: no host sync
 * between faiss_index.search, reference src/stage1_retriever.py:380, and stage 2).  Negative: error.          */
int ts_index_filter_path(const ts_index* h, int32_t k);
int ts_index_finish(ts_index* h, void* stream, int64_t* failed_tickets, int32_t max_failed,
                    int32_t* n_failed);

/* ---- introspection -------------------------------------------------------
 * faiss_index.ntotal / .d                                                    */
int64_t ts_index_ntotal(const ts_index* h);
int32_t ts_index_dim(const ts_index* h);
int32_t ts_index_dtype(const ts_index* h);
/* row-shard support: ids reported by search = local row + offset            */
int ts_index_set_id_offset(ts_index* h, int64_t offset);
/* copy rows [row0, row0+n) back out as row-major float32 (host or device):
 * used by save_index (reference src/stage1_retriever.py:421-441)            */
int ts_index_reconstruct(ts_index* h, int64_t row0, int64_t n, float* out,
                         uint32_t flags, void* stream);
/* counters of the last search on this handle: [0] path taken (low 4 bits: 0 dense,
 * 1 filter, 2 filter-then-dense fallback; +16 when the filter ran as the
 * one-launch scan), [1] max candidates per query, [2] sample rows,
 * [3] sample rank m                                                          */
int ts_index_last_search_info(const ts_index* h, int64_t info[4]);

/* Per-phase device timing of searches, measured with HIP events recorded on the
 * search's own stream (bench.py's roofline leg).  Phases: 0 query prep,
 * 1 sample scan, 2 thresholds, 3 fused scan+filter (the dominant kernel),
 * 4 candidate select, 5 dense path (scan+select), 6-7 unused.  ms[i] is the sum
 * over counts[i] occurrences since the last reset.  on = N > 1 times every N-th
 * search only (a timing event costs a few microseconds in the stream);
 * asynchronous searches record phase 3 only.                                 */
int ts_index_set_profiling(ts_index* h, int32_t on);
int ts_index_get_timings(ts_index* h, double ms[8], int64_t counts[8], int32_t reset);

/* Read-bandwidth ceiling of THIS box for the scan's access pattern (SURVEY.md 8d asks for a measured peak in
 * the same report as the roofline fraction): a kernel that only reads the index's tiled corpus — the scan's grid,
 * block order and non-temporal 16-byte loads, no LDS, no matrix cores, no epilogue — timed with HIP events on
 * `stream`, `reps` passes after one warm-up pass.  *bytes = bytes one pass reads (the algorithmic bytes of one
 * ts_index_search launch on this index).  No reference counterpart: measurement aid of bench.py.            */
int ts_index_read_probe(ts_index* h, int32_t reps, double* ms_avg, double* ms_best, int64_t* bytes,
                        void* stream);

/* ---- merge of per-shard partial top-k lists --------------------------------
 * New for the row-sharded multi-GPU path: `scores`/`ids` are
 * device arrays [nlists, nq, k] (e.g. the output of an RCCL all-gather of each
 * rank's ts_index_search result); writes the global top-k [nq, k] in the same
 * canonical order.  Entries with id < 0 are padding and ignored.  k <= 8192;
 * any number of lists (more than 16384 / k of them are merged in groups, then the
 * groups' results: the order is total, so the result is the same).            */
int ts_merge_topk(const float* scores, const int64_t* ids, int32_t nlists,
                  int32_t nq, int32_t k, float* out_scores, int64_t* out_ids,
                  int32_t device, void* stream);

/* Same, for lists that are not densely packed: list r's scores start at
 * scores + r*score_list_stride (floats) and its ids at ids + r*id_list_stride
 * (int64s) — e.g. each rank's [scores | ids] bytes gathered into one buffer, so
 * the merge reads the all-gather output in place.                            */
int ts_merge_topk_strided(const float* scores, const int64_t* ids, int32_t nlists,
                          int32_t nq, int32_t k, int64_t score_list_stride,
                          int64_t id_list_stride, float* out_scores, int64_t* out_ids,
                          int32_t device, void* stream);

/* ---- stage-2 MaxSim ------------------------------------------------------
 * replaces ColBERTScorer._maxsim_score / _colbert_score applied per candidate
 * (reference src/stage2_rescorer.py:167-201, loop at :268-276).
 * q:       device [Lq, H] token embeddings of the query (dtype)
 * docs:    device [sum(Ld_i), H] token embeddings of all candidates, packed
 * doc_off: device int32 [n_docs+1] row offsets into docs
 * mode:    0 = maxsim (mean_i max_j cos), 1 = colbert (softmax-weighted)
 * out:     device float32 [n_docs]                                           */
int ts_maxsim(const void* q, int32_t Lq, const void* docs,
              const int32_t* doc_off, int32_t n_docs, int32_t H, int32_t dtype,
              int32_t mode, float* out, int32_t device, void* stream);

/* Same scores for candidates that already live in a resident token store
 * (token matrices computed once at add time instead of per query,
 * reference src/stage2_rescorer.py:254-259): document i occupies rows
 * [starts[i], starts[i]+lens[i]) of `store` [rows, H].  starts: device int64[n_docs],
 * lens: device int32[n_docs].  No gather copy: the kernel reads the store in place. */
int ts_maxsim_indexed(const void* q, int32_t Lq, const void* store, const int64_t* starts,
                      const int32_t* lens, int32_t n_docs, int32_t H, int32_t dtype,
                      int32_t mode, float* out, int32_t device, void* stream);

/* The same for several queries in ONE launch (the batched caller: RetrievalPipeline.search_many):
 * query j has tokens [q_off[j], q_off[j+1]) of q [sum Lq, H] and candidates
 * [cand_off[j], cand_off[j+1]) of starts / lens / out.  q_off and cand_off are HOST arrays of
 * nq+1 int32 starting at 0 (they travel inside the kernel arguments, 64 queries per launch, so the
 * caller may free them on return); everything else is device memory.  A single query's launch is
 * dominated by fixed costs (~25 of ~50 us at 1000 candidates); batched, they overlap with the
 * other queries' streaming.                                                              */
int ts_maxsim_indexed_batch(const void* q, const int32_t* q_off, int32_t nq, const void* store,
                            const int64_t* starts, const int32_t* lens, const int32_t* cand_off,
                            int32_t H, int32_t dtype, int32_t mode, float* out, int32_t device,
                            void* stream);

/* ---- BM25 (the lexical half of stage 1) ---------------------------------------
 * replaces BM25Index.search (reference src/stage1_retriever.py:103-112, called at
 * :385-388): same float64 arithmetic and ordering (score desc, doc id asc), but the
 * postings live in HBM (CSR) and a query touches only its terms' postings.
 * ts_bm25_set_index takes HOST arrays: term_off[V+1], post_doc/post_tf[nnz] sorted
 * by term, idf[V], len_norm[N] = k1*(1-b+b*len/avg), k1p1 = k1+1.
 * ts_bm25_search: term_ids = query tokens as vocabulary ids in query order (host);
 * writes up to k (score, doc) pairs to HOST arrays; *n_out < k means all documents
 * with a non-zero score were returned (the rest score exactly 0.0).  A ts_bm25
 * handle keeps ONE set of accumulator / candidate workspaces: calls on one handle must
 * not overlap (one caller at a time; different handles are independent).        */
typedef struct ts_bm25 ts_bm25;
int ts_bm25_create(int32_t device, ts_bm25** out);
int ts_bm25_destroy(ts_bm25* h);
int ts_bm25_set_index(ts_bm25* h, int64_t N, int64_t V, int64_t nnz, const int64_t* term_off,
                      const int32_t* post_doc, const float* post_tf, const double* idf,
                      const double* len_norm, double k1p1);
int ts_bm25_search(ts_bm25* h, const int32_t* term_ids, int32_t n_terms, int32_t k,
                   double* out_scores, int64_t* out_ids, int32_t* n_out, void* stream);
/* The same for nq queries with ONE synchronisation: query q's terms are
 * term_ids[term_off[q] .. term_off[q+1]) (host arrays; term_off has nq+1 entries), its
 * results out_scores / out_ids [q*k .. q*k + n_out[q]).  Each query is scored exactly as by
 * ts_bm25_search (same kernels, same order of additions per query); the queries of a batch
 * run side by side — one launch per token position for up to 64 of them, each with its own
 * accumulator (20 bytes per document and query, at most 4 GiB per handle) — instead of one
 * chain of launches per query.                                                           */
int ts_bm25_search_batch(ts_bm25* h, const int32_t* term_ids, const int64_t* term_off, int32_t nq,
                         int32_t k, double* out_scores, int64_t* out_ids, int32_t* n_out,
                         void* stream);

/* ---- fused residual add + LayerNorm (between the GEMMs of the encoder forwards) ---
 * The cross-encoder forward the reference reaches through CrossEncoder.predict
 * (src/stage3_reranker.py:127-131) is PyTorch-ROCm GEMMs and attention here too; the
 * residual add, LayerNorm and the cast for the next GEMM between them are one pass:
 *   y = LayerNorm(x + residual) * gamma + beta  over the last dimension H (fp32
 *   statistics and arithmetic, like torch.layer_norm);
 * x [rows, H] of x_dtype; residual fp32 [rows, H] or NULL; gamma / beta fp32 [H];
 * y is written as fp32 (out_f32, may be NULL) and / or in lp_dtype (out_lp, TS_F16 or
 * TS_BF16, may be NULL).  beta may be NULL (a LayerNorm without bias).  H a multiple of
 * 4, <= 2048; pointers 16-byte aligned (device).                                    */
int ts_add_layernorm(const void* x, int32_t x_dtype, const float* residual, const float* gamma,
                     const float* beta, float eps, int64_t rows, int32_t H, float* out_f32,
                     void* out_lp, int32_t lp_dtype, int32_t device, void* stream);

/* The same pass for a pre-LN model (ModernBERT, the reference's default stage-2 token
 * encoder, src/stage2_rescorer.py:30): out_sum (fp32, may be NULL) receives x + residual
 * — the residual stream — and out_lp the normalised row for the next GEMM.           */
int ts_add_prenorm(const void* x, int32_t x_dtype, const float* residual, const float* gamma,
                   const float* beta, float eps, int64_t rows, int32_t H, float* out_sum,
                   void* out_lp, int32_t lp_dtype, int32_t device, void* stream);

/* The embedding layer of those models the same way: row r of the output is
 *   LayerNorm((word[ids[r]] + type[type_ids[r]]) + position[pos_ids[r]]) * gamma + beta
 * (the order of additions of BertEmbeddings / RobertaEmbeddings); ids / pos_ids /
 * type_ids int64 [rows] on the device (type_ids NULL = type 0), tables fp32 [*, H]
 * — indices are NOT range checked; outputs as for ts_add_layernorm.                */
int ts_embed_layernorm(const int64_t* ids, const int64_t* pos_ids, const int64_t* type_ids,
                       const float* word_tab, const float* pos_tab, const float* typ_tab,
                       const float* gamma, const float* beta, float eps, int64_t rows, int32_t H,
                       float* out_f32, void* out_lp, int32_t lp_dtype, int32_t device, void* stream);

/* ---- self-attention of a right-padded batch (the attention of those forwards) -----
 * softmax(Q K^T * scale) V per head over the first lens[b] tokens of sequence b only —
 * what torch's scaled_dot_product_attention computes under the padding mask of
 * CrossEncoder.predict's tokenizer batch (src/stage3_reranker.py:127-131), without
 * a mask tensor and without work on the padding.  qkv [B, L, 3, heads, dh] of dtype
 * (TS_F16 / TS_BF16): the output of the fused Q/K/V projection, read in place;
 * lens int32 [B] (device; clamped to L; 0 = nothing written for that sequence);
 * out [B, L, heads*dh] of dtype — rows at padded positions are NOT written.  fp32
 * softmax statistics and accumulation, probabilities rounded to dtype before the
 * P V product (as flash attention does).  dh 32 or 64; K and V^T of one sequence and
 * head must fit the 160 KB of LDS: L <= 1120 at dh 32, L <= 576 at dh 64;
 * pointers 16-byte aligned, B <= 65535.  window > 0: query q only sees the keys k with
 * |q - k| <= window (the bidirectional sliding window of ModernBERT's local layers:
 * window = local_attention / 2); 0 = all keys.  rope_cos / rope_sin (fp32 [L, dh], both or
 * neither): the rotary embedding of ts_rope_inplace applied to q and k as they are
 * loaded — same arithmetic, same results as ts_rope_inplace followed by this call
 * without tables, minus one pass over q and k (qkv itself is left as it is).
 * offs (int32 [B] on the device, or NULL): a PACKED batch — qkv is [T, 3, heads, dh] and out
 * [T, heads*dh] with sequence b occupying tokens offs[b] .. offs[b] + lens[b]; L is then
 * only the upper bound of lens (it sizes the LDS tiles).  NULL: the padded layout above.  */
int ts_attention_varlen(const void* qkv, const int32_t* lens, int32_t B, int32_t L, int32_t heads,
                        int32_t dh, int32_t dtype, float scale, int32_t window,
                        const float* rope_cos, const float* rope_sin, const int32_t* offs, void* out,
                        int32_t device, void* stream);

/* Rotary position embedding of the q and k thirds of qkv [B, L, 3, heads, dh] in place:
 * x <- x * cos + rotate_half(x) * sin with fp32 tables cos / sin [L, dh] (row = token
 * position), computed in fp32 like transformers' apply_rotary_pos_emb (products and sum
 * each rounded, no fused multiply-add), rounded once to dtype.  dh a multiple of 8.   */
int ts_rope_inplace(void* qkv, int32_t dtype, const float* cos_tab, const float* sin_tab, int64_t B,
                    int32_t L, int32_t heads, int32_t dh, int32_t device, void* stream);

/* Gated GELU of ModernBertMLP: u [rows, 2 I] -> out [rows, I] = gelu(u[:, :I]) * u[:, I:]
 * (erf GELU in fp32 rounded to dtype, then the product rounded to dtype: the two
 * roundings of the two torch ops).  I a multiple of 8.                               */
int ts_geglu(const void* u, int32_t dtype, int64_t rows, int32_t I, void* out, int32_t device,
             void* stream);

/* ---- linear layers with a short reduction dimension (the projections of those forwards) ---
 * out[M, N] = act(x[M, K] w[N, K]^T + bias[N]) for the Q/K/V, attention-output and feed-
 * forward "up" projections of MiniLM-class encoders (K <= 384 is where it beats the
 * library GEMM, 1.2-1.5x; at K = 768 hipBLASLt's stream-K kernels win and the Python
 * host keeps them; any K that is a multiple of 128 up to 2176 is accepted), act 0 = none,
 * 1 = erf GELU (BertIntermediate: no separate activation pass over the M x N result).
 * The weight is re-tiled ONCE with ts_linear_tile_weight (w [N, K] in torch.nn.Linear
 * layout -> out, N*K elements of the same dtype) and then streamed from L2 by every
 * workgroup while the workgroup's rows of x sit in LDS (the structure of the stage-1 scan,
 * DESIGN.md 4.7).  x, bias (may be NULL), out of dtype (TS_F16 / TS_BF16), fp32 accumulation;
 * sum + bias is rounded to dtype before the activation, the result rounded again — the
 * roundings of linear followed by gelu.  N a multiple of 32; pointers 16-byte aligned
 * (bias 8).                                                                              */
int ts_linear_tile_weight(const void* w, int32_t dtype, int32_t N, int32_t K, void* out,
                          int32_t device, void* stream);
int ts_linear_act(const void* w_tiled, const void* x, const void* bias, int32_t dtype, int64_t M,
                  int32_t N, int32_t K, int32_t act, void* out, int32_t device, void* stream);

/* BertSelfOutput / BertOutput of a post-LN encoder in ONE kernel (the cross-encoder the reference reaches through
 * CrossEncoder.predict, /root/reference/src/stage3_reranker.py:127-131; transformers' modeling_bert
 * BertSelfOutput.forward / BertOutput.forward: dense -> dropout -> LayerNorm(hidden + input)):
 *     y = LayerNorm(round_dtype(a(x)[M, K] w[N, K]^T + bias[N]) + residual[M, N]) * gamma + beta
 * w_tiled from ts_linear_tile_weight; x, bias (may be NULL) of dtype (TS_F16 / TS_BF16); residual fp32 (may be
 * NULL), gamma fp32 [N], beta fp32 [N] or NULL; y is written as fp32 (out_f32, the next residual) and / or in
 * dtype (out_lp, the next GEMM's input) — at least one of them.  act_in 0: a(x) = x; 1: a(x) = round_dtype(
 * erf GELU(x)) applied as the rows are staged — BertIntermediate's activation folded into BertOutput, x being
 * the up projection's output BEFORE its activation (ts_linear_act with act 0): the activation costs no pass
 * of its own and its arithmetic runs beside this kernel's matrix instructions.  The roundings are those of
 * (gelu,) ts_linear_act, ts_add_layernorm, and for N > 128 so are the bits (below, the fp32 row statistics may
 * differ in the last place: 1e-6).  N a multiple of 32 up to 384 (a workgroup owns
 * whole rows: the projection's output never goes to HBM), K a multiple of 384; pointers 16-byte aligned (bias 8). */
int ts_linear_add_layernorm(const void* w_tiled, const void* x, const void* bias, const float* residual,
                            const float* gamma, const float* beta, float eps, int32_t dtype, int64_t M,
                            int32_t N, int32_t K, int32_t act_in, float* out_f32, void* out_lp, int32_t device,
                            void* stream);

/* BertIntermediate + BertOutput — the whole feed-forward block of a post-LN encoder layer — in ONE kernel:
 *     y = LayerNorm(round(gelu(round(x[M, H] w1[I, H]^T + b1[I])) w2[H, I]^T + b2[H]) + residual[M, H]) * gamma + beta
 * (erf GELU; round = to dtype).  The M x I intermediate never leaves the CU (as separate kernels it is written to HBM
 * and read back: 40 % of a layer's traffic).  w1_tiled / w2_tiled from ts_linear_tile_weight; x, b1, b2 (may be
 * NULL) of dtype (TS_F16 / TS_BF16); residual fp32 (may be NULL), gamma fp32 [H], beta fp32 [H] or NULL; outputs as
 * for ts_linear_add_layernorm.  The roundings and the accumulation order are those of ts_linear_act (act 1) followed
 * by ts_linear_add_layernorm: the same bits.  H = 384 (MiniLM-class: a workgroup owns whole rows), I a multiple of
 * 384; pointers 16-byte aligned (biases 8).                                                                     */
int ts_mlp_add_layernorm(const void* w1_tiled, const void* b1, const void* w2_tiled, const void* b2, const void* x,
                         const float* residual, const float* gamma, const float* beta, float eps, int32_t dtype,
                         int64_t M, int32_t H, int32_t I, float* out_f32, void* out_lp, int32_t device, void* stream);

/* Frees the internal MaxSim scratch buffers kept per (device, stream) (all devices
 * if device < 0).  No MaxSim launch may be pending on that device.               */
int ts_maxsim_release_scratch(int32_t device);

/* ---- diagnostics (no GPU needed) -------------------------------------------
 * Exercises the per-device one-time table that guards hipFuncSetAttribute with
 * n_threads racing host threads over n_devices device numbers; 0 = every
 * (kernel, device) action ran exactly once, failed actions were retried.      */
int ts_selftest_device_once(int32_t n_threads, int32_t n_devices);

/* ---- misc ---------------------------------------------------------------- */
const char* ts_last_error(void);
int ts_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TRISTAGE_H_ */
