/*
 * mi355dr.h -- C ABI of the MI355X-native dense-retrieval core (libmi355dr.so).
 *
 * This is the drop-in boundary for AutoRAG-Research's Vector Search hot path.  The reference
 * has no FFI of its own for this path: it dispatches two SQL operators to PostgreSQL
 * extensions.  Each entry point below cites the reference call it replaces
 * (paths relative to /root/reference):
 *
 *   mi355dr_search / _device      <->  BaseVectorRepository.vector_search_with_scores
 *                                      autorag_research/orm/repository/base.py:378-426
 *                                      (SELECT id, embedding <=> q AS distance ... ORDER BY distance LIMIT k)
 *   mi355dr_search_maxsim         <->  BaseVectorRepository.maxsim_search / maxsim_search_with_ids
 *                                      autorag_research/orm/repository/base.py:487-535, 537-571
 *                                      (embeddings @# ARRAY[...] AS distance ... ORDER BY distance LIMIT k)
 *   mi355dr_add_rows / _device    <->  the `embedding VECTOR(d)` column fill
 *                                      autorag_research/orm/service/base_ingestion.py:199-247
 *   mi355dr_add_multivec          <->  BaseVectorRepository.set_multi_vector_embedding(s_batch)
 *                                      autorag_research/orm/repository/base.py:428-485
 *
 * Semantics (identical to oracle/oracle.c, which is the checker):
 *   - cosine distance = pgvector cosine_distance: fp32 accumulators dot/|q|^2/|c|^2 (k-ascending
 *     fused-multiply-add chains), double sim = dot/sqrt(nq*nc) clamped to [-1,1], distance = 1-sim
 *     returned as double (float8), NaN when a norm is zero.
 *   - inner product distance = (double)dot * -1 (pgvector <#>).
 *   - MaxSim distance = sum over query vectors of min over doc vectors of (-dot), fp32 (VectorChord @#).
 *   - exact brute force (the reference never builds an ANN index), total order
 *     (distance asc, NaN last, row index asc).
 *   - rows are addressed by dense row index in insertion order; the caller owns the
 *     row-index -> chunk-id table (ids may be BIGINT or VARCHAR, schema_factory.py:63-76).
 *
 * Conventions: every function returns 0 on success or a negative MI355DR_E_* code; the text of the
 * last error is available from mi355dr_last_error().  The caller owns all host buffers; the
 * library owns device memory.  One search in flight per handle (internal mutex); independent
 * handles are independent.  No Python / torch types cross this boundary.
 */
#ifndef MI355DR_H
#define MI355DR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi355dr_index mi355dr_index; /* opaque */

enum {
    MI355DR_OK = 0,
    MI355DR_E_INVALID = -1,     /* bad argument */
    MI355DR_E_HIP = -2,         /* HIP runtime error (text in last_error) */
    MI355DR_E_NOMEM = -3,       /* device or host allocation failed */
    MI355DR_E_UNSUPPORTED = -4, /* valid request this build cannot serve */
    MI355DR_E_INTERNAL = -5
};

enum { MI355DR_METRIC_COSINE = 0, MI355DR_METRIC_IP = 1 };

/* Search strategies (option "path"). */
enum {
    MI355DR_PATH_AUTO = 0,   /* screen where it applies, else scan */
    MI355DR_PATH_SCREEN = 1, /* MFMA screen over the normalised shadow corpus + exact fp32 re-score (both metrics) */
    MI355DR_PATH_SCAN = 2    /* exact fp32 chain per (query,row); slow, guaranteed, also the in-library fallback */
};

/* Element type of the screen pass (option "screen_dtype").  Both are exact end to end: the screen only
 * decides which rows are re-scored, under a rigorous per-query bound on |screen value - exact cosine|. */
enum {
    MI355DR_SCREEN_AUTO = 0, /* int8 when the corpus quantises within the residual limit, else bf16 */
    MI355DR_SCREEN_BF16 = 1, /* v_mfma_f32_32x32x16_bf16 over the bf16 shadow (bound ~0.0082 at d=768) */
    MI355DR_SCREEN_I8 = 2    /* v_mfma_i32_32x32x32_i8 over the int8 shadow (bound ~0.023): half the bytes, twice the rate */
};

/* ---- lifetime ---- */
int mi355dr_create(mi355dr_index** out, int device_id, int dim, int metric);
void mi355dr_destroy(mi355dr_index* idx);
/* idx may be NULL: returns the text of the last error raised before a handle existed */
const char* mi355dr_last_error(const mi355dr_index* idx);
int mi355dr_version(void);

/* ---- corpus (single-vector) ---- */
int mi355dr_reserve(mi355dr_index* idx, int64_t n_rows);
/* rows: host, row-major [n, dim] fp32.  Appends; copies to HBM; precomputes |c|^2 and the bf16 / int8 shadows. */
int mi355dr_add_rows(mi355dr_index* idx, const float* rows, int64_t n);
/* same, rows already resident on this index's device (e.g. an embedding model's output tensor) */
int mi355dr_add_rows_device(mi355dr_index* idx, const float* rows_dev, int64_t n);
int64_t mi355dr_size(const mi355dr_index* idx);
int mi355dr_dim(const mi355dr_index* idx);
/* copy stored rows back (testing / cpu baseline): out host [n, dim] */
int mi355dr_get_rows(mi355dr_index* idx, int64_t row0, int64_t n, float* out);

/* ---- search (single-vector) ----
 * queries: [B, dim] fp32.  out_dist: [B, k] double, out_rows: [B, k] int64; slots beyond the
 * number of stored rows hold NaN / -1.  row_offset is added to every returned row (shards). */
int mi355dr_search(mi355dr_index* idx, const float* queries, int B, int k, double* out_dist, int64_t* out_rows);
/* device-resident queries and outputs; stream = hipStream_t (NULL: the index's own stream).
 * Returns after the work is complete on that stream (it checks the device-side status word). */
int mi355dr_search_device(mi355dr_index* idx, const float* queries_dev, int B, int k, double* out_dist_dev,
                          int64_t* out_rows_dev, void* stream);
/* The same in two halves, so that consecutive blocks follow each other on the GPU without the host's round trip (the
 * reference's caller issues its queries back to back: pipelines/retrieval/vector_search.py:157-169; here a page of query
 * blocks).  _async puts the whole search on `stream` and returns a ticket without synchronising; the outputs are valid
 * after mi355dr_search_wait(ticket) returned MI355DR_OK: it waits for the block(s), reads their status words and, in the
 * rare case that a query overflowed a candidate list or cannot be screened, recomputes those queries (blocking) into the
 * same output buffers.  queries_dev and the output buffers must stay valid and untouched until the wait.  Up to 4 blocks
 * of 1024 queries may be in flight (a fifth first completes the oldest); blocks must be issued on ONE stream (a block on
 * another stream first completes what is in flight).  mi355dr_search_wait(idx, t) completes every block up to ticket t;
 * the synchronous entry points complete whatever is in flight first. */
int mi355dr_search_device_async(mi355dr_index* idx, const float* queries_dev, int B, int k, double* out_dist_dev,
                                int64_t* out_rows_dev, void* stream, int64_t* ticket);
int mi355dr_search_wait(mi355dr_index* idx, int64_t ticket);

/* ---- corpus + search (multi-vector, MaxSim) ----
 * vecs: host [sum_T, dim] fp32, offsets: [n_docs+1] (doc i owns rows offsets[i]..offsets[i+1]). */
int mi355dr_add_multivec(mi355dr_index* idx, const float* vecs, const int64_t* offsets, int64_t n_docs);
/* the same from DEVICE memory (an encoder's output never leaves HBM: embeddings/colpali.py:168-245 `embed_image(s)` /
 * `embed_documents` -> [T,128] patch / token tensors): vecs_dev = device [sum_T, dim], offsets = HOST [n_docs+1] */
int mi355dr_add_multivec_device(mi355dr_index* idx, const float* vecs_dev, const int64_t* offsets, int64_t n_docs);
int64_t mi355dr_size_multivec(const mi355dr_index* idx);
/* qtok: host [sum_nq, dim], q_offsets: [B+1].  out_dist: [B,k] fp32 (= -sum_i max_j <q_i,d_j>). */
int mi355dr_search_maxsim(mi355dr_index* idx, const float* qtok, const int32_t* q_offsets, int B, int k,
                          float* out_dist, int64_t* out_rows);
/* The same with the query vectors and the results in DEVICE memory of the index's GPU (an encoder's output tensor in, the
 * packed block of a row-sharded search out: BaseVectorRepository.maxsim_search behind a multi-GPU caller,
 * orm/repository/base.py:487-535).  qtok_dev: device [sum_nq, dim]; q_offsets: HOST [B+1]; out_dist_dev [B,k] fp32 and
 * out_rows_dev [B,k] int64 on the device, complete on return.  `stream`: the stream that produced qtok_dev (waited for
 * before the vectors are read; NULL: none).  The query side of a pass is small (a few hundred KiB) and its screen bound is
 * evaluated in double on the host, so the vectors are read back once; the results never leave HBM. */
int mi355dr_search_maxsim_device(mi355dr_index* idx, const float* qtok_dev, const int32_t* q_offsets, int B, int k,
                                 float* out_dist_dev, int64_t* out_rows_dev, void* stream);

/* exact MaxSim distance of every query to an explicit list of docs (candidate re-scoring: HEAVEN stage 2,
 * autorag_research/pipelines/retrieval/heaven.py:244-266 `_score_candidates`, score = -distance / n_q; GQR pools,
 * gqr_hybrid.py:366-406).  doc_ids: host [B, m] global rows (as returned by the searches; other values are skipped),
 * out_dist: host [B, m] fp32, NaN for skipped ids, docs without vectors and queries without vectors. */
int mi355dr_maxsim_subset(mi355dr_index* idx, const float* qtok, const int32_t* q_offsets, int B, const int64_t* doc_ids,
                          int m, float* out_dist);
/* The same with flags.  MI355DR_MAXSIM_CLAMP0: every query vector contributes max(0, max_j <q_i, d_j>) -- the ColBERT
 * reranker's MaxSim (autorag_research/rerankers/colbert.py:63-84: padding masked, `clamp(min=0)`, mean over the VALID query
 * tokens): pass only the valid tokens of query and documents; score = -distance / n_valid_query_tokens. */
#define MI355DR_MAXSIM_CLAMP0 1
int mi355dr_maxsim_subset_ex(mi355dr_index* idx, const float* qtok, const int32_t* q_offsets, int B, const int64_t* doc_ids,
                             int m, int flags, float* out_dist);

/* ---- Guided Query Refinement of candidate pools (GQR hybrid pipeline) ----
 * Replaces the per-query numpy loops of autorag_research/pipelines/retrieval/gqr_hybrid.py: `_optimize_query_embedding`
 * (:321-340), `_optimize_query_multi_embedding` (:342-362), `_optimize_in_score_space` (:306-319).  Float64 throughout
 * (the reference's arithmetic type); a block of B queries is one launch, one workgroup per query; the candidate vectors
 * are the rows already resident in HBM, named by global row id.  Pools: host [B, P], live ids first, -1 padding after
 * (P <= 2048); comp_dist: host [B, P] complementary distribution (gqr_hybrid.py:436); out_scores: host [B, P], NaN at
 * padding.  n_steps > 0, learning_rate > 0, temperature > 0, 0 <= mixture_alpha <= 1 (:202-216), else MI355DR_E_INVALID.
 *   mi355dr_gqr_refine         queries: host [B, dim] float64; candidates = single-vector rows; out = refined cosine
 *   mi355dr_gqr_refine_maxsim  qtok: host [sum_nq, dim] float64, q_offsets [B+1] (every query >= 1 vector); candidates =
 *                              multi-vector docs (each must have vectors); out = refined mean-of-max late-interaction score
 *   mi355dr_gqr_refine_scores  no vectors: primary_scores host [B, P] float64 are the variables, counts[b] live entries */
int mi355dr_gqr_refine(mi355dr_index* idx, const double* queries, int B, const int64_t* cand_rows, int P,
                       const double* comp_dist, int n_steps, double learning_rate, double temperature,
                       double mixture_alpha, double* out_scores);
int mi355dr_gqr_refine_maxsim(mi355dr_index* idx, const double* qtok, const int32_t* q_offsets, int B,
                              const int64_t* doc_ids, int P, const double* comp_dist, int n_steps, double learning_rate,
                              double temperature, double mixture_alpha, double* out_scores);
int mi355dr_gqr_refine_scores(mi355dr_index* idx, const double* primary_scores, const int32_t* counts, int B, int P,
                              const double* comp_dist, int n_steps, double learning_rate, double temperature,
                              double mixture_alpha, double* out_scores);

/* ---- shard merge (multi-GPU): [world, B, k] gathered (dist,row) device buffers -> [B, k] ----
 * The merge functions only enqueue work on `stream` (NULL: the index's stream); synchronise that stream (or call
 * mi355dr_synchronize for the index stream) before reading the outputs on the host.  The index's stream is NON-BLOCKING: the
 * legacy default stream (handle 0 -- what NULL means here, and what a framework's "current stream" often is) is ordered
 * against it by nothing.  A caller whose inputs were produced on the default stream either passes a real stream of its own
 * (the merge then runs ON it, behind its producers and in front of its consumers) or synchronises on both sides itself. */
int mi355dr_merge_topk_device(mi355dr_index* idx, const double* dist_all_dev, const int64_t* rows_all_dev, int world,
                              int B, int k, double* out_dist_dev, int64_t* out_rows_dev, void* stream);

/* same merge on the PACKED layout one rank produces for a single all-gather: int64 [world][2][B][k], plane 0 = the
 * float8 distance bit patterns, plane 1 = rows.  mi355dr_pack_topk_device builds one rank's [2][B][k] block. */
int mi355dr_pack_topk_device(mi355dr_index* idx, const double* dist_dev, const int64_t* rows_dev, int B, int k,
                             int64_t* packed_dev, void* stream);
int mi355dr_merge_topk_packed_device(mi355dr_index* idx, const int64_t* packed_all_dev, int world, int B, int k,
                                     double* out_dist_dev, int64_t* out_rows_dev, void* stream);

/* ---- row-sharded search inside the library: one index = one shard on one GPU, one process per GPU.
 * Replaces nothing in the reference (its engine is one PostgreSQL server); it is the data-parallel form of
 * `ORDER BY distance LIMIT k` (orm/repository/base.py:409-415): every rank searches its rows with "row_offset" set, ONE
 * ncclAllGather (RCCL over xGMI) of the packed [2,B,k] (float8 distance bits, int64 global row) block per rank, then the
 * world*k -> k merge under the same total order: bit-identical to the single-GPU result on every rank.
 * RCCL is bound at run time (dlopen): a host that never calls these needs no librccl.  The 128-byte ncclUniqueId is
 * created on one rank (mi355dr_comm_unique_id) and handed to every rank through the host's own channel. */
int mi355dr_comm_unique_id(void* out_128_bytes, size_t len);
int mi355dr_comm_init(mi355dr_index* idx, int rank, int world, const void* nccl_unique_id, size_t id_len);
int mi355dr_comm_world(const mi355dr_index* idx);  /* 0 before mi355dr_comm_init */
/* the rank count RCCL itself reports for the communicator (ncclCommCount): what a caller logs to show that the collective
 * really spans N ranks; 0 before mi355dr_comm_init */
int mi355dr_comm_count(mi355dr_index* idx, int* out);
/* Transport plug-in: the same sharded search over an all-gather the HOST provides instead of RCCL -- MPI / UCX hosts, a node
 * whose ranks cannot meet in one RCCL communicator (RCCL refuses two ranks on one device: the two-ranks-on-one-GPU parity test
 * of tests/test_gpu_world2.py runs mi355dr_search_sharded_device through this entry).  `fn` gathers `bytes_per_rank` bytes from
 * every rank's `send_dev` into `recv_dev` (rank-major, device memory), either ordered on `stream` or complete on return, and
 * returns 0 on success; it is called from the thread that calls mi355dr_search_sharded_device, once per block, in the same
 * order on every rank.  Replaces a communicator set by mi355dr_comm_init (and vice versa); mi355dr_comm_count then reports
 * `world` as given. */
typedef int (*mi355dr_allgather_fn)(const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream, void* user);
int mi355dr_comm_init_custom(mi355dr_index* idx, int rank, int world, mi355dr_allgather_fn fn, void* user);
/* device buffers in / out like mi355dr_search_device; every rank passes the same queries and receives the same result.
 * Blocks of 1024 queries are software-pipelined: block i + 1 is searched while block i's all-gather + merge run on the
 * index's communication stream (two packed / gathered buffers).  Asynchronous on `stream` (NULL = the index's own stream):
 * on return `stream` has been made to wait for the last merge. */
int mi355dr_search_sharded_device(mi355dr_index* idx, const float* queries_dev, int B, int k, double* out_dist_dev,
                                  int64_t* out_rows_dev, void* stream);

/* ---- options / stats / timing ----
 * options: "path" (MI355DR_PATH_*), "screen_dtype" (MI355DR_SCREEN_*), "row_offset", "profile" (0/1: HIP-event
 *          timing of the dominant kernel), "chunk0_rows", "chunk_growth", "cand_cap", "prefilter16" (1: int8 screen only, a bf16 second
 *          screen of the surviving candidates before the exact re-score; default 0, same results), "maxsim_screen" (1: bf16 MFMA screen
 *          over every doc + exact re-score of the candidates [default], 0: exact kernel over every doc; same results),
 *          "screen_stream" (1 [default]: query blocks of at most 64 are screened by the streaming kernel -- resident query
 *          block, ring of row stages --, 0: by the tile kernel; same results), "screen_rq" (1 [default]: query blocks of more
 *          than 128 over an int8 shadow of at most 768 bytes per row are screened by k_screen_rq -- query operand resident in
 *          registers, rows alone through the LDS --, 0: by k_screen256c; same results).
 *          Round 3, all with identical results (A/B switches of the pass schedule): "starter" (1 [default]: sampled threshold
 *          estimator instead of the three smallest chunks, k <= 32), "defer_round_b" (1 [default]: prunes before the last one
 *          carry their survivors over instead of re-scoring them), "prune_companion" (1 [default]: general-form prune launch
 *          behind every one-wave prune), "scan_dma" (1 [default]: k_scan32, LDS-DMA staging, for dims that are a multiple
 *          of 32), "maxsim_persistent" (0 [default]), "maxsim_coop" (exact MaxSim on candidate lists: -1 [default] one workgroup per
 *          candidate for stores of long documents, 0 one wave, 1 always one workgroup), "i8_min_budget_x100" (AUTO keeps the int8 screen while the chunk-growth
 *          budget at this k is at least value / 100; default 25 = k <= 133).
 *          Round 4, MaxSim, all with identical results (A/B switches of the 16-query pass, dims <= 128): "maxsim_pass_groups"
 *          (1..4 [default 4] groups of <= 4 queries per screen pass), "maxsim_wg" (screen form from 8/9 column blocks of query
 *          vectors up: -1 [default] workgroup form, per-document sums parked for short documents / at once for long ones,
 *          0 one wave per document, 1 parked, 2 at once), "maxsim_wg_bps" (4 [default] / 2 token blocks per ring stage),
 *          "maxsim_wg_pipe" (1 [default]: a block's maxima folded between the next block's MFMAs), "maxsim_wg_min" (8 [default]
 *          / 9: fewest column blocks that take the workgroup form), "maxsim_aligned" (1 [default]: queries that are exactly
 *          one 32-column block are summed by the wave that holds them), "maxsim_tighten" (1 [default]: candidate band from the
 *          exact distances of the screen's top-k); round 6: "prune_wide" (1 [default]: passes at 33 <= k <= 128 use the
 *          two-wave prune, a starter over "starter_rows_wide" [65536] rows and chunk ratios up to 4; 0 = the round-5 schedule),
 *          "screen_flush_sync" (1 [default]: the waves of a k_screen_rq workgroup flush their hit-lane queues at the same tiles;
 *          "screen_flush_lanes" [48] / "screen_flush_alone" [40] tune the period), "chunk_taper_x100" (0 [default] = 120 for
 *          prune_wide passes, 100 = uniform chunk ratios otherwise), "wide_inflation_x10" (the budget's inflation figure);
 *          "maxsim_pack8" (MaxSim screen, passes of 32-vector queries in the workgroup form and passes of up to four column blocks: a second bf16 shadow whose documents are
 *          rounded up to 8-token granules instead of 32-token blocks, built on the first such pass and extended by the next one after an add -- -1
 *          [default]: when it has at least 5 % fewer blocks than the padded copy and its memory is there, 1: always, 0: never;
 *          identical results).
 * stats:   "screen_launches", "screen_ns" (profile=1), "screen_rows" (all screen launches) and their k_screen256 share
 *          "screen256_launches", "screen256_ns", "screen256_rows"; "candidates", "rescored",
 *          "fallback_queries" (queries recomputed by the exact scan), "retry_queries" (queries whose candidate list
 *          overflowed and that were re-screened with the bf16 bound and slower chunk growth first), "i8_demoted" (AUTO
 *          gave up the int8 screen for this index after > 1 % of a block overflowed -- from "i8_demoted_k", the k of that
 *          block, upwards; smaller k keep int8), "starters", "chunks", "passes", "irregular_rows", "loose_rows" (rows outside the int8 shadow,
 *          irregular ones included), "screen_dtype_active" (MI355DR_SCREEN_BF16 / _I8: what AUTO resolves to now),
 *          "maxsim_screened" (queries served by the MaxSim screen), "maxsim_candidates" (docs re-scored exactly for them),
 *          "maxsim_fallbacks" (queries re-run by the exact full scan), "maxsim_screen_launches" / "maxsim_screen_ns" /
 *          "maxsim_exact_launches" / "maxsim_exact_ns" (profile=1), "maxsim_packed_launches" / "maxsim_packed_blocks" / "maxsim_packed_built" (screen launches over the
 *          granule-packed copy / its 32-token blocks / blocks written into it so far: a store that grows is packed from its new granules on), "maxsim_screen_cols" (query columns the screen launches
 *          multiplied every token by),
 *          "hbm_bytes_resident". */
int mi355dr_set_option(mi355dr_index* idx, const char* key, int64_t value);
int mi355dr_get_stat(mi355dr_index* idx, const char* key, int64_t* out);
int mi355dr_reset_stats(mi355dr_index* idx);
/* HIP-event stopwatch on the index's stream (bench.py: kernel-side time of a timed region) */
int mi355dr_timer_start(mi355dr_index* idx);
int mi355dr_timer_stop(mi355dr_index* idx, double* elapsed_ms);
int mi355dr_synchronize(mi355dr_index* idx);

/* ---- raw device memory helpers for hosts that have no device allocator of their own (tests, bench) ---- */
int mi355dr_dev_alloc(mi355dr_index* idx, size_t bytes, void** out);
int mi355dr_dev_free(mi355dr_index* idx, void* p);
int mi355dr_dev_upload(mi355dr_index* idx, void* dst_dev, const void* src_host, size_t bytes);
int mi355dr_dev_download(mi355dr_index* idx, void* dst_host, const void* src_dev, size_t bytes);

/* ---- test hooks (used by tests/ only; exercise the production kernels on small inputs) ----
 * dense screen values t[b, r] for rows [row0,row0+n): runs the screen kernel with thresholds at -inf. */
int mi355dr_debug_screen_dense(mi355dr_index* idx, const float* queries, int B, int64_t row0, int64_t n, float* out_t);
/* the per-query screen bound E of the active screen dtype: exact cosine <= screen value + E  (bf16 screen: also
 * |screen value - exact cosine| <= E; int8 screen: the screen value already carries its row group's share of the bound) */
int mi355dr_debug_screen_bound(mi355dr_index* idx, const float* queries, int B, float* out_E);
/* int8 screen: per query the step S_q and the factor kq = 1.0001 + 3 e_q; per group of 32 rows [g0, g0 + n_groups) the
 * step S_g and the measured residual norm e_g.  Screen value of (query, row) = S_q S_g (q8 . c8) + e_g kq. */
int mi355dr_debug_i8_state(mi355dr_index* idx, const float* queries, int B, float* out_sq, float* out_kq, int64_t g0,
                           int64_t n_groups, float* out_step, float* out_err);
/* exact fp32 chain + distance for explicit (query,row) pairs, computed by the re-score device code */
int mi355dr_debug_rescore(mi355dr_index* idx, const float* queries, int B, const int32_t* pair_q,
                          const int64_t* pair_row, int64_t n_pairs, float* out_dot, double* out_dist);

/* ---- measurement support (bench.py; SURVEY 8(d): "re-measure with ... an MFMA microbench on the box and use the measured
 * peaks in the report") -- no reference counterpart.  A BARE stream of the MFMA instruction a screen kernel issues (format 0:
 * v_mfma_i32_32x32x32_i8 on Gaussian int8 operands like the int8 shadow's; 1: v_mfma_f32_32x32x16_bf16), operands in registers,
 * every CU busy, no memory traffic, for `seconds`; *out_tops = the settled rate in 10^12 operations per second.  Needs no
 * index; allocates and frees its own 4 MiB. */
int mi355dr_diag_mfma_stream(int device, int format, double seconds, double* out_tops);

#ifdef __cplusplus
}
#endif
#endif /* MI355DR_H */
