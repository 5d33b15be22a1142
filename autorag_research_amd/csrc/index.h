// index.h -- the opaque handle behind mi355dr_index and the host-side error helpers (internal).
#pragma once
#include <climits>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "dev_common.h"
#include "mi355dr.h"

namespace mi355 {
struct EventPair {
    hipEvent_t a, b;
    int big;  // 1: the launch went to k_screen256 (the dominant kernel), 0: k_screen
};
struct MultiVecStore;  // mi355dr_maxsim.hip
// one block between enqueue_block() and complete_block() (mi355dr.hip)
struct Pending {
    bool active = false;
    hipEvent_t done = nullptr;
    int* status_host = nullptr;   // pinned [kQBlockMax + 1]: the block's per-query status words + their OR
    hipStream_t stream = nullptr;
    const float* q_dev = nullptr; // the caller's queries (valid until the wait: a fix-up gathers from them)
    int B = 0, k = 0, level = 0;
    double* out_dist = nullptr;
    int64_t* out_rows = nullptr;
    bool used_screen = false, was_i8 = false;
};
constexpr int kPendingRing = 4;
}  // namespace mi355

struct mi355dr_index {
    int device = 0, dim = 0, dpad = 0, metric = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::string err;

    // corpus (single-vector)
    int64_t n = 0, cap_rows = 0;
    float* rows = nullptr;
    uint16_t* shadow = nullptr;
    float* nrm2 = nullptr;
    int32_t* irr_rows = nullptr;
    int* irr_count = nullptr;
    int irr_n = 0;
    unsigned* n2max_dev = nullptr;  // largest regular |c|^2 (float bits)
    float cmax = 0.0f;              // its square root, inflated: inner-product thresholds
    unsigned* bf16_res2_dev = nullptr;  // largest squared residual norm |c_hat - bf16(c_hat)|^2 over the rows (float bits)
    float bf16_ec = 0.00390625f;        // its square root, inflated: the corpus half of the bf16 screen bound
    // int8 screen: second shadow, one step for the whole corpus; rows it cannot hold are flagged and listed
    int dpad8 = 0;
    int8_t* shadow8 = nullptr;     // [cap_rows, dpad8]
    uint8_t* flag8 = nullptr;      // [cap_rows]
    mi355::I8Group* grp8 = nullptr;  // [cap_rows / 32] int8 step + residual norm per group of 32 rows
    int32_t* irr8_rows = nullptr;  // [kIrrCap] irregular + loose rows
    int* irr8_count = nullptr;
    int irr8_n = 0;

    // per-search state (sized for one block of kQBlockMax queries)
    bool qstate_ready = false;
    mi355::QueryState st{};
    float* qdev = nullptr;       // [kQBlockMax, dim]
    int32_t* cand_row = nullptr; // [kQBlockMax, cap]
    float* cand_val = nullptr;
    int* qlist_dev = nullptr;    // [kQBlockMax]
    int* status_or_dev = nullptr;
    int* status_host = nullptr;  // pinned [kQBlockMax + 1]
    double* out_dist_dev = nullptr;  // [kQBlockMax, kKMax]
    int64_t* out_rows_dev = nullptr;
    unsigned long long* stat_dev = nullptr;  // [2*kQBlockMax]: per query (candidates, re-scored)
    int* prune_skip = nullptr;   // [2 + 2*kQBlockMax] hand-over lists of k_prune (PruneArgs::skip_list)
    int prune_parity = 0;

    // options
    int path = 0;  // MI355DR_PATH_AUTO
    int screen_dtype = 0;  // MI355DR_SCREEN_AUTO
    int retry_level = 0;   // > 0 while overflowed queries are re-screened: bf16 bound, slower chunk growth
    int i8_backoff = 0, i8_probation = 0;  // demotion is not for ever: after i8_probation more blocks at such a k AUTO tries int8 again;
                                           // the wait doubles (16 ... 4096 blocks) every time that try overflows again
    int i8_demoted_k = INT_MAX;  // AUTO saw the int8 bound overflow on this corpus' score distribution at this k: bf16 from there up
    double i8_min_budget = 0.25;  // AUTO keeps the int8 screen while growth_budget(k, int8) stays above this (k <= 133)
    // buffers of the sub-block a search at retry level L re-screens (one set per level: the nested call owns the next)
    float* retry_q[3] = {nullptr, nullptr, nullptr};      // [kQBlockMax, dim] queries being re-screened / re-scanned
    double* retry_dist[3] = {nullptr, nullptr, nullptr};  // [kQBlockMax, kKMax]
    int64_t* retry_rows[3] = {nullptr, nullptr, nullptr};
    int* retry_map[3] = {nullptr, nullptr, nullptr};      // [kQBlockMax] position of each sub-block query in its block
    // blocks in flight: sequence numbers [seq_done, seq_next) live in pend[seq % kPendingRing]; sub_pend[level]: the
    // synchronous blocks of the fix-up levels
    mi355::Pending pend[mi355::kPendingRing];
    mi355::Pending sub_pend[4];
    int64_t seq_next = 0, seq_done = 0;
    int k_now = 10;        // k of the search in progress (the screen element type and the chunk growth depend on it)
    int maxsim_coop = -1;       // exact MaxSim on candidate lists: one workgroup per candidate (1), one wave (0), by document length (-1)
    int maxsim_persistent = 0;  // MaxSim screen (dims <= 128): persistent workgroups walking the docs in rounds.  A/B on one box
                                // (interleaved A/B through this option, 1 M text docs / 100 k pages): text +-0, pages 4 % SLOWER -- off
    // MaxSim screen, 9..16 column blocks: the workgroup-cooperative form (k_maxsim_wg.h).  -1 (default): yes, per-document epilogue
    // by document length; 1: parked epilogue; 2: immediate epilogue; 0: one wave per document with the query fragments in LDS
    // (k_maxsim16_d128<NCB, 8>) -- option "maxsim_wg", A/B and tests
    int maxsim_wg = -1;
    int maxsim_tighten = 1;  // MaxSim fast path: narrow the candidate band with the exact distances of the screen's top-k (0: band 2E)
    int maxsim_aligned = 1;  // k_maxsim16_wg: when every query of a pass is one column block, a wave sums its own two queries (0: A/B)
    int maxsim_wg_min = 8;   // fewest column blocks of a pass that take the workgroup form (8: short documents only; 9)
    int maxsim_wg_pipe = 1;  // k_maxsim16_wg, 4 blocks per stage: fold block j under the MFMAs of block j + 1 (0: the unpipelined form, A/B)
    // MaxSim screen, aligned passes that take the workgroup form: the granule-packed bf16 copy (k_maxsim_wg8.h; a second shadow,
    // built on first use).  -1 (default): when it removes at least 5 % of the padded copy's blocks and its memory is there; 1: always; 0: never
    int maxsim_pack8 = -1;
    int maxsim_wg_bps = 4;  // k_maxsim16_wg: 32-token blocks per ring stage (2: 7 stages of 16 KiB, 4: 4 stages of 32 KiB); option, A/B
    int maxsim_pass_groups = 4;  // groups of <= 4 queries one pass of the MaxSim screen serves (1 .. 4; option "maxsim_pass_groups", A/B and tests)
    int maxsim_screen = 1; // 1: bf16 MFMA screen + exact re-score of the candidates, 0: exact kernel over every doc
    int64_t row_offset = 0;
    int round_a = 0;      // k_prune: rows re-scored before the cut is known (0 = max(32, 2k)); tuning option "round_a"
    int screen_rq = 1;      // query blocks above 128, int8 shadow of <= 768 B per row: k_screen_rq (query operand in registers) instead of k_screen256c (option "screen_rq")
    int screen_stream = 1;  // query blocks of at most 64: k_screen_stream instead of k_screen (option "screen_stream", A/B and tests)
    int prefilter16 = 0;  // int8 screen: bf16 second screen of the surviving candidates inside k_prune (option "prefilter16";
                          // off: measured +1.4 % at 1.25 M rows, +0.2 % at 10 M -- the prune is bound by batch latency, not bytes)
    int profile = 0;
    int starter = 1;          // pass schedule: sampled threshold estimator instead of the smallest chunks (option "starter", A/B and tests)
    // 1 (default): the general-form prune is launched behind every one-wave prune (4 us when it finds nothing to do); 0: the
    // one-wave form alone, what it cannot hold (> 1024 entries) is flagged and re-screened -- measured at N = 10 M, k = 10: a few
    // queries per block exceed the 1024 entries in some chunk, and their re-screen pass costs 0.25 ms per block (0.85 ms with
    // carried survivors in the lists), against 20 us of companion launches
    int prune_companion = 1;
    int scan_dma = 1;         // exact scan: the LDS-DMA form (k_scan32) when dim % 32 == 0 (option "scan_dma", A/B and tests)
    int defer_round_b = 1;    // k_prune before the pass's last one carries the survivors of its cut over instead of re-scoring them
    int chunk0_set = 0;       // the first chunk's size was set by the caller: emit-all ladder, no starter
    int64_t chunk0_rows = 1024;
    int64_t chunk_growth = 3;
    int chunk_growth_set = 0;  // the option was set by the caller: no small-block override
    int64_t small_chunk_rows = 16384;  // chunks up to this many rows go through the 128x128 kernel (dense hits: per-lane appends)
    int cap = mi355::kCandCap;
    int cap_set = 0;          // option "cand_cap" was set by the caller: every pass uses it (else wide-prune passes use kCandCapWide)
    // 33 <= k <= 128: the two-wave prune (k_prune_wide.h: 4096 entries, round A of up to 128 rows, no companion launch), the
    // starter over a 64 k-row sample and chunk ratios up to 4 (option "prune_wide"; 0 = the round-5 schedule, A/B and tests)
    int prune_wide = 1;
    // k_screen_rq: the waves of a workgroup flush their hit-lane queues at the same tiles, every `period` tiles with
    // period = the largest power of two below screen_flush_lanes / (expected hit lanes per wave and tile) (1 ... 64);
    // option "screen_flush_sync" (0 = every wave on its own, round 5), "screen_flush_lanes" (tuning)
    int screen_flush_sync = 1, screen_flush_lanes = 48, screen_flush_alone = 40;
    int wide_inflation_x10 = 100;  // growth budget of the two-wave prune's passes: inflation of the int8 bound it plans for, x 10 (tuning)
    int flush_mask_now = -1;     // ... of the chunk being launched (run_screen)
    // pass schedule: ratio of chunk i = (geometric mean) x taper^((n-1)/2 - i) -- early chunks (cheap hits, loose bound) larger,
    // late chunks (expensive hits) smaller; 100 = uniform ratios (option "chunk_taper_x100")
    int chunk_taper_x100 = 0;   // 0 = auto: 120 for passes of the two-wave prune (measured -1 % at k = 100), 100 otherwise (k = 10: no gain)
    int64_t starter_rows_wide = 65536;  // the starter's sample at 33 <= k <= 128 (option "starter_rows_wide", tuning: 4096 ... 262144)

    // stats
    int64_t s_screen_launches = 0, s_screen_ns = 0, s_screen_rows = 0, s_fallback_queries = 0, s_chunks = 0,
            s_passes = 0, s_candidates = 0, s_rescored = 0, s_starters = 0;
    int64_t s_big_launches = 0, s_big_ns = 0, s_big_rows = 0;  // the k_screen256 share of the three above
    int screen_rq_split_tests = 1;  // k_screen_rq: a block test's maxima ride the MFMAs of the other row half (0: in one piece; d = 768 only, A/B; option "screen_rq_split_tests")
    int64_t debug_park = 0;     // diagnostic (option "debug_park_thresholds" = rows): k_screen_rq launches of at least that many rows run with every threshold at +inf -- what such a launch costs without hits; results are wrong while it is set
    float* park_thr = nullptr;
    int screen_drift = 3;   // k_screen_rq: tiles a workgroup may run ahead of its slowest sibling (0 = no limiter; option "screen_drift")
    int* rq_progress = nullptr;  // [kRqProgressWords] the limiter's progress words
    int rq_epoch = 0;            // launch stamp of the last k_screen_rq launch (1 ... 4095)
    int64_t s_rq_launches = 0;  // of them: k_screen_rq launches (stat "screen_rq_launches")
    int64_t s_retry_queries = 0;  // queries whose candidate list overflowed and that were re-screened with the bf16 bound
    int64_t s_ms_screened = 0, s_ms_candidates = 0, s_ms_fallbacks = 0;  // MaxSim: queries screened, docs re-scored, full re-runs
    // option "profile": HIP-event time of the MaxSim screen launches (k_maxsim16*) and of the exact launches on candidate lists
    int64_t s_ms_screen_ns = 0, s_ms_screen_launches = 0, s_ms_exact_ns = 0, s_ms_exact_launches = 0, s_ms_pack_ns = 0;
    int64_t s_ms_packed_launches = 0;  // screen launches that took the granule-packed copy (k_maxsim_wg8.h)
    int64_t s_ms_packed_blocks = 0;    // ... and the 32-token blocks of that copy (0: none built)
    int64_t s_ms_packed_built = 0;     // blocks k_ms_pack8 has written since the index was created (a store that grows is packed from its new granules on)
    int64_t s_ms_screen_cols = 0;  // query-vector columns (whole blocks of 32) the screen launches multiplied every token by
    hipEvent_t ms_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<mi355::EventPair> ev_pool, ev_pending;
    hipEvent_t t0 = nullptr, t1 = nullptr;

    // RCCL communicator of a row-sharded index (mi355dr_comm.hip)
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    mi355dr_allgather_fn comm_custom = nullptr;  // the host's own all-gather instead of RCCL (mi355dr_comm_init_custom)
    void* comm_custom_user = nullptr;
    // two of each: block i's all-gather + merge run on comm_stream under block i + 1's search (mi355dr_search_sharded_device)
    int64_t* comm_packed[2] = {nullptr, nullptr};      // [2, kQBlockMax, k] this rank's packed block
    int64_t* comm_packed_all[2] = {nullptr, nullptr};  // [world, 2, kQBlockMax, k]
    hipStream_t comm_stream = nullptr;
    hipEvent_t comm_done[2] = {nullptr, nullptr};      // gather + merge of the block that used buffer b have been enqueued / finished
    bool comm_done_armed[2] = {false, false};
    size_t comm_cap = 0;

    // multi-vector store (MaxSim), owned by mi355dr_maxsim.hip
    mi355::MultiVecStore* mv = nullptr;
};


namespace mi355 {

int fail(mi355dr_index* idx, int code, const std::string& msg);  // mi355dr.hip
void multivec_destroy(mi355dr_index* idx);                       // mi355dr_maxsim.hip
void comm_destroy(mi355dr_index* idx);                           // mi355dr_comm.hip
// read-only view of the multi-vector store for kernels outside mi355dr_maxsim.hip (GQR refinement)
struct MultiVecView {
    const float* tok;             // [blocks*32, dpad] device
    const int64_t* blk_off;       // [n_docs+1] device
    const int64_t* blk_off_host;  // same, host
    int dpad;
    int64_t n_docs;
};
bool multivec_view(const mi355dr_index* idx, MultiVecView* out);  // false: no store yet
int multivec_col_perm(int j);  // position j of a stored token row holds original column multivec_col_perm(j)

#define HIPCHECK(idx, expr)                                                                              \
    do {                                                                                                 \
        hipError_t e__ = (expr);                                                                         \
        if (e__ != hipSuccess) {                                                                         \
            char buf__[512];                                                                             \
            snprintf(buf__, sizeof(buf__), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                     __LINE__);                                                                          \
            return mi355::fail(idx, e__ == hipErrorOutOfMemory ? MI355DR_E_NOMEM : MI355DR_E_HIP, buf__); \
        }                                                                                                \
    } while (0)

#define CHECK(expr)                          \
    do {                                     \
        int rc__ = (expr);                   \
        if (rc__ != MI355DR_OK) return rc__; \
    } while (0)

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

}  // namespace mi355
