// mi355dr.hip -- host side of libmi355dr.so: the C ABI declared in include/mi355dr.h, device-memory
// management, and the launch schedule of the search (chunked screen -> prune, exact fallback).
// gfx950 only.  Build: see __graft_entry__.build().
#include "index.h"
#include "k_prep.h"
#include "k_scan.h"
#include "k_screen.h"
#include "k_screen256c.h"
#include "k_screen_rq.h"

// K-step counts k_screen_rq is built for (int8 shadow rows of 128 ... 768 bytes)
#define MI355_RQ_FORMS(X) X(1) X(2) X(3) X(4) X(5) X(6)
inline bool screen_rq_has(int ksteps) { return ksteps >= 1 && ksteps <= 6; }
#include "k_screen_stream.h"
#include "k_select.h"
#include "k_prune_wide.h"

using namespace mi355;

namespace {
std::mutex g_err_mu;
std::string g_err;  // errors raised before a handle exists
}  // namespace

namespace mi355 {
int fail(mi355dr_index* idx, int code, const std::string& msg) {
    if (idx) idx->err = msg;
    else {
        std::lock_guard<std::mutex> g(g_err_mu);
        g_err = msg;
    }
    return code;
}
}  // namespace mi355

namespace {


int ensure_capacity(mi355dr_index* idx, int64_t want_rows) {
    if (want_rows <= idx->cap_rows) return MI355DR_OK;
    int64_t new_cap = std::max<int64_t>(want_rows, idx->cap_rows + idx->cap_rows / 2);
    new_cap = round_up(std::max<int64_t>(new_cap, kT2), kT2);  // whole 256-row screen tiles
    float* rows = nullptr;
    uint16_t* shadow = nullptr;
    float* nrm2 = nullptr;
    int8_t* shadow8 = nullptr;
    uint8_t* flag8 = nullptr;
    I8Group* grp8 = nullptr;
    {   // all six or none: a failed allocation must not leak the ones before it
        hipError_t e = hipMalloc(&rows, (size_t)new_cap * idx->dim * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&shadow, (size_t)new_cap * idx->dpad * sizeof(uint16_t));
        if (e == hipSuccess) e = hipMalloc(&nrm2, (size_t)new_cap * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&shadow8, (size_t)new_cap * idx->dpad8);
        if (e == hipSuccess) e = hipMalloc(&flag8, (size_t)new_cap);
        if (e == hipSuccess) e = hipMalloc(&grp8, (size_t)(new_cap / kI8GroupRows) * sizeof(I8Group));
        if (e != hipSuccess) {
            for (void* p : {(void*)rows, (void*)shadow, (void*)nrm2, (void*)shadow8, (void*)flag8, (void*)grp8})
                if (p) (void)hipFree(p);
            HIPCHECK(idx, e);
        }
    }
    HIPCHECK(idx, hipMemsetAsync(shadow, 0, (size_t)new_cap * idx->dpad * sizeof(uint16_t), idx->stream));
    HIPCHECK(idx, hipMemsetAsync(shadow8, 0, (size_t)new_cap * idx->dpad8, idx->stream));
    HIPCHECK(idx, hipMemsetAsync(flag8, 0, (size_t)new_cap, idx->stream));
    HIPCHECK(idx, hipMemsetAsync(grp8, 0, (size_t)(new_cap / kI8GroupRows) * sizeof(I8Group), idx->stream));
    if (idx->n > 0) {
        HIPCHECK(idx, hipMemcpyAsync(rows, idx->rows, (size_t)idx->n * idx->dim * sizeof(float),
                                     hipMemcpyDeviceToDevice, idx->stream));
        HIPCHECK(idx, hipMemcpyAsync(shadow, idx->shadow, (size_t)idx->n * idx->dpad * sizeof(uint16_t),
                                     hipMemcpyDeviceToDevice, idx->stream));
        HIPCHECK(idx, hipMemcpyAsync(nrm2, idx->nrm2, (size_t)idx->n * sizeof(float), hipMemcpyDeviceToDevice,
                                     idx->stream));
        HIPCHECK(idx, hipMemcpyAsync(shadow8, idx->shadow8, (size_t)idx->n * idx->dpad8, hipMemcpyDeviceToDevice,
                                     idx->stream));
        HIPCHECK(idx, hipMemcpyAsync(flag8, idx->flag8, (size_t)idx->n, hipMemcpyDeviceToDevice, idx->stream));
        HIPCHECK(idx, hipMemcpyAsync(grp8, idx->grp8, (size_t)((idx->n + kI8GroupRows - 1) / kI8GroupRows) * sizeof(I8Group),
                                     hipMemcpyDeviceToDevice, idx->stream));
    }
    HIPCHECK(idx, hipStreamSynchronize(idx->stream));
    if (idx->rows) (void)hipFree(idx->rows);
    if (idx->shadow) (void)hipFree(idx->shadow);
    if (idx->nrm2) (void)hipFree(idx->nrm2);
    if (idx->shadow8) (void)hipFree(idx->shadow8);
    if (idx->flag8) (void)hipFree(idx->flag8);
    if (idx->grp8) (void)hipFree(idx->grp8);
    idx->shadow8 = shadow8;
    idx->flag8 = flag8;
    idx->grp8 = grp8;
    idx->rows = rows;
    idx->shadow = shadow;
    idx->nrm2 = nrm2;
    idx->cap_rows = new_cap;
    return MI355DR_OK;
}

int ensure_qstate(mi355dr_index* idx) {
    if (idx->qstate_ready) return MI355DR_OK;
    const size_t B = kQBlockMax;
    HIPCHECK(idx, hipMalloc(&idx->st.qn, B * sizeof(float)));
    HIPCHECK(idx, hipMalloc(&idx->st.qhat, B * idx->dpad * sizeof(uint16_t)));
    HIPCHECK(idx, hipMalloc(&idx->st.thr, B * sizeof(float)));
    HIPCHECK(idx, hipMalloc(&idx->st.cnt, B * sizeof(int)));
    HIPCHECK(idx, hipMalloc(&idx->st.best_n, B * sizeof(int)));
    HIPCHECK(idx, hipMalloc(&idx->st.best_key, B * kKMax * sizeof(uint64_t)));
    HIPCHECK(idx, hipMalloc(&idx->st.best_row, B * kKMax * sizeof(int32_t)));
    HIPCHECK(idx, hipMalloc(&idx->st.thr_key, B * sizeof(uint64_t)));
    HIPCHECK(idx, hipMalloc(&idx->st.thr_row, B * sizeof(int32_t)));
    HIPCHECK(idx, hipMalloc(&idx->st.status, (B + 1) * sizeof(int)));  // [B]: the block's OR-ed status word
    idx->status_or_dev = idx->st.status + B;
    HIPCHECK(idx, hipMalloc(&idx->st.E, B * sizeof(float)));
    HIPCHECK(idx, hipMalloc(&idx->st.E16, B * sizeof(float)));
    HIPCHECK(idx, hipMalloc(&idx->st.sc, B * sizeof(float)));
    HIPCHECK(idx, hipMalloc(&idx->st.kq, B * sizeof(float)));
    HIPCHECK(idx, hipMalloc(&idx->rq_progress, kRqProgressWords * sizeof(int)));
    HIPCHECK(idx, hipMemset(idx->rq_progress, 0, kRqProgressWords * sizeof(int)));
    HIPCHECK(idx, hipMalloc(&idx->st.qhat8, B * idx->dpad8));
    HIPCHECK(idx, hipMalloc(&idx->st.carry, B * sizeof(int)));
    HIPCHECK(idx, hipMalloc(&idx->qdev, B * idx->dim * sizeof(float)));
    HIPCHECK(idx, hipMalloc(&idx->cand_row, B * kCandCapWide * sizeof(int32_t)));
    HIPCHECK(idx, hipMalloc(&idx->cand_val, B * kCandCapWide * sizeof(float)));
    HIPCHECK(idx, hipMalloc(&idx->qlist_dev, 2 * B * sizeof(int)));  // second half: overflow re-runs
    HIPCHECK(idx, hipHostMalloc(&idx->status_host, (B + 1) * sizeof(int)));
    HIPCHECK(idx, hipMalloc(&idx->out_dist_dev, B * kKMax * sizeof(double)));
    HIPCHECK(idx, hipMalloc(&idx->out_rows_dev, B * kKMax * sizeof(int64_t)));
    HIPCHECK(idx, hipMalloc(&idx->prune_skip, (2 + 2 * B) * sizeof(int)));
    HIPCHECK(idx, hipMalloc(&idx->stat_dev, 2 * B * sizeof(unsigned long long)));
    HIPCHECK(idx, hipMemsetAsync(idx->stat_dev, 0, 2 * B * sizeof(unsigned long long), idx->stream));
    // the prune / scan kernels use more than the default 64 KiB of dynamic LDS
    if (prune_lds_bytes(idx->dim, kPruneBigThreads, kPruneBigSort, 0) > 160 * 1024 || scan_lds_bytes(idx->dim, 1) > 160 * 1024)
        return fail(idx, MI355DR_E_UNSUPPORTED, "dim too large for the select kernels' LDS budget");
    HIPCHECK(idx, hipFuncSetAttribute((const void*)k_prune<kPruneBigThreads, kPruneBigSort>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)prune_lds_bytes(idx->dim, kPruneBigThreads, kPruneBigSort, 0)));
    HIPCHECK(idx, hipFuncSetAttribute((const void*)k_prune<kPruneSmallThreads, kPruneSmallSort>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)prune_lds_bytes(idx->dim, kPruneSmallThreads, kPruneSmallSort, idx->dpad)));
    HIPCHECK(idx, hipFuncSetAttribute((const void*)k_prune_wide<2, 32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)prune_wide_lds_bytes(idx->dim, 2)));
    {
        int per = kScanQ;
        while (per > 1 && scan_lds_bytes(idx->dim, per) > 150 * 1024) per >>= 1;
        HIPCHECK(idx, hipFuncSetAttribute((const void*)k_scan, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)scan_lds_bytes(idx->dim, per)));
    }
    HIPCHECK(idx, hipFuncSetAttribute((const void*)k_scan32, hipFuncAttributeMaxDynamicSharedMemorySize, kScan32Lds));
    HIPCHECK(idx, hipFuncSetAttribute((const void*)k_screen<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreenLds));
    HIPCHECK(idx, hipFuncSetAttribute((const void*)k_screen<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreenLds));
    {
        const int stream_lds = kStreamQueryBytesMax + kStreamStages * kStreamStageBytes;
        HIPCHECK(idx, hipFuncSetAttribute((const void*)k_screen_stream<false, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, stream_lds));
        HIPCHECK(idx, hipFuncSetAttribute((const void*)k_screen_stream<true, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, stream_lds));
        HIPCHECK(idx, hipFuncSetAttribute((const void*)k_screen_stream<false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, stream_lds));
        HIPCHECK(idx, hipFuncSetAttribute((const void*)k_screen_stream<true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, stream_lds));
    }
    HIPCHECK(idx, hipFuncSetAttribute((const void*)k_screen256c<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kScreen256Lds));
    HIPCHECK(idx, hipFuncSetAttribute((const void*)k_screen256c<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kScreen256Lds));
#define MI355_RQ_ATTR(KS)                                                                                              \
    HIPCHECK(idx, hipFuncSetAttribute((const void*)k_screen_rq<KS, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, rq_lds(KS)));
    MI355_RQ_FORMS(MI355_RQ_ATTR)
#undef MI355_RQ_ATTR
    HIPCHECK(idx, hipFuncSetAttribute((const void*)k_screen_rq<6, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, rq_lds(6)));
    HIPCHECK(idx, hipFuncSetAttribute((const void*)k_merge_topk, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kSortMax * 12));
    idx->qstate_ready = true;
    return MI355DR_OK;
}

// bf16 screen bound E(d): |t - exact cosine key| <= 2^-7 + 2^-15 + 8*d*2^-24   (DESIGN.md "Screen bounds").
// bf16 keeps 8 significand bits: round-to-nearest has unit roundoff 2^-8 PER OPERAND, so a product of two rounded
// operands is off by up to 2^-7 + 2^-16 (relative), and by Cauchy-Schwarz so is the dot product of unit vectors.
inline float screen_bound(int d) {
    return (float)(std::ldexp(1.0, -7) + std::ldexp(1.0, -15) + 8.0 * d * std::ldexp(1.0, -24));
}

EventPair take_events(mi355dr_index* idx) {
    if (!idx->ev_pool.empty()) {
        EventPair p = idx->ev_pool.back();
        idx->ev_pool.pop_back();
        return p;
    }
    EventPair p{};
    (void)hipEventCreate(&p.a);
    (void)hipEventCreate(&p.b);
    return p;
}

void drain_events(mi355dr_index* idx) {  // the pairs whose launch has finished (a later block's may still be in flight)
    std::vector<EventPair> keep;
    for (auto& p : idx->ev_pending) {
        if (hipEventQuery(p.b) == hipErrorNotReady) {
            keep.push_back(p);
            continue;
        }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            idx->s_screen_ns += (int64_t)(ms * 1e6);
            if (p.big) idx->s_big_ns += (int64_t)(ms * 1e6);
        }
        idx->ev_pool.push_back(p);
    }
    idx->ev_pending.swap(keep);
}

// Candidates a chunk appends per query ~ k * (chunk / rows seen before) * inflation, where the inflation is how much
// the screen's bound widens the tail it has to keep: measured ~4-5 for the bf16 bound and ~14-16 for the int8 bound on
// Gaussian data.  The chunk growth is capped so that this stays inside what one prune of the one-wave kernel holds.
constexpr double kInflationBf16 = 5.0, kInflationI8 = 16.0;
constexpr double kSmallBlockBudget = 1.6;  // (the measured inflations are ~4 and ~9-10: the budget above carries that much slack)
// Round 6: passes at 33 <= k <= 128 prune with the two-wave form (k_prune_wide.h).  Its 4096 entries hold a chunk's appends
// AND the survivors carried from the chunk before; the first chunk behind the starter screens its rows from row 0, so it
// appends ~ k * inflation * ratio.  The inflation of the int8 bound at these k: 5.5 (k-th best of 64 k rows) ... 9.6 (of 10 M)
// by the Gaussian tail ratio, ~6 measured over the round-5 pass at k = 100 -- budgeted as 10 with a quarter of the entries spare.
constexpr double kInflationI8Wide = 10.0, kInflationBf16Wide = 4.0;
inline bool wide_ok(const mi355dr_index* idx, int k) { return idx->prune_wide && k >= kWideKMin && k <= kWideKMax; }
// the prunes of the search in progress take the two-wave form (the exact scan keeps the general form and its 2048-slot lists)
inline bool wide_now(const mi355dr_index* idx) { return wide_ok(idx, idx->k_now) && idx->path != MI355DR_PATH_SCAN; }
// candidate slots per query (= the lists' stride) of the search in progress
inline int cap_now(const mi355dr_index* idx) { return !idx->cap_set && wide_now(idx) ? kCandCapWide : idx->cap; }
inline double growth_budget(const mi355dr_index* idx, int k, bool i8) {
    if (wide_ok(idx, k) && idx->path != MI355DR_PATH_SCAN) {
        const int room = std::min(kWideEntries, idx->cap_set ? idx->cap : kCandCapWide);
        return 0.75 * room / ((double)k * (i8 ? idx->wide_inflation_x10 / 10.0 : kInflationBf16Wide)) - 1.0;  // (1 + growth = the chunk ratio)
    }
    const int room = k < kPruneSmallSort / 2 ? kPruneSmallSort - k : idx->cap;  // (large k: the general prune, whole buffer)
    return 0.6 * std::min(room, idx->cap) / ((double)k * (i8 ? kInflationI8 : kInflationBf16));
}

// which screen the search in progress uses: int8 needs a corpus that quantised within the limit and (in AUTO) a k small
// enough that its wider bound still allows chunks to grow (k <= 133 at the default budget line); larger k keeps bf16
inline bool i8_available(const mi355dr_index* idx) { return idx->irr8_n <= kIrrCap; }
inline bool use_i8(const mi355dr_index* idx) {
    if (idx->retry_level > 0) return false;  // re-screening overflowed queries: the ~3x tighter bf16 bound
    if (idx->screen_dtype == MI355DR_SCREEN_I8) return true;
    // (measured round 3, N = 10 M, 1024 queries: int8 7.4 / 8.0 / 9.5 / 11.7 / 12.7 ms at k = 10 / 32 / 64 / 100 / 128 against
    // 13-14.4 ms for bf16, whose kernel alone is 11 ms; 22 against 16 at k = 200, where the int8 chunks hardly grow any more:
    // the cross-over is near k = 160, budget 0.2; round 2 drew the line at k = 24, budget 1.5)
    return idx->screen_dtype == MI355DR_SCREEN_AUTO && i8_available(idx) && idx->k_now < idx->i8_demoted_k &&
           growth_budget(idx, idx->k_now, true) >= idx->i8_min_budget;
}

int launch_prune(mi355dr_index* idx, hipStream_t s, int nblocks, const int* qlist, int k, int exact, bool thr_only = false,
                 bool one_wave_only = false, bool defer_b = false) {
    PruneArgs pa{};
    pa.thr_only = thr_only ? 1 : 0;
    pa.one_wave_only = one_wave_only ? 1 : 0;
    pa.defer_b = defer_b && idx->defer_round_b ? 1 : 0;
    pa.rows = idx->rows;
    pa.nrm2 = idx->nrm2;
    pa.q = idx->qdev;
    pa.st = idx->st;
    pa.cand_row = idx->cand_row;
    pa.cand_val = idx->cand_val;
    pa.qlist = qlist;
    pa.stat = idx->stat_dev;
    pa.cap = cap_now(idx);
    pa.d = idx->dim;
    pa.k = k;
    pa.metric = idx->metric;
    pa.exact = exact;
    // (flags exist only for loose rows, and every loose row is counted: a corpus without any -- the usual case -- spares each
    // candidate the dependent flag8[row] load, one memory round trip of the prune's latency chain)
    pa.flag8 = (use_i8(idx) && idx->irr8_n > 0) ? idx->flag8 : nullptr;
    pa.cscale = idx->metric == MI355DR_METRIC_IP ? idx->cmax : 1.0f;
    // int8 screen, cosine: candidates that survive the exact cut are screened once more on their bf16 shadow rows
    // (half the bytes of an fp32 row, a bound ~5x tighter) before the exact re-score
    pa.shadow16 = (use_i8(idx) && idx->metric == 0 && !exact && idx->prefilter16) ? idx->shadow : nullptr;
    pa.dpad = idx->dpad;
    pa.round_a = idx->round_a;
    // the general form walks the (usually empty) list of queries the one-wave form left, on a small grid
    if (!exact && wide_now(idx)) {  // 33 <= k <= 128: the two-wave form alone (what it cannot hold is flagged for the re-screen)
        pa.shadow16 = nullptr;
        hipLaunchKernelGGL((k_prune_wide<2, 32>), dim3(nblocks), dim3(2 * kWave), prune_wide_lds_bytes(idx->dim, 2), s, pa);
        HIPCHECK(idx, hipGetLastError());
        return MI355DR_OK;
    }
    const bool list_mode = qlist == nullptr;
    pa.skip_list = list_mode ? idx->prune_skip : nullptr;
    pa.skip_parity = list_mode ? (idx->prune_parity ^= 1) : 0;
    // small instantiation first (common case, whole block resident), then the large one for what it skipped
    hipLaunchKernelGGL((k_prune<kPruneSmallThreads, kPruneSmallSort>), dim3(nblocks), dim3(kPruneSmallThreads),
                       prune_lds_bytes(idx->dim, kPruneSmallThreads, kPruneSmallSort, pa.shadow16 ? idx->dpad : 0), s, pa);
    HIPCHECK(idx, hipGetLastError());
    if (one_wave_only) return MI355DR_OK;  // (what the one-wave form cannot hold is flagged for the host's re-screen)
    hipLaunchKernelGGL((k_prune<kPruneBigThreads, kPruneBigSort>), dim3(list_mode ? std::min(nblocks, 64) : nblocks),
                       dim3(kPruneBigThreads), prune_lds_bytes(idx->dim, kPruneBigThreads, kPruneBigSort, 0), s, pa);
    HIPCHECK(idx, hipGetLastError());
    return MI355DR_OK;
}

int launch_prep(mi355dr_index* idx, hipStream_t s, int B, int Bpad, int metric, int cnt0 = 0) {
    hipLaunchKernelGGL(k_prep_queries, dim3(Bpad), dim3(64), (size_t)idx->dim * sizeof(float), s, idx->qdev, B, idx->dim,
                       idx->dpad, metric, idx->st, idx->dpad8, use_i8(idx) ? 1 : 0, idx->bf16_ec, idx->status_or_dev,
                       idx->prune_skip, cnt0, idx->metric == MI355DR_METRIC_IP ? idx->cmax : 1.0f);
    idx->prune_parity = 0;
    HIPCHECK(idx, hipGetLastError());
    return MI355DR_OK;
}

// tile edge used for a block of B queries: the 256x256 ping-pong kernel from 129 queries up, else 128x128
inline int screen_tile(int B) { return B > kTileN ? kT2 : kTileM; }

constexpr int kRetryLevels = 2;  // re-screens of an overflowed query (bf16, growth/2, then growth 0.25) before the exact scan

// launch one screen pass over rows [r0, r_end) (r0 a multiple of the tile edge)
__global__ void k_set_counts(int* cnt, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cnt[i] = v;
}

int launch_screen(mi355dr_index* idx, hipStream_t s, int B, int64_t r0, int64_t r_end, int cap, int emit_mode) {
    const bool emit_all = emit_mode != 0;  // (both special epilogues live in k_screen)
    // the emit-all first chunk always goes through the 128x128 kernel (k_screen256 has no emit-all epilogue)
    // ... and so do chunks of a few thousand rows: their thresholds are still so low that a good part of the tile is a
    // hit, which the per-lane global append of k_screen handles better than k_screen256's small per-wave queues
    const int tile = (emit_all || r_end - r0 <= idx->small_chunk_rows) ? kTileM : screen_tile(B);
    const bool i8 = use_i8(idx);
    ScreenArgs2 sa{};
    sa.status = idx->st.status;
    sa.shadow = i8 ? (const void*)idx->shadow8 : (const void*)idx->shadow;
    sa.qhat = i8 ? (const void*)idx->st.qhat8 : (const void*)idx->st.qhat;
    sa.thr = idx->st.thr;
    sa.sc = idx->st.sc;
    sa.kq = idx->st.kq;
    sa.grp = idx->grp8;
    sa.flag8 = idx->flag8;
    sa.cnt = idx->st.cnt;
    sa.cand_row = idx->cand_row;
    sa.cand_val = idx->cand_val;
    sa.row_bytes = i8 ? idx->dpad8 : idx->dpad * 2;
    sa.ksteps = sa.row_bytes / kRowB;
    sa.cap = cap;
    sa.ct0 = (int)(r0 / tile);
    sa.n_ctiles = (int)(round_up(r_end, tile) / tile) - sa.ct0;
    sa.n_qtiles = (int)(round_up(B, tile) / tile);
    sa.row_end = r_end;
    sa.row0 = r0;
    sa.emit_all = emit_mode;
    const int64_t grid = round_up(sa.n_ctiles, 8) * sa.n_qtiles;
    if (tile == kT2 && i8 && idx->screen_rq && screen_rq_has(sa.ksteps)) {
        // query operand resident in registers, 128-row tiles (k_screen_rq.h): int8 shadows of at most 768 bytes per row
        sa.ct0 = (int)(r0 / kRqRows);
        sa.n_ctiles = (int)(round_up(r_end, kRqRows) / kRqRows) - sa.ct0;
        const unsigned g2 = screen_rq_grid(sa.n_ctiles, sa.n_qtiles);
        idx->s_rq_launches++;
        if (idx->debug_park > 0 && r_end - r0 >= idx->debug_park) {  // (diagnostic: thresholds at +inf for a launch of at least that many rows -- its cost without a single hit; results are WRONG)
            if (!idx->park_thr) {
                std::vector<float> inf(kQBlockMax, INFINITY);
                HIPCHECK(idx, hipMalloc(&idx->park_thr, kQBlockMax * sizeof(float)));
                HIPCHECK(idx, hipMemcpy(idx->park_thr, inf.data(), kQBlockMax * sizeof(float), hipMemcpyHostToDevice));
            }
            sa.thr = idx->park_thr;
        }
        sa.progress = idx->rq_progress;  // sibling drift limiter: words of older launches carry another stamp and are ignored
        sa.epoch = idx->rq_epoch = idx->rq_epoch % 4095 + 1;
        // (the stamp has 12 bits: when it wraps, words left by launches 4095 stamps ago are cleared so that none of them can
        // pass for a sibling of this launch -- a stale word could only cost a capped wait, never a result)
        if (sa.epoch == 1) HIPCHECK(idx, hipMemsetAsync(idx->rq_progress, 0, kRqProgressWords * sizeof(int), s));
        sa.drift = idx->screen_drift;
        sa.flush_mask = idx->flush_mask_now;
        sa.flush_alone = idx->screen_flush_alone;
        if (sa.ksteps == 6 && !idx->screen_rq_split_tests) {  // (A/B form, d = 768 only: every block test in one piece)
            hipLaunchKernelGGL((k_screen_rq<6, false, true>), dim3(g2), dim3(512), rq_lds(6), s, sa);
        } else {
            switch (sa.ksteps) {
#define MI355_RQ_LAUNCH(KS) \
    case KS: hipLaunchKernelGGL((k_screen_rq<KS, true, true>), dim3(g2), dim3(512), rq_lds(KS), s, sa); break;
                MI355_RQ_FORMS(MI355_RQ_LAUNCH)
#undef MI355_RQ_LAUNCH
            }
        }
    } else if (tile == kT2) {
        const unsigned g2 = screen256_grid(sa.n_ctiles, sa.n_qtiles);  // persistent: <= one workgroup per CU
        if (i8) hipLaunchKernelGGL((k_screen256c<true>), dim3(g2), dim3(512), kScreen256Lds, s, sa);
        else hipLaunchKernelGGL((k_screen256c<false>), dim3(g2), dim3(512), kScreen256Lds, s, sa);
    } else if (!emit_all && idx->screen_stream && B <= 64 && sa.ksteps >= 1 &&
               (B <= 32 ? 32 : 64) * sa.row_bytes <= kStreamQueryBytesMax) {
        // small query blocks: the streaming form (resident query block, deep row ring, one persistent workgroup per CU)
        const int nq = B <= 32 ? 32 : 64;
        const unsigned gs = (unsigned)std::min(sa.n_ctiles, 256);
        const size_t lds = screen_stream_lds(nq, sa.row_bytes);
        if (i8 && nq == 32) hipLaunchKernelGGL((k_screen_stream<true, 32>), dim3(gs), dim3(256), lds, s, (ScreenArgs)sa);
        else if (i8) hipLaunchKernelGGL((k_screen_stream<true, 64>), dim3(gs), dim3(256), lds, s, (ScreenArgs)sa);
        else if (nq == 32) hipLaunchKernelGGL((k_screen_stream<false, 32>), dim3(gs), dim3(256), lds, s, (ScreenArgs)sa);
        else hipLaunchKernelGGL((k_screen_stream<false, 64>), dim3(gs), dim3(256), lds, s, (ScreenArgs)sa);
    } else {
        if (i8) hipLaunchKernelGGL(k_screen<true>, dim3((unsigned)grid), dim3(256), kScreenLds, s, (ScreenArgs)sa);
        else hipLaunchKernelGGL(k_screen<false>, dim3((unsigned)grid), dim3(256), kScreenLds, s, (ScreenArgs)sa);
    }
    HIPCHECK(idx, hipGetLastError());
    if (emit_mode == kEmitAll) {  // every row of the chunk was stored at slot row-r0 for every query
        // (the starter's one-candidate-per-slab count is what k_prep_queries initialised the lists with: starter_count())
        const int per_query = (int)(r_end - r0);
        hipLaunchKernelGGL(k_set_counts, dim3((B + 255) / 256), dim3(256), 0, s, idx->st.cnt, B, per_query);
        HIPCHECK(idx, hipGetLastError());
    }
    return MI355DR_OK;
}

// ---- the pass schedule --------------------------------------------------------------------------------------------
// Thresholds are frozen during a launch, so the corpus is walked in geometrically growing chunks, each followed by the exact
// re-score + select of what it appended (k_prune), which publishes the next chunk's thresholds.  Round 3:
//  * STARTER instead of the three smallest chunks (k <= kStarterKMax, first attempt only): one k_screen launch over the first
//    S <= 16 k rows that keeps, per query, the best value of every 64-row slab (S / 64 candidates, no thresholds, no atomics)
//    + one k_prune (thr_only) that re-scores the best-looking of them exactly and publishes the threshold their k-th best
//    gives -- valid whatever the sample missed -- and keeps nothing; the first regular chunk then starts at row 0.  Two
//    launches (~80 us) where the ladder 1 024 -> 4 096 -> 16 384 took six (~290 us), at every shard size.
//  * The chunk ends are PLANNED: n = the fewest steps of ratio <= 1 + growth from the starter's sample (or the emit-all first
//    chunk) to the end, then one uniform ratio (N / S)^(1/n) -- no short last chunk with a prune of its own.
//  * (Option prune_companion = 0: no general-form launch behind the one-wave prune, what it cannot hold is flagged and
//    re-screened -- measured slower: see index.h.)
constexpr int kStarterKMax = 32;          // the starter's round A re-scores max(32, 2k) <= 64 rows: one batch of the one-wave form
constexpr int64_t kStarterRows = 16384;   // sample size (256 slabs); a corpus must hold at least 4 samples
// (a pass at 33 <= k <= 128 samples idx->starter_rows_wide rows: 65536 = 1024 slabs by default)
struct PassPlan {
    int64_t sample = 0;          // > 0: starter over rows [0, sample)
    std::vector<int64_t> ends;   // chunk ends, ascending, last = n; the first chunk starts at 0 (starter) or is the emit-all one
    bool emit_all_first = false;
};
PassPlan plan_pass(const mi355dr_index* idx, int B, int k, double growth) {
    PassPlan p;
    const int64_t n = idx->n;
    const int tile = screen_tile(B);
    int64_t seen = 0;  // rows whose k-th best the first planned chunk's threshold comes from
    // the starter leaves one candidate per 64-row slab of its sample in every query's list: the list must hold them
    // (option cand_cap goes down to 16: with fewer slots than slabs every list would start past its end and the whole block
    // would be flagged for a re-screen), and the sample must offer at least k of them, else its prune publishes no threshold
    // and the first regular chunk would run at thr = -inf WITHOUT the emit-all epilogue -- correct, pathologically slow
    // (4096 <= n < 8192 with k in 17..32).  Either way: the emit-all ladder.
    // Round 6, 33 <= k <= 128 (two-wave prune): the same estimator over a 64 k-row sample -- 1024 slab maxima per query, of
    // which the prune re-scores the 128 best-looking; their k-th best exact score is (about) the k-th best of the sample.
    const bool wide = wide_now(idx);
    const int cap = cap_now(idx);
    const int64_t starter_rows = std::min<int64_t>(wide ? idx->starter_rows_wide : kStarterRows, n / 4) / kTileM * kTileM;
    const int64_t starter_slabs = (starter_rows + kSlabRows - 1) / kSlabRows;
    if (idx->starter && (k <= kStarterKMax || wide) && idx->retry_level == 0 && idx->chunk0_set == 0 && n >= 4 * 1024 &&
        starter_slabs <= cap && starter_slabs >= k) {
        p.sample = starter_rows;
        seen = p.sample;
    } else {
        const int64_t c0 = std::min<int64_t>(n, round_up(std::max<int64_t>(tile, std::min<int64_t>(idx->chunk0_rows, cap)), tile));
        p.emit_all_first = c0 <= cap;
        p.ends.push_back(c0);
        seen = c0;
        if (c0 >= n) return p;
    }
    const double rmax = 1.0 + growth;
    const double span = (double)n / (double)seen;
    // (0.15: a span a hair above a power of the ratio -- 1.25 M rows behind a 16 k sample, the 8-way shard of the headline corpus --
    // takes the smaller number of chunks: 3 instead of 4 there, 1.136 against 1.156 ms per pass with k_screen_rq, whose hits cost
    // less than a chunk boundary; 5 M rows take 4 instead of 5: 3.50 against 3.49 ms; profiles/r05_chunk_sweep.txt)
    int steps = std::max(1, (int)std::ceil(std::log(span) / std::log(rmax) - 0.15));
    const double r = std::pow(span, 1.0 / steps);
    // tapered ratios (same product): r_i = r * t^((steps-1)/2 - i); the first (largest) one stays within what the lists hold
    double taper = std::max(1.0, (idx->chunk_taper_x100 > 0 ? idx->chunk_taper_x100 : (wide ? 120 : 100)) / 100.0);
    if (steps > 1 && taper > 1.0) {
        const double room = std::max(1.0, rmax * 1.25 / r);  // (the budget line already keeps a quarter / 40 % of the entries spare)
        taper = std::min(taper, std::pow(room, 2.0 / (steps - 1)));
    } else {
        taper = 1.0;
    }
    double pos = (double)seen;
    int64_t prev = p.sample > 0 ? 0 : seen;
    for (int i = 1; i <= steps; ++i) {
        pos *= r * std::pow(taper, (steps - 1) / 2.0 - (i - 1));
        int64_t end = i == steps ? n : std::min<int64_t>(n, round_up((int64_t)pos, tile));
        if (tile == kT2 && end < n) {
            // whole rounds of the persistent grid: 8 XCDs x (32 / n_qtiles) corpus tiles are in flight at a time, and a chunk
            // of 4.3 rounds costs 5 (k_screen256c's launches of 70 k rows ran at 1.6 ns per row against 0.68 in long ones)
            const int n_qtiles = (int)(round_up(B, kT2) / kT2);
            const int64_t round_rows = (int64_t)8 * (32 / n_qtiles) * kT2;
            const int64_t len = end - prev;
            if (len >= 2 * round_rows) end = std::min<int64_t>(n, prev + (len + round_rows / 2) / round_rows * round_rows);
        }
        if (end <= prev) continue;
        p.ends.push_back(end);
        prev = end;
        if (end >= n) break;
    }
    if (p.ends.empty() || p.ends.back() < n) p.ends.push_back(n);
    return p;
}

inline int starter_count(const PassPlan& p) { return (int)((p.sample + kSlabRows - 1) / kSlabRows); }

PassPlan make_plan(const mi355dr_index* idx, int B, int k) {
    // (measured round 3, N = 10 M, k = 10: a ratio of 4 per step -- 5 chunks -- 7.43 ms, the budget's 4.8 -- 4 chunks -- 7.54)
    double growth = std::max(0.25, std::min((double)idx->chunk_growth, growth_budget(idx, k, use_i8(idx))));
    // small query blocks: a pass is one stream over the shadow rows plus one latency-bound re-score launch per chunk, and an
    // append costs nothing -- fewer, larger chunks (the k-dependent budget alone bounds the growth: x7 per step at k = 10)
    if (B <= 64 && idx->retry_level == 0 && idx->chunk_growth_set == 0)
        growth = std::max(growth, std::min(8.0, growth_budget(idx, k, use_i8(idx)) * kSmallBlockBudget));
    if (idx->retry_level == 1) growth = std::max(0.25, growth * 0.5);
    if (idx->retry_level >= 2) growth = 0.25;  // (every chunk then holds <= 20 % of the rows: a dense neighbourhood is split up)
    return plan_pass(idx, B, k, growth);
}

// screen path over all rows for the B queries prepared in idx->st / idx->qdev (candidate counts initialised to
// starter_count(plan) by k_prep_queries when the plan has a starter)
int run_screen(mi355dr_index* idx, hipStream_t s, int B, int k, const PassPlan& plan) {
    int64_t kept_all_below = 0;  // rows the emit-all first chunk already turned into candidates
    const bool i8 = use_i8(idx);
    const int side_n = i8 ? idx->irr8_n : idx->irr_n;  // rows this screen cannot see
    constexpr int kSideMerge = 32;
    bool side_done = false;
    // one-wave prune alone: small k, first attempt (a retry keeps the general form: it is the last screen before the exact scan)
    const bool lean = plan.sample > 0 && idx->prune_companion == 0;
    auto timed = [&](bool big, int64_t rows, auto&& launch) -> int {
        EventPair ev{};
        if (idx->profile) {
            ev = take_events(idx);
            HIPCHECK(idx, hipEventRecord(ev.a, s));
        }
        CHECK(launch());
        if (idx->profile) {
            HIPCHECK(idx, hipEventRecord(ev.b, s));
            ev.big = big ? 1 : 0;
            idx->ev_pending.push_back(ev);
        }
        idx->s_screen_launches++;
        idx->s_screen_rows += rows;
        if (big) {
            idx->s_big_launches++;
            idx->s_big_rows += rows;
        }
        return MI355DR_OK;
    };
    if (plan.sample > 0) {
        CHECK(timed(false, plan.sample, [&] { return launch_screen(idx, s, B, 0, plan.sample, cap_now(idx), kEmitSlabMax); }));
        CHECK(launch_prune(idx, s, B, nullptr, k, /*exact=*/0, /*thr_only=*/true, /*one_wave_only=*/true));
        idx->s_starters++;
    }
    int64_t done = 0;
    for (size_t ci = 0; ci < plan.ends.size(); ++ci) {
        const int64_t end = plan.ends[ci];
        const bool emit_all = ci == 0 && plan.emit_all_first;
        if (emit_all) kept_all_below = end;
        // the first chunk has no threshold yet: it keeps every row (direct stores) as long as it fits the buffer
        const bool big = !emit_all && end - done > idx->small_chunk_rows && screen_tile(B) == kT2;
        {   // hit lanes a wave of k_screen_rq expects per tile (32 queries x 128 rows): rows above the threshold of `seen` rows
            // ~ k x inflation / seen per row and query (inflation of the int8 bound ~8)
            const int64_t seen = ci == 0 ? (plan.sample > 0 ? plan.sample : end) : done;
            const double lanes = 8.0 * k * 4096.0 / (double)std::max<int64_t>(seen, 1);
            int period = 1;
            while (period < 64 && period * 2 * lanes <= idx->screen_flush_lanes) period *= 2;
            idx->flush_mask_now = idx->screen_flush_sync ? period - 1 : -1;
        }
        CHECK(timed(big, end - done, [&] { return launch_screen(idx, s, B, done, end, cap_now(idx), emit_all ? kEmitAll : 0); }));
        idx->s_chunks++;
        if (end >= idx->n && side_n > 0 && side_n <= kSideMerge) {
            // rows this screen cannot see (irregular; for int8 also loose): a handful of them ride the LAST chunk's prune
            // as "no bound" candidates instead of costing a prune pass of their own (0.12 ms per block at N = 10 M)
            hipLaunchKernelGGL(k_emit_irregular, dim3(B), dim3(64), 0, s, i8 ? idx->irr8_rows : idx->irr_rows, side_n, idx->st,
                               idx->cand_row, idx->cand_val, cap_now(idx), (int)kept_all_below);
            HIPCHECK(idx, hipGetLastError());
            side_done = true;
        }
        // every prune but the pass's last one carries its survivors over instead of re-scoring them (k_prune: defer_b)
        const bool last = end >= idx->n && (side_n == 0 || side_done);
        CHECK(launch_prune(idx, s, B, nullptr, k, /*exact=*/0, false, lean, !last));
        done = end;
    }
    if (side_n > 0 && !side_done) {
        hipLaunchKernelGGL(k_emit_irregular, dim3(B), dim3(64), 0, s, i8 ? idx->irr8_rows : idx->irr_rows, side_n, idx->st,
                           idx->cand_row, idx->cand_val, cap_now(idx), (int)kept_all_below);
        HIPCHECK(idx, hipGetLastError());
        CHECK(launch_prune(idx, s, B, nullptr, k, 0, false, lean));
    }
    idx->s_passes++;
    return MI355DR_OK;
}

__global__ void k_reset_queries(QueryState st, const int* qlist, int nq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const int q = qlist[i];
    st.cnt[q] = 0;
    st.carry[q] = 0;
    st.best_n[q] = 0;
    st.thr_key[q] = kKeyNaN;
    st.thr_row[q] = 0x7FFFFFFF;
    st.status[q] &= ~kStOverflow;
}

// exact scan of rows [r0,r1) for the <= kScanQ queries in qlist_dev[off..off+nq), then exact prune
int scan_range(mi355dr_index* idx, hipStream_t s, int off, int nq, int k, int64_t r0, int64_t r1) {
    ScanArgs sa{};
    sa.rows = idx->rows;
    sa.nrm2 = idx->nrm2;
    sa.q = idx->qdev;
    sa.st = idx->st;
    sa.cand_row = idx->cand_row;
    sa.cand_val = idx->cand_val;
    sa.qlist = idx->qlist_dev + off;
    sa.nq = nq;
    sa.cap = cap_now(idx);
    sa.d = idx->dim;
    sa.metric = idx->metric;
    sa.row0 = r0;
    sa.row1 = r1;
    const int64_t grid = (r1 - r0 + kScanThreads - 1) / kScanThreads;
    if (idx->dim % kScan32PieceCols == 0 && idx->scan_dma)  // rows are whole 128-byte pieces: the LDS-DMA form
        hipLaunchKernelGGL(k_scan32, dim3((unsigned)grid), dim3(kScanThreads), kScan32Lds, s, sa, idx->n);
    else
        hipLaunchKernelGGL(k_scan, dim3((unsigned)grid), dim3(kScanThreads), scan_lds_bytes(idx->dim, nq), s, sa);
    HIPCHECK(idx, hipGetLastError());
    return launch_prune(idx, s, nq, idx->qlist_dev + off, k, /*exact=*/1);
}

// guaranteed exact path for the queries listed in `qs` (indices into the current block)
int run_scan(mi355dr_index* idx, hipStream_t s, const std::vector<int>& qs, int k) {
    if (qs.empty()) return MI355DR_OK;
    HIPCHECK(idx, hipMemcpyAsync(idx->qlist_dev, qs.data(), qs.size() * sizeof(int), hipMemcpyHostToDevice, s));
    const int nq_all = (int)qs.size();
    hipLaunchKernelGGL(k_reset_queries, dim3((nq_all + 255) / 256), dim3(256), 0, s, idx->st, idx->qlist_dev, nq_all);
    HIPCHECK(idx, hipGetLastError());
    const int per = kScanQ;  // queries per launch: one 32-column MFMA block (LDS use does not depend on the dimension)
    for (int off = 0; off < nq_all; off += per) {
        const int nq = std::min(per, nq_all - off);
        int64_t done = 0;
        const int cap = cap_now(idx);
        int64_t chunk = std::min<int64_t>(idx->chunk0_rows, cap);
        while (done < idx->n) {
            const int64_t end = std::min<int64_t>(idx->n, done + chunk);
            CHECK(scan_range(idx, s, off, nq, k, done, end));
            if (end - done > cap) {  // only a chunk larger than the buffer can overflow
                HIPCHECK(idx, hipMemcpyAsync(idx->status_host, idx->st.status, kQBlockMax * sizeof(int),
                                             hipMemcpyDeviceToHost, s));
                HIPCHECK(idx, hipStreamSynchronize(s));
                std::vector<int> redo;
                for (int j = 0; j < nq; ++j)
                    if (idx->status_host[qs[off + j]] & kStOverflow) redo.push_back(qs[off + j]);
                if (!redo.empty()) {
                    // re-run this range for the overflowed queries in buffer-sized pieces (cannot overflow);
                    // the other queries of the group already committed it.  Uses the tail of qlist_dev.
                    const int roff = kQBlockMax;
                    HIPCHECK(idx, hipMemcpyAsync(idx->qlist_dev + roff, redo.data(), redo.size() * sizeof(int),
                                                 hipMemcpyHostToDevice, s));
                    // clear the overflow bit but keep the kept list (prune did not commit the failed chunk)
                    for (int q : redo) idx->status_host[q] &= ~kStOverflow;
                    for (int q : redo)
                        HIPCHECK(idx, hipMemcpyAsync(idx->st.status + q, idx->status_host + q, sizeof(int),
                                                     hipMemcpyHostToDevice, s));
                    for (int64_t p = done; p < end; p += cap)
                        CHECK(scan_range(idx, s, roff, (int)redo.size(), k, p, std::min<int64_t>(end, p + cap)));
                }
            }
            done = end;
            // exact keys: a chunk only appends the rows that enter the running top-k (k ln(ratio) of them on unordered data),
            // so the ladder can be steep (x64) -- 3 launches and prunes for 2 M rows instead of 7 (an adversarial order overflows
            // the list and takes the buffer-sized re-run above)
            chunk = std::max<int64_t>(chunk, done * (idx->chunk_growth_set ? idx->chunk_growth : 63));
        }
    }
    idx->s_fallback_queries += nq_all;
    return MI355DR_OK;
}

// rows `map[j]` of src -> row j of dst (d floats each)
__global__ void k_gather_queries(const float* src, const int* map, int d, float* dst) {
    const int j = blockIdx.x;
    for (int c = threadIdx.x; c < d; c += blockDim.x) dst[(int64_t)j * d + c] = src[(int64_t)map[j] * d + c];
}
// result j of the re-screened sub-block -> slot map[j] of the block's outputs
__global__ void k_scatter_results(const double* sd, const int64_t* sr, const int* map, int k, double* od, int64_t* orow) {
    const int j = blockIdx.x;
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        od[(int64_t)map[j] * k + i] = sd[(int64_t)j * k + i];
        orow[(int64_t)map[j] * k + i] = sr[(int64_t)j * k + i];
    }
}

// ---- a block in flight -------------------------------------------------------------------------------------------------
// Round 3: a block no longer ends with a host synchronisation.  enqueue_block() puts the whole pass on the stream -- prepare,
// starter, chunks, finalize, a copy of the per-query status words into the block's own pinned buffer, an event -- and returns;
// complete_block() waits for the event and only then looks at the status: queries that overflowed a candidate list are
// re-screened (tighter bound, slower growth), queries the screen cannot rank are recomputed by the exact scan, each as a
// sub-block of its own whose results are scattered into the block's outputs.  Between the two calls the caller may enqueue
// the NEXT block (same stream): the GPU goes from one block's last kernel to the next one's first without waiting for the
// host's round trip (~50 us per block: 0.6 % of the 10 M-row pass, 4 % at the 8-way shard size).  The per-search state is
// single: blocks follow each other in stream order, and a fix-up (enqueued behind whatever is in flight) gathers its
// queries from the CALLER's buffer, which therefore stays valid until the wait.
int pending_alloc(mi355dr_index* idx, Pending& p) {
    if (!p.status_host) HIPCHECK(idx, hipHostMalloc(&p.status_host, (kQBlockMax + 1) * sizeof(int)));
    if (!p.done) HIPCHECK(idx, hipEventCreateWithFlags(&p.done, hipEventDisableTiming));
    return MI355DR_OK;
}

int enqueue_block(mi355dr_index* idx, hipStream_t s, const float* q_dev, int B, int k, double* out_dist_dev,
                  int64_t* out_rows_dev, Pending& p) {
    CHECK(ensure_qstate(idx));
    CHECK(pending_alloc(idx, p));
    idx->k_now = k;
    // a demoted int8 screen (AUTO) is on probation: first attempts at a demoted k count it down, then int8 gets another try
    if (idx->retry_level == 0 && idx->screen_dtype == MI355DR_SCREEN_AUTO && k >= idx->i8_demoted_k && --idx->i8_probation <= 0)
        idx->i8_demoted_k = INT_MAX;
    const int Bpad = (int)round_up(B, screen_tile(B));
    if (idx->screen_dtype == MI355DR_SCREEN_I8 && !i8_available(idx) && idx->path != MI355DR_PATH_SCAN)
        return fail(idx, MI355DR_E_UNSUPPORTED, "int8 screen unavailable: too many rows outside the residual limit");
    // (inner product rides the same cosine screens: thresholds become cos >= dot_k / (|q| cmax), see k_prune)
    const bool screen_possible = use_i8(idx) ? i8_available(idx) : idx->irr_n <= kIrrCap;
    const bool use_screen = idx->n > 0 && screen_possible && idx->path != MI355DR_PATH_SCAN;
    if (idx->path == MI355DR_PATH_SCREEN && !screen_possible && idx->n > 0)
        return fail(idx, MI355DR_E_UNSUPPORTED, "screen path unavailable (metric or too many irregular rows)");
    if (q_dev != idx->qdev)
        HIPCHECK(idx, hipMemcpyAsync(idx->qdev, q_dev, (size_t)B * idx->dim * sizeof(float), hipMemcpyDeviceToDevice, s));
    PassPlan plan;
    if (use_screen) plan = make_plan(idx, B, k);
    // (also re-arms the status word and the prune's hand-over counters, and starts the lists at the starter's count)
    CHECK(launch_prep(idx, s, B, Bpad, idx->metric, starter_count(plan)));
    if (idx->n > 0) {
        if (use_screen) {
            CHECK(run_screen(idx, s, B, k, plan));
        } else {
            std::vector<int> all(B);
            for (int i = 0; i < B; ++i) all[i] = i;
            CHECK(run_scan(idx, s, all, k));  // (blocks on the host only where a chunk larger than the buffer overflowed)
        }
    }
    hipLaunchKernelGGL(k_finalize, dim3(B), dim3(64), 0, s, idx->st, k, idx->row_offset, out_dist_dev, out_rows_dev,
                       idx->status_or_dev);
    HIPCHECK(idx, hipGetLastError());
    // (one copy: the per-query words and, behind them, their OR)
    HIPCHECK(idx, hipMemcpyAsync(p.status_host, idx->st.status, (size_t)(kQBlockMax + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHECK(idx, hipEventRecord(p.done, s));
    p.active = true;
    p.stream = s;
    p.q_dev = q_dev;
    p.B = B;
    p.k = k;
    p.out_dist = out_dist_dev;
    p.out_rows = out_rows_dev;
    p.used_screen = use_screen;
    p.was_i8 = use_screen && use_i8(idx);
    p.level = idx->retry_level;
    return MI355DR_OK;
}

int search_block(mi355dr_index* idx, hipStream_t s, const float* q_dev, int B, int k, double* out_dist_dev,
                 int64_t* out_rows_dev);

int complete_block(mi355dr_index* idx, Pending& p) {
    if (!p.active) return MI355DR_OK;
    p.active = false;
    HIPCHECK(idx, hipEventSynchronize(p.done));
    drain_events(idx);
    if (!p.used_screen || p.status_host[kQBlockMax] == 0) return MI355DR_OK;
    // some query overflowed its candidate buffer or has an irregular norm
    hipStream_t s = p.stream;
    const int B = p.B, k = p.k, level = p.level;
    std::vector<int> sub;  // [todo ... | retry ...]
    int n_todo = 0;
    {
        std::vector<int> retry;
        for (int i = 0; i < B; ++i) {
            const int st = p.status_host[i];
            if (st == 0) continue;
            // an overflow at the first attempt is re-screened with the tighter bound and slower growth; whatever
            // overflows again, and every query the screen cannot rank (irregular norm), is recomputed exactly
            if (level < kRetryLevels && !(st & kStIrregular)) retry.push_back(i);
            else sub.push_back(i);
        }
        n_todo = (int)sub.size();
        sub.insert(sub.end(), retry.begin(), retry.end());
    }
    const int n_sub = (int)sub.size(), n_retry = n_sub - n_todo;
    if (n_sub == 0) return MI355DR_OK;
    if (!idx->retry_q[level]) {
        HIPCHECK(idx, hipMalloc(&idx->retry_q[level], (size_t)kQBlockMax * idx->dim * sizeof(float)));
        HIPCHECK(idx, hipMalloc(&idx->retry_dist[level], (size_t)kQBlockMax * kKMax * sizeof(double)));
        HIPCHECK(idx, hipMalloc(&idx->retry_rows[level], (size_t)kQBlockMax * kKMax * sizeof(int64_t)));
        HIPCHECK(idx, hipMalloc(&idx->retry_map[level], (size_t)kQBlockMax * sizeof(int)));
    }
    // both sub-blocks' queries are gathered from the caller's buffer BEFORE either runs (a nested search overwrites qdev)
    HIPCHECK(idx, hipMemcpyAsync(idx->retry_map[level], sub.data(), n_sub * sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_gather_queries, dim3(n_sub), dim3(128), 0, s, p.q_dev, idx->retry_map[level], idx->dim,
                       idx->retry_q[level]);
    HIPCHECK(idx, hipGetLastError());
    HIPCHECK(idx, hipStreamSynchronize(s));  // `sub` (pageable) was read by the copy
    const int saved_level = idx->retry_level, saved_path = idx->path;
    int rc = MI355DR_OK;
    if (n_todo > 0) {  // guaranteed exact path
        idx->retry_level = level + 1;  // (buffers of the next level; the scan itself never re-screens)
        idx->path = MI355DR_PATH_SCAN;
        rc = search_block(idx, s, idx->retry_q[level], n_todo, k, idx->retry_dist[level], idx->retry_rows[level]);
        idx->path = saved_path;
        idx->retry_level = saved_level;
        CHECK(rc);
    }
    if (n_retry > 0) {
        idx->s_retry_queries += n_retry;
        // AUTO gives the int8 screen up (from this k upwards) when more than 1 % of a block overflowed under its bound: the
        // re-screen is a bf16 pass of its own, and lists that overflow are lists that cost -- at d = 2048 (the int8 bound is
        // absolute, ~0.0175, the spread of the scores shrinks like 1 / sqrt(d)) 1.5 % of the queries overflowed and the pass
        // took 17.1 ms against 12.9 on bf16; at d = 768 nothing overflows.  (Round 2: 5 %.)
        // Not for ever -- a burst of queries into one dense neighbourhood must not cost a Gaussian-like corpus its int8 screen
        // (7.4 against 12.8 ms per block at the headline size): after 16 more blocks at such a k int8 gets another try, and the
        // wait doubles (up to 4096 blocks) whenever that try overflows again.
        if (p.was_i8 && idx->screen_dtype == MI355DR_SCREEN_AUTO && n_retry * 100 > B) {
            idx->i8_demoted_k = std::min(idx->i8_demoted_k, k);
            idx->i8_backoff = idx->i8_backoff == 0 ? 16 : std::min(idx->i8_backoff * 2, 4096);
            idx->i8_probation = idx->i8_backoff;
        }
        idx->retry_level = level + 1;
        rc = search_block(idx, s, idx->retry_q[level] + (size_t)n_todo * idx->dim, n_retry, k,
                          idx->retry_dist[level] + (size_t)n_todo * k, idx->retry_rows[level] + (size_t)n_todo * k);
        idx->retry_level = saved_level;
        CHECK(rc);
    }
    hipLaunchKernelGGL(k_scatter_results, dim3(n_sub), dim3(128), 0, s, idx->retry_dist[level], idx->retry_rows[level],
                       idx->retry_map[level], k, p.out_dist, p.out_rows);
    HIPCHECK(idx, hipGetLastError());
    HIPCHECK(idx, hipStreamSynchronize(s));
    return MI355DR_OK;
}

// blocks still in flight (mi355dr_search_device_async) are finished before anything that is not the next block on the same
// stream touches the per-search state or the corpus buffers
int drain_pending(mi355dr_index* idx) {
    int rc = MI355DR_OK;
    while (idx->seq_done < idx->seq_next) {
        const int r = complete_block(idx, idx->pend[idx->seq_done % kPendingRing]);
        if (rc == MI355DR_OK) rc = r;
        idx->seq_done++;
    }
    return rc;
}

// one block of B <= kQBlockMax device-resident queries -> device outputs [B,k], complete on return
int search_block(mi355dr_index* idx, hipStream_t s, const float* q_dev, int B, int k, double* out_dist_dev,
                 int64_t* out_rows_dev) {
    Pending& p = idx->sub_pend[std::min(idx->retry_level, kRetryLevels + 1)];
    CHECK(enqueue_block(idx, s, q_dev, B, k, out_dist_dev, out_rows_dev, p));
    return complete_block(idx, p);
}

// k_merge_topk sorts next_pow2(world * k) entries: LDS and threads by need (80 entries at 8 ranks x k = 10 -- a 256-thread
// workgroup with 48 KiB of LDS per query kept the merge from slipping in beside other work)
inline size_t merge_lds(int world, int k) {
    size_t np = 1;
    while (np < (size_t)world * k) np <<= 1;
    return np * 12;
}
inline int merge_threads(int world, int k) { return (int64_t)world * k <= 128 ? 64 : 256; }

int check_search_args(mi355dr_index* idx, const void* q, int B, int k, const void* od, const void* orow) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    if (B < 0 || k <= 0) return fail(idx, MI355DR_E_INVALID, "B must be >= 0 and k > 0");
    if (k > kKMax) return fail(idx, MI355DR_E_UNSUPPORTED, "k exceeds 1024");
    if (B > 0 && (!q || !od || !orow)) return fail(idx, MI355DR_E_INVALID, "null buffer");
    return MI355DR_OK;
}

}  // namespace

extern "C" {

int mi355dr_version(void) { return 100; }

const char* mi355dr_last_error(const mi355dr_index* idx) {
    if (idx) return idx->err.c_str();
    std::lock_guard<std::mutex> g(g_err_mu);
    return g_err.c_str();
}

int mi355dr_create(mi355dr_index** out, int device_id, int dim, int metric) {
    if (!out) return fail(nullptr, MI355DR_E_INVALID, "out is null");
    *out = nullptr;
    if (dim <= 0 || dim > 16384) return fail(nullptr, MI355DR_E_INVALID, "dim must be in [1,16384]");
    if (metric != MI355DR_METRIC_COSINE && metric != MI355DR_METRIC_IP)
        return fail(nullptr, MI355DR_E_INVALID, "metric must be 0 (cosine) or 1 (inner product)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, MI355DR_E_HIP, "no HIP device available (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, MI355DR_E_INVALID, "device_id out of range");
    mi355dr_index* idx = new mi355dr_index();
    idx->device = device_id;
    idx->dim = dim;
    idx->dpad = (int)round_up(dim, kStepK);
    idx->dpad8 = (int)round_up(dim, kRowB);
    idx->metric = metric;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&idx->irr_rows, kIrrCap * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&idx->irr_count, sizeof(int));
    if (e == hipSuccess) e = hipMemset(idx->irr_count, 0, sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&idx->n2max_dev, sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(idx->n2max_dev, 0, sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc(&idx->bf16_res2_dev, sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(idx->bf16_res2_dev, 0, sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc(&idx->irr8_rows, kIrrCap * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&idx->irr8_count, sizeof(int));
    if (e == hipSuccess) e = hipMemset(idx->irr8_count, 0, sizeof(int));
    if (e == hipSuccess) e = hipEventCreate(&idx->t0);
    if (e == hipSuccess) e = hipEventCreate(&idx->t1);
    if (e != hipSuccess) {
        std::string m = std::string("device setup failed: ") + hipGetErrorString(e);
        delete idx;
        return fail(nullptr, MI355DR_E_HIP, m);
    }
    *out = idx;
    return MI355DR_OK;
}

void mi355dr_destroy(mi355dr_index* idx) {
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    (void)drain_pending(idx);
    if (idx->stream) (void)hipStreamSynchronize(idx->stream);
    for (Pending* pp : {&idx->pend[0], &idx->pend[1], &idx->pend[2], &idx->pend[3], &idx->sub_pend[0], &idx->sub_pend[1],
                        &idx->sub_pend[2], &idx->sub_pend[3]}) {
        if (pp->done) (void)hipEventDestroy(pp->done);
        if (pp->status_host) (void)hipHostFree(pp->status_host);
    }
    void* ptrs[] = {idx->retry_q[2], idx->retry_dist[2], idx->retry_rows[2], idx->retry_map[2],
                    idx->n2max_dev, idx->bf16_res2_dev, idx->retry_q[0], idx->retry_dist[0], idx->retry_rows[0], idx->retry_map[0], idx->retry_q[1],
                    idx->retry_dist[1], idx->retry_rows[1], idx->retry_map[1], idx->shadow8, idx->flag8, idx->grp8, idx->irr8_rows, idx->irr8_count, idx->st.E, idx->st.E16, idx->st.sc, idx->st.kq,
                    idx->st.qhat8, idx->st.carry,
                    idx->rows, idx->shadow, idx->nrm2, idx->irr_rows, idx->irr_count, idx->st.qn, idx->st.qhat,
                    idx->st.thr, idx->st.cnt, idx->st.best_n, idx->st.best_key, idx->st.best_row, idx->st.thr_key,
                    idx->st.thr_row, idx->st.status, idx->qdev, idx->cand_row, idx->cand_val, idx->qlist_dev,
                    idx->out_dist_dev, idx->out_rows_dev, idx->stat_dev, idx->prune_skip, idx->rq_progress, idx->park_thr};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    multivec_destroy(idx);
    comm_destroy(idx);
    if (idx->status_host) (void)hipHostFree(idx->status_host);
    for (auto& p : idx->ev_pool) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (auto& p : idx->ev_pending) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (hipEvent_t e : idx->ms_ev)
        if (e) (void)hipEventDestroy(e);
    if (idx->t0) (void)hipEventDestroy(idx->t0);
    if (idx->t1) (void)hipEventDestroy(idx->t1);
    if (idx->stream) (void)hipStreamDestroy(idx->stream);
    delete idx;
}

int mi355dr_reserve(mi355dr_index* idx, int64_t n_rows) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    HIPCHECK(idx, hipSetDevice(idx->device));
    if (n_rows < 0) return fail(idx, MI355DR_E_INVALID, "negative row count");
    if (n_rows >= (int64_t)1 << 31) return fail(idx, MI355DR_E_UNSUPPORTED, "more than 2^31-1 rows per index");
    CHECK(drain_pending(idx));
    return ensure_capacity(idx, n_rows);
}

static int add_rows_impl(mi355dr_index* idx, const float* rows, int64_t n, hipMemcpyKind kind) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (n < 0) return fail(idx, MI355DR_E_INVALID, "negative row count");
    if (n == 0) return MI355DR_OK;
    if (!rows) return fail(idx, MI355DR_E_INVALID, "rows is null");
    if (idx->n + n >= (int64_t)1 << 31) return fail(idx, MI355DR_E_UNSUPPORTED, "more than 2^31-1 rows per index");
    HIPCHECK(idx, hipSetDevice(idx->device));
    CHECK(drain_pending(idx));
    CHECK(ensure_capacity(idx, idx->n + n));
    hipStream_t s = idx->stream;
    HIPCHECK(idx, hipMemcpyAsync(idx->rows + idx->n * idx->dim, rows, (size_t)n * idx->dim * sizeof(float), kind, s));
    // the per-row build kernels run one workgroup per row: a grid is kept below 2^22 rows (gridDim.x * blockDim.x < 2^32)
    constexpr int64_t kBuildSlice = (int64_t)1 << 22;
    for (int64_t r0 = 0; r0 < n; r0 += kBuildSlice) {
        const int64_t m = std::min(kBuildSlice, n - r0), first = idx->n + r0;
        hipLaunchKernelGGL(k_row_nrm2, dim3((unsigned)((m + kWave - 1) / kWave)), dim3(kWave), 0, s, idx->rows, first, m,
                           idx->dim, idx->nrm2, idx->n2max_dev);
        HIPCHECK(idx, hipGetLastError());
        hipLaunchKernelGGL(k_build_shadow, dim3((unsigned)m), dim3(256), 0, s, idx->rows, idx->nrm2, first, m, idx->dim,
                           idx->dpad, idx->shadow, idx->irr_rows, idx->irr_count, idx->bf16_res2_dev,
                           idx->metric == MI355DR_METRIC_IP ? 1 : 0);
        HIPCHECK(idx, hipGetLastError());
    }
    {   // int8 shadow: whole groups of 32 rows, from the (possibly partly filled) group the first new row falls into
        const int64_t g_lo = idx->n / kI8GroupRows, g_hi = (idx->n + n + kI8GroupRows - 1) / kI8GroupRows;
        for (int64_t g0 = g_lo; g0 < g_hi; g0 += kBuildSlice) {
            const int64_t m = std::min(kBuildSlice, g_hi - g0);
            hipLaunchKernelGGL(k_build_shadow8, dim3((unsigned)m), dim3(256), 0, s, idx->rows, idx->nrm2, g0, idx->n + n,
                               idx->n, idx->dim, idx->dpad8, idx->shadow8, idx->flag8, idx->grp8, idx->irr8_rows,
                               idx->irr8_count, idx->metric == MI355DR_METRIC_IP ? 1 : 0);
            HIPCHECK(idx, hipGetLastError());
        }
    }
    int irr = 0, irr8 = 0;
    HIPCHECK(idx, hipMemcpyAsync(&irr, idx->irr_count, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHECK(idx, hipMemcpyAsync(&irr8, idx->irr8_count, sizeof(int), hipMemcpyDeviceToHost, s));
    float res2 = 0.0f;
    HIPCHECK(idx, hipMemcpyAsync(&res2, idx->bf16_res2_dev, sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHECK(idx, hipStreamSynchronize(s));
    float n2max = 0.0f;
    HIPCHECK(idx, hipMemcpy(&n2max, idx->n2max_dev, sizeof(float), hipMemcpyDeviceToHost));
    idx->cmax = std::sqrt(n2max) * 1.000001f;
    // (a-priori cap: 2^-8 |c_hat|; inner product: the shadow holds the rows themselves, residuals in their units)
    idx->bf16_ec = std::min(std::sqrt(res2) * 1.001f, 0.00390625f * 1.0001f * (idx->metric == MI355DR_METRIC_IP ? idx->cmax : 1.0f));
    idx->irr_n = irr;
    idx->irr8_n = irr8;
    idx->n += n;
    return MI355DR_OK;
}

int mi355dr_add_rows(mi355dr_index* idx, const float* rows, int64_t n) {
    return add_rows_impl(idx, rows, n, hipMemcpyHostToDevice);
}
int mi355dr_add_rows_device(mi355dr_index* idx, const float* rows_dev, int64_t n) {
    return add_rows_impl(idx, rows_dev, n, hipMemcpyDeviceToDevice);
}

int64_t mi355dr_size(const mi355dr_index* idx) { return idx ? idx->n : -1; }
int mi355dr_dim(const mi355dr_index* idx) { return idx ? idx->dim : -1; }

int mi355dr_get_rows(mi355dr_index* idx, int64_t row0, int64_t n, float* out) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (row0 < 0 || n < 0 || row0 + n > idx->n || (n > 0 && !out)) return fail(idx, MI355DR_E_INVALID, "bad row range");
    if (n == 0) return MI355DR_OK;
    HIPCHECK(idx, hipSetDevice(idx->device));
    HIPCHECK(idx, hipMemcpyAsync(out, idx->rows + row0 * idx->dim, (size_t)n * idx->dim * sizeof(float),
                                 hipMemcpyDeviceToHost, idx->stream));
    HIPCHECK(idx, hipStreamSynchronize(idx->stream));
    return MI355DR_OK;
}

int mi355dr_search(mi355dr_index* idx, const float* queries, int B, int k, double* out_dist, int64_t* out_rows) {
    CHECK(check_search_args(idx, queries, B, k, out_dist, out_rows));
    std::lock_guard<std::mutex> g(idx->mu);
    HIPCHECK(idx, hipSetDevice(idx->device));
    CHECK(drain_pending(idx));
    CHECK(ensure_qstate(idx));
    hipStream_t s = idx->stream;
    for (int b0 = 0; b0 < B; b0 += kQBlockMax) {
        const int nb = std::min(kQBlockMax, B - b0);
        HIPCHECK(idx, hipMemcpyAsync(idx->qdev, queries + (int64_t)b0 * idx->dim, (size_t)nb * idx->dim * sizeof(float),
                                     hipMemcpyHostToDevice, s));
        CHECK(search_block(idx, s, idx->qdev, nb, k, idx->out_dist_dev, idx->out_rows_dev));
        HIPCHECK(idx, hipMemcpyAsync(out_dist + (int64_t)b0 * k, idx->out_dist_dev, (size_t)nb * k * sizeof(double),
                                     hipMemcpyDeviceToHost, s));
        HIPCHECK(idx, hipMemcpyAsync(out_rows + (int64_t)b0 * k, idx->out_rows_dev, (size_t)nb * k * sizeof(int64_t),
                                     hipMemcpyDeviceToHost, s));
        HIPCHECK(idx, hipStreamSynchronize(s));
    }
    return MI355DR_OK;
}

static int search_device_async_locked(mi355dr_index* idx, const float* queries_dev, int B, int k, double* out_dist_dev,
                                      int64_t* out_rows_dev, hipStream_t s, int64_t* ticket) {
    // a block in flight on ANOTHER stream is finished first: the per-search state is shared and only stream order protects it
    if (idx->seq_done < idx->seq_next && idx->pend[(idx->seq_next - 1) % kPendingRing].stream != s) CHECK(drain_pending(idx));
    for (int b0 = 0; b0 < B; b0 += kQBlockMax) {
        const int nb = std::min(kQBlockMax, B - b0);
        if (idx->seq_next - idx->seq_done >= kPendingRing) {  // the ring is full: finish the oldest block
            CHECK(complete_block(idx, idx->pend[idx->seq_done % kPendingRing]));
            idx->seq_done++;
        }
        Pending& p = idx->pend[idx->seq_next % kPendingRing];
        CHECK(enqueue_block(idx, s, queries_dev + (int64_t)b0 * idx->dim, nb, k, out_dist_dev + (int64_t)b0 * k,
                            out_rows_dev + (int64_t)b0 * k, p));
        idx->seq_next++;
    }
    if (ticket) *ticket = idx->seq_next;  // every block below this sequence number belongs to (or precedes) this call
    return MI355DR_OK;
}

int mi355dr_search_device_async(mi355dr_index* idx, const float* queries_dev, int B, int k, double* out_dist_dev,
                                int64_t* out_rows_dev, void* stream, int64_t* ticket) {
    CHECK(check_search_args(idx, queries_dev, B, k, out_dist_dev, out_rows_dev));
    if (!ticket) return fail(idx, MI355DR_E_INVALID, "ticket is null");
    std::lock_guard<std::mutex> g(idx->mu);
    HIPCHECK(idx, hipSetDevice(idx->device));
    return search_device_async_locked(idx, queries_dev, B, k, out_dist_dev, out_rows_dev,
                                      stream ? (hipStream_t)stream : idx->stream, ticket);
}

int mi355dr_search_wait(mi355dr_index* idx, int64_t ticket) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (ticket < 0 || ticket > idx->seq_next) return fail(idx, MI355DR_E_INVALID, "unknown ticket");
    HIPCHECK(idx, hipSetDevice(idx->device));
    int rc = MI355DR_OK;
    while (idx->seq_done < ticket) {
        const int r = complete_block(idx, idx->pend[idx->seq_done % kPendingRing]);
        if (rc == MI355DR_OK) rc = r;
        idx->seq_done++;
    }
    return rc;
}

int mi355dr_search_device(mi355dr_index* idx, const float* queries_dev, int B, int k, double* out_dist_dev,
                          int64_t* out_rows_dev, void* stream) {
    CHECK(check_search_args(idx, queries_dev, B, k, out_dist_dev, out_rows_dev));
    std::lock_guard<std::mutex> g(idx->mu);
    HIPCHECK(idx, hipSetDevice(idx->device));
    int64_t ticket = 0;
    CHECK(search_device_async_locked(idx, queries_dev, B, k, out_dist_dev, out_rows_dev,
                                     stream ? (hipStream_t)stream : idx->stream, &ticket));
    return drain_pending(idx);
}

int mi355dr_merge_topk_device(mi355dr_index* idx, const double* dist_all_dev, const int64_t* rows_all_dev, int world,
                              int B, int k, double* out_dist_dev, int64_t* out_rows_dev, void* stream) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (world <= 0 || B < 0 || k <= 0) return fail(idx, MI355DR_E_INVALID, "bad merge shape");
    if ((int64_t)world * k > kSortMax) return fail(idx, MI355DR_E_UNSUPPORTED, "world*k exceeds 4096");
    if (B == 0) return MI355DR_OK;
    HIPCHECK(idx, hipSetDevice(idx->device));
    CHECK(ensure_qstate(idx));
    hipStream_t s = stream ? (hipStream_t)stream : idx->stream;
    hipLaunchKernelGGL(k_merge_topk, dim3(B), dim3(merge_threads(world, k)), merge_lds(world, k), s, dist_all_dev, rows_all_dev,
                       (int64_t)B * k, world, B, k, out_dist_dev, out_rows_dev);
    HIPCHECK(idx, hipGetLastError());
    return MI355DR_OK;  // asynchronous on `s`
}

int mi355dr_pack_topk_device(mi355dr_index* idx, const double* dist_dev, const int64_t* rows_dev, int B, int k,
                             int64_t* packed_dev, void* stream) {
    if (!idx || !dist_dev || !rows_dev || !packed_dev) return fail(idx, MI355DR_E_INVALID, "null argument");
    std::lock_guard<std::mutex> g(idx->mu);
    if (B < 0 || k <= 0) return fail(idx, MI355DR_E_INVALID, "bad shape");
    HIPCHECK(idx, hipSetDevice(idx->device));
    hipStream_t s = stream ? (hipStream_t)stream : idx->stream;
    const size_t bytes = (size_t)B * k * 8;
    if (bytes) {
        HIPCHECK(idx, hipMemcpyAsync(packed_dev, dist_dev, bytes, hipMemcpyDeviceToDevice, s));
        HIPCHECK(idx, hipMemcpyAsync(packed_dev + (size_t)B * k, rows_dev, bytes, hipMemcpyDeviceToDevice, s));
    }
    return MI355DR_OK;
}

int mi355dr_merge_topk_packed_device(mi355dr_index* idx, const int64_t* packed_all_dev, int world, int B, int k,
                                     double* out_dist_dev, int64_t* out_rows_dev, void* stream) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (world <= 0 || B < 0 || k <= 0) return fail(idx, MI355DR_E_INVALID, "bad merge shape");
    if ((int64_t)world * k > kSortMax) return fail(idx, MI355DR_E_UNSUPPORTED, "world*k exceeds 4096");
    if (B == 0) return MI355DR_OK;
    HIPCHECK(idx, hipSetDevice(idx->device));
    CHECK(ensure_qstate(idx));
    hipStream_t s = stream ? (hipStream_t)stream : idx->stream;
    const int64_t plane = (int64_t)B * k;
    hipLaunchKernelGGL(k_merge_topk, dim3(B), dim3(merge_threads(world, k)), merge_lds(world, k), s, (const double*)packed_all_dev,
                       packed_all_dev + plane, 2 * plane, world, B, k, out_dist_dev, out_rows_dev);
    HIPCHECK(idx, hipGetLastError());
    return MI355DR_OK;  // asynchronous on `s`
}

int mi355dr_set_option(mi355dr_index* idx, const char* key, int64_t value) {
    if (!idx || !key) return fail(idx, MI355DR_E_INVALID, "null argument");
    std::lock_guard<std::mutex> g(idx->mu);
    if (idx->seq_done < idx->seq_next) {  // (a fix-up of a block in flight must see the options it was enqueued under)
        HIPCHECK(idx, hipSetDevice(idx->device));
        CHECK(drain_pending(idx));
    }
    const std::string k(key);
    if (k == "path") {
        if (value < 0 || value > 2) return fail(idx, MI355DR_E_INVALID, "path must be 0,1,2");
        idx->path = (int)value;
    } else if (k == "screen_dtype") {
        if (value < 0 || value > 2) return fail(idx, MI355DR_E_INVALID, "screen_dtype must be 0,1,2");
        idx->screen_dtype = (int)value;
        idx->i8_demoted_k = INT_MAX;  // (setting the option again re-arms AUTO)
        idx->i8_backoff = idx->i8_probation = 0;
    } else if (k == "i8_min_budget_x100") {
        if (value < 1 || value > 1000) return fail(idx, MI355DR_E_INVALID, "i8_min_budget_x100 must be in 1..1000");
        idx->i8_min_budget = (double)value / 100.0;
    } else if (k == "maxsim_screen") {
        idx->maxsim_screen = value != 0;
    } else if (k == "maxsim_coop") {
        if (value < -1 || value > 1) return fail(idx, MI355DR_E_INVALID, "maxsim_coop must be -1 (by document length), 0 or 1");
        idx->maxsim_coop = (int)value;
    } else if (k == "maxsim_persistent") {
        idx->maxsim_persistent = value != 0;
    } else if (k == "maxsim_wg") {
        if (value < -1 || value > 2) return fail(idx, MI355DR_E_INVALID, "maxsim_wg must be -1 (by document length), 0, 1 or 2");
        idx->maxsim_wg = (int)value;
    } else if (k == "maxsim_tighten") {
        idx->maxsim_tighten = value != 0;
    } else if (k == "maxsim_aligned") {
        idx->maxsim_aligned = value != 0;
    } else if (k == "maxsim_wg_min") {
        if (value < 8 || value > 9) return fail(idx, MI355DR_E_INVALID, "maxsim_wg_min must be 8 or 9");
        idx->maxsim_wg_min = (int)value;
    } else if (k == "maxsim_wg_pipe") {
        idx->maxsim_wg_pipe = value != 0;
    } else if (k == "maxsim_pack8") {
        if (value < -1 || value > 1) return fail(idx, MI355DR_E_INVALID, "maxsim_pack8 must be -1 (when it pays), 0 or 1");
        idx->maxsim_pack8 = (int)value;
    } else if (k == "maxsim_wg_bps") {
        if (value != 2 && value != 4) return fail(idx, MI355DR_E_INVALID, "maxsim_wg_bps must be 2 or 4");
        idx->maxsim_wg_bps = (int)value;
    } else if (k == "maxsim_pass_groups") {
        if (value < 1 || value > 4) return fail(idx, MI355DR_E_INVALID, "maxsim_pass_groups must be in 1..4");
        idx->maxsim_pass_groups = (int)value;
    } else if (k == "row_offset") {
        idx->row_offset = value;
    } else if (k == "profile") {
        idx->profile = value != 0;
    } else if (k == "chunk0_rows") {
        if (value < 1) return fail(idx, MI355DR_E_INVALID, "chunk0_rows must be >= 1");
        idx->chunk0_rows = value;
        idx->chunk0_set = 1;  // (an explicit first chunk means the emit-all ladder: no starter)
    } else if (k == "starter") {
        idx->starter = value != 0;
    } else if (k == "prune_companion") {
        idx->prune_companion = value != 0;
    } else if (k == "scan_dma") {
        idx->scan_dma = value != 0;
    } else if (k == "defer_round_b") {
        idx->defer_round_b = value != 0;
    } else if (k == "chunk_growth") {
        if (value < 1) return fail(idx, MI355DR_E_INVALID, "chunk_growth must be >= 1");
        idx->chunk_growth = value;
        idx->chunk_growth_set = 1;
    } else if (k == "screen_stream") {
        idx->screen_stream = value != 0;
    } else if (k == "screen_rq") {
        idx->screen_rq = value != 0;
    } else if (k == "debug_park_thresholds") {
        idx->debug_park = value;
    } else if (k == "screen_rq_split_tests") {
        idx->screen_rq_split_tests = value != 0;
    } else if (k == "screen_drift") {
        if (value < 0 || value > 1024) return fail(idx, MI355DR_E_INVALID, "screen_drift: 0 ... 1024 tiles");
        idx->screen_drift = (int)value;
    } else if (k == "small_chunk_rows") {
        if (value < 0) return fail(idx, MI355DR_E_INVALID, "small_chunk_rows must be >= 0");
        idx->small_chunk_rows = value;
    } else if (k == "round_a") {
        if (value < 0 || value > 64) return fail(idx, MI355DR_E_INVALID, "round_a must be in [0,64]");
        idx->round_a = (int)value;
    } else if (k == "prefilter16") {
        idx->prefilter16 = value != 0;
    } else if (k == "cand_cap") {
        if (value < 16 || value > kCandCap) return fail(idx, MI355DR_E_INVALID, "cand_cap must be in [16,2048]");
        idx->cap = (int)value;
        idx->cap_set = 1;
    } else if (k == "prune_wide") {
        idx->prune_wide = value != 0;
    } else if (k == "screen_flush_sync") {
        idx->screen_flush_sync = value != 0;
    } else if (k == "screen_flush_lanes") {
        if (value < 1 || value > 64) return fail(idx, MI355DR_E_INVALID, "screen_flush_lanes must be in [1, 64]");
        idx->screen_flush_lanes = (int)value;
    } else if (k == "screen_flush_alone") {
        if (value < 8 || value > 60) return fail(idx, MI355DR_E_INVALID, "screen_flush_alone must be in [8, 60]");
        idx->screen_flush_alone = (int)value;
    } else if (k == "wide_inflation_x10") {
        if (value < 20 || value > 400) return fail(idx, MI355DR_E_INVALID, "wide_inflation_x10 must be in [20, 400]");
        idx->wide_inflation_x10 = (int)value;
    } else if (k == "chunk_taper_x100") {
        if (value != 0 && (value < 100 || value > 300)) return fail(idx, MI355DR_E_INVALID, "chunk_taper_x100 must be 0 (auto) or in [100, 300]");
        idx->chunk_taper_x100 = (int)value;
    } else if (k == "starter_rows_wide") {
        if (value < 4096 || value > 262144) return fail(idx, MI355DR_E_INVALID, "starter_rows_wide must be in [4096, 262144]");
        idx->starter_rows_wide = value;
    } else {
        return fail(idx, MI355DR_E_INVALID, "unknown option: " + k);
    }
    return MI355DR_OK;
}

int mi355dr_get_stat(mi355dr_index* idx, const char* key, int64_t* out) {
    if (!idx || !key || !out) return fail(idx, MI355DR_E_INVALID, "null argument");
    std::lock_guard<std::mutex> g(idx->mu);
    const std::string k(key);
    if (k == "candidates" || k == "rescored") {
        unsigned long long v[2] = {0, 0};
        if (idx->stat_dev) {
            std::vector<unsigned long long> per(2 * kQBlockMax);
            HIPCHECK(idx, hipSetDevice(idx->device));
            HIPCHECK(idx, hipMemcpy(per.data(), idx->stat_dev, per.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            for (int i = 0; i < kQBlockMax; ++i) {
                v[0] += per[2 * i];
                v[1] += per[2 * i + 1];
            }
        }
        *out = (int64_t)(k == "candidates" ? v[0] : v[1]);
    } else if (k == "screen_launches") *out = idx->s_screen_launches;
    else if (k == "screen_ns") *out = idx->s_screen_ns;
    else if (k == "screen_rows") *out = idx->s_screen_rows;
    else if (k == "screen256_launches") *out = idx->s_big_launches;
    else if (k == "screen_rq_launches") *out = idx->s_rq_launches;
    else if (k == "screen256_ns") *out = idx->s_big_ns;
    else if (k == "screen256_rows") *out = idx->s_big_rows;
    else if (k == "fallback_queries") *out = idx->s_fallback_queries;
    else if (k == "chunks") *out = idx->s_chunks;
    else if (k == "passes") *out = idx->s_passes;
    else if (k == "starters") *out = idx->s_starters;
    else if (k == "retry_queries") *out = idx->s_retry_queries;
    else if (k == "i8_demoted") *out = idx->i8_demoted_k != INT_MAX ? 1 : 0;
    else if (k == "i8_demoted_k") *out = idx->i8_demoted_k == INT_MAX ? 0 : idx->i8_demoted_k;
    else if (k == "maxsim_screened") *out = idx->s_ms_screened;
    else if (k == "maxsim_candidates") *out = idx->s_ms_candidates;
    else if (k == "maxsim_fallbacks") *out = idx->s_ms_fallbacks;
    else if (k == "maxsim_screen_ns") *out = idx->s_ms_screen_ns;
    else if (k == "maxsim_pack_ns") *out = idx->s_ms_pack_ns;
    else if (k == "maxsim_screen_launches") *out = idx->s_ms_screen_launches;
    else if (k == "maxsim_exact_ns") *out = idx->s_ms_exact_ns;
    else if (k == "maxsim_exact_launches") *out = idx->s_ms_exact_launches;
    else if (k == "maxsim_screen_cols") *out = idx->s_ms_screen_cols;
    else if (k == "maxsim_packed_launches") *out = idx->s_ms_packed_launches;
    else if (k == "maxsim_packed_blocks") *out = idx->s_ms_packed_blocks;
    else if (k == "maxsim_packed_built") *out = idx->s_ms_packed_built;
    else if (k == "irregular_rows") *out = idx->irr_n;
    else if (k == "loose_rows") *out = idx->irr8_n;
    else if (k == "screen_dtype_active") *out = use_i8(idx) ? MI355DR_SCREEN_I8 : MI355DR_SCREEN_BF16;
    else if (k == "hbm_bytes_resident")
        *out = idx->cap_rows * ((int64_t)idx->dim * 4 + (int64_t)idx->dpad * 2 + (int64_t)idx->dpad8 + 5) +
               idx->cap_rows / kI8GroupRows * (int64_t)sizeof(I8Group);
    else return fail(idx, MI355DR_E_INVALID, "unknown stat: " + k);
    return MI355DR_OK;
}

int mi355dr_reset_stats(mi355dr_index* idx) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    idx->s_screen_launches = idx->s_screen_ns = idx->s_screen_rows = idx->s_fallback_queries = idx->s_chunks =
        idx->s_passes = idx->s_rq_launches = idx->s_big_launches = idx->s_big_ns = idx->s_big_rows = idx->s_starters = 0;
    idx->s_ms_screened = idx->s_ms_candidates = idx->s_ms_fallbacks = idx->s_retry_queries = 0;
    idx->s_ms_screen_ns = idx->s_ms_screen_launches = idx->s_ms_exact_ns = idx->s_ms_exact_launches = idx->s_ms_screen_cols = idx->s_ms_pack_ns = idx->s_ms_packed_launches = 0;
    if (idx->stat_dev) {
        HIPCHECK(idx, hipSetDevice(idx->device));
        HIPCHECK(idx, hipMemset(idx->stat_dev, 0, 2 * kQBlockMax * sizeof(unsigned long long)));
    }
    return MI355DR_OK;
}

int mi355dr_timer_start(mi355dr_index* idx) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    HIPCHECK(idx, hipSetDevice(idx->device));
    HIPCHECK(idx, hipEventRecord(idx->t0, idx->stream));
    return MI355DR_OK;
}
int mi355dr_timer_stop(mi355dr_index* idx, double* elapsed_ms) {
    if (!idx || !elapsed_ms) return fail(idx, MI355DR_E_INVALID, "null argument");
    HIPCHECK(idx, hipSetDevice(idx->device));
    HIPCHECK(idx, hipEventRecord(idx->t1, idx->stream));
    HIPCHECK(idx, hipEventSynchronize(idx->t1));
    float ms = 0.f;
    HIPCHECK(idx, hipEventElapsedTime(&ms, idx->t0, idx->t1));
    *elapsed_ms = ms;
    return MI355DR_OK;
}
int mi355dr_synchronize(mi355dr_index* idx) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    HIPCHECK(idx, hipSetDevice(idx->device));
    HIPCHECK(idx, hipStreamSynchronize(idx->stream));
    return MI355DR_OK;
}

int mi355dr_dev_alloc(mi355dr_index* idx, size_t bytes, void** out) {
    if (!idx || !out) return fail(idx, MI355DR_E_INVALID, "null argument");
    HIPCHECK(idx, hipSetDevice(idx->device));
    HIPCHECK(idx, hipMalloc(out, bytes ? bytes : 1));
    return MI355DR_OK;
}
int mi355dr_dev_free(mi355dr_index* idx, void* p) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    HIPCHECK(idx, hipSetDevice(idx->device));
    if (p) HIPCHECK(idx, hipFree(p));
    return MI355DR_OK;
}
int mi355dr_dev_upload(mi355dr_index* idx, void* dst_dev, const void* src_host, size_t bytes) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    HIPCHECK(idx, hipSetDevice(idx->device));
    HIPCHECK(idx, hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return MI355DR_OK;
}
int mi355dr_dev_download(mi355dr_index* idx, void* dst_host, const void* src_dev, size_t bytes) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    HIPCHECK(idx, hipSetDevice(idx->device));
    HIPCHECK(idx, hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return MI355DR_OK;
}

int mi355dr_debug_screen_dense(mi355dr_index* idx, const float* queries, int B, int64_t row0, int64_t n, float* out_t) {
    if (!idx || !queries || !out_t) return fail(idx, MI355DR_E_INVALID, "null argument");
    std::lock_guard<std::mutex> g(idx->mu);
    if (B <= 0 || B > kQBlockMax || n <= 0 || n > kCandCap || row0 < 0 || row0 % screen_tile(B) != 0 || row0 + n > idx->n)
        return fail(idx, MI355DR_E_INVALID,
                    "debug_screen_dense: need 1<=B<=1024, 1<=n<=2048, row0 a multiple of the tile (128; 256 if B>128)");
    HIPCHECK(idx, hipSetDevice(idx->device));
    CHECK(drain_pending(idx));
    CHECK(ensure_qstate(idx));
    hipStream_t s = idx->stream;
    HIPCHECK(idx, hipMemcpyAsync(idx->qdev, queries, (size_t)B * idx->dim * sizeof(float), hipMemcpyHostToDevice, s));
    const int Bpad = (int)round_up(B, screen_tile(B));
    CHECK(launch_prep(idx, s, B, Bpad, /*metric=*/2));  // test hook: thresholds at -inf for every query
    CHECK(launch_screen(idx, s, B, row0, row0 + n, kCandCap, /*emit_mode=*/0));
    std::vector<int> cnt(B);
    std::vector<int32_t> crow((size_t)B * kCandCap);
    std::vector<float> cval((size_t)B * kCandCap);
    HIPCHECK(idx, hipMemcpyAsync(cnt.data(), idx->st.cnt, B * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHECK(idx, hipMemcpyAsync(crow.data(), idx->cand_row, crow.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIPCHECK(idx, hipMemcpyAsync(cval.data(), idx->cand_val, cval.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    std::vector<uint8_t> flag(n, 0);  // int8 screen: rows outside the shadow carry a stale 0 (k_prune drops them)
    if (use_i8(idx)) HIPCHECK(idx, hipMemcpyAsync(flag.data(), idx->flag8 + row0, n, hipMemcpyDeviceToHost, s));
    HIPCHECK(idx, hipStreamSynchronize(s));
    for (int64_t i = 0; i < (int64_t)B * n; ++i) out_t[i] = NAN;
    for (int b = 0; b < B; ++b) {
        const int c = std::min(cnt[b], kCandCap);
        for (int j = 0; j < c; ++j) {
            const int64_t r = crow[(size_t)b * kCandCap + j] - row0;
            if (r >= 0 && r < n && !flag[r]) out_t[(int64_t)b * n + r] = cval[(size_t)b * kCandCap + j];
        }
    }
    return MI355DR_OK;
}

int mi355dr_debug_screen_bound(mi355dr_index* idx, const float* queries, int B, float* out_E) {
    if (!idx || !queries || !out_E) return fail(idx, MI355DR_E_INVALID, "null argument");
    std::lock_guard<std::mutex> g(idx->mu);
    if (B <= 0 || B > kQBlockMax) return fail(idx, MI355DR_E_INVALID, "need 1<=B<=1024");
    HIPCHECK(idx, hipSetDevice(idx->device));
    CHECK(drain_pending(idx));
    CHECK(ensure_qstate(idx));
    hipStream_t s = idx->stream;
    HIPCHECK(idx, hipMemcpyAsync(idx->qdev, queries, (size_t)B * idx->dim * sizeof(float), hipMemcpyHostToDevice, s));
    CHECK(launch_prep(idx, s, B, (int)round_up(B, screen_tile(B)), idx->metric));
    HIPCHECK(idx, hipMemcpyAsync(out_E, idx->st.E, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHECK(idx, hipStreamSynchronize(s));
    return MI355DR_OK;
}

int mi355dr_debug_i8_state(mi355dr_index* idx, const float* queries, int B, float* out_sq, float* out_kq, int64_t g0,
                           int64_t n_groups, float* out_step, float* out_err) {
    if (!idx || !queries || !out_sq || !out_kq || !out_step || !out_err) return fail(idx, MI355DR_E_INVALID, "null argument");
    std::lock_guard<std::mutex> g(idx->mu);
    if (B <= 0 || B > kQBlockMax) return fail(idx, MI355DR_E_INVALID, "need 1<=B<=1024");
    if (g0 < 0 || n_groups < 0 || (g0 + n_groups) * kI8GroupRows > idx->cap_rows)
        return fail(idx, MI355DR_E_INVALID, "group range outside the index");
    if (!use_i8(idx)) return fail(idx, MI355DR_E_UNSUPPORTED, "the int8 screen is not active");
    HIPCHECK(idx, hipSetDevice(idx->device));
    CHECK(drain_pending(idx));
    CHECK(ensure_qstate(idx));
    hipStream_t s = idx->stream;
    HIPCHECK(idx, hipMemcpyAsync(idx->qdev, queries, (size_t)B * idx->dim * sizeof(float), hipMemcpyHostToDevice, s));
    CHECK(launch_prep(idx, s, B, (int)round_up(B, screen_tile(B)), idx->metric));
    HIPCHECK(idx, hipMemcpyAsync(out_sq, idx->st.sc, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHECK(idx, hipMemcpyAsync(out_kq, idx->st.kq, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, s));
    std::vector<I8Group> gr((size_t)n_groups);
    if (n_groups > 0)
        HIPCHECK(idx, hipMemcpyAsync(gr.data(), idx->grp8 + g0, (size_t)n_groups * sizeof(I8Group), hipMemcpyDeviceToHost, s));
    HIPCHECK(idx, hipStreamSynchronize(s));
    for (int64_t i = 0; i < n_groups; ++i) {
        out_step[i] = gr[(size_t)i].step;
        out_err[i] = gr[(size_t)i].err;
    }
    return MI355DR_OK;
}

int mi355dr_debug_rescore(mi355dr_index* idx, const float* queries, int B, const int32_t* pair_q,
                          const int64_t* pair_row, int64_t n_pairs, float* out_dot, double* out_dist) {
    if (!idx || !queries || !pair_q || !pair_row || !out_dot || !out_dist)
        return fail(idx, MI355DR_E_INVALID, "null argument");
    std::lock_guard<std::mutex> g(idx->mu);
    if (B <= 0 || B > kQBlockMax || n_pairs <= 0) return fail(idx, MI355DR_E_INVALID, "bad shape");
    for (int64_t i = 0; i < n_pairs; ++i)
        if (pair_q[i] < 0 || pair_q[i] >= B || pair_row[i] < 0 || pair_row[i] >= idx->n)
            return fail(idx, MI355DR_E_INVALID, "pair out of range");
    HIPCHECK(idx, hipSetDevice(idx->device));
    CHECK(drain_pending(idx));
    CHECK(ensure_qstate(idx));
    hipStream_t s = idx->stream;
    HIPCHECK(idx, hipMemcpyAsync(idx->qdev, queries, (size_t)B * idx->dim * sizeof(float), hipMemcpyHostToDevice, s));
    CHECK(launch_prep(idx, s, B, (int)round_up(B, screen_tile(B)), idx->metric));
    int32_t* pq = nullptr;
    int64_t* pr = nullptr;
    float* od = nullptr;
    double* ods = nullptr;
    HIPCHECK(idx, hipMalloc(&pq, n_pairs * sizeof(int32_t)));
    HIPCHECK(idx, hipMalloc(&pr, n_pairs * sizeof(int64_t)));
    HIPCHECK(idx, hipMalloc(&od, n_pairs * sizeof(float)));
    HIPCHECK(idx, hipMalloc(&ods, n_pairs * sizeof(double)));
    HIPCHECK(idx, hipMemcpyAsync(pq, pair_q, n_pairs * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHECK(idx, hipMemcpyAsync(pr, pair_row, n_pairs * sizeof(int64_t), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_rescore_pairs, dim3((unsigned)((n_pairs + kWave - 1) / kWave)), dim3(kWave), 0, s, idx->rows,
                       idx->nrm2, idx->qdev, idx->st.qn, pq, pr, n_pairs, idx->dim, idx->metric, od, ods);
    HIPCHECK(idx, hipGetLastError());
    HIPCHECK(idx, hipMemcpyAsync(out_dot, od, n_pairs * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHECK(idx, hipMemcpyAsync(out_dist, ods, n_pairs * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHECK(idx, hipStreamSynchronize(s));
    (void)hipFree(pq);
    (void)hipFree(pr);
    (void)hipFree(od);
    (void)hipFree(ods);
    return MI355DR_OK;
}

/* ---- multi-vector (MaxSim): implemented in mi355dr_maxsim.hip ---- */

}  // extern "C"
