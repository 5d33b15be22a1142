// tools/k_screen_rq_abl.h -- the ABLATION build of csrc/k_screen_rq.h: the same kernel with its timing forms (template parameter ABL) for
// tools/screen_ab.hip.  Not part of the library: the product header carries the kernel alone (round 6).  Keep in step by hand.
// k_screen_rq.h -- the large-block screen with the QUERY operand resident in registers ("rq"): rows are the only operand
// that goes through the LDS.
//
// Why (round-4 timing builds of k_screen256c, profiles/r04_kstep_ab.txt): 1.7-2.0 ms of its 6.2 ms are the L2 -> LDS staging,
// and half of the staged bytes are the QUERY half-tiles, fetched again from L2 for every K-step of every corpus tile although
// a persistent workgroup never changes its query tile.  At d <= 768 int8 a query is <= 768 B: the 24 16-byte MFMA fragments
// of 32 queries are 96 registers per lane, loaded ONCE per launch.
//
//   * Workgroup tile 128 corpus rows x 256 queries, 8 waves (two per SIMD); wave w owns queries [32 w, +32) for ALL 128
//     rows: 4 accumulator blocks (64 registers) + 4 KS query fragments (KS = K-steps of 128 B; 96 registers at d = 768) + a
//     ring of row fragments (32).  Every row byte staged into the LDS feeds 256 queries, as in k_screen256c, but nothing
//     else is staged: 16 KiB per K-step and CU instead of 64 KiB per K-step of a tile of twice the rows -- HALF the L2 -> LDS
//     bytes per multiply-add (1/256 B), 2 LDS-DMA pieces per wave and K-step instead of 9.  LDS reads: one ds_read_b128 per
//     MFMA (128 B/clk/CU at the MFMA peak, half of what ds_read_b128 delivers on gfx950).
//   * The LDS holds a ring of NST stages of 16 KiB (one K-step of the tile's 128 rows each; NST = 6 at KS = 6: a whole
//     tile), filled NST K-steps ahead: a piece has ~5 K-steps (~3000 cycles) to land, counted vmcnt never waits in steady
//     state.  One barrier per K-step (hand-over of the stage just read), placed between two MFMA bursts of the same wave --
//     except in a tile's last K-step, whose stage is handed over with the next one's (so that all four block tests of a tile
//     fall between two barriers: see kSkipLast).
//   * K-step = 8 micro-steps m = 4 I + kk of 2 MFMAs (row blocks 2 I, 2 I + 1; K sub-step kk): row-half-major as in
//     k_screen256c, so that a finished tile's blocks are tested UNDER MFMAs of the other row half (both waves of a SIMD
//     reach a tile's end together: tests behind the last MFMA would idle the matrix pipe).  Row fragments are read three
//     micro-steps ahead under counted lgkmcnt.
//   * Hits: a per-wave queue of HIT LANES (k_screen.h: screen_test_block_lq) -- five LDS stores per lane and no call in the hot
//     loop, expanded into candidates 64 entries at a time --; same persistent XCD-aware walk as k_screen256c (the query
//     tiles of one row tile run on one XCD: the shadow comes from HBM once), same int8 row-group records by one small LDS-DMA
//     piece per tile.
// Restrictions: row_bytes = 128 KS with KS a template parameter (the fragments are indexed at compile time); instantiated
// for the int8 shadow at d <= 768 (the bf16 shadow of d = 768 would need 192 query registers).  Everything else keeps
// k_screen256c.  ct0 / n_ctiles of the arguments count 128-ROW tiles here.
// Measured (tools/screen_ab, interleaved with k_screen256c on the same operands, 10 M rows x 1024 queries, profiles/r05_kstep_ab.txt):
// true Gaussian int8 -17 % per launch, -8 % sustained (both kernels run at the socket power cap: the gain is the energy of the
// L2 -> LDS bytes no longer moved), zeros -26 % (4.3 POP/s = 0.86 of the int8 peak: in cycles the staging no longer shows).
#pragma once
#include "k_screen256_common.h"

namespace mi355 {

constexpr int kRqRows = 128;                         // corpus rows per tile
constexpr int kRqStageBytes = kRqRows * kRowB;       // 16 KiB: one K-step of the tile
__host__ __device__ constexpr int rq_stages(int ks) { return ks == 4 ? 4 : ks == 5 ? 5 : 6; }  // a multiple of KS
constexpr int kRqRecSlots = 8;                       // ring of row-group records (one 256-B slot per tile)
__host__ __device__ constexpr int rq_que_off(int ks) { return rq_stages(ks) * kRqStageBytes; }
__host__ __device__ constexpr int rq_rec_off(int ks) { return rq_que_off(ks) + 8 * kLaneQueueBytes; }  // + one lane queue per wave
__host__ __device__ constexpr int rq_prog_off(int ks) { return rq_rec_off(ks) + kRqRecSlots * 256; }  // + 256 B of sibling progress words
__host__ __device__ constexpr int rq_lds(int ks) { return rq_prog_off(ks) + 256; }
// Sibling drift limiter.  The n_qtiles workgroups that walk the same row tiles (same XCD: the tile is meant to come from HBM
// once and from that XCD's L2 for the others) are independent programs: every flush of a hit-lane queue sets one of them back
// by ~3 us, and over a launch of hundreds of tiles the gaps add up until the laggard finds the tile evicted (measured before the
// limiter: L2 hit rate 60 % instead of 75 %, 1.8 x the shadow from HBM in the 7.2 M-row launch, profiles/r05_traffic_i8.json).
// Each workgroup publishes (launch stamp, tiles begun) in a global word per tile; wave 0 fetches its siblings' words with one
// LDS-DMA dword piece per tile (agent scope; no register, no vector-memory wait in the loop -- it is older than the pieces the
// next hand-over waits for) and, when a sibling of THIS launch is more than `drift` tiles behind, sleeps until it has caught
// up.  Progress only: no result depends on it; a wait is capped (kRqSpinCap polls) and a capped wait switches the limiter off
// for the rest of the launch (a sibling that is not resident -- never the case on an idle GPU: the grid is one workgroup per CU).
constexpr int kRqProgressWords = 32 * 8 * 8;  // row-tile slots per XCD x XCDs x query tiles (limiter off above 8 query tiles)
constexpr int kRqSpinCap = 256;
constexpr int kRqDoneTiles = 0xFFFFF;
static_assert(rq_lds(6) <= 160 * 1024, "LDS per workgroup");

// persistent grid in 128-row tiles: 8 XCDs x L workgroups (same rule as screen256_grid)
__host__ __device__ inline unsigned screen_rq_grid(int n_ctiles, int n_qtiles) { return screen256_grid(n_ctiles, n_qtiles); }

// ABL (timing builds for the A/B table; 0 = the kernel): 1 no fragment reads, 4 no tests, 8 no barrier, 16 no LDS-DMA in the
// loop, 32 no vmcnt at the hand-over, 64 every test reads its own row-group record (the form before RQ_LOAD_REC); 2048 a hand-over in EVERY K-step (the form before kSkipLast); 4096 no drift limiter; 8192 every test in one piece (the form before the maxima rode the MFMAs); bits 8, 9: the hit path without its stores (256) / its stores ALWAYS issued under EXEC = hit lanes instead of behind a branch (512: measured +15 % with thresholds parked -- stores under an empty EXEC are not free).  (Cache policies nt / sc0 / sc1 on the row pieces, measured in round 5:
// +1 ... +3 % on Gaussian operands, profiles/r05_kstep_ab.txt -- the default policy stays.)
template <int KS, int ABL, bool I8>
__global__ __launch_bounds__(512, 2) void k_screen_rq(ScreenArgs2 a) {
    constexpr int NST = rq_stages(KS);
    static_assert(NST % KS == 0, "a tile never wraps the ring");
    static_assert((NST + KS - 1) / KS + 2 <= kRqRecSlots, "records ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lq = lds_addr(smem + rq_que_off(KS) + wave * kLaneQueueBytes);  // this wave's queue of hit lanes (k_screen.h)
    int lq_n = 0, lq_ovf = 0;                                                       // entries in it; "a block did not fit" (wave-uniform)

    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int l = b >> 3;
    const int cslot = l / a.n_qtiles;
    const int qt = l - cslot * a.n_qtiles;
    const int cstep = ((int)(gridDim.x >> 3) / a.n_qtiles) * 8;  // row tiles between two visits
    int ctl = cslot * 8 + xcd;
    if (ctl >= a.n_ctiles) return;
    const int q0 = qt * kT2 + 32 * wave;  // this wave's first query
    const int row_bytes = a.row_bytes;

    // ---- the wave's query operand: 4 KS fragments of 16 B per lane (query lane & 31, K bytes 32 i + 16 (lane >> 5))
    bf16x8 fB[4 * KS];
    {
        const char* qp = (const char*)a.qhat + (int64_t)(q0 + (lane & 31)) * row_bytes + 16 * (lane >> 5);
#pragma unroll
        for (int i = 0; i < 4 * KS; ++i) fB[i] = __builtin_bit_cast(bf16x8, *(const uint4*)(qp + 32 * i));
        // the fragments are complete HERE: left to the waitcnt pass, the waits for these loads land in front of their first use
        // inside the tile loop -- a vmcnt(23) ... vmcnt(0) ladder that drains the LDS-DMA ring on every tile
#pragma unroll
        for (int i = 0; i < 4 * KS; ++i) asm volatile("" : "+v"(fB[i]));
    }
    // ---- DMA sources: this wave stages local rows [16 wave + 8 u, +8) of every stage
    unsigned voff[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = (2 * wave + u) * 8 + (lane >> 3);  // local row 0..127
        const int c = (lane & 7) ^ ((r >> 1) & 7);       // source chunk for this LDS slot (swizzle)
        voff[u] = (unsigned)(r * row_bytes + c * 16);
    }
    // ---- fragment read offset inside a stage: row block rb adds 4096 (the key (r >> 1) & 7 does not see it), K sub-step kk
    // flips bits 5, 6
    int offA;
    {
        const int r = lane & 31, g = lane >> 5;
        offA = r * kRowB + ((g ^ ((r >> 1) & 7)) << 4);
    }
    const int q_lane = q0 + (lane & 31);
    float th = a.thr[q_lane];
    float scq = I8 ? a.sc[q_lane] : 1.0f, kqq = I8 ? a.kq[q_lane] : 1.0f;
    asm volatile("" ::"v"(th), "v"(kqq), "v"(scq));

    f32x16 acc[4];  // row blocks 0..3 of the tile
    int tg = 0;     // running maximum of the block under test (screen_block_max_part)
    float rec_m[4] = {1.0f, 1.0f, 1.0f, 1.0f}, rec_e[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // their int8 constants (RQ_LOAD_REC)
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
    acc[2] = zero16, acc[3] = zero16;  // (the first tile takes the maxima of "the previous tile's" row half 1 like every other)
    const unsigned lds0 = lds_addr(smem);
    const unsigned rec_lds = lds0 + rq_rec_off(KS);
    const unsigned rec_voff = (unsigned)((lane & 7) * 4);  // the tile's 4 records = 8 dwords, eight copies per slot
    // drift limiter (wave 0): the slot's 8 progress words, this workgroup's is word qt
    bool lim = (ABL & 4096) == 0 && a.drift > 0 && a.progress != nullptr && a.n_qtiles > 1 && a.n_qtiles <= 8 && wave == 0;
    const char* const prog_base = (const char*)(a.progress + (cslot * 8 + xcd) * 8);
    const unsigned prog_lds = lds0 + rq_prog_off(KS);
    const unsigned prog_voff = (unsigned)(((lane & 7) < a.n_qtiles ? (lane & 7) : 0) * 4);
    const int stamp = a.epoch << 20;

#define RQ_PIN() __builtin_amdgcn_sched_barrier(0)
    // staging cursor: the NEXT K-step to stage = (tile base, K offset, K-steps done in that tile, ring stage, tile counter);
    // past the last tile it stays on it (dummy re-stage of valid memory, drained before the exit)
    const int64_t tile_bytes = (int64_t)kRqRows * row_bytes;
    const char* c_base = (const char*)a.shadow + (int64_t)(a.ct0 + ctl) * tile_bytes;
    const int64_t tile_stride_bytes = (int64_t)cstep * tile_bytes;
    int c_k = 0, c_n = 0, c_ctl = ctl, c_tc = 0;
    unsigned c_dst = lds0 + (unsigned)(2 * wave) * 1024u;  // LDS address of this wave's first piece in the cursor's stage
#define RQ_PIECE(U)                                                                                   \
    do {                                                                                              \
        if constexpr ((ABL & 16) == 0) glds16_saddr(c_base + c_k, voff[U], c_dst + (U) * 1024u);       \
    } while (0)
#define RQ_REC()                                                                                      \
    do {                                                                                              \
        if constexpr (I8 && (ABL & 16) == 0)                                                          \
            if (c_n == 0)                                                                             \
                glds4_saddr((const char*)a.grp + (int64_t)(a.ct0 + c_ctl) * (kRqRows / kI8GroupRows * (int)sizeof(I8Group)), \
                            rec_voff, rec_lds + (unsigned)(c_tc & (kRqRecSlots - 1)) * 256u);         \
    } while (0)
#define RQ_ADVANCE()                                                                                  \
    do {                                                                                              \
        c_k += kRowB;                                                                                 \
        c_dst += kRqStageBytes;                                                                       \
        if (c_dst >= lds0 + (unsigned)(NST * kRqStageBytes)) c_dst -= (unsigned)(NST * kRqStageBytes); \
        if (++c_n == KS) {                                                                            \
            c_n = 0;                                                                                  \
            c_k = 0;                                                                                  \
            ++c_tc;                                                                                   \
            if (c_ctl + cstep < a.n_ctiles) {                                                         \
                c_ctl += cstep;                                                                       \
                c_base += tile_stride_bytes;                                                          \
            }                                                                                         \
        }                                                                                             \
    } while (0)

    // ---- row fragments: ring of four micro-steps x 2 blocks, read kPF micro-steps ahead
    constexpr int kPF = 3;
    bf16x8 fAq[4][2];
    if constexpr ((ABL & 1) != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" : "=v"(fAq[i][j]));
    }
// reads of micro-step M2 (0..7 this K-step, 8..10 = micro-steps 0..2 of the next one) from stage byte offsets SB / SBN
#define RQ_PREFETCH(M2, SB, SBN)                                                                      \
    do {                                                                                              \
        if constexpr ((ABL & 1) == 0) {                                                               \
            constexpr int m2__ = (M2) & 7;                                                            \
            const char* r__ = smem + ((M2) >= 8 ? (SBN) : (SB)) + (2 * (m2__ >> 2)) * 4096;           \
            _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                          \
                fAq[m2__ & 3][rb] = __builtin_bit_cast(bf16x8, *(const uint4*)(r__ + rb * 4096 + (offA ^ ((m2__ & 3) * 32)))); \
        }                                                                                             \
    } while (0)
#define RQ_MM(M, TT, ZERO)                                                                            \
    do {                                                                                              \
        _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                              \
            acc[2 * ((M) >> 2) + rb] = screen_mfma<I8>(fAq[(M) & 3][rb], fB[4 * (TT) + ((M) & 3)],    \
                                                       (ZERO) ? zero16 : acc[2 * ((M) >> 2) + rb]);   \
    } while (0)
// ... the same with the running maximum of finished block TB folded in behind the two MFMAs (parts 2 (M & 1), 2 (M & 1) + 1 of
// screen_block_max_part: a block's four parts ride two consecutive micro-steps).  ABL bit 13: the test in one piece (the form
// before, A/B).
#define RQ_MM_T(M, TT, ZERO, TB)                                                                      \
    do {                                                                                              \
        if constexpr ((ABL & 4) != 0 || (ABL & 8192) != 0) {                                          \
            RQ_MM(M, TT, ZERO);                                                                       \
        } else {                                                                                      \
            acc[2 * ((M) >> 2)] = screen_mfma<I8>(fAq[(M) & 3][0], fB[4 * (TT) + ((M) & 3)], (ZERO) ? zero16 : acc[2 * ((M) >> 2)]); \
            RQ_PIN();                                                                                 \
            tg = screen_block_max_part<I8, 2 * ((M) & 1)>(acc[TB], tg);                               \
            RQ_PIN();                                                                                 \
            acc[2 * ((M) >> 2) + 1] = screen_mfma<I8>(fAq[(M) & 3][1], fB[4 * (TT) + ((M) & 3)], (ZERO) ? zero16 : acc[2 * ((M) >> 2) + 1]); \
            RQ_PIN();                                                                                 \
            tg = screen_block_max_part<I8, 2 * ((M) & 1) + 1>(acc[TB], tg);                           \
        }                                                                                             \
    } while (0)
// The four blocks' constants (m = S_g S_q, ek = e_g kq) of the tile under test, read from the records ring ONE micro-step
// before the tile's first test instead of inside each test: a test that reads its record itself waits with lgkmcnt(0) for a
// read queued BEHIND the six to eight fragment reads in flight (LDS returns in order) -- ~150 cycles of this wave, four times
// per tile.  (ABL bit 6: the old form, for the A/B.)
#define RQ_LOAD_REC(TC)                                                                               \
    do {                                                                                              \
        if constexpr (I8 && (ABL & 4) == 0 && (ABL & 64) == 0) {                                      \
            const float4* rp__ = (const float4*)(smem + rq_rec_off(KS) + ((TC) & (kRqRecSlots - 1)) * 256); \
            const float4 r0__ = rp__[0], r1__ = rp__[1];                                              \
            rec_m[0] = r0__.x * scq, rec_e[0] = r0__.y * kqq, rec_m[1] = r0__.z * scq, rec_e[1] = r0__.w * kqq; \
            rec_m[2] = r1__.x * scq, rec_e[2] = r1__.y * kqq, rec_m[3] = r1__.z * scq, rec_e[3] = r1__.w * kqq; \
        }                                                                                             \
    } while (0)
// test row block RB of the tile whose first row is ROW0 (records slot of tile counter TC)
#define RQ_TEST(RB, ROW0, TC)                                                                         \
    do {                                                                                              \
        if constexpr ((ABL & 4) == 0) {                                                               \
            int lane_e = lane;                                                                        \
            asm volatile("" : "+v"(lane_e));                                                          \
            const int q__ = q0 + (lane_e & 31);                                                       \
            const int rbase__ = (ROW0) + 32 * (RB) + 4 * (lane_e >> 5);                               \
            I8Blk blk__{1.0f, 0.0f};                                                                  \
            if constexpr (I8 && (ABL & 64) != 0) {                                                    \
                const I8Group g__ = ((const I8Group*)(smem + rq_rec_off(KS) + ((TC) & (kRqRecSlots - 1)) * 256))[RB]; \
                blk__ = i8_blk(g__, scq, kqq);                                                        \
            } else if constexpr (I8) {                                                                \
                blk__ = I8Blk{rec_m[RB], rec_e[RB]};                                                  \
            }                                                                                         \
            if constexpr ((ABL & 8192) != 0)                                                          \
                screen_test_block_lq<I8, (ABL >> 8) & 3>(a, a.status, row_end, acc[RB], q__, rbase__, th, blk__, lq, lq_n, lq_ovf); \
            else                                                                                      \
                screen_test_block_lq_max<I8, (ABL >> 8) & 3>(a, a.status, row_end, acc[RB], tg, q__, rbase__, th, blk__, lq, lq_n, lq_ovf); \
        }                                                                                             \
    } while (0)
#define RQ_MICRO(M, TT, ZERO, SB, SBN)                                                                \
    do {                                                                                              \
        RQ_PREFETCH((M) + kPF, SB, SBN);                                                              \
        RQ_PIN();                                                                                     \
        RQ_MM(M, TT, ZERO);                                                                           \
        RQ_PIN();                                                                                     \
    } while (0)
#define RQ_MICRO_T(M, TT, ZERO, SB, SBN, TB)                                                          \
    do {                                                                                              \
        RQ_PREFETCH((M) + kPF, SB, SBN);                                                              \
        RQ_PIN();                                                                                     \
        RQ_MM_T(M, TT, ZERO, TB);                                                                     \
        RQ_PIN();                                                                                     \
    } while (0)

    // Hand-over schedule.  kSkipLast (KS >= 2; ABL bit 11 = the per-K-step form, A/B): NO hand-over in a tile's LAST K-step -- the
    // ring stage it reads is handed over together with the next K-step's, in the next tile's first K-step.  The four block tests
    // of a tile sit in its last K-step (row half 0) and in the next tile's first (row half 1), i.e. now BETWEEN two consecutive
    // barriers: a wave with hits is late once per tile instead of once per test site, and the delays of different waves overlap
    // instead of adding up (the eight waves meet at every barrier: whatever a hit costs one wave is paid by all -- with one hit
    // per 3 ... 30 blocks some wave of the eight has one at most test sites; profiles/r05_hit_path.txt).
    constexpr bool kSkipLast = KS >= 2 && (ABL & 2048) == 0;
    constexpr int kPrologueSteps = kSkipLast ? NST - 1 : NST;  // (the first hand-over then frees two stages like every tile's first)
    // ---- prologue: K-steps 0 .. kPrologueSteps-1 into the ring; K-step 0 landed and visible; fragments of micro-steps 0..2
#pragma unroll
    for (int s = 0; s < kPrologueSteps; ++s) {
        RQ_PIECE(0);
        RQ_PIECE(1);
        RQ_REC();
        RQ_ADVANCE();
    }
    if constexpr ((ABL & 16) == 0) {
        // this wave's pieces of K-step 0 (the oldest) have landed: at most the younger ones (+ records) are out
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (kPrologueSteps - 1)) : "memory");
    }
    MI355_BARRIER();
    int s0b = 0;  // ring byte offset of the tile's first K-step (compile-time 0 when a tile is the whole ring)
    int tc = 0;   // tile counter of this workgroup (records slot)
    RQ_PREFETCH(0, 0, 0);
    RQ_PREFETCH(1, 0, 0);
    RQ_PREFETCH(2, 0, 0);

    const int row_end = (int)a.row_end;
    int row0_cur = (a.ct0 + ctl) * kRqRows, row0_prev = row0_cur;  // rows < 2^31 (checked by the host)
    bool have_prev = false;  // a finished tile's row half 1 is waiting for its tests
    for (;;) {
        // Flush of the hit-lane queue.  A flush costs its wave one returning atomic's round trip, and the workgroup's other seven
        // waves wait for it at the next barrier; waves that flush when THEIR queue is full do so at different tiles, so at
        // 0.7 hit lanes per block (k = 100) the workgroup paid for a flush at almost every tile.  Round 6: all eight flush at the
        // same tiles (a.flush_mask, period from the host's estimate of the hit density) -- their round trips overlap; a wave
        // whose queue fills faster than foreseen still flushes on its own (at a.flush_alone entries).
        const bool flush_due = a.flush_mask >= 0 ? ((tc & a.flush_mask) == 0 && lq_n > 0) || lq_n > a.flush_alone
                                                 : lq_n > kLaneQueueFlushAt;
        if (((ABL >> 8) & 3) != 1 && flush_due) {  // wave-uniform
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (((ABL >> 8) & 3) != 1) lane_queue_flush<I8>(a, lq, lq_n, row_end);
            lq_n = 0;
        }
        if (((ABL >> 8) & 3) == 2 && lq_ovf) {  // (branch-free A/B form only) a burst did not fit the queue -- this wave's queries are re-screened by the host
            __hip_atomic_fetch_or(&a.status[q_lane], kStOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lq_ovf = 0;
        }
        // drift limiter, wave 0, every tile.  Here -- not behind a barrier: measured behind the first K-step's hand-over, where the
        // words could be read under the wait for the fragments, the same work cost +4.7 % (zeros: +13.7 %): right behind a
        // barrier all eight waves start together and the one that has extra work is late at the next; at a tile's start the
        // waves are staggered (the last barrier is a K-step back) and most of it is absorbed.  (Acting on every 2nd / 4th tile
        // only was SLOWER, +0.8 / +2.9 % per step: profiles/r05_drift_limiter.txt.)
        if (lim) {
            const int mine = stamp | (tc + 1);
            if (lane == 0) asm volatile("global_store_dword %0, %1, %2 sc1" ::"v"(qt * 4), "v"(mine), "s"(prog_base) : "memory");
            if (tc > 0) {  // the words fetched one tile ago (landed: twelve younger pieces have been issued and waited down to 8)
                int w = *(const int*)(smem + rq_prog_off(KS) + (lane & 7) * 4);
                if (__builtin_amdgcn_ballot_w64((w >> 20) == a.epoch && (w & kRqDoneTiles) + a.drift < tc + 1)) {
                    int spins = 0;
                    do {
                        __builtin_amdgcn_s_sleep(16);
                        w = __hip_atomic_load((const int*)(prog_base + prog_voff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } while (__builtin_amdgcn_ballot_w64((w >> 20) == a.epoch && (w & kRqDoneTiles) + a.drift < tc + 1) &&
                             ++spins < kRqSpinCap);
                    if (spins >= kRqSpinCap) lim = false;
                }
            }
            glds4_saddr_sc1(prog_base, prog_voff, prog_lds);
        }
        const int tile_sb = (NST == KS) ? 0 : s0b;
        const int next_tile_sb = (NST == KS) ? 0 : (s0b + KS * kRqStageBytes >= NST * kRqStageBytes ? 0 : s0b + KS * kRqStageBytes);
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const bool first = t == 0, last = t + 1 == KS;
            const int sb = tile_sb + t * kRqStageBytes;
            const int sbn = last ? next_tile_sb : sb + kRqStageBytes;
            const bool tp = first && have_prev;  // test the previous tile's row half 1 under this K-step's row half 0
            // (the previous tile's row half 1 -- blocks 2, 3 -- is tested under this tile's first K-step, row half 0: the blocks'
            // maxima ride micro-steps 0, 1 / 2, 3, the rest of each test follows its second micro-step.  In the first tile
            // the maxima are taken of zeros and dropped)
            if (first) {
                RQ_MICRO_T(0, t, first, sb, sbn, 2);
                RQ_MICRO_T(1, t, false, sb, sbn, 2);
                if (tp) RQ_TEST(2, row0_prev, tc - 1);
                RQ_MICRO_T(2, t, false, sb, sbn, 3);
                RQ_MICRO_T(3, t, false, sb, sbn, 3);
                if (tp) RQ_TEST(3, row0_prev, tc - 1);
            } else {
                RQ_MICRO(0, t, first, sb, sbn);
                RQ_MICRO(1, t, false, sb, sbn);
                RQ_MICRO(2, t, false, sb, sbn);
                RQ_MICRO(3, t, false, sb, sbn);
            }
            if (last) RQ_LOAD_REC(tc);  // (behind the previous tile's last test -- micro-step 3 of ITS next K-step -- also at KS = 1)
            if (last) RQ_MICRO_T(4, t, first, sb, sbn, 0);
            else RQ_MICRO(4, t, first, sb, sbn);
            // ---- hand-over: every read of this K-step's stage has been issued (the last ones kPF micro-steps before its end)
            const bool hand = !(kSkipLast && last);
            if (hand) {
                // own pieces landed: of the next K-step -- and of the one after it when that one has no hand-over of its own
                // (t = KS - 2).  In flight before this K-step's issue: through K-step g + NST - 1 (g + NST - 2 in a tile's first
                // K-step: the previous one issued nothing).  Record pieces only make the wait stricter.
                const int allowed = 2 * (NST - 2 - ((kSkipLast && first) ? 1 : 0) - ((kSkipLast && t == KS - 2) ? 1 : 0));
                if constexpr ((ABL & 48) == 0) {
                    if (allowed >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else if (allowed == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else if (allowed == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if (allowed == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and the last fragments of this one are in registers
                if constexpr ((ABL & 8) == 0) MI355_BARRIER();
            }
            RQ_PIN();
            if (last) RQ_MICRO_T(5, t, false, sb, sbn, 0);
            else RQ_MICRO(5, t, false, sb, sbn);
            if (last) RQ_TEST(0, row0_cur, tc);
            if (hand) RQ_PIECE(0);
            RQ_PIN();
            if (last) RQ_MICRO_T(6, t, false, sb, sbn, 1);
            else RQ_MICRO(6, t, false, sb, sbn);
            if (hand) RQ_PIECE(1);
            RQ_PIN();
            if (hand && kSkipLast && first) {  // two stages were handed over: the second K-step's pieces ride micro-step 7
                RQ_REC();
                RQ_ADVANCE();
                RQ_PIECE(0);
                RQ_PIN();
            }
            if (last) RQ_MICRO_T(7, t, false, sb, sbn, 1);
            else RQ_MICRO(7, t, false, sb, sbn);
            if (last) RQ_TEST(1, row0_cur, tc);
            if (hand && kSkipLast && first) RQ_PIECE(1);
            if (hand) {
                RQ_REC();
                RQ_ADVANCE();
            }
            RQ_PIN();
        }
        row0_prev = row0_cur;
        have_prev = true;
        ++tc;
        if constexpr (NST != KS) s0b = next_tile_sb;
        if (ctl + cstep >= a.n_ctiles) break;
        ctl += cstep;
        row0_cur = (a.ct0 + ctl) * kRqRows;
    }
    // the last tile's row half 1 (no MFMAs left to hide under: the maxima in one piece)
#define RQ_TEST_WHOLE(RB)                                                                             \
    do {                                                                                              \
        if constexpr ((ABL & 4) == 0 && (ABL & 8192) == 0) {                                          \
            tg = screen_block_max_part<I8, 0>(acc[RB], tg);                                           \
            tg = screen_block_max_part<I8, 1>(acc[RB], tg);                                           \
            tg = screen_block_max_part<I8, 2>(acc[RB], tg);                                           \
            tg = screen_block_max_part<I8, 3>(acc[RB], tg);                                           \
        }                                                                                             \
        RQ_TEST(RB, row0_prev, tc - 1);                                                               \
    } while (0)
    RQ_TEST_WHOLE(2);
    RQ_TEST_WHOLE(3);
#undef RQ_TEST_WHOLE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy prefetches must land before the LDS is freed
    if ((ABL & 4096) == 0 && a.drift > 0 && a.progress != nullptr && wave == 0 && lane == 0)  // done: nobody waits for this one
        __hip_atomic_store((int*)prog_base + qt, stamp | kRqDoneTiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lane_queue_flush<I8>(a, lq, lq_n, row_end);
    if constexpr ((ABL & 4) != 0) {  // timing build without tests: the accumulators stay live
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(acc[i]));
    }

#undef RQ_PIN
#undef RQ_PIECE
#undef RQ_REC
#undef RQ_ADVANCE
#undef RQ_PREFETCH
#undef RQ_MM
#undef RQ_MM_T
#undef RQ_MICRO_T
#undef RQ_TEST
#undef RQ_MICRO
}

}  // namespace mi355
