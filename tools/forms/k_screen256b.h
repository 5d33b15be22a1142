// k_screen256b.h -- second form of the large-block screen (256 corpus rows x 256 queries per tile, 8 waves, persistent).
//
// Same contract, tile, wave layout, LDS image and ping-pong as k_screen256 (k_screen256.h); what changes is WHEN things
// happen inside a K-step, after reading the first form's ISA and timeline (DESIGN.md 4.1 "second form"):
//
//  * Both query halves stay in registers (fb0, fb1: +16 VGPRs), so B0 is read ONCE per K-step instead of twice
//    (-4 of 28 ds_read_b128 per wave and K-step) -- and, more important, every ring slot now has its LAST reader in the
//    first three phases of a K-step.
//  * Ring slots are re-staged as soon as they are free, with the data of the K-step TWO ahead (first form: one ahead,
//    staged in the same order every K-step, which left the vmcnt(4) wait only 2 phases of DMA flight).  Schedule for
//    K-step g, ring parity P = g&1:
//        phase 0: read A0(P) B0(P) | stage B1(g+1) -> (P^1)   [slot last read in phase 1 of K-step g-1]
//        phase 1: read B1(P)       | stage A1(g+1) -> (P^1)   [last read in phase 2 of g-1]
//        phase 2: read A1(P)       | stage A0(g+2) -> (P)     [last read in phase 0 of THIS K-step: 2 barriers ago]
//        phase 3: (no reads)       | stage B0(g+2) -> (P)     [last read in phase 0 of this K-step]
//    Every half-tile is needed 5-6 phases after its issue; the wait in front of each barrier is vmcnt(8) = FOUR
//    half-tiles (64 KiB per workgroup) still in flight, twice the first form's lead.  Hazard rules unchanged: a
//    half-tile is read in the phase after the wait that covers it (2 barriers for the waiting wave, 1 for the other
//    group), a slot is re-staged >= 2 barriers after its last reader's lgkmcnt(0).
//  * The LOAD half issues its ds_reads BEFORE the two LDS-DMA instructions (their issue alone costs 60-180 cycles each
//    under load; the reads are what the next MFMA half waits for).
//  * No per-tile epilogue block.  A quadrant (i,j) of the accumulators is final after phase p(i,j) of the tile's last
//    K-step and is overwritten in phase p(i,j) of the next tile's first K-step, so its threshold test runs in a LOAD half
//    in between (quadrants 00, 01, 11 in phases 1, 2, 3 of the last K-step, quadrant 10 in phase 1 of the next tile's first
//    K-step) and its 32 accumulator registers are zeroed in the LOAD half right before their first MFMA: ~25 + 32 VALU
//    instructions next to the OTHER group's MFMA half, instead of ~300 in one block during which the matrix pipe of
//    both groups idles (at d = 768 int8 a tile is only 6 K-steps long).
//  * A burst that overflows a wave's LDS candidate queue inside one tile no longer falls back to returning global atomics
//    inside the K loop: the query is flagged kStOverflow and re-screened by the host's retry path (exactness kept, the
//    hot loop has no vector-memory instruction besides the DMA).
#pragma once
#include "k_screen256.h"

namespace mi355 {

// ABL (developer switches for tools/screen_bench / screen_trace, 0 in the library):
//   bit2 (4)   no s_setprio            bit3 (8)   wait for the ds_reads before the barrier instead of after it
//   bit4 (16)  timeline trace          bit5 (32)  NM = 1: the second 1-KiB piece of every half-tile is issued inside the
//   MFMA half (after the 4th MFMA)     bit6 (64)  NM = 2: both pieces inside the MFMA half (after the 2nd and 6th MFMA)
//   bit8 (256) TAIL = 1 / bit9 (512) TAIL = 2: the phase's second barrier is passed before the last 2 / 4 MFMAs are issued,
//              so the other group's first MFMAs queue behind them and the matrix pipe does not drain during the hand-over
//   bit7 (128) every workgroup starts its K walk at a different K-step (sharers of a tile do not ask the L2 for the same
//              lines at the same moment)
constexpr int kScreen256bAbl = 64 | 1024 | 2048;  // what the library runs: NM = 2, SADDR, FAN (A/B in tools/screen_bench)
template <int ABL, bool I8>
__global__ __launch_bounds__(512, 2) void k_screen256b(ScreenArgs2 a) {
    constexpr int NM = (ABL & 64) ? 2 : ((ABL & 32) ? 1 : 0);
    constexpr bool SADDR = (ABL & 1024) != 0;  // LDS-DMA through the SGPR-base + 32-bit-offset form (inline asm)
    constexpr bool FAN = (ABL & 2048) != 0;    // balance the LOAD halves: 8/4/8/4 ds_reads instead of 12/4/8/0
    constexpr int TAIL = (ABL & 512) ? 2 : ((ABL & 256) ? 1 : 0);  // K sub-steps (2 MFMAs each) issued AFTER the second barrier
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2;  // 0 leads, 1 runs one barrier behind
    const int wr = group, wc = wave & 3;
    int32_t* const que = (int32_t*)(smem + kRingBytes + wave * (kWaveQueueCap * 12));  // [q | row | value bits]
    int que_n = 0;                                                                       // wave-uniform

    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int l = b >> 3;
    const int cslot = l / a.n_qtiles;
    const int qt = l - cslot * a.n_qtiles;
    const int cstep = ((int)(gridDim.x >> 3) / a.n_qtiles) * 8;  // corpus tiles between two visits
    int ctl = cslot * 8 + xcd;
    if (ctl >= a.n_ctiles) return;
    const int q0 = qt * kT2;
    const int64_t row_bytes = a.row_bytes;

    // ---- DMA sources (as in k_screen256): this wave stages local rows [16*wave + 8u, +8) of every half-tile
    unsigned voffA[2], voffB[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = (2 * wave + u) * 8 + (lane >> 3);      // local row 0..127
        const int c = (lane & 7) ^ ((r >> 1) & 7);           // source chunk for this LDS slot (swizzle)
        const int arow0 = 128 * (r >> 6) + (r & 63);         // + 64*i
        const int bcol0 = 64 * (r >> 5) + (r & 31);          // + 32*j
        voffA[u] = (unsigned)(arow0 * (int)row_bytes + c * 16);
        voffB[u] = (unsigned)(bcol0 * (int)row_bytes + c * 16);
    }
    const char* const baseB = (const char*)a.qhat + (int64_t)q0 * row_bytes;
    const int64_t tile_stride_bytes = (int64_t)cstep * kT2 * row_bytes;
    const int64_t half_A = 64 * row_bytes, half_B = 32 * row_bytes;
    int offA[2], offB;
    {
        const int g = lane >> 5;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int r = wr * 64 + rb * 32 + (lane & 31);
            offA[rb] = r * kRowB + ((g ^ ((r >> 1) & 7)) << 4);
        }
        const int r = wc * 32 + (lane & 31);
        offB = r * kRowB + ((g ^ ((r >> 1) & 7)) << 4);
    }
    float th[2], scq[2], kqq[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = q0 + 64 * wc + 32 * j + (lane & 31);
        th[j] = a.thr[q];
        scq[j] = I8 ? a.sc[q] : 1.0f;
        kqq[j] = I8 ? a.kq[q] : 1.0f;
    }
    // These loads must be complete IN THE COMPILER'S BOOKS before the first LDS-DMA is issued: its waitcnt insertion does
    // not see the counted asm waits below, and would otherwise put s_waitcnt vmcnt(0) in front of the first use of a
    // threshold -- inside the K loop, draining the whole prefetch (tests/test_build_pipeline.py checks the ISA).
    asm volatile("" ::"v"(th[0]), "v"(th[1]), "v"(kqq[0]), "v"(kqq[1]), "v"(scq[0]), "v"(scq[1]));

    // accumulators start at "never a hit": the first tile's phase 1 tests quadrant (1,0) of a tile that does not exist
    f32x16 acc[2][2][2];  // [row half i][row block rb][query half j]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][rb][j][r] = I8 ? __int_as_float((int)0x80000000) : -__builtin_inff();
    bf16x8 fa[2][4], fb0[4], fb1[4];  // typed bf16x8 also for int8 data: see the NOTE in k_screen.h (waitcnt insertion)
    bf16x8 fan[2][2];                 // FAN: K sub-steps 0,1 of the NEXT K-step's A0, read one phase early (phase 3)
    const int T = a.ksteps;
    const int kend = T * kRowB;

#define KB_READ_A(I, PAR)                                                                             \
    do {                                                                                              \
        const char* s__ = smem + (4 * (PAR) + ((I) ? 3 : 0)) * kHalfBytes;                            \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                              \
            _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                          \
                fa[rb][kk] = __builtin_bit_cast(bf16x8, *(const uint4*)(s__ + (offA[rb] ^ (kk * 32)))); \
    } while (0)
// A0 split for FAN: K sub-steps [K0, K0+2) of A0 in parity PAR into DST[rb][kk - KD]
#define KB_READ_A0_PART(PAR, K0, KD, DST)                                                             \
    do {                                                                                              \
        const char* s__ = smem + (4 * (PAR)) * kHalfBytes;                                            \
        _Pragma("unroll") for (int kk = (K0); kk < (K0) + 2; ++kk)                                    \
            _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                          \
                DST[rb][kk - (KD)] = __builtin_bit_cast(bf16x8, *(const uint4*)(s__ + (offA[rb] ^ (kk * 32)))); \
    } while (0)
#define KB_READ_B(J, PAR, FB)                                                                         \
    do {                                                                                              \
        const char* s__ = smem + (4 * (PAR) + 1 + (J)) * kHalfBytes;                                  \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                              \
            FB[kk] = __builtin_bit_cast(bf16x8, *(const uint4*)(s__ + (offB ^ (kk * 32))));           \
    } while (0)
// the phase's half-tile: type SS into parity SPAR from SRC (first row of the half-tile + K offset), rows by VOFF
#define KB_STG(U) kb_stage<SS__, SADDR>(smem, wave, spar__, ssrc__, *svoff__, U)
#define KB_PIN() __builtin_amdgcn_sched_barrier(0)
#define KB_ZERO(I, J)                                                                                 \
    do {                                                                                              \
        asm volatile("; zero quadrant");  /* keeps this a branch: if-converted it is 32 v_cndmask in EVERY K-step */ \
        _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                              \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[I][rb][J][r] = 0.0f;                   \
    } while (0)
// threshold test of quadrant (I,J) of the tile whose first row is ROW0; G[rb] = the row-group records of its two blocks
#define KB_TEST(I, J, ROW0, G)                                                                        \
    do {                                                                                              \
        int lane_e = lane;                                                                            \
        asm volatile("" : "+v"(lane_e));                                                              \
        const int q__ = q0 + 64 * wc + 32 * (J) + (lane_e & 31);                                      \
        _Pragma("unroll") for (int rb = 0; rb < 2; ++rb) {                                            \
            const int rbase__ = (ROW0) + 128 * wr + 64 * (I) + 32 * rb + 4 * (lane_e >> 5);           \
            I8Blk blk__{1.0f, 0.0f};                                                                  \
            if constexpr (I8) blk__ = i8_blk((G)[rb], scq[J], kqq[J]);                                \
            screen_queue_block<I8, true>(a, a.status, acc[I][rb][J], q__, rbase__, row_end, th[J], blk__, que, que_n); \
        }                                                                                             \
    } while (0)
// int8: the four row-group records of this wave's rows of tile ROW0 (rows 128 wr + 32 (2 I + rb)) -- scalar loads.
// They are issued in EVERY K-step, for the tile of the NEXT K-step, at the end of phase 3's LOAD half, and waited for (by
// the compiler: KB_GROUPS_LANDED is a use) at the end of its MFMA half -- a barrier and eight MFMAs later, the one MFMA
// half without counted LDS waits: a scalar load that may or may not be in flight (one
// under `if (first)`) makes the compiler turn the counted lgkmcnt waits in front of the next MFMAs into lgkmcnt(0) -- in
// every K-step (measured: -4 %).  Half 1 of the outgoing records is kept for the quadrant tested one K-step late.
#define KB_LOAD_GROUPS(ROW0)                                                                          \
    do {                                                                                              \
        if constexpr (I8) {                                                                           \
            KB_PIN();                                                                                 \
            gprev[0] = gcur[1][0];                                                                    \
            gprev[1] = gcur[1][1];                                                                    \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                             \
                _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                      \
                    gcur[i][rb] = i8_group_of(a.grp, (int64_t)(ROW0) + 128 * wr + 64 * i + 32 * rb);  \
            KB_PIN();                                                                                 \
        }                                                                                             \
    } while (0)
#define KB_GROUPS_LANDED()                                                                            \
    do {                                                                                              \
        if constexpr (I8)                                                                             \
            asm volatile("" ::"s"(gcur[0][0].step), "s"(gcur[0][0].err), "s"(gcur[0][1].step), "s"(gcur[0][1].err), \
                         "s"(gcur[1][0].step), "s"(gcur[1][0].err), "s"(gcur[1][1].step), "s"(gcur[1][1].err));     \
    } while (0)
#define KB_WAIT_VM_N()                                                                                \
    do {                                                                                              \
        if constexpr (NM == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                       \
        else if constexpr (NM == 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");                  \
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                         \
    } while (0)
// One phase.  LOAD half: READS (ds_reads first), then the pieces of this phase's half-tile that are not issued inside the
// MFMA half, EXTRA (accumulator zeroing / threshold tests at tile boundaries), the counted DMA wait (WAIT = 1) | barrier |
// MFMA half: 8 MFMAs on quadrant (I,J), the remaining DMA pieces between them | barrier.
#define KB_PHASE(READS, SS, SPAR, SSRC, SVOFF, EXTRA, WAIT, I, J, FB, MIDA, MIDB)                      \
    do {                                                                                              \
        constexpr int SS__ = (SS);                                                                    \
        const int spar__ = (SPAR);                                                                    \
        const char* const ssrc__ = (SSRC);                                                            \
        const unsigned(*svoff__)[2] = &(SVOFF);                                                       \
        MI355_TR_STAMP(tr_on, tr_addr);                                                               \
        READS;                                                                                        \
        if constexpr (NM < 2) KB_STG(0);                                                              \
        if constexpr (NM < 1) KB_STG(1);                                                              \
        EXTRA;                                                                                        \
        if (ABL & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                               \
        if (WAIT) KB_WAIT_VM_N();                                                                     \
        MI355_TR_ISSUE(tr_on, tr_l);                                                                  \
        MI355_BARRIER();                                                                              \
        MI355_TR_STAMP(tr_on, tr_addr);                                                               \
        MI355_TR_STORE(tr_on, tr_addr, tr_l);                                                         \
        if (!(ABL & 4)) __builtin_amdgcn_s_setprio(1);                                                \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                            \
            if (kk == 4 - TAIL) { /* hand the matrix pipe over while the last MFMAs are still to be issued */ \
                MI355_TR_STAMP(tr_on, tr_addr);                                                       \
                MI355_BARRIER();                                                                      \
            }                                                                                         \
            _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                          \
                acc[I][rb][J] = screen_mfma<I8>((FAN && (I) == 0 && kk < 2) ? fan[rb][kk] : fa[rb][kk], FB[kk],  \
                                                acc[I][rb][J]);                                       \
            if constexpr (NM == 2) {                                                                  \
                if (kk == 0) { KB_PIN(); KB_STG(0); KB_PIN(); }                                       \
                if (kk == (TAIL >= 2 ? 1 : 2)) { KB_PIN(); KB_STG(1); KB_PIN(); }                     \
            }                                                                                         \
            if constexpr (NM == 1) {                                                                  \
                if (kk == 1) { KB_PIN(); KB_STG(1); KB_PIN(); }                                       \
            }                                                                                         \
            if (kk == 1) { MIDA; }                                                                    \
        }                                                                                             \
        MIDB;                                                                                         \
        if (!(ABL & 4)) __builtin_amdgcn_s_setprio(0);                                                \
        if (TAIL == 0) {                                                                              \
            MI355_TR_STAMP(tr_on, tr_addr);                                                           \
            MI355_BARRIER();                                                                          \
        }                                                                                             \
    } while (0)

    // staging cursors: position g+1 (c1) and g+2 (c2) of the workgroup's K-step sequence = (tile base, K offset, K-steps
    // done in that tile); past the last tile they stay on it (dummy re-stage of valid memory, drained before the exit)
    const int rot = (ABL & 128) ? (l % T) * kRowB : 0;  // K offset the walk starts at (the sum over K-steps is order-free)
    const char* c1_base = (const char*)a.shadow + (int64_t)(a.ct0 + ctl) * kT2 * row_bytes;
    int c1_k = rot, c1_n = 0, c1_ctl = ctl;
    const char* c2_base;
    int c2_k, c2_n, c2_ctl;
#define KB_ADVANCE(BASE, K, N, CTL)                                                                   \
    do {                                                                                              \
        K += kRowB;                                                                                   \
        if (K == kend) K = 0;                                                                         \
        if (++N == T) {                                                                               \
            N = 0;                                                                                    \
            if (CTL + cstep < a.n_ctiles) {                                                           \
                CTL += cstep;                                                                         \
                BASE += tile_stride_bytes;                                                            \
            }                                                                                         \
        }                                                                                             \
    } while (0)

    // ---- prologue: K-step 0 completely (parity 0), A0 and B0 of K-step 1 (parity 1)
#pragma unroll
    for (int u = 0; u < 2; ++u) kb_stage<0, SADDR>(smem, wave, 0, c1_base + c1_k, voffA, u);
#pragma unroll
    for (int u = 0; u < 2; ++u) kb_stage<1, SADDR>(smem, wave, 0, baseB + c1_k, voffB, u);
#pragma unroll
    for (int u = 0; u < 2; ++u) kb_stage<2, SADDR>(smem, wave, 0, baseB + half_B + c1_k, voffB, u);
#pragma unroll
    for (int u = 0; u < 2; ++u) kb_stage<3, SADDR>(smem, wave, 0, c1_base + half_A + c1_k, voffA, u);
    KB_ADVANCE(c1_base, c1_k, c1_n, c1_ctl);  // c1 = position 1
#pragma unroll
    for (int u = 0; u < 2; ++u) kb_stage<0, SADDR>(smem, wave, 1, c1_base + c1_k, voffA, u);
#pragma unroll
    for (int u = 0; u < 2; ++u) kb_stage<1, SADDR>(smem, wave, 1, baseB + c1_k, voffB, u);
    c2_base = c1_base;
    c2_k = c1_k;
    c2_n = c1_n;
    c2_ctl = c1_ctl;
    KB_ADVANCE(c2_base, c2_k, c2_n, c2_ctl);  // c2 = position 2
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // K-step 0 has landed
    MI355_BARRIER();
    if constexpr (FAN) KB_READ_A0_PART(0, 0, 0, fan);  // (every later K-step gets these in phase 3 of its predecessor)
    if (group == 1) MI355_BARRIER();  // stagger: group 1's LOAD halves line up with group 0's MFMA halves

    const int row_end = (int)a.row_end;
    int par = 0, t = 0;
    int gk = 0;  // developer trace (ABL bit 4): see MI355_TR_* in k_screen256.h
    bool tr_on = false;
    unsigned tr_addr = lds_addr(smem + kTraceOff + group * (kTraceStamps * 8));
    unsigned long long tr_l = 0;
    int row0_cur = (a.ct0 + ctl) * kT2, row0_prev = row0_cur;  // rows < 2^31 (checked by the host)
    I8Group gcur[2][2] = {}, gprev[2] = {};  // [row half i][row block rb]; gprev: half 1 of the previous tile (tested one K-step late)
    KB_LOAD_GROUPS(row0_cur);
    KB_GROUPS_LANDED();
    for (;;) {
        const bool first = t == 0, last = t + 1 == T;
        if (first && que_n > kWaveQueueCap / 2) {  // wave-uniform, rare: this wave stalls on vector memory once
            wave_queue_flush(a, que, min(que_n, kWaveQueueCap));
            que_n = 0;
        }
        if (ABL & 16) tr_on = blockIdx.x == 0 && (wave & 3) == 0 && gk >= kTraceG0 && gk < kTraceG0 + kTraceSteps;
        // phase 0: quadrant (0,0); stage B1 of K-step g+1
        KB_PHASE(KB_READ_B(0, par, fb0); if constexpr (FAN) KB_READ_A0_PART(par, 2, 0, fa); else KB_READ_A(0, par), 2, par ^ 1,
                 baseB + half_B + c1_k, voffB,
                 if (first) KB_ZERO(0, 0), 1, 0, 0, fb0, , );
        // phase 1: quadrant (0,1); stage A1 of K-step g+1; the previous tile's last quadrant is tested here
        KB_PHASE(KB_READ_B(1, par, fb1), 3, par ^ 1, c1_base + half_A + c1_k, voffA,
                 if (first) { KB_TEST(1, 0, row0_prev, gprev); KB_ZERO(0, 1); } if (last) KB_TEST(0, 0, row0_cur, gcur[0]), 1, 0, 1, fb1, , );
        // phase 2: quadrant (1,1); stage A0 of K-step g+2 (its slot was read in phase 0 of this K-step)
        KB_PHASE(KB_READ_A(1, par), 0, par, c2_base + c2_k, voffA,
                 if (first) KB_ZERO(1, 1); if (last) KB_TEST(0, 1, row0_cur, gcur[0]), FAN ? 1 : 0, 1, 1, fb1, , );
        // phase 3: quadrant (1,0); stage B0 of K-step g+2
        KB_PHASE(if constexpr (FAN) KB_READ_A0_PART(par ^ 1, 0, 0, fan), 1, par, baseB + c2_k, voffB,
                 if (first) KB_ZERO(1, 0); if (last) KB_TEST(1, 1, row0_cur, gcur[1]);
                 if constexpr (!(ABL & 4096)) { if (!(ABL & 8192) || last) KB_LOAD_GROUPS((a.ct0 + c1_ctl) * kT2); }, 1, 1, 0, fb0,
                 , if constexpr (!(ABL & 4096)) { if constexpr ((ABL & 16384) != 0) KB_PIN(); KB_GROUPS_LANDED(); });

        par ^= 1;
        ++gk;
        c1_base = c2_base;
        c1_k = c2_k;
        c1_n = c2_n;
        c1_ctl = c2_ctl;
        KB_ADVANCE(c2_base, c2_k, c2_n, c2_ctl);
        if (last) {
            row0_prev = row0_cur;
            if (ctl + cstep >= a.n_ctiles) break;
            ctl += cstep;
            row0_cur = (a.ct0 + ctl) * kT2;
            t = 0;
        } else {
            ++t;
        }
    }
    KB_TEST(1, 0, row0_prev, gprev);  // the last tile's last quadrant
    if (group == 0) MI355_BARRIER();  // balance the stagger barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy prefetches must land before the LDS is freed
    if ((ABL & 16) && blockIdx.x == 0 && (wave & 3) == 0) {  // dump the trace
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long* src = (const unsigned long long*)(smem + kTraceOff + group * (kTraceStamps * 8));
        for (int i = lane; i < kTraceStamps; i += 64) g_trace_out[group * kTraceStamps + i] = src[i];
    }

#undef KB_READ_A
#undef KB_READ_B
#undef KB_READ_A0_PART
#undef KB_STG
#undef KB_PIN
#undef KB_ZERO
#undef KB_TEST
#undef KB_LOAD_GROUPS
#undef KB_GROUPS_LANDED
#undef KB_WAIT_VM_N
#undef KB_PHASE
#undef KB_ADVANCE

    // ---- flush this wave's candidate queue: one global atomic per entry
    wave_queue_flush(a, que, min(que_n, kWaveQueueCap));
}

}  // namespace mi355
