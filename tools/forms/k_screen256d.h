// k_screen256d.h -- fourth form of the large-block screen: the free-running waves of k_screen256c on a ring of FOUR
// half-K-step stages instead of two K-steps.
//
// Why: with the operands of a K-step staged as one unit, a ring slot is released at that K-step's one
// barrier and the LDS-DMA pieces that refill it must land by the next one -- the staging engine idles between "all issued"
// and "slot released", and every K-step waits for its slowest piece.  A DMA-only build of the third form needs 1.06 us per
// K-step of 64 KiB per CU (~32 B/clk, corpus in the Infinity Cache; 1.4 us from HBM); its 64 MFMAs per SIMD need 1.08 us:
// both fit in 1.1 us only if they overlap all the time.  Here a stage is 64 B of K (two MFMA sub-steps): 4 half-tiles x
// 128 rows x 64 B = 32 KiB, the ring holds four (the same 128 KiB), a slot is refilled with the stage FOUR ahead as soon as
// its last fragment has been read, and the hand-over only asks for the NEXT stage: the two stages behind it (8 pieces per
// wave, 64 KiB per CU) stay in flight across the barrier -- a piece has three stages (~1.5 K-steps) to land.
//
// LDS image of a half-tile stage: 128 rows x 64 B, row pitch 64 B, physical 16-B chunk pc = c ^ ((row >> 2) & 3) for logical
//   chunk c: the 16 rows of every ds_read_b128 lane group ({0-3,12-15,20-27} ...) have 16 distinct (row & 15), so the group
//   covers all 64 banks; a DMA piece is lane-linear: lane -> (row 16 w + (lane >> 2), pc = lane & 3), one piece per wave and
//   half-tile stage.  K sub-step kk (0, 1) of lane half g reads logical chunk 2 kk + g: offset ^ (kk * 32), as before.
// K-step = 2 stages = 8 micro-steps mu = 4 s2 + 2 I + kk of 4 MFMAs (row half I, blocks (rb, j)); fragments: the row side
//   through a ring of four micro-steps read two ahead, the query side per stage (double-buffered by stage parity).
// Hand-over of stage S, between its micro-steps 1 and 2 (all its reads are issued by then): vmcnt(8 [+1 records]) --
//   this wave's pieces of stage S+1 have landed --, lgkmcnt(0), barrier; then the 4 pieces of stage S+4 into the slot of S,
//   one per micro-step.
// Tests: a tile's row half 0 is final after micro-step 5 of its last K-step and tested in micro-steps 6, 7; row half 1 is
//   final after micro-step 7 and tested in micro-steps 0, 1 of the next tile's first K-step (its registers restart from
//   C = 0 in micro-step 2).  Append path out of line (k_screen.h: screen_queue_hits).
#pragma once
#include "k_screen256b.h"
#include "k_screen256c.h"

namespace mi355 {

constexpr int kScreen256dAbl = 0;
constexpr int kStageHalf = 128 * 64;        // one half-tile stage: 128 rows x 64 B
constexpr int kStageSlot = 4 * kStageHalf;  // A0 B0 B1 A1
static_assert(4 * kStageSlot == kRingBytes, "the stage ring is the same 128 KiB");

template <int ABL, bool I8>
__global__ __launch_bounds__(512, 2) void k_screen256d(ScreenArgs2 a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
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

    // ---- DMA sources: this wave stages rows [16 w, 16 w + 16) of every half-tile stage
    unsigned voffA, voffB;
    {
        const int r = 16 * wave + (lane >> 2);                // local row 0..127 of the half-tile
        const int c = (lane & 3) ^ ((r >> 2) & 3);            // logical chunk that lives in this lane's LDS position
        const int arow0 = 128 * (r >> 6) + (r & 63);          // + 64 * i
        const int bcol0 = 64 * (r >> 5) + (r & 31);           // + 32 * j
        voffA = (unsigned)(arow0 * (int)row_bytes + c * 16);
        voffB = (unsigned)(bcol0 * (int)row_bytes + c * 16);
    }
    const char* const baseB = (const char*)a.qhat + (int64_t)q0 * row_bytes;
    const int64_t tile_stride_bytes = (int64_t)cstep * kT2 * row_bytes;
    const int64_t half_A = 64 * row_bytes, half_B = 32 * row_bytes;
    // ---- fragment offsets inside a half-tile stage (kk folded in with ^ 32)
    int offA[2], offB;
    {
        const int g = lane >> 5;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int r = wr * 64 + rb * 32 + (lane & 31);
            offA[rb] = r * 64 + ((g ^ ((r >> 2) & 3)) << 4);
        }
        const int r = wc * 32 + (lane & 31);
        offB = r * 64 + ((g ^ ((r >> 2) & 3)) << 4);
    }
    float th[2], scq[2], kqq[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = q0 + 64 * wc + 32 * j + (lane & 31);
        th[j] = a.thr[q];
        scq[j] = I8 ? a.sc[q] : 1.0f;
        kqq[j] = I8 ? a.kq[q] : 1.0f;
    }
    asm volatile("" ::"v"(th[0]), "v"(th[1]), "v"(kqq[0]), "v"(kqq[1]), "v"(scq[0]), "v"(scq[1]));

    f32x16 acc[2][2][2];  // [row half i][row block rb][query half j]
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
    bf16x8 fAq[4][2], fBk[2][2][2];  // [micro-step & 3][rb]; [stage parity][kk][j]
    const int T = a.ksteps;          // K-steps (of two stages) per tile
    const int kend = T * kRowB;
    const unsigned rec_lds = lds_addr(smem + kRecOff);
    const unsigned rec_voff = (unsigned)((lane & 15) * 4);  // the tile's 8 records = 16 dwords, four copies per slot
    int gpos = 0;  // K-step counter (ring position of the records: gpos & 3)

#define KD_PIN() __builtin_amdgcn_sched_barrier(0)
    // staging cursor: the NEXT stage to stage = (tile base, K offset in bytes, stages done in that tile); past the last tile it
    // stays on it (dummy re-stage of valid memory, drained before the exit)
    const char* c_base = (const char*)a.shadow + (int64_t)(a.ct0 + ctl) * kT2 * row_bytes;
    int c_k = 0, c_ctl = ctl;
#define KD_ADVANCE()                                                                                  \
    do {                                                                                              \
        c_k += 64;                                                                                    \
        if (c_k == kend) {                                                                            \
            c_k = 0;                                                                                  \
            if (c_ctl + cstep < a.n_ctiles) {                                                         \
                c_ctl += cstep;                                                                       \
                c_base += tile_stride_bytes;                                                          \
            }                                                                                         \
        }                                                                                             \
    } while (0)
// piece P (0..3: A0 B0 B1 A1; 4: the tile's records, int8 only) of the cursor's stage into ring slot SLOT
#define KD_PIECE(SLOT, P, RECPOS)                                                                     \
    do {                                                                                              \
        if constexpr ((ABL & 16) == 0) {                                                              \
            const unsigned d__ = lds_addr(smem + (SLOT) * kStageSlot + (P) * kStageHalf + wave * 1024); \
            if ((P) == 0) glds16_saddr(c_base + c_k, voffA, d__);                                     \
            else if ((P) == 1) glds16_saddr(baseB + c_k, voffB, d__);                                 \
            else if ((P) == 2) glds16_saddr(baseB + half_B + c_k, voffB, d__);                        \
            else if ((P) == 3) glds16_saddr(c_base + half_A + c_k, voffA, d__);                       \
            else if constexpr (I8)                                                                    \
                glds4_saddr((const char*)a.grp + (int64_t)(a.ct0 + c_ctl) * (kT2 / kI8GroupRows * (int)sizeof(I8Group)), rec_voff, \
                            rec_lds + (unsigned)((RECPOS) & 3) * 256u);                               \
        }                                                                                             \
    } while (0)
// fragment reads for micro-step MU (0..9; 8, 9 = micro-steps 0, 1 of the next K-step): stage slot SL
#define KD_PREFETCH(MU, SL)                                                                           \
    do {                                                                                              \
        constexpr int mu__ = (MU) & 7;                                                                \
        constexpr int i__ = (mu__ >> 1) & 1, kk__ = mu__ & 1, sp__ = (mu__ >> 2) & 1;                 \
        const char* r__ = smem + (SL) * kStageSlot;                                                   \
        if (i__ == 0) {                                                                               \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                             \
                fBk[sp__][kk__][j] = __builtin_bit_cast(bf16x8, *(const uint4*)(r__ + (1 + j) * kStageHalf + (offB ^ (kk__ * 32)))); \
        }                                                                                             \
        _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                              \
            fAq[mu__ & 3][rb] = __builtin_bit_cast(bf16x8, *(const uint4*)(r__ + (i__ ? 3 : 0) * kStageHalf + (offA[rb] ^ (kk__ * 32)))); \
    } while (0)
#define KD_MM(MU, ZERO)                                                                               \
    do {                                                                                              \
        constexpr int i__ = ((MU) >> 1) & 1, kk__ = (MU) & 1, sp__ = ((MU) >> 2) & 1;                 \
        _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                              \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                             \
                acc[i__][rb][j] = screen_mfma<I8>(fAq[(MU) & 3][rb], fBk[sp__][kk__][j], (ZERO) ? zero16 : acc[i__][rb][j]); \
    } while (0)
// test block (I, RB, J) of the tile whose first row is ROW0, records slot RSLOT
#define KD_TEST1(I, RB, J, ROW0, RSLOT)                                                               \
    do {                                                                                              \
        if constexpr ((ABL & 4) == 0) {                                                               \
            int lane_e = lane;                                                                        \
            asm volatile("" : "+v"(lane_e));                                                          \
            const int q__ = q0 + 64 * wc + 32 * (J) + (lane_e & 31);                                  \
            const int rbase__ = (ROW0) + 128 * wr + 64 * (I) + 32 * (RB) + 4 * (lane_e >> 5);         \
            I8Blk blk__{1.0f, 0.0f};                                                                  \
            if constexpr (I8) {                                                                       \
                const I8Group g__ = ((const I8Group*)(smem + kRecOff + ((RSLOT) & 3) * 256))[4 * wr + 2 * (I) + (RB)]; \
                blk__ = i8_blk(g__, scq[J], kqq[J]);                                                  \
            }                                                                                         \
            screen_test_block<I8>(a.status, acc[I][RB][J], q__, rbase__, row_end, th[J], blk__, que, que_n); \
        }                                                                                             \
    } while (0)
// one micro-step: [reads for MU + 2 from slot SLR] [4 MFMAs] [one DMA piece P (or none: P < 0) into slot SLW]
#define KD_MICRO(MU, ZERO, SLR, SLW, P, RECPOS)                                                       \
    do {                                                                                              \
        KD_PREFETCH((MU) + 2, SLR);                                                                   \
        KD_PIN();                                                                                     \
        KD_MM(MU, ZERO);                                                                              \
        KD_PIN();                                                                                     \
        if ((P) >= 0) KD_PIECE(SLW, P, RECPOS);                                                       \
        KD_PIN();                                                                                     \
    } while (0)
// hand-over in the middle of a stage: the next stage is visible afterwards, this stage's slot is free
#define KD_HANDOVER()                                                                                 \
    do {                                                                                              \
        if constexpr ((ABL & 32) == 0) {                                                              \
            if constexpr (I8) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");                        \
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                     \
        }                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
        if constexpr ((ABL & 8) == 0) MI355_BARRIER();                                                \
        KD_PIN();                                                                                     \
    } while (0)

    // ---- prologue: stages 0..2 into slots 0..2 (records with the even stages) and the first two pieces of stage 3 (the loop's
    // first K-step issues the other two, as every K-step does for the stage it finds half staged); stage 0 landed and visible;
    // fragments of micro-steps 0 and 1
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int p = 0; p < 4; ++p) KD_PIECE(s, p, 0);
        if ((s & 1) == 0) KD_PIECE(s, 4, s >> 1);
        KD_ADVANCE();
    }
    KD_PIECE(3, 0, 0);
    KD_PIECE(3, 1, 0);
    if constexpr (I8) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");  // stage 0 (+ the first tile's records) has landed
    else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    MI355_BARRIER();
    KD_PREFETCH(0, 0);
    KD_PREFETCH(1, 0);

    const int row_end = (int)a.row_end;
    int t = 0;
    int sl = 0;  // ring slot of the K-step's first stage: 0 or 2
    int row0_cur = (a.ct0 + ctl) * kT2, row0_prev = row0_cur;  // rows < 2^31 (checked by the host)
    bool have_prev = false;  // a finished tile's row half 1 is waiting for its tests
    for (;;) {
        const bool first = t == 0, last = t + 1 == T;
        if (first && que_n > kWaveQueueCap / 2) {  // wave-uniform, rare: this wave stalls on vector memory once
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_queue_flush(a, que, min(que_n, kWaveQueueCap));
            que_n = 0;
        }
        const bool tp = first && have_prev;  // test the previous tile's row half 1 under this K-step's first micro-steps
        const int s0 = sl, s1 = sl + 1, s2 = sl ^ 2;  // slots of this K-step's two stages and of the next K-step's first
        // Staging (one piece per micro-step, cursor = the stage being staged): micro-steps 0, 1 finish stage 2g+3 (slot s2 + 1,
        // released by the previous K-step's second hand-over); 2..5 stage 2g+4 into s0 (+ its K-step's records); 6, 7 start
        // stage 2g+5 in s1.
        if (first) {  // (a tile's first MFMA per block starts from C = 0 -- an inline constant -- instead of zeroing registers)
            asm volatile("; first K-step of a tile, row half 0");
            KD_MICRO(0, true, s0, s2 + 1, 2, 0);
        } else {
            KD_MICRO(0, false, s0, s2 + 1, 2, 0);
        }
        if (tp) {
            KD_TEST1(1, 0, 0, row0_prev, gpos - 1);
            KD_TEST1(1, 0, 1, row0_prev, gpos - 1);
        }
        KD_MICRO(1, false, s0, s2 + 1, 3, 0);
        KD_ADVANCE();
        if (tp) {
            KD_TEST1(1, 1, 0, row0_prev, gpos - 1);
            KD_TEST1(1, 1, 1, row0_prev, gpos - 1);
        }
        KD_HANDOVER();  // stage 1 (slot s1) visible, slot s0 free: stage + 4 goes there
        if (first) {
            asm volatile("; first K-step of a tile, row half 1");
            KD_MICRO(2, true, s1, s0, 0, 0);
        } else {
            KD_MICRO(2, false, s1, s0, 0, 0);
        }
        KD_MICRO(3, false, s1, s0, 1, 0);
        // ---- stage 1 of the K-step (slot s1)
        KD_MICRO(4, false, s1, s0, 2, 0);
        KD_MICRO(5, false, s1, s0, 3, 0);
        KD_PIECE(s0, 4, gpos + 2);  // the records of the K-step whose first stage was just staged
        KD_ADVANCE();
        KD_HANDOVER();  // the next K-step's first stage (slot s2) visible, slot s1 free
        KD_MICRO(6, false, s2, s1, 0, 0);
        if (last) {
            KD_TEST1(0, 0, 0, row0_cur, gpos);
            KD_TEST1(0, 0, 1, row0_cur, gpos);
        }
        KD_MICRO(7, false, s2, s1, 1, 0);
        if (last) {
            KD_TEST1(0, 1, 0, row0_cur, gpos);
            KD_TEST1(0, 1, 1, row0_cur, gpos);
        }
        sl ^= 2;
        ++gpos;
        if (last) {
            row0_prev = row0_cur;
            have_prev = true;
            if (ctl + cstep >= a.n_ctiles) break;
            ctl += cstep;
            row0_cur = (a.ct0 + ctl) * kT2;
            t = 0;
        } else {
            ++t;
        }
    }
    // the last tile's row half 1
    KD_TEST1(1, 0, 0, row0_prev, gpos - 1);
    KD_TEST1(1, 0, 1, row0_prev, gpos - 1);
    KD_TEST1(1, 1, 0, row0_prev, gpos - 1);
    KD_TEST1(1, 1, 1, row0_prev, gpos - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy prefetches must land before the LDS is freed
    wave_queue_flush(a, que, min(que_n, kWaveQueueCap));

#undef KD_PIN
#undef KD_ADVANCE
#undef KD_PIECE
#undef KD_PREFETCH
#undef KD_MM
#undef KD_TEST1
#undef KD_MICRO
#undef KD_HANDOVER
}

}  // namespace mi355
