// k_screen_rq1.h -- EXPERIMENT (round 5): k_screen_rq at ONE wave per SIMD.  Workgroup = 4 waves; wave w = queries [64 w, +64) of
// the 256-query tile x all 128 rows: 8 accumulator blocks (128 registers) + 8 KS query fragments (192 at d = 768) + the ring of
// row fragments (32) -- a 512-register wave (the form round 4's verdict sketched first).  Every row fragment read from the LDS
// feeds TWO MFMAs (half the LDS reads per multiply-add of k_screen_rq) and four waves meet at the K-step barrier instead of
// eight.  Same tile, ring, persistent walk, hit-lane queues, records and drift limiter as k_screen_rq.h -- see there for
// everything not said here.  Measured against k_screen_rq in tools/screen_ab (variant 200 + ABL): profiles/r05_rq1_ab.txt.
// NOT launched by the library.
#pragma once
#include "k_screen_rq_abl.h"

namespace mi355 {

// ABL: 1 no fragment reads, 4 no tests, 8 no barrier, 16 no LDS-DMA, 32 no vmcnt at the hand-over, 4096 no drift limiter
template <int KS, int ABL, bool I8>
__global__ __launch_bounds__(256, 1) void k_screen_rq1(ScreenArgs2 a) {
    constexpr int NST = rq_stages(KS);
    static_assert(NST % KS == 0, "a tile never wraps the ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..3
    const unsigned lq = lds_addr(smem + rq_que_off(KS) + wave * kLaneQueueBytes);
    int lq_n = 0, lq_ovf = 0;

    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int l = b >> 3;
    const int cslot = l / a.n_qtiles;
    const int qt = l - cslot * a.n_qtiles;
    const int cstep = ((int)(gridDim.x >> 3) / a.n_qtiles) * 8;
    int ctl = cslot * 8 + xcd;
    if (ctl >= a.n_ctiles) return;
    const int q0 = qt * kT2 + 64 * wave;  // this wave's first query; half h = queries q0 + 32 h + (lane & 31)
    const int row_bytes = a.row_bytes;

    // ---- the wave's query operand: 2 halves x 4 KS fragments
    bf16x8 fB[2][4 * KS];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const char* qp = (const char*)a.qhat + (int64_t)(q0 + 32 * h + (lane & 31)) * row_bytes + 16 * (lane >> 5);
#pragma unroll
        for (int i = 0; i < 4 * KS; ++i) fB[h][i] = __builtin_bit_cast(bf16x8, *(const uint4*)(qp + 32 * i));
    }
    // (complete here: see k_screen_rq.h.  Register halves are left to the allocator: it keeps the accumulators in AGPRs, so every
    // block test pays 16 v_accvgpr_read; pinning the query operand to AGPRs and the accumulators to VGPRs with "+a" / "+v"
    // constraints at every tile start made it worse -- profiles/r05_rq1_ab.txt)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4 * KS; ++i) asm volatile("" : "+v"(fB[h][i]));
    // ---- DMA sources: this wave stages local rows [32 wave + 8 u, +8), u = 0..3, of every stage
    unsigned voff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = (4 * wave + u) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        voff[u] = (unsigned)(r * row_bytes + c * 16);
    }
    int offA;
    {
        const int r = lane & 31, g = lane >> 5;
        offA = r * kRowB + ((g ^ ((r >> 1) & 7)) << 4);
    }
    float th[2], scq[2], kqq[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ql = q0 + 32 * h + (lane & 31);
        th[h] = a.thr[ql];
        scq[h] = I8 ? a.sc[ql] : 1.0f;
        kqq[h] = I8 ? a.kq[ql] : 1.0f;
        asm volatile("" ::"v"(th[h]), "v"(kqq[h]), "v"(scq[h]));
    }

    f32x16 acc[4][2];  // [row block][query half]
    int tg = 0;
    float rec_s[4] = {1.0f, 1.0f, 1.0f, 1.0f}, rec_e[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // the tile's raw records (step, err)
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
    acc[2][0] = zero16, acc[2][1] = zero16, acc[3][0] = zero16, acc[3][1] = zero16;
    const unsigned lds0 = lds_addr(smem);
    const unsigned rec_lds = lds0 + rq_rec_off(KS);
    const unsigned rec_voff = (unsigned)((lane & 7) * 4);
    bool lim = (ABL & 4096) == 0 && a.drift > 0 && a.progress != nullptr && a.n_qtiles > 1 && a.n_qtiles <= 8 && wave == 0;
    const char* const prog_base = (const char*)(a.progress + (cslot * 8 + xcd) * 8);
    const unsigned prog_lds = lds0 + rq_prog_off(KS);
    const unsigned prog_voff = (unsigned)(((lane & 7) < a.n_qtiles ? (lane & 7) : 0) * 4);
    const int stamp = a.epoch << 20;

#define R1_PIN() __builtin_amdgcn_sched_barrier(0)
    const int64_t tile_bytes = (int64_t)kRqRows * row_bytes;
    const char* c_base = (const char*)a.shadow + (int64_t)(a.ct0 + ctl) * tile_bytes;
    const int64_t tile_stride_bytes = (int64_t)cstep * tile_bytes;
    int c_k = 0, c_n = 0, c_ctl = ctl, c_tc = 0;
    unsigned c_dst = lds0 + (unsigned)(4 * wave) * 1024u;
#define R1_PIECE(U)                                                                                   \
    do {                                                                                              \
        if constexpr ((ABL & 16) == 0) glds16_saddr(c_base + c_k, voff[U], c_dst + (U) * 1024u);       \
    } while (0)
#define R1_REC()                                                                                      \
    do {                                                                                              \
        if constexpr (I8 && (ABL & 16) == 0)                                                          \
            if (c_n == 0)                                                                             \
                glds4_saddr((const char*)a.grp + (int64_t)(a.ct0 + c_ctl) * (kRqRows / kI8GroupRows * (int)sizeof(I8Group)), \
                            rec_voff, rec_lds + (unsigned)(c_tc & (kRqRecSlots - 1)) * 256u);         \
    } while (0)
#define R1_ADVANCE()                                                                                  \
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

    constexpr int kPF = 3;
    bf16x8 fAq[4][2];
    if constexpr ((ABL & 1) != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" : "=v"(fAq[i][j]));
    }
#define R1_PREFETCH(M2, SB, SBN)                                                                      \
    do {                                                                                              \
        if constexpr ((ABL & 1) == 0) {                                                               \
            constexpr int m2__ = (M2) & 7;                                                            \
            const char* r__ = smem + ((M2) >= 8 ? (SBN) : (SB)) + (2 * (m2__ >> 2)) * 4096;           \
            _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                          \
                fAq[m2__ & 3][rb] = __builtin_bit_cast(bf16x8, *(const uint4*)(r__ + rb * 4096 + (offA ^ ((m2__ & 3) * 32)))); \
        }                                                                                             \
    } while (0)
// the four MFMAs of micro-step M (row blocks 2 (M >> 2) + rb, query halves h); with TEST: the four parts of block (TB, TH)'s
// maximum ride behind them
#define R1_MFMA(M, TT, ZERO, RB, H)                                                                   \
    acc[2 * ((M) >> 2) + (RB)][H] = screen_mfma<I8>(fAq[(M) & 3][RB], fB[H][4 * (TT) + ((M) & 3)],    \
                                                    (ZERO) ? zero16 : acc[2 * ((M) >> 2) + (RB)][H])
#define R1_MM(M, TT, ZERO, TEST, TB, TH)                                                              \
    do {                                                                                              \
        R1_MFMA(M, TT, ZERO, 0, 0);                                                                   \
        if constexpr ((TEST) && (ABL & 4) == 0) { R1_PIN(); tg = screen_block_max_part<I8, 0>(acc[TB][TH], tg); R1_PIN(); } \
        R1_MFMA(M, TT, ZERO, 0, 1);                                                                   \
        if constexpr ((TEST) && (ABL & 4) == 0) { R1_PIN(); tg = screen_block_max_part<I8, 1>(acc[TB][TH], tg); R1_PIN(); } \
        R1_MFMA(M, TT, ZERO, 1, 0);                                                                   \
        if constexpr ((TEST) && (ABL & 4) == 0) { R1_PIN(); tg = screen_block_max_part<I8, 2>(acc[TB][TH], tg); R1_PIN(); } \
        R1_MFMA(M, TT, ZERO, 1, 1);                                                                   \
        if constexpr ((TEST) && (ABL & 4) == 0) { R1_PIN(); tg = screen_block_max_part<I8, 3>(acc[TB][TH], tg); }           \
    } while (0)
#define R1_LOAD_REC(TC)                                                                               \
    do {                                                                                              \
        if constexpr (I8 && (ABL & 4) == 0) {                                                         \
            const float4* rp__ = (const float4*)(smem + rq_rec_off(KS) + ((TC) & (kRqRecSlots - 1)) * 256); \
            const float4 r0__ = rp__[0], r1__ = rp__[1];                                              \
            rec_s[0] = r0__.x, rec_e[0] = r0__.y, rec_s[1] = r0__.z, rec_e[1] = r0__.w;               \
            rec_s[2] = r1__.x, rec_e[2] = r1__.y, rec_s[3] = r1__.z, rec_e[3] = r1__.w;               \
        }                                                                                             \
    } while (0)
#define R1_TEST(RB, H, ROW0)                                                                          \
    do {                                                                                              \
        if constexpr ((ABL & 4) == 0) {                                                               \
            int lane_e = lane;                                                                        \
            asm volatile("" : "+v"(lane_e));                                                          \
            const int q__ = q0 + 32 * (H) + (lane_e & 31);                                            \
            const int rbase__ = (ROW0) + 32 * (RB) + 4 * (lane_e >> 5);                               \
            I8Blk blk__{1.0f, 0.0f};                                                                  \
            if constexpr (I8) blk__ = I8Blk{rec_s[RB] * scq[H], rec_e[RB] * kqq[H]};                  \
            screen_test_block_lq_max<I8, 0>(a, a.status, row_end, acc[RB][H], tg, q__, rbase__, th[H], blk__, lq, lq_n, lq_ovf); \
        }                                                                                             \
    } while (0)
#define R1_MICRO(M, TT, ZERO, SB, SBN, TEST, TB, TH)                                                  \
    do {                                                                                              \
        R1_PREFETCH((M) + kPF, SB, SBN);                                                              \
        R1_PIN();                                                                                     \
        R1_MM(M, TT, ZERO, TEST, TB, TH);                                                             \
        R1_PIN();                                                                                     \
    } while (0)

    constexpr bool kSkipLast = KS >= 2;
    constexpr int kPrologueSteps = kSkipLast ? NST - 1 : NST;
#pragma unroll
    for (int s = 0; s < kPrologueSteps; ++s) {
        R1_PIECE(0);
        R1_PIECE(1);
        R1_PIECE(2);
        R1_PIECE(3);
        R1_REC();
        R1_ADVANCE();
    }
    if constexpr ((ABL & 16) == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kPrologueSteps - 1)) : "memory");
    MI355_BARRIER();
    int s0b = 0;
    int tc = 0;
    R1_PREFETCH(0, 0, 0);
    R1_PREFETCH(1, 0, 0);
    R1_PREFETCH(2, 0, 0);

    const int row_end = (int)a.row_end;
    int row0_cur = (a.ct0 + ctl) * kRqRows, row0_prev = row0_cur;
    bool have_prev = false;
    for (;;) {
        if (lq_n > kLaneQueueFlushAt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lane_queue_flush<I8>(a, lq, lq_n, row_end);
            lq_n = 0;
        }
        if (lim) {
            const int mine = stamp | (tc + 1);
            if (lane == 0) asm volatile("global_store_dword %0, %1, %2 sc1" ::"v"(qt * 4), "v"(mine), "s"(prog_base) : "memory");
            if (tc > 0) {
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
            const bool tp = first && have_prev;
            // the previous tile's row half 1 (blocks (2, 0), (2, 1), (3, 0), (3, 1)): one block per micro-step 0..3 of the first K-step
            if (first) {
                R1_MICRO(0, t, first, sb, sbn, true, 2, 0);
                if (tp) R1_TEST(2, 0, row0_prev);
                R1_MICRO(1, t, false, sb, sbn, true, 2, 1);
                if (tp) R1_TEST(2, 1, row0_prev);
                R1_MICRO(2, t, false, sb, sbn, true, 3, 0);
                if (tp) R1_TEST(3, 0, row0_prev);
                R1_MICRO(3, t, false, sb, sbn, true, 3, 1);
                if (tp) R1_TEST(3, 1, row0_prev);
            } else {
                R1_MICRO(0, t, first, sb, sbn, false, 0, 0);
                R1_MICRO(1, t, false, sb, sbn, false, 0, 0);
                R1_MICRO(2, t, false, sb, sbn, false, 0, 0);
                R1_MICRO(3, t, false, sb, sbn, false, 0, 0);
            }
            if (last) R1_LOAD_REC(tc);
            if (last) {
                R1_MICRO(4, t, first, sb, sbn, true, 0, 0);
                R1_TEST(0, 0, row0_cur);
            } else {
                R1_MICRO(4, t, first, sb, sbn, false, 0, 0);
            }
            const bool hand = !(kSkipLast && last);
            if (hand) {
                const int allowed = 4 * (NST - 2 - ((kSkipLast && first) ? 1 : 0) - ((kSkipLast && t == KS - 2) ? 1 : 0));
                if constexpr ((ABL & 48) == 0) {
                    if (allowed >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    else if (allowed == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else if (allowed == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else if (allowed == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr ((ABL & 8) == 0) MI355_BARRIER();
            }
            R1_PIN();
            if (last) {
                R1_MICRO(5, t, false, sb, sbn, true, 0, 1);
                R1_TEST(0, 1, row0_cur);
            } else {
                R1_MICRO(5, t, false, sb, sbn, false, 0, 0);
            }
            if (hand) {
                R1_PIECE(0);
                R1_PIECE(1);
            }
            R1_PIN();
            if (last) {
                R1_MICRO(6, t, false, sb, sbn, true, 1, 0);
                R1_TEST(1, 0, row0_cur);
            } else {
                R1_MICRO(6, t, false, sb, sbn, false, 0, 0);
            }
            if (hand) {
                R1_PIECE(2);
                R1_PIECE(3);
            }
            R1_PIN();
            if (hand && kSkipLast && first) {  // two stages were handed over: the second K-step's pieces ride micro-step 7
                R1_REC();
                R1_ADVANCE();
                R1_PIECE(0);
                R1_PIECE(1);
                R1_PIN();
            }
            if (last) {
                R1_MICRO(7, t, false, sb, sbn, true, 1, 1);
                R1_TEST(1, 1, row0_cur);
            } else {
                R1_MICRO(7, t, false, sb, sbn, false, 0, 0);
            }
            if (hand && kSkipLast && first) {
                R1_PIECE(2);
                R1_PIECE(3);
            }
            if (hand) {
                R1_REC();
                R1_ADVANCE();
            }
            R1_PIN();
        }
        row0_prev = row0_cur;
        have_prev = true;
        ++tc;
        if constexpr (NST != KS) s0b = next_tile_sb;
        if (ctl + cstep >= a.n_ctiles) break;
        ctl += cstep;
        row0_cur = (a.ct0 + ctl) * kRqRows;
    }
    // the last tile's row half 1
#define R1_TEST_WHOLE(RB, H)                                                                          \
    do {                                                                                              \
        if constexpr ((ABL & 4) == 0) {                                                               \
            tg = screen_block_max_part<I8, 0>(acc[RB][H], tg);                                        \
            tg = screen_block_max_part<I8, 1>(acc[RB][H], tg);                                        \
            tg = screen_block_max_part<I8, 2>(acc[RB][H], tg);                                        \
            tg = screen_block_max_part<I8, 3>(acc[RB][H], tg);                                        \
        }                                                                                             \
        R1_TEST(RB, H, row0_prev);                                                                    \
    } while (0)
    R1_TEST_WHOLE(2, 0);
    R1_TEST_WHOLE(2, 1);
    R1_TEST_WHOLE(3, 0);
    R1_TEST_WHOLE(3, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((ABL & 4096) == 0 && a.drift > 0 && a.progress != nullptr && wave == 0 && lane == 0)
        __hip_atomic_store((int*)prog_base + qt, stamp | kRqDoneTiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lane_queue_flush<I8>(a, lq, lq_n, row_end);
    if constexpr ((ABL & 4) != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) asm volatile("" ::"v"(acc[i][h]));
    }
#undef R1_PIN
#undef R1_PIECE
#undef R1_REC
#undef R1_ADVANCE
#undef R1_PREFETCH
#undef R1_MFMA
#undef R1_MM
#undef R1_LOAD_REC
#undef R1_TEST
#undef R1_MICRO
#undef R1_TEST_WHOLE
}

}  // namespace mi355
