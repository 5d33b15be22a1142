// k_screen256c.h -- third form of the large-block screen: same tile (256 corpus rows x 256 queries, 8 waves, persistent),
// same LDS image, same epilogue as k_screen256b -- but NO ping-pong.
//
// What the second form's A/B table says: its two groups of four waves hand the matrix pipe back and
// forth at a barrier twice per phase, every half-phase is ~400 cycles of which 256 are MFMA time, and the halves are
// balanced so tightly that every instruction added to either is paid in full -- a ceiling of ~65 % of the instruction's
// rate.  Here every wave runs its own software pipeline and the two waves that share a SIMD cover each other's stalls:
//
//   per K-step (128 B of K = 4 sub-steps kk) and wave: 32 MFMAs (128 rows x 64 queries), 24 ds_read_b128, 8 LDS-DMA
//   pieces, ONE workgroup barrier.
//       kk = 0..2:  issue the 6 fragment reads of sub-step kk+1 (4 x A, 2 x B: 24 VGPRs, double-buffered), then the
//                   8 MFMAs of sub-step kk
//       kk = 3:     [own DMA pieces of K-step g+1 landed: vmcnt(0)] [barrier: everybody's landed, everybody has read
//                   the last fragment of K-step g]  -> fragment reads of sub-step 0 of K-step g+1 (other ring parity),
//                   the 8 DMA pieces of K-step g+2 into the parity just freed, between the 8 MFMAs of sub-step 3
//   The barrier sits between two MFMA bursts of the same wave, the fragments of the next K-step are already on their way
//   when its first MFMA issues, and a DMA piece has a whole K-step (~2000 cycles) to land: the lead-time probe of round 2
//   (counted waits shortened to two phases / one phase of the second form) shows the data is there after ~800.
//   Threshold tests and accumulator zeroing happen once per tile, after the last MFMA of its last K-step; while one
//   wave of a SIMD tests, nothing stops the other one.
// int8 row-group records (step, residual norm of every 32 rows): one more LDS-DMA piece per K-step -- a dword per lane, the
//   256-row tile's 64-byte line four times over -- into a ring of 4 x 256 B, for the tile of the K-step being staged; every
//   wave issues it (identical bytes to the same slot), so the vmcnt(0) + barrier that publish the operands publish it too.
//   Read back with a plain ds_read when the tile is tested.  (A scalar load inside the K loop would turn the counted
//   lgkmcnt waits that keep one sub-step of fragment reads in flight into lgkmcnt(0): DESIGN.md 4.1b.)
// LDS: the ring of 8 half-tile slots (2 parities x A0 B0 B1 A1), the per-wave candidate queues, 1 KiB of records.
#pragma once
#include "k_screen256b.h"

namespace mi355 {

constexpr int kScreen256cAbl = 1024;  // SADDR (the only staging form this kernel has)

template <int ABL, bool I8>
__global__ __launch_bounds__(512, 2) void k_screen256c(ScreenArgs2 a) {
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
    asm volatile("" ::"v"(th[0]), "v"(th[1]), "v"(kqq[0]), "v"(kqq[1]), "v"(scq[0]), "v"(scq[1]));

    f32x16 acc[2][2][2];  // [row half i][row block rb][query half j]
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.0f;
    bf16x8 fA[2][2][2], fB[2][2];  // [buffer][i][rb], [buffer][j]: fragments of one K sub-step, double-buffered
    const int T = a.ksteps;
    const int kend = T * kRowB;
    const unsigned rec_lds = lds_addr(smem + kRecOff);
    const unsigned rec_voff = (unsigned)((lane & 15) * 4);  // the tile's 8 records = 16 dwords, four copies per slot
    int gpos = 0;  // K-step counter (ring position of the records: gpos & 3)
// records of the cursor's tile into slot POS & 3
#define KC_REC_STAGE(POS)                                                                             \
    do {                                                                                              \
        if constexpr (I8)                                                                             \
            glds4_saddr((const char*)a.grp + (int64_t)(a.ct0 + c_ctl) * (kT2 / kI8GroupRows * (int)sizeof(I8Group)), rec_voff, \
                        rec_lds + (unsigned)((POS) & 3) * 256u);                                      \
    } while (0)

#define KC_PIN() __builtin_amdgcn_sched_barrier(0)
// the 6 fragment reads of K sub-step KK of the K-step in ring parity PAR into buffer BUF
#define KC_READ(BUF, PAR, KK)                                                                         \
    do {                                                                                              \
        const char* s__ = smem + (4 * (PAR)) * kHalfBytes;                                            \
        if constexpr ((ABL & 2) != 0) {                                                               \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                             \
                fB[BUF][j] = __builtin_bit_cast(bf16x8, *(const uint4*)(s__ + (1 + j) * kHalfBytes + (offB ^ ((KK) * 32)))); \
        }                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                 \
            _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                          \
                fA[BUF][i][rb] = __builtin_bit_cast(bf16x8, *(const uint4*)(s__ + (i ? 3 : 0) * kHalfBytes + (offA[rb] ^ ((KK) * 32)))); \
        if constexpr ((ABL & 2) == 0) {                                                               \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                             \
                fB[BUF][j] = __builtin_bit_cast(bf16x8, *(const uint4*)(s__ + (1 + j) * kHalfBytes + (offB ^ ((KK) * 32)))); \
        }                                                                                             \
    } while (0)
// the 8 MFMAs of one K sub-step on the fragments in buffer BUF
#define KC_MFMA(BUF)                                                                                  \
    do {                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                 \
            _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                          \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                         \
                    acc[i][rb][j] = screen_mfma<I8>(fA[BUF][i][rb], fB[BUF][j], acc[i][rb][j]);       \
    } while (0)
// one interleaved K sub-step: reads of the next sub-step first, then the MFMAs; the scheduler is told to alternate
#define KC_SUBSTEP(BUF, PAR, KKNEXT)                                                                  \
    do {                                                                                              \
        KC_READ((BUF) ^ 1, PAR, KKNEXT);                                                              \
        KC_MFMA(BUF);                                                                                 \
        if constexpr ((ABL & 1) != 0) { /* burst: all six reads, then the eight MFMAs */              \
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                        \
        } else {                                                                                      \
            _Pragma("unroll") for (int s__ = 0; s__ < 6; ++s__) {                                     \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); /* one DS read  */                 \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* one MFMA     */                 \
            }                                                                                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                        \
        }                                                                                             \
        KC_PIN();                                                                                     \
    } while (0)

    // staging cursor: position of the NEXT K-step to stage = (tile base, K offset, K-steps done in that tile); past the
    // last tile it stays on it (dummy re-stage of valid memory, drained before the exit)
    const char* c_base = (const char*)a.shadow + (int64_t)(a.ct0 + ctl) * kT2 * row_bytes;
    int c_k = 0, c_n = 0, c_ctl = ctl;
#define KC_ADVANCE()                                                                                  \
    do {                                                                                              \
        c_k += kRowB;                                                                                 \
        if (c_k == kend) c_k = 0;                                                                     \
        if (++c_n == T) {                                                                             \
            c_n = 0;                                                                                  \
            if (c_ctl + cstep < a.n_ctiles) {                                                         \
                c_ctl += cstep;                                                                       \
                c_base += tile_stride_bytes;                                                          \
            }                                                                                         \
        }                                                                                             \
    } while (0)
// piece P (0..8) of the cursor's K-step into ring parity PAR: A0 A0 B0 | B0 B1 B1 | A1 A1 records
#define KC_PIECE(PAR, P, RECPOS)                                                                      \
    do {                                                                                              \
        if constexpr ((ABL & 16) == 0) {                                                              \
            if ((P) == 0 || (P) == 1) kb_stage<0, true>(smem, wave, PAR, c_base + c_k, voffA, (P) & 1);           \
            else if ((P) == 2 || (P) == 3) kb_stage<1, true>(smem, wave, PAR, baseB + c_k, voffB, (P) & 1);        \
            else if ((P) == 4 || (P) == 5) kb_stage<2, true>(smem, wave, PAR, baseB + half_B + c_k, voffB, (P) & 1); \
            else if ((P) == 6 || (P) == 7) kb_stage<3, true>(smem, wave, PAR, c_base + half_A + c_k, voffA, (P) & 1); \
            else KC_REC_STAGE(RECPOS);                                                                \
        }                                                                                             \
    } while (0)
// One K sub-step, written out: the 6 fragment reads of sub-step KKNEXT of ring parity RPAR into buffer BUF^1 and up to
// three DMA pieces [P0, P0+NP) of the K-step being staged (parity SPAR) between the 8 MFMAs on buffer BUF.
// Order: reads B0 B1 | A00 A01 | A10 | A11 in front of the MFMA pairs (0,0) (0,1) (1,0) (1,1); one DMA piece behind
// each of the first NP pairs.
#define KC_STEP(BUF, RPAR, KKNEXT, SPAR, P0, NP, RECPOS, ZERO)                                        \
    do {                                                                                              \
        const char* r__ = smem + (4 * (RPAR)) * kHalfBytes;                                           \
        _Pragma("unroll") for (int g__ = 0; g__ < 4; ++g__) {                                         \
            const int i__ = g__ >> 1, rb__ = g__ & 1;                                                 \
            if (g__ == 0) {                                                                           \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                         \
                    fB[(BUF) ^ 1][j] = __builtin_bit_cast(bf16x8, *(const uint4*)(r__ + (1 + j) * kHalfBytes + (offB ^ ((KKNEXT) * 32)))); \
            } else if (g__ == 1) {                                                                    \
                _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                      \
                    fA[(BUF) ^ 1][0][rb] = __builtin_bit_cast(bf16x8, *(const uint4*)(r__ + (offA[rb] ^ ((KKNEXT) * 32)))); \
            } else {                                                                                  \
                fA[(BUF) ^ 1][1][g__ - 2] = __builtin_bit_cast(bf16x8, *(const uint4*)(r__ + 3 * kHalfBytes + (offA[g__ - 2] ^ ((KKNEXT) * 32)))); \
            }                                                                                         \
            KC_PIN();                                                                                 \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                             \
                acc[i__][rb__][j] = screen_mfma<I8>(fA[BUF][i__][rb__], fB[BUF][j], (ZERO) ? zero16 : acc[i__][rb__][j]); \
            KC_PIN();                                                                                 \
            if (g__ < (NP)) {                                                                         \
                KC_PIECE(SPAR, (P0) + g__, RECPOS);                                                   \
                KC_PIN();                                                                             \
            }                                                                                         \
        }                                                                                             \
    } while (0)

    // ---- prologue: K-step 0 completely into parity 0, the first three pieces of K-step 1 into parity 1; K-step 0 landed
    // and visible; first fragments
#pragma unroll
    for (int p = 0; p < 9; ++p) KC_PIECE(0, p, 0);
    KC_ADVANCE();
#pragma unroll
    for (int p = 0; p < 3; ++p) KC_PIECE(1, p, 1);
    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // K-step 0 (+ its records) has landed (this wave's pieces)
    MI355_BARRIER();
    KC_READ(0, 0, 0);

    const int row_end = (int)a.row_end;
    int par = 0, t = 0;
    int row0_cur = (a.ct0 + ctl) * kT2;  // rows < 2^31 (checked by the host)
    for (;;) {
        const bool first = t == 0, last = t + 1 == T;
        if (first && que_n > kWaveQueueCap / 2) {  // wave-uniform, rare: this wave stalls on vector memory once
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_queue_flush(a, que, min(que_n, kWaveQueueCap));
            que_n = 0;
        }
        // K-step g = gpos, ring parity par.  The K-step being staged: g+1 (pieces 3..8, into par^1) in sub-steps 0 and 1, then
        // g+2 (pieces 0..2, into par) behind the barrier of sub-step 3 -- three LDS-DMA instructions per sub-step: all nine
        // in one sub-step (8 waves x 9 through the one texture-address path of the CU) held the matrix pipe up for 20 %.
        // (a tile's first sub-step starts from C = 0 -- an inline constant of the MFMA -- instead of zeroing 128 registers)
        if (first) {
            asm volatile("; first sub-step of a tile");
            KC_STEP(0, par, 1, par ^ 1, 3, 3, gpos + 1, true);
        } else {
            KC_STEP(0, par, 1, par ^ 1, 3, 3, gpos + 1, false);
        }
        KC_STEP(1, par, 2, par ^ 1, 6, 3, gpos + 1, false);
        KC_ADVANCE();
        KC_STEP(0, par, 3, par, 0, 0, 0, false);
        // ---- sub-step 3: hand-over of the ring
        if constexpr ((ABL & 32) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of K-step g+1 have landed
        else if constexpr ((ABL & 64) == 0) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");  // (probe: one K-step more)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // ... and its last fragments of K-step g are in registers
        if constexpr ((ABL & 8) == 0) MI355_BARRIER();
        KC_PIN();
        KC_STEP(1, par ^ 1, 0, par, 0, 3, gpos + 2, false);  // reads: sub-step 0 of K-step g+1; pieces 0..2 of K-step g+2
        par ^= 1;
        if (last && (ABL & 4)) {
            if (ctl + cstep >= a.n_ctiles) break;
            ctl += cstep;
            row0_cur = (a.ct0 + ctl) * kT2;
            t = 0;
        } else if (last) {
            // ---- the tile is complete: threshold tests of its eight blocks (k_screen.h: screen_queue_block)
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            I8Group gcur[2][2] = {};
            if constexpr (I8) {  // this tile's records: staged two K-steps ago (or by the prologue), published by the barriers since
                const I8Group* rp = (const I8Group*)(smem + kRecOff + (gpos & 3) * 256) + 4 * wr;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) gcur[i][rb] = rp[2 * i + rb];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = q0 + 64 * wc + 32 * j + (lane_e & 31);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) {
                        const int rbase = row0_cur + 128 * wr + 64 * i + 32 * rb + 4 * (lane_e >> 5);
                        I8Blk blk{1.0f, 0.0f};
                        if constexpr (I8) blk = i8_blk(gcur[i][rb], scq[j], kqq[j]);
                        screen_queue_block<I8, true>(a, a.status, acc[i][rb][j], q, rbase, row_end, th[j], blk, que, que_n);
                    }
            }
            if (ctl + cstep >= a.n_ctiles) break;
            ctl += cstep;
            row0_cur = (a.ct0 + ctl) * kT2;
            t = 0;
        } else {
            ++t;
        }
        ++gpos;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy prefetches must land before the LDS is freed
    wave_queue_flush(a, que, min(que_n, kWaveQueueCap));

#undef KC_PIN
#undef KC_READ
#undef KC_MFMA
#undef KC_SUBSTEP
#undef KC_ADVANCE
#undef KC_PIECE
#undef KC_STEP
#undef KC_REC_STAGE
}

}  // namespace mi355
