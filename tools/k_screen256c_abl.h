// tools/k_screen256c_abl.h -- the ABLATION build of csrc/k_screen256c.h: the same kernel with its timing forms (template parameter ABL) for
// tools/screen_ab.hip.  Not part of the library: the product header carries the kernel alone (round 6).  Keep in step by hand.
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
#include "k_screen256_common.h"

namespace mi355 {

constexpr int kScreen256cAbl = 1024;  // SADDR (the only staging form this kernel has)

// LDS-DMA pieces per micro-step slot {5, 6, 7 | 0, 1, 2, 3, 4} for schedule id (0 = what the library runs; the others are the
// A/B of round 2 (interleaved): +1.5 ... +5 %.  More pieces right behind the hand-over, or a thinner, longer spread: both lose.)
__host__ __device__ constexpr int kc_sched(int id, int slot) {
    constexpr int t[4][8] = {{2, 2, 0, 2, 2, 1, 0, 0}, {2, 2, 0, 1, 1, 1, 1, 1}, {1, 1, 1, 1, 1, 2, 2, 0}, {2, 2, 0, 2, 1, 1, 1, 0}};
    return t[id][slot];
}
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
// piece P (0..8) of the cursor's K-step into ring parity PAR.  Issue order: the corpus half-tiles first (A0 A0 A1 A1), then
// the query half-tiles (B0 B0 B1 B1), then the records -- the pieces that may have to come from HBM get the longest lead
// (-1.6 % against A0 B0 | B1 A1).  
#define KC_PIECE(PAR, P, RECPOS)                                                                      \
    do {                                                                                              \
        if constexpr ((ABL & 16) == 0) {                                                              \
            const int c__ = (P);                                                                      \
            if (c__ == 0 || c__ == 1) kb_stage<0, true>(smem, wave, PAR, c_base + c_k, voffA, c__ & 1);            \
            else if (c__ == 2 || c__ == 3) kb_stage<3, true>(smem, wave, PAR, c_base + half_A + c_k, voffA, c__ & 1); \
            else if (c__ == 4 || c__ == 5) kb_stage<1, true>(smem, wave, PAR, baseB + c_k, voffB, c__ & 1);        \
            else if (c__ == 6 || c__ == 7) kb_stage<2, true>(smem, wave, PAR, baseB + half_B + c_k, voffB, c__ & 1); \
            else KC_REC_STAGE(RECPOS);                                                                \
        }                                                                                             \
    } while (0)
// ---- I-major K-step: 8 micro-steps m = 4 I + kk of 4 MFMAs each -- first the row half I = 0 over the four K sub-steps,
// then I = 1 -- so that a finished tile's blocks are tested UNDER MFMAs of the other row half: I = 0 (final after micro-step 3
// of the tile's last K-step) during micro-steps 4..7 of that K-step, I = 1 during micro-steps 0..3 of the next tile's first
// K-step.  (All eight tests after the last MFMA, both waves of every SIMD at once, idled the matrix pipe for 20 % of the
// kernel.)  The query fragments of a K-step stay in registers (fBk, 8 x 16 B), the row fragments go through a ring of four
// micro-steps (fAq), read three micro-steps ahead (kPF).
#define KC_RD_A(SLOT, I, RPAR, KK)                                                                    \
    do {                                                                                              \
        const char* r__ = smem + (4 * (RPAR) + ((I) ? 3 : 0)) * kHalfBytes;                           \
        _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                              \
            fAq[SLOT][rb] = __builtin_bit_cast(bf16x8, *(const uint4*)(r__ + (offA[rb] ^ ((KK) * 32)))); \
    } while (0)
#define KC_RD_B(RPAR, KK)                                                                             \
    do {                                                                                              \
        const char* r__ = smem + (4 * (RPAR) + 1) * kHalfBytes;                                       \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                 \
            fBk[KK][j] = __builtin_bit_cast(bf16x8, *(const uint4*)(r__ + j * kHalfBytes + (offB ^ ((KK) * 32)))); \
    } while (0)
// reads for micro-step M2 (0..9; 8, 9 = micro-steps 0, 1 of the next K-step, other ring parity)
#define KC_PREFETCH(M2)                                                                               \
    do {                                                                                              \
        if constexpr ((ABL & 1) == 0) {  /* (bit 0: timing build without fragment reads) */            \
            constexpr int m2__ = (M2) & 7;                                                            \
            const int rp__ = (M2) >= 8 ? (par ^ 1) : par;                                             \
            if (m2__ < 4) KC_RD_B(rp__, m2__ & 3);                                                    \
            KC_RD_A(m2__ & 3, m2__ >> 2, rp__, m2__ & 3);                                             \
        }                                                                                             \
    } while (0)
#define KC_MM(M, ZERO)                                                                                \
    do {                                                                                              \
        _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                              \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                             \
                acc[(M) >> 2][rb][j] = screen_mfma<I8>(fAq[(M) & 3][rb], fBk[(M) & 3][j], (ZERO) ? zero16 : acc[(M) >> 2][rb][j]); \
    } while (0)
// test block (I, RB, J) of the tile whose first row is ROW0, records slot RSLOT
#define KC_TEST1(I, RB, J, ROW0, RSLOT)                                                               \
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
            if constexpr ((ABL & 4096) != 0)                                                          \
                screen_test_block_cold<I8>(a.status, acc[I][RB][J], q__, rbase__, row_end, th[J], blk__, que, que_n); \
            else                                                                                      \
                screen_test_block<I8>(a.status, acc[I][RB][J], q__, rbase__, row_end, th[J], blk__, que, que_n); \
        }                                                                                             \
    } while (0)
// one micro-step: [reads for M + 2] [4 MFMAs] [DMA pieces P0, P0 + 1 (NP of them)] [one block test]
#define KC_MICRO(M, ZERO, SPAR, P0, NP, RECPOS)                                                       \
    do {                                                                                              \
        KC_PREFETCH((M) + kPF);                                                                       \
        KC_PIN();                                                                                     \
        KC_MM(M, ZERO);                                                                               \
        KC_PIN();                                                                                     \
        _Pragma("unroll") for (int p__ = 0; p__ < (NP); ++p__) KC_PIECE(SPAR, (P0) + p__, RECPOS);    \
        KC_PIN();                                                                                     \
    } while (0)

    // pieces per micro-step 5, 6, 7 (behind the hand-over) | 0, 1, 2, 3, 4 (next K-step); bits 7, 8: the schedule id
    constexpr int kSched = (ABL >> 7) & 3;
    constexpr int kN5 = kc_sched(kSched, 0), kN6 = kc_sched(kSched, 1), kN7 = kc_sched(kSched, 2), kN0 = kc_sched(kSched, 3),
                  kN1 = kc_sched(kSched, 4), kN2 = kc_sched(kSched, 5), kN3 = kc_sched(kSched, 6), kN4 = kc_sched(kSched, 7);
    constexpr int kNLate = kN5 + kN6 + kN7;
    static_assert(kNLate + kN0 + kN1 + kN2 + kN3 + kN4 == 9, "nine pieces per K-step");
    constexpr int kPF = (ABL & 2048) ? 2 : 3;  // fragment reads run this many micro-steps ahead (3: -1.7 % against 2, bit 11)
    bf16x8 fAq[4][2], fBk[4][2];
    if constexpr ((ABL & 1) != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" : "=v"(fAq[i][j]), "=v"(fBk[i][j]));
    }
    // ---- prologue: K-step 0 completely into parity 0, the first four pieces of K-step 1 into parity 1; K-step 0 landed
    // and visible; fragments of micro-steps 0 and 1
#pragma unroll
    for (int p = 0; p < 9; ++p) KC_PIECE(0, p, 0);
    KC_ADVANCE();
#pragma unroll
    for (int p = 0; p < kNLate; ++p) KC_PIECE(1, p, 1);
    if constexpr (kNLate == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (kNLate == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // K-step 0 (+ its records) has landed (this wave's pieces)
    MI355_BARRIER();
    int par = 0, t = 0;
    KC_PREFETCH(0);
    KC_PREFETCH(1);
    if constexpr (kPF == 3) KC_PREFETCH(2);

    const int row_end = (int)a.row_end;
    int row0_cur = (a.ct0 + ctl) * kT2, row0_prev = row0_cur;  // rows < 2^31 (checked by the host)
    bool have_prev = false;  // a finished tile's row half 1 is waiting for its tests
    for (;;) {
        const bool first = t == 0, last = t + 1 == T;
        if (first && que_n > kWaveQueueCap / 2) {  // wave-uniform, rare: this wave stalls on vector memory once
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_queue_flush(a, que, min(que_n, kWaveQueueCap));
            que_n = 0;
        }
        // K-step g = gpos, ring parity par.  Staging: pieces 4..8 of K-step g+1 (into par^1) in micro-steps 0..2, pieces 0..3
        // of K-step g+2 (into par) behind the hand-over in micro-steps 5, 6: at most two LDS-DMA instructions per micro-step
        // (all nine at once -- 8 waves x 9 through the one texture-address path of the CU -- held the matrix pipe up).
        const bool tp = first && have_prev;  // test the previous tile's row half 1 under this K-step's row half 0
        if (first) {  // (a tile's first MFMA per block starts from C = 0 -- an inline constant -- instead of zeroing registers)
            asm volatile("; first K-step of a tile");
            KC_MICRO(0, true, par ^ 1, kNLate, kN0, gpos + 1);
        } else {
            KC_MICRO(0, false, par ^ 1, kNLate, kN0, gpos + 1);
        }
        if (tp) KC_TEST1(1, 0, 0, row0_prev, gpos - 1);
        KC_MICRO(1, false, par ^ 1, kNLate + kN0, kN1, gpos + 1);
        if (tp) KC_TEST1(1, 0, 1, row0_prev, gpos - 1);
        KC_MICRO(2, false, par ^ 1, kNLate + kN0 + kN1, kN2, gpos + 1);
        if (tp) KC_TEST1(1, 1, 0, row0_prev, gpos - 1);
        KC_MICRO(3, false, par ^ 1, kNLate + kN0 + kN1 + kN2, kN3, gpos + 1);
        if (tp) KC_TEST1(1, 1, 1, row0_prev, gpos - 1);
        if (first) {
            asm volatile("; first K-step of a tile, row half 1");
            KC_MICRO(4, true, par ^ 1, kNLate + kN0 + kN1 + kN2 + kN3, kN4, gpos + 1);
        } else {
            KC_MICRO(4, false, par ^ 1, kNLate + kN0 + kN1 + kN2 + kN3, kN4, gpos + 1);
        }
        if (last) KC_TEST1(0, 0, 0, row0_cur, gpos);
#define KC_HANDOVER()                                                                                 \
    do {                                                                                              \
        if constexpr ((ABL & 32) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  /* this wave's pieces of K-step g+1 have landed */ \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  /* ... and its last fragments of K-step g are in registers */ \
        if constexpr ((ABL & 8) == 0) MI355_BARRIER();                                                \
        KC_PIN();                                                                                     \
    } while (0)
        // ---- hand-over of the ring: every read of K-step g has been issued (the last ones kPF micro-steps before the end)
        KC_ADVANCE();
        if constexpr (kPF == 3) KC_HANDOVER();
        KC_MICRO(5, false, par, 0, kPF == 3 ? kN5 : 0, gpos + 2);
        if (last) KC_TEST1(0, 0, 1, row0_cur, gpos);
        if constexpr (kPF == 2) KC_HANDOVER();
        KC_MICRO(6, false, par, kPF == 3 ? kN5 : 0, kPF == 3 ? kN6 : 2, gpos + 2);
        if (last) KC_TEST1(0, 1, 0, row0_cur, gpos);
        KC_MICRO(7, false, par, kPF == 3 ? kN5 + kN6 : 2, kPF == 3 ? kN7 : 2, gpos + 2);
#undef KC_HANDOVER
        if (last) KC_TEST1(0, 1, 1, row0_cur, gpos);
        par ^= 1;
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
    KC_TEST1(1, 0, 0, row0_prev, gpos - 1);
    KC_TEST1(1, 0, 1, row0_prev, gpos - 1);
    KC_TEST1(1, 1, 0, row0_prev, gpos - 1);
    KC_TEST1(1, 1, 1, row0_prev, gpos - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy prefetches must land before the LDS is freed
    wave_queue_flush(a, que, min(que_n, kWaveQueueCap));

#undef KC_RD_A
#undef KC_RD_B
#undef KC_PREFETCH
#undef KC_MM
#undef KC_TEST1
#undef KC_MICRO
#undef KC_PIN
#undef KC_ADVANCE
#undef KC_PIECE
#undef KC_REC_STAGE
}

}  // namespace mi355
