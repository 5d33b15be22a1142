// k_maxsim_wgp.h -- k_maxsim16_wg over the PACKED bf16 token copy: the documents' tokens back to back, the stream cut into
// 32-token blocks wherever they fall, so a block may hold the end of one document and the start of the next (or many short ones).
//
// Why it was built, and what it measured (round 4).  The padded copy rounds every document up to whole 32-token blocks (the
// tail repeats the last token, so no mask is ever needed): a store of 32..180-token passages is 13 % padding (3.82 blocks per
// document where 3.31 would do), 1 030-token pages 2.5 %.  Every padded row is matrix-pipe time, so the packed stream should
// have been up to 13 % faster.  It is NOT: interleaved on one box, pages 9.0 ms per 16-query screen against 8.7 ms over the
// padded copy, text 14.2 against 14.0 -- 13 % fewer blocks, each 17 % slower.  The screen is not bound by the count of MFMAs
// but by what a wave does between its MFMA bursts (fold, bookkeeping, fragment reads: k_maxsim_wg.h's software-pipelined form
// is the answer to THAT, and bought 5 %), and a boundary block adds to exactly that part.  The form stays as an option
// (`maxsim_packed`, default off), bit-identical in its results, for stores whose padding is far worse than the survey's shapes.
// The MFMAs run over real tokens only; the block maximum of a boundary block is taken per row range:
//   * a block whose 32 rows belong to the current document (the usual case): the same 16-accumulator maximum as before;
//   * a boundary block: for every document that has rows in it, the maximum over ITS rows, the mask applied by one more
//     MFMA per range and column block (mwp_rows_max below).
// A workgroup's range of documents is cut by TOKENS; its first and last block may be shared with the neighbour workgroups
// (each takes the rows of its own documents: 2 x 256 blocks multiplied twice per launch).
// Everything else -- query fragments in registers, the LDS-DMA ring of 4 stages x 4 blocks, counted waits, parked and staggered
// per-document sums -- is k_maxsim_wg.h's; the one addition: more than BPS documents may end between two ring barriers now
// (short documents), the parking lot then takes a barrier of its own.
#pragma once
#include "k_maxsim_wg.h"

namespace mi355 {

// maximum over the token rows [r0, re) of a block, per query column, of one lane's 16 accumulators (r0, re wave-uniform).
// The mask rides the matrix pipe: one more MFMA adds u (x) 1 to the block's products -- u = 0 for the rows of the range,
// -2^100 for the others (k = 0 of the A fragment: lane = row, lower half; the B fragment holds 1.0 at k = 0 of every column) --
// so the rows outside drop out of the plain 16-way maximum.  acc + 0 is exact: the rows inside keep their values.  A masked
// maximum on the VALU (compare-selects per accumulator) cost a boundary block 100+ VALU operations per column block; this
// costs one of the 9 MFMAs the block then has, and the 8 max3 it had anyway.
__device__ __forceinline__ float mwp_rows_max(const f32x16& acc, int r0, int re, int lane) {
    typedef unsigned mwp_u32x4 __attribute__((ext_vector_type(4)));
    const int row = lane & 31;
    const bool lower = lane < 32;
    const mwp_u32x4 u = {lower && (row < r0 || row >= re) ? 0xF180u : 0u, 0u, 0u, 0u};  // bf16 -2^100 in element 0
    const mwp_u32x4 one = {lower ? 0x3F80u : 0u, 0u, 0u, 0u};                              // bf16 1.0 in element 0
    const f32x16 t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ms_bf16x8, u), __builtin_bit_cast(ms_bf16x8, one), acc, 0, 0, 0);
    float m = t[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fmaxf(m, t[r]);
    return m;
}

constexpr int mwp_park_floats = 512 + 4;  // a parked document: 16 column blocks x 32 maxima + its id
__host__ __device__ constexpr int mwp_lds(int bps) { return mw_stages(bps) * mw_stage_bytes(bps) + mw_park(bps) * mwp_park_floats * (int)sizeof(float); }
static_assert(mwp_lds(4) <= 160 * 1024 && mwp_lds(2) <= 160 * 1024, "LDS per workgroup");

template <int NCB, int BPS>
__global__ __launch_bounds__(512, 2) void k_maxsim16_wgp(Ms16Args a, int64_t n_tok) {
    static_assert(BPS == 2 || BPS == 4, "blocks per ring stage");
    static_assert(NCB >= 9 && NCB <= 16, "this form serves 9..16 column blocks");
    constexpr int kMwStages = mw_stages(BPS), kMwStageBytes = mw_stage_bytes(BPS), kMwColmaxOff = kMwStages * kMwStageBytes;
    constexpr int kPark = mw_park(BPS), kParkFloats = mwp_park_floats;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb0 = wave, cb1 = wave + 8;
    const bool two = cb1 < NCB;
    float* const colmax = (float*)(smem + kMwColmaxOff);

    // ---- this wave's query fragments: registers for the whole launch ("used" before any LDS-DMA flies: k_maxsim_wg.h)
    ms_bf16x8 qf0[8], qf1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        qf0[i] = __builtin_bit_cast(ms_bf16x8, a.qfrag[(cb0 * 8 + i) * 64 + lane]);
        qf1[i] = two ? __builtin_bit_cast(ms_bf16x8, a.qfrag[(cb1 * 8 + i) * 64 + lane]) : qf0[i];
    }
    typedef int mw_i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        asm volatile("" ::"v"(__builtin_bit_cast(mw_i32x4, qf0[i])));
        asm volatile("" ::"v"(__builtin_bit_cast(mw_i32x4, qf1[i])));
    }

    // token offsets through the scalar cache (nothing in this launch writes them; indices made wave-uniform by hand)
    typedef const __attribute__((address_space(4))) int64_t c_i64;
    c_i64* const toff = (c_i64*)(const int64_t*)a.tok_off;
    auto uni = [](int64_t x) -> int64_t {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uint64_t)x);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uint64_t)x >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    auto first_doc_at = [&](int64_t t) -> int64_t {  // first doc whose first token is >= t
        int64_t lo = 0, hi = a.n_docs;               // tok_off[n_docs] = n_tok >= t
        while (lo < hi) {
            const int64_t mid = uni((lo + hi) >> 1);
            if (toff[mid] >= t) hi = mid;
            else lo = mid + 1;
        }
        return uni(lo);
    };
    const int64_t g = blockIdx.x, G = gridDim.x;
    const int64_t d0 = first_doc_at(n_tok * g / G);
    const int64_t d1 = g + 1 == G ? a.n_docs : first_doc_at(n_tok * (g + 1) / G);
    if (d0 >= d1) return;  // (workgroup-uniform)
    const int64_t t_begin = toff[d0], t_end = toff[d1];

    const float kNaN = __uint_as_float(0x7FC00000u);
    const int qa = wave, qb = wave + 8;  // queries whose sums this wave writes
    auto write_doc = [&](int64_t doc, float va, float vb) __attribute__((always_inline)) {
        if (lane == 63) {  // (the lane the DPP sums end in)
            if (qa < a.nq_launch) a.dist[(int64_t)qa * a.n_docs + doc] = va;
            if (qb < a.nq_launch) a.dist[(int64_t)qb * a.n_docs + doc] = vb;
        }
    };

    // ---- document cursor (TOKEN positions): cur = the doc the stream is in, end_cur = its end, end_next = the next doc's
    int64_t cur = d0;
    int64_t end_cur = toff[cur + 1];
    int64_t end_next = cur + 2 <= a.n_docs ? toff[cur + 2] : end_cur;
    auto advance_doc = [&]() __attribute__((always_inline)) {  // to the next doc with tokens; empty docs on the way get NaN (the select skips them)
        for (;;) {
            cur = uni(cur + 1);
            if (cur >= d1) return;
            const int64_t prev_end = end_cur;
            end_cur = end_next;
            end_next = cur + 2 <= a.n_docs ? toff[cur + 2] : end_cur;
            if (end_cur != prev_end) return;
            write_doc(cur, kNaN, kNaN);
        }
    };
    while (cur < d1 && end_cur == t_begin) {  // leading empty docs
        write_doc(cur, kNaN, kNaN);
        cur = uni(cur + 1);
        if (cur >= d1) break;
        end_cur = end_next;
        end_next = cur + 2 <= a.n_docs ? toff[cur + 2] : end_cur;
    }
    if (t_end == t_begin) return;
    const int64_t b_begin = t_begin >> 5, b_end = (t_end + 31) >> 5;  // blocks of the packed stream this range touches
    const int64_t n_my = b_end - b_begin, b_last = b_end - 1;
    const int64_t n_stages = (n_my + BPS - 1) / BPS;
    int64_t pos = b_begin;  // next block of the stream to be multiplied
    // 32-bit view of the cursor for the per-block test: tokens of the current document from the current block's first row on
    // (saturated), and the rows of the FIRST block that belong to the workgroup before this one
    auto left_from = [&](int64_t base) __attribute__((always_inline)) -> int {
        const int64_t l = end_cur - base;
        return (int)uni(l < (int64_t)1 << 30 ? l : (int64_t)1 << 30);
    };
    int left = left_from(b_begin << 5);
    int r_first = (int)uni(t_begin - (b_begin << 5));

    float run0 = -__builtin_inff(), run1 = -__builtin_inff();
    // Parked and staggered per-document sums (k_maxsim_wg.h), kept as three COUNTERS: documents finished (n_fin), published by
    // a barrier (n_pub), summed by this wave (n_done).  Finished document n parks its column maxima -- and its id -- in buffer
    // n mod 2 BPS.  Between two ring barriers lie BPS blocks, and a block of the packed stream may END any number of
    // documents: when BPS documents are parked and unpublished, a barrier of its own publishes them (so a buffer is never
    // rewritten before every wave has summed it: at most BPS documents per barrier interval, 2 BPS buffers).
    int n_fin = 0, n_pub = 0, n_done = 0;
    auto flush_to = [&](int upto) __attribute__((always_inline)) {
#pragma unroll 1
        for (; n_done < upto; ++n_done) {
            const float* cm = colmax + (n_done & (kPark - 1)) * kParkFloats;
            const int64_t doc = *(const int64_t*)(cm + 512);
            float out[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int qi = wave + 8 * s;
                float part = 0.0f;
                if (qi < a.nq_launch) {  // wave-uniform
                    const int c0 = a.q_col0[qi], len = a.q_len[qi];
                    if (lane < len) part += cm[c0 + lane];
                    if (lane + 64 < len) part += cm[c0 + 64 + lane];
                    part = mw_wave_sum_lane63(part);
                }
                out[s] = -part;
            }
            write_doc(doc, out[0], out[1]);
        }
    };
    auto finish_doc = [&]() __attribute__((always_inline)) {
        if (n_fin - n_pub == BPS) {  // (workgroup-uniform) the parking lot is full
#pragma unroll 1
            for (int rep = 0; rep < 2; ++rep) {
                flush_to(n_pub);  // rep 0: what the last barrier published and this wave has not summed yet; rep 1: the lot
                if (rep == 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    MI355_BARRIER();
                    n_pub = n_fin;
                }
            }
        }
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(run0), __float_as_uint(run1), false, false);
        const float r = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        float* cm = colmax + (n_fin & (kPark - 1)) * kParkFloats;
        if (lane < 32) cm[cb0 * 32 + lane] = r;
        else if (two) cm[cb1 * 32 + lane - 32] = r;
        if (tid == 0) *(int64_t*)(cm + 512) = cur;
        ++n_fin;
        run0 = run1 = -__builtin_inff();
    };

    // ---- staging: wave w moves k-group fragment w of every block (1 KiB per instruction); past the range: the last block again
    const char* const tokbase = (const char*)a.tok16;
    const unsigned voff = (unsigned)lane * 16u;
    int64_t s_issue = 0;
    int slot_issue = 0;
    auto issue_stage = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < BPS; ++u) {
            int64_t bi = b_begin + BPS * s_issue + u;
            if (bi > b_last) bi = b_last;
            glds16_saddr(tokbase + ((bi * 8 + wave) << 10), voff,
                         lds_addr(smem + slot_issue * kMwStageBytes + u * 8192 + wave * 1024));
        }
        ++s_issue;
        if (++slot_issue == kMwStages) slot_issue = 0;
    };
#pragma unroll 1
    for (int s = 0; s < kMwStages; ++s) issue_stage();

    ms_bf16x8 tfA[8], tfB[8];
    auto read_block = [&](ms_bf16x8(&tf)[8], int slot, int u) __attribute__((always_inline)) {
        const char* p = smem + slot * kMwStageBytes + u * 8192 + lane * 16;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) tf[kk] = __builtin_bit_cast(ms_bf16x8, *(const uint4*)(p + kk * 1024));
    };
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto do_block = [&](const ms_bf16x8(&tf)[8]) __attribute__((always_inline)) {  // multiply block `pos`, fold it into the documents with rows in it
        f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[0], qf0[0], zero, 0, 0, 0);
#pragma unroll
        for (int i = 1; i < 8; ++i) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[i], qf0[i], acc0, 0, 0, 0);
        f32x16 acc1;  // (only defined and only used when this wave has a second column block)
        if (two) {  // wave-uniform
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[0], qf1[0], zero, 0, 0, 0);
#pragma unroll
            for (int i = 1; i < 8; ++i) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[i], qf1[i], acc1, 0, 0, 0);
        }
        const int64_t base = pos << 5;
        ++pos;
        if (left >= 32 && r_first == 0) {  // (scalar) all 32 rows belong to the current document
            if (two) {
                float m1 = acc1[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m1 = fmaxf(m1, acc1[r]);
                run1 = fmaxf(run1, m1);
            }
            float m0 = acc0[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m0 = fmaxf(m0, acc0[r]);
            run0 = fmaxf(run0, m0);
            if (left == 32) {
                finish_doc();
                advance_doc();
                left = left_from(base);  // (of the next document, still from THIS block's first row: - 32 below)
            }
        } else {
            // a boundary block: every document of this range with rows in it takes the maximum over ITS rows.  (Kept in the
            // else-branch on purpose: as one loop behind the whole-block case the compiler hoisted this path's work in front of
            // the loop, i.e. into EVERY block.)
            int r0 = r_first;
            r_first = 0;
#pragma unroll 1
            for (;;) {  // (workgroup-uniform trip count)
                const int re = left < 32 ? left : 32;  // left > r0: the current document has tokens and starts at or before row r0
                run0 = fmaxf(run0, mwp_rows_max(acc0, r0, re, lane));
                if (two) run1 = fmaxf(run1, mwp_rows_max(acc1, r0, re, lane));
                if (left > 32) break;  // the document goes on in the next block
                finish_doc();
                advance_doc();
                left = left_from(base);
                r0 = re;
                if (cur >= d1 || r0 >= 32) break;  // (the rest of the block is the next workgroup's / the block is used up)
            }
        }
        left -= 32;  // from the next block's first row on
    };

    // stage 0 has landed (this wave's pieces: the BPS (stages - 1) youngest may still fly) and is visible
    static_assert(BPS * (kMwStages - 1) == 12, "the counted wait below");
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    MI355_BARRIER();
    int slot = 0;
    read_block(tfA, 0, 0);
    read_block(tfB, 0, 1);
    for (int64_t s = 0; s < n_stages; ++s) {
        const int slot_n = slot + 1 == kMwStages ? 0 : slot + 1;
        auto pair = [&](int jp) __attribute__((always_inline)) {  // blocks 2 jp (fragments A) and 2 jp + 1 (fragments B) of the stage
            const bool last = jp == BPS / 2 - 1;
            if (jp == 0 || pos < b_end) do_block(tfA);  // workgroup-uniform (the range may end inside a stage)
            if (last) {
                // ---- hand-over: this wave's pieces of stage s + 1 have landed, its last fragments of stage s are in registers
                if constexpr (BPS == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                MI355_BARRIER();
                issue_stage();
                n_pub = n_fin;  // the documents that ended since the previous barrier
                if (wave < 4) flush_to(n_pub);
            }
            read_block(tfA, last ? slot_n : slot, last ? 0 : 2 * jp + 2);
            if (pos < b_end) do_block(tfB);
            if (last && wave >= 4) flush_to(n_pub);  // (one block behind the other wave of this SIMD)
            read_block(tfB, last ? slot_n : slot, last ? 1 : 2 * jp + 3);
        };
#pragma unroll
        for (int jp = 0; jp < BPS / 2; ++jp) pair(jp);
        slot = slot_n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy stages must land before the LDS is freed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MI355_BARRIER();
    flush_to(n_fin);  // the range's last document(s)
}

}  // namespace mi355
