// k_maxsim_wg8.h -- k_maxsim16_wg (software-pipelined, 4 blocks per ring stage) over the GRANULE-PACKED bf16 token copy, for passes
// whose queries are each exactly one column block (`aligned`: ColBERT's 32-vector queries).
//
// Why (round 6, third session).  The padded copy rounds every document up to whole 32-token blocks: a store of 32..180-token
// passages is 13 % padding, and every padded row is matrix-pipe time at the socket power cap.  Round 4's packed copy
// (k_maxsim_wgp.h, removed in round 6) packed to the TOKEN and paid for it in every boundary block -- a mask MFMA per document
// range and column block plus a fold outside the MFMA shadow: 13 % fewer blocks, each 17 % slower.  This form packs to the
// GRANULE of 8 tokens instead (a document's tail granule repeats its last token: no mask is ever needed), which is exactly the
// accumulator layout of v_mfma_f32_32x32x16_bf16: lane l holds query column l % 32 and, in register 4 j + i, token row
// 8 j + 4 (l / 32) + i -- REGISTER QUAD j of both wave halves IS granule j of the block.  A block's fold produces four granule
// maxima per column block (2 VALU operations per quad: 16 per block and wave, one in the shadow of every MFMA, the count the
// padded form's 16-way maximum has); a document boundary inside a block then is a choice of which quads go to which running
// maximum, made by scalar branches behind the fold.  No extra MFMA, no lane masks, 3.6 % padding instead of 13 %.
// The values are the padded copy's (same bf16 roundings, copied fragment by fragment), maxima are order-free and the
// per-document sums are k_maxsim16_wg's own DPP sequence (aligned form): the screen distances are BIT-IDENTICAL to k_maxsim16_wg's on the padded
// copy (tests/test_gpu_maxsim.py::test_pack8_*).
// A workgroup's documents are a contiguous range holding ~1/gridDim of the granules; its first and last block may be shared with
// the neighbours (each folds the granules of its own documents only: at most 2 x 256 blocks multiplied twice per launch).
#pragma once
#include "k_maxsim_wg.h"

namespace mi355 {

constexpr int kGranRows = 8;                       // token rows per granule
constexpr int kGranPerBlk = kMsBlkRows / kGranRows;  // 4

__global__ __launch_bounds__(512, 2) void k_maxsim16_wg8(Ms16Args a, Ms16Pack pk, int ncb) {
    constexpr int BPS = 4;
    constexpr int kMwStages = mw_stages(BPS), kMwStageBytes = mw_stage_bytes(BPS);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb0 = wave, cb1 = wave + 8;
    const bool two = cb1 < ncb;

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

    // granule offsets through the scalar cache (nothing in this launch writes them; indices made wave-uniform by hand)
    typedef const __attribute__((address_space(4))) int64_t c_i64;
    c_i64* const goff = (c_i64*)(const int64_t*)pk.goff;
    auto uni = [](int64_t x) -> int64_t {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uint64_t)x);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uint64_t)x >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    auto first_doc_at = [&](int64_t t) -> int64_t {  // first doc whose first granule is >= t (goff[n_docs] = n_gran >= t)
        int64_t lo = 0, hi = a.n_docs;
        while (lo < hi) {
            const int64_t mid = uni((lo + hi) >> 1);
            if (goff[mid] >= t) hi = mid;
            else lo = mid + 1;
        }
        return uni(lo);
    };
    const int64_t g = blockIdx.x, G = gridDim.x;
    const int64_t d0 = first_doc_at(pk.n_gran * g / G);
    const int64_t d1 = g + 1 == G ? a.n_docs : first_doc_at(pk.n_gran * (g + 1) / G);
    if (d0 >= d1) return;  // (workgroup-uniform)
    const int64_t gs = goff[d0], ge = goff[d1];  // this workgroup's granules
    const int64_t b_begin = gs / kGranPerBlk, b_endx = (ge + kGranPerBlk - 1) / kGranPerBlk;
    const int n_my = ge > gs ? (int)(b_endx - b_begin) : 0;  // this workgroup's blocks (32-bit from here on: scalar compares)
    const int rb_last = n_my > 0 ? n_my - 1 : 0;
    const int n_stages = (n_my + BPS - 1) / BPS;
    const int n_docs_i = (int)a.n_docs;

    const float kNaN = __uint_as_float(0x7FC00000u);
    const int qa = wave, qb = wave + 8;  // the queries whose sums this wave writes (= its column blocks)
    const int len_a = qa < a.nq_launch ? a.q_len[qa] : 0, len_b = qb < a.nq_launch ? a.q_len[qb] : 0;
    int len_mine = lane < 32 ? len_a : len_b;
    asm volatile("" : "+v"(len_mine));
    auto write_doc = [&](int doc, float va, float vb) {
        if (lane == 63) {
            if (qa < a.nq_launch) a.dist[(int64_t)qa * a.n_docs + doc] = va;
            if (qb < a.nq_launch) a.dist[(int64_t)qb * a.n_docs + doc] = vb;
        }
    };

    // ---- document cursor.  Everything the block loop compares is a 32-bit SCALAR: documents by index (< 2^31: the store refuses
    // more), granules relative to this workgroup's first block (64-bit compares are VALU compares + a branch on vcc: ~10 of them
    // per block were a third of what a boundary block cost more than a plain one)
    const int64_t base = b_begin * kGranPerBlk;
    const int g0 = (int)(gs - base);  // 0..3: granules of the first block in front of this workgroup's first document
    const int d1i = (int)d1;
    auto rel = [&](int64_t gidx) -> int { return (int)(gidx - base); };
    // cur = the doc the stream is in, end_cur = its end; nxt_raw = goff[cur + 2], the NEXT doc's end, read one document ahead and
    // left untouched until a block later (after_block): a scalar load consumed where it is issued is an s_waitcnt lgkmcnt(0) --
    // the load's latency AND the block's fragment reads -- in the one place where both waves of a SIMD stand still
    int cur = (int)d0;
    int end_cur = rel(goff[cur + 1]);
    int64_t nxt_raw = cur + 2 <= n_docs_i ? goff[cur + 2] : base + end_cur;
    auto advance_doc = [&]() {  // to the next doc; empty docs on the way get NaN (the select skips them)
        for (;;) {
            cur = cur + 1;
            if (cur >= d1i) return;
            const int prev_end = end_cur;
            end_cur = rel(nxt_raw);
            nxt_raw = cur + 2 <= n_docs_i ? goff[cur + 2] : base + end_cur;
            if (end_cur != prev_end) return;
            write_doc(cur, kNaN, kNaN);
        }
    };
    while (cur < d1i && end_cur == g0) {  // leading empty docs
        write_doc(cur, kNaN, kNaN);
        cur = cur + 1;
        if (cur >= d1i) break;
        end_cur = rel(nxt_raw);
        nxt_raw = cur + 2 <= n_docs_i ? goff[cur + 2] : base + end_cur;
    }
    if (n_my == 0) return;

    // ---- per-document epilogue, DEFERRED: a finished document's running maxima are set aside (two registers and its index) and
    // summed in the shadow of the NEXT block's MFMAs -- the two waves of a SIMD walk the stream in lockstep, so an epilogue at the
    // document's end (a permlane swap, six dependent DPP steps with their wait states, a store: ~300 cycles) is time in which the
    // SIMD issues no MFMA at all.  Only the store stays behind the burst (its lane mask is a branch).
    float run0 = -__builtin_inff(), run1 = -__builtin_inff();
    float pend0 = run0, pend1 = run1;
    int pend_doc = 0;
    bool pend_valid = false;
    auto epilogue_value = [&](float p0, float p1) -> float {
        // lanes 0..31: column block cb0 = query `wave`, lanes 32..63: cb1 = query `wave + 8` (k_maxsim_wg.h, aligned form):
        // four row_shr steps and one row_bcast:15 leave the two sums in lanes 31 and 63
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
        const float r = fmaxf(fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])), -__builtin_inff());  // (one v_max3: no canonicalising v_max_f32 x, x, x)
        float part = (lane & 31) < len_mine ? r : 0.0f;
        part += mw_dpp<0x111, 0xF, true>(part);
        part += mw_dpp<0x112, 0xF, true>(part);
        part += mw_dpp<0x114, 0xF, true>(part);
        part += mw_dpp<0x118, 0xF, true>(part);
        part += mw_dpp<0x142, 0xA, false>(part);
        return part;
    };
    auto store_pending = [&](float part) {
        if ((lane & 31) == 31 && len_mine > 0) a.dist[(int64_t)(wave + 8 * (lane >> 5)) * a.n_docs + pend_doc] = -part;
        pend_valid = false;
    };
    auto retire_doc = [&]() {  // the document `cur` is complete
        if (pend_valid) store_pending(epilogue_value(pend0, pend1));  // (a second document ending inside one block: short documents)
        pend0 = run0;
        pend1 = run1;
        pend_doc = cur;
        pend_valid = true;
        run0 = run1 = -__builtin_inff();
    };

    // ---- staging: wave w moves k-group fragment w of every block (1 KiB per instruction); past the range: the last block again
    const char* const tokbase = (const char*)pk.tok16p + ((b_begin * 8) << 10);  // this workgroup's first block
    const unsigned voff = (unsigned)lane * 16u;
    int s_issue = 0;
    int slot_issue = 0;
    auto issue_stage = [&]() {
#pragma unroll
        for (int u = 0; u < BPS; ++u) {
            int rb = BPS * s_issue + u;
            if (rb > rb_last) rb = rb_last;
            glds16_saddr(tokbase + (((int64_t)rb * 8 + wave) << 10), voff,
                         lds_addr(smem + slot_issue * kMwStageBytes + u * 8192 + wave * 1024));
        }
        ++s_issue;
        if (++slot_issue == kMwStages) slot_issue = 0;
    };
#pragma unroll 1
    for (int s = 0; s < kMwStages; ++s) issue_stage();

    ms_bf16x8 tfA[8], tfB[8];
    auto read_block = [&](ms_bf16x8(&tf)[8], int slot, int u) {
        const char* p = smem + slot * kMwStageBytes + u * 8192 + lane * 16;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) tf[kk] = __builtin_bit_cast(ms_bf16x8, *(const uint4*)(p + kk * 1024));
    };
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // first granule (relative) of the next block to be folded.  The pipelined loop folds one block BEHIND its MFMAs, so its first
    // fold is of a block that does not exist: it is given the four granules in front of b_begin -- all before g0, all skipped.
    int pos = -kGranPerBlk;
    // one block's granule maxima meet the document cursor (everything here is workgroup-uniform control flow)
    auto after_block = [&](const float (&m0)[kGranPerBlk], const float (&m1)[kGranPerBlk], auto two_c) __attribute__((always_inline)) {
        constexpr bool TWO = decltype(two_c)::value;
        const int gb = pos;
        pos += kGranPerBlk;
        const int left = end_cur - gb;  // granules of the current document from this block's first one on
        // (the first branch behind the MFMAs has the granule maxima live on BOTH sides: with a side that does not need them the
        // compiler sinks the fold out of the MFMAs' basic block, away from their shadow)
        if (cur < d1i && gb >= g0 && left >= kGranPerBlk) {  // the usual case: the whole block lies inside the current document
            run0 = fmaxf(fmaxf(fmaxf(fmaxf(run0, m0[0]), m0[1]), m0[2]), m0[3]);  // (two v_max3)
            if constexpr (TWO) run1 = fmaxf(fmaxf(fmaxf(fmaxf(run1, m1[0]), m1[1]), m1[2]), m1[3]);
            if (left == kGranPerBlk) {
                retire_doc();
                advance_doc();
            }
            return;
        }
        if (cur + 1 < d1i && gb >= g0 && left > 0 && rel(nxt_raw) - gb > kGranPerBlk) {
            // ONE boundary inside the block, `left` (1..3) granules in: the current document ends there, the next one runs past the
            // block -- what a store of passages (every document longer than a block) has in a quarter of its blocks.  The maxima of
            // the two sides by selects on the scalar `left`; the first operand of each v_max3 rides again where a quad is not its side's.
            const bool l2 = left >= 2, l3 = left >= 3;
            const float lo0 = fmaxf(fmaxf(m0[0], l2 ? m0[1] : m0[0]), l3 ? m0[2] : m0[0]);
            const float hi0 = fmaxf(fmaxf(m0[3], l3 ? m0[3] : m0[2]), l2 ? m0[3] : m0[1]);
            run0 = fmaxf(run0, lo0);
            float lo1 = 0.0f, hi1 = 0.0f;
            if constexpr (TWO) {
                lo1 = fmaxf(fmaxf(m1[0], l2 ? m1[1] : m1[0]), l3 ? m1[2] : m1[0]);
                hi1 = fmaxf(fmaxf(m1[3], l3 ? m1[3] : m1[2]), l2 ? m1[3] : m1[1]);
                run1 = fmaxf(run1, lo1);
            }
            retire_doc();
            advance_doc();
            run0 = hi0;
            if constexpr (TWO) run1 = hi1;
            return;
        }
        // anything else (a shared first or last block, several documents ending in one block): granule by granule
#pragma unroll 1
        for (int j = 0; j < kGranPerBlk; ++j) {
            if (cur >= d1i) break;
            if (gb + j < g0) continue;  // (granules before g0 belong to the previous workgroup's last document)
            const float a0 = j == 0 ? m0[0] : j == 1 ? m0[1] : j == 2 ? m0[2] : m0[3];  // (scalar conditions: v_cndmask, no indexing)
            run0 = fmaxf(run0, a0);
            if constexpr (TWO) {
                const float a1 = j == 0 ? m1[0] : j == 1 ? m1[1] : j == 2 ? m1[2] : m1[3];
                run1 = fmaxf(run1, a1);
            }
            if (gb + j + 1 == end_cur) {
                retire_doc();
                advance_doc();
            }
        }
    };

    static_assert(BPS * (kMwStages - 1) == 12, "the counted wait below");
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    MI355_BARRIER();
    auto stage_loop = [&](auto two_c) __attribute__((always_inline)) {
        constexpr bool TWO = decltype(two_c)::value;
        f32x16 x0, x1, y0, y1;
#pragma unroll
        for (int r = 0; r < 16; ++r) x0[r] = x1[r] = y0[r] = y1[r] = -__builtin_inff();
        auto quads = [&](const f32x16& p, float (&m)[kGranPerBlk]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < kGranPerBlk; ++j) {
                // two v_max3 (the first operand rides twice): a two-operand v_max_f32 of raw MFMA results costs a canonicalising
                // v_max_f32 x, x, x per operand under IEEE mode -- four VALU operations per quad instead of two
                const float t = fmaxf(fmaxf(p[4 * j], p[4 * j + 1]), p[4 * j + 2]);
                m[j] = fmaxf(fmaxf(t, p[4 * j + 3]), p[4 * j]);
            }
        };
        auto slot_body = [&](const ms_bf16x8(&tf)[8], f32x16& n0, f32x16& n1, const f32x16& p0, const f32x16& p1)
                             __attribute__((always_inline)) {
            n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[0], qf0[0], zero, 0, 0, 0);
#pragma unroll
            for (int i = 1; i < 8; ++i) n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[i], qf0[i], n0, 0, 0, 0);
            if constexpr (TWO) {
                n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[0], qf1[0], zero, 0, 0, 0);
#pragma unroll
                for (int i = 1; i < 8; ++i) n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[i], qf1[i], n1, 0, 0, 0);
            }
            float m0[kGranPerBlk], m1[kGranPerBlk];
            quads(p0, m0);
            if constexpr (TWO) quads(p1, m1);
            else {
#pragma unroll
                for (int j = 0; j < kGranPerBlk; ++j) m1[j] = -__builtin_inff();
            }
            const float part = epilogue_value(pend0, pend1);  // (of the document that ended in the block before, if one did)
#pragma unroll
            for (int i = 0; i < (TWO ? 16 : 8); ++i) {  // VALU operations in the shadow of every MFMA
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, TWO ? 2 : 3, 0);
            }
            // (the fold's results are "used" HERE, in the MFMAs' basic block: left to their uses behind after_block's branches the
            // compiler sinks the whole fold out of the MFMAs' shadow; input-only, so the values stay known to it)
#pragma unroll
            for (int j = 0; j < kGranPerBlk; ++j) {
                asm volatile("" ::"v"(m0[j]));
                if constexpr (TWO) asm volatile("" ::"v"(m1[j]));
            }
            asm volatile("" ::"v"(part));
            if (pend_valid) store_pending(part);
            after_block(m0, m1, two_c);
        };
        int slot = 0;
        read_block(tfA, 0, 0);
        read_block(tfB, 0, 1);
        for (int s = 0; s < n_stages; ++s) {
            const int slot_n = slot + 1 == kMwStages ? 0 : slot + 1;
#pragma unroll
            for (int j = 0; j < BPS; ++j) {
                if (j & 1) slot_body(tfB, y0, y1, x0, x1);
                else slot_body(tfA, x0, x1, y0, y1);
                if (j == BPS - 2) {
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    MI355_BARRIER();
                    issue_stage();
                }
                const int jn = j + 2;
                if (j & 1) read_block(tfB, jn < BPS ? slot : slot_n, jn < BPS ? jn : jn - BPS);
                else read_block(tfA, jn < BPS ? slot : slot_n, jn < BPS ? jn : jn - BPS);
            }
            slot = slot_n;
        }
        // the last block's products (BPS is even: they sit in the y set)
        float m0[kGranPerBlk], m1[kGranPerBlk];
        quads(y0, m0);
        if constexpr (TWO) quads(y1, m1);
        else {
#pragma unroll
            for (int j = 0; j < kGranPerBlk; ++j) m1[j] = -__builtin_inff();
        }
        if (pend_valid) store_pending(epilogue_value(pend0, pend1));
        after_block(m0, m1, two_c);
        if (pend_valid) store_pending(epilogue_value(pend0, pend1));
    };
    if (two) stage_loop(std::true_type{});
    else stage_loop(std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy stages must land before the LDS is freed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MI355_BARRIER();
}

}  // namespace mi355
