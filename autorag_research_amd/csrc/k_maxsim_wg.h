// k_maxsim_wg.h -- the MaxSim bf16 screen for 8..16 column blocks of query vectors per pass (up to 16 queries x 32 vectors),
// dims <= 128: ONE workgroup of 8 waves walks a contiguous range of documents as ONE stream of 32-token blocks; the waves
// split the query COLUMNS, not the documents.
//
// Why a second form (round 4).  k_maxsim16_d128 gives every wave its own documents: the wave pulls its token fragments
// HBM -> VGPR and multiplies them with every column block, reading each query fragment from LDS -- one ds_read_b128 per MFMA --
// and pays the per-document epilogue (16 masked butterflies at 16 queries) alone.  At 16 queries per pass the pass is no
// longer bound by the token stream but by the matrix pipe at the socket power cap (bench: 16 queries per pass ran no faster per
// query than 8), and -- the round's A/Bs: profiles/r04_maxsim_ab.txt -- by the time a wave spends BETWEEN its MFMA bursts:
//   * the query fragments of a wave's OWN column blocks (w and w + 8: at most two) stay in REGISTERS for the whole launch;
//   * the token blocks go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction = one k-group fragment
//     of a block, which the fragment-ordered bf16 copy stores contiguously), once per WORKGROUP: a ring of 4 stages of four
//     blocks (BPS = 4; or 7 stages of two, BPS = 2: -3 %), 96 KiB per CU in flight either way, counted vmcnt waits, one
//     barrier per stage; every wave reads a block's 8 fragments from LDS once and uses each for its one or two column
//     blocks: 0.5 .. 1 LDS read per MFMA instead of 1, no token fragment ever crosses a VGPR on its way in;
//   * PIPE (the default): the 16-way maxima of block j are folded between the MFMAs of block j + 1 -- two accumulator sets,
//     the fold in the burst's own basic block, one VALU operation per MFMA by sched_group_barrier (-5 .. 6 %);
//   * the per-document epilogue is shared: the waves' per-column maxima meet in 2 KiB of LDS, wave w then sums the columns of
//     queries w and w + 8 by DPP -- parked behind the ring's NEXT stage barrier and staggered between the two waves of a SIMD
//     for short documents (DEFER), behind a barrier of its own for long ones;
//   * and not at all when every query of the pass IS one column block (`aligned`: ColBERT's 32-vector queries): the wave that
//     holds a query's columns sums them itself (-5 % on a store of 32..180-token documents).
// Column blocks w and w + 8 sit on the same wave, waves w and w + 4 on the same SIMD: 12 column blocks (sixteen 24-vector
// queries) are 3 per SIMD, 16 are 4 per SIMD -- balanced.  What did NOT help: fewer MFMAs (round 4's packed copy without per-document padding: 13 % fewer blocks on text, 2-3 % slower; removed in round 6),
// wave priorities, interleaving the two accumulator chains.
// The per-document sums add the column maxima in another order than k_maxsim16_d128 and the exact kernel do; the screen's
// bound covers any order (search_maxsim_impl: e_acc).  Bit-exactness of the RESULTS is the exact re-score's business, as before.
#pragma once
#include <type_traits>

#include "k_screen256_common.h"

namespace mi355 {

// ring geometry by blocks per stage (BPS): 2 -> 7 stages of 16 KiB (6 in flight), 4 -> 4 stages of 32 KiB (3 in flight): 96 KiB
// of token stream per CU on its way either way; BPS = 4 halves the barriers and hand-overs per block
__host__ __device__ constexpr int mw_stages(int bps) { return bps == 2 ? 7 : 4; }
__host__ __device__ constexpr int mw_stage_bytes(int bps) { return bps * 8192; }  // 32-token blocks of 128 dims bf16, fragment order
__host__ __device__ constexpr int mw_park(int bps) { return 2 * bps; }            // parked documents' buffers (2 KiB each)
__host__ __device__ constexpr int mw_lds(int bps) { return mw_stages(bps) * mw_stage_bytes(bps) + mw_park(bps) * 512 * (int)sizeof(float); }
static_assert(mw_lds(2) <= 160 * 1024 && mw_lds(4) <= 160 * 1024, "LDS per workgroup");

// (mw_dpp / mw_wave_sum_lane63: the DPP sums, defined by the including file next to Ms16Args)

template <int NCB, bool DEFER, int BPS, bool PIPE = false>
__global__ __launch_bounds__(512, 2) void k_maxsim16_wg(Ms16Args a, int64_t n_blocks) {
    static_assert(BPS == 2 || BPS == 4, "blocks per ring stage");
    constexpr int kMwStages = mw_stages(BPS), kMwStageBytes = mw_stage_bytes(BPS), kMwColmaxOff = kMwStages * kMwStageBytes;
    constexpr int kPark = mw_park(BPS);
    static_assert(NCB >= 8 && NCB <= 16, "this form serves 8..16 column blocks");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb0 = wave, cb1 = wave + 8;
    const bool two = cb1 < NCB;
    float* const colmax = (float*)(smem + kMwColmaxOff);

    // ---- this wave's query fragments: registers for the whole launch
    ms_bf16x8 qf0[8], qf1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        qf0[i] = __builtin_bit_cast(ms_bf16x8, a.qfrag[(cb0 * 8 + i) * 64 + lane]);
        qf1[i] = two ? __builtin_bit_cast(ms_bf16x8, a.qfrag[(cb1 * 8 + i) * 64 + lane]) : qf0[i];
    }
    // The fragments are "used" here, before any LDS-DMA is in flight: left to their first use, the compiler's s_waitcnt vmcnt(0)
    // for these loads sits inside the block loop and drains the ring once per block (tests/test_build_pipeline.py pins the loop).
    typedef int mw_i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        asm volatile("" ::"v"(__builtin_bit_cast(mw_i32x4, qf0[i])));
        asm volatile("" ::"v"(__builtin_bit_cast(mw_i32x4, qf1[i])));
    }

    // Block offsets are read through the SCALAR cache (constant address space: nothing in this launch writes them, and the
    // index is made wave-uniform by hand): as vector loads their s_waitcnt vmcnt(0) -- one per document -- drained the
    // LDS-DMA ring, and a text document is two stages long.
    typedef const __attribute__((address_space(4))) int64_t c_i64;
    c_i64* const boff = (c_i64*)(const int64_t*)a.blk_off;
    auto uni = [](int64_t x) -> int64_t {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uint64_t)x);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uint64_t)x >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    // ---- this workgroup's documents: a contiguous range holding ~1/gridDim of the token blocks
    auto first_doc_at = [&](int64_t t) -> int64_t {  // first doc whose first block is >= t (wave-uniform binary search)
        int64_t lo = 0, hi = a.n_docs;               // blk_off[n_docs] = n_blocks >= t
        while (lo < hi) {
            const int64_t mid = uni((lo + hi) >> 1);
            if (boff[mid] >= t) hi = mid;
            else lo = mid + 1;
        }
        return uni(lo);
    };
    const int64_t g = blockIdx.x, G = gridDim.x;
    const int64_t d0 = first_doc_at(n_blocks * g / G);
    const int64_t d1 = g + 1 == G ? a.n_docs : first_doc_at(n_blocks * (g + 1) / G);
    if (d0 >= d1) return;  // (workgroup-uniform)
    const int64_t b_begin = boff[d0], b_end = boff[d1];
    const int64_t n_my = b_end - b_begin;
    const int64_t b_last = n_my > 0 ? b_end - 1 : 0;
    const int64_t n_stages = (n_my + BPS - 1) / BPS;

    const float kNaN = __uint_as_float(0x7FC00000u);
    // queries whose sums this wave writes: wave and wave + 8
    const int qa = wave, qb = wave + 8;
    const bool aligned = a.aligned != 0;  // (wave-uniform) query r = column block r
    const int len_a = qa < a.nq_launch ? a.q_len[qa] : 0, len_b = qb < a.nq_launch ? a.q_len[qb] : 0;  // (scalar loads: wave-uniform indices)
    int len_mine = lane < 32 ? len_a : len_b;
    asm volatile("" : "+v"(len_mine));  // (materialised HERE: left to its first use its wait lands behind the ring's prologue)
    auto write_doc = [&](int64_t doc, float va, float vb) {
        if (lane == 63) {  // (the lane the DPP sums end in)
            if (qa < a.nq_launch) a.dist[(int64_t)qa * a.n_docs + doc] = va;
            if (qb < a.nq_launch) a.dist[(int64_t)qb * a.n_docs + doc] = vb;
        }
    };

    // ---- document cursor: cur = the doc the stream is in, end_cur = its end block, end_next = the next doc's (one ahead)
    int64_t cur = d0;
    int64_t end_cur = boff[cur + 1];
    int64_t end_next = cur + 2 <= a.n_docs ? boff[cur + 2] : end_cur;
    int64_t pos = b_begin;  // next block of the stream to be multiplied
    auto advance_doc = [&]() {  // to the next doc; empty docs on the way get NaN (the select skips them)
        for (;;) {
            cur = uni(cur + 1);
            if (cur >= d1) return;
            const int64_t prev_end = end_cur;
            end_cur = end_next;
            end_next = cur + 2 <= a.n_docs ? boff[cur + 2] : end_cur;
            if (end_cur != prev_end) return;
            write_doc(cur, kNaN, kNaN);
        }
    };
    while (cur < d1 && end_cur == pos) {  // leading empty docs
        write_doc(cur, kNaN, kNaN);
        cur = uni(cur + 1);
        if (cur >= d1) break;
        end_cur = end_next;
        end_next = cur + 2 <= a.n_docs ? boff[cur + 2] : end_cur;
    }
    if (n_my == 0) return;

    float run0 = -__builtin_inff(), run1 = -__builtin_inff();
    // A finished document's column maxima are PARKED in one of 2 BPS 2-KiB buffers and summed after the NEXT stage barrier --
    // the barrier the ring needs anyway publishes them; a barrier (and an LDS drain) of its own per document cost a text store,
    // whose documents are four blocks long, a tenth of the kernel.  Between two stage barriers lie BPS blocks, so at most BPS
    // documents end; a buffer is rewritten 2 BPS documents, i.e. at least two barriers, later.
    // The sums are STAGGERED: waves 0..3 take them right behind the barrier, waves 4..7 one block later -- waves w and w + 4
    // share a SIMD, so while one of the two walks through its sums (LDS reads, two DPP chains, the store: ~300 cycles in which
    // it issues no MFMA) the other keeps the matrix pipe busy; all eight at once left the pipe idle for that long per document.
    int n_fin = 0, pend_n = 0, ready_n = 0;
    bool flush_pending_now = false;
    int64_t pend_doc[BPS], ready_doc[BPS];
    int pend_buf[BPS], ready_buf[BPS];
#pragma unroll
    for (int i = 0; i < BPS; ++i) {
        pend_doc[i] = ready_doc[i] = 0;
        pend_buf[i] = ready_buf[i] = 0;
    }
    auto finish_doc = [&]() {
        // the two halves of the wave hold different token rows of the same query column: one v_permlane32_swap puts the halves
        // of BOTH running maxima side by side (lanes 0..31: column block cb0, lanes 32..63: cb1)
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(run0), __float_as_uint(run1), false, false);
        const float r = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        if (aligned) {
            // every query of the pass IS one column block (32-vector queries: ColBERT's padded queries): this wave already
            // holds every column of its two queries -- lanes 0..31 query `wave`, lanes 32..63 query `wave + 8` -- and sums them
            // itself: four row_shr steps and one row_bcast:15 leave the two sums in lanes 31 and 63.  No LDS, no parking, no
            // barrier: the per-document exchange below exists for queries that straddle column blocks.
            float part = (lane & 31) < len_mine ? r : 0.0f;
            part += mw_dpp<0x111, 0xF, true>(part);
            part += mw_dpp<0x112, 0xF, true>(part);
            part += mw_dpp<0x114, 0xF, true>(part);
            part += mw_dpp<0x118, 0xF, true>(part);
            part += mw_dpp<0x142, 0xA, false>(part);
            if ((lane & 31) == 31 && len_mine > 0) a.dist[(int64_t)(wave + 8 * (lane >> 5)) * a.n_docs + cur] = -part;
            run0 = run1 = -__builtin_inff();
            return;
        }
        const int bw = n_fin & (kPark - 1);
        float* cm = colmax + bw * 512;
        if (lane < 32) cm[cb0 * 32 + lane] = r;
        else if (two) cm[cb1 * 32 + lane - 32] = r;
#pragma unroll
        for (int i = 0; i < BPS; ++i)  // (static indices: the arrays stay in registers)
            if (i == pend_n) {
                pend_doc[i] = cur;
                pend_buf[i] = bw;
            }
        ++pend_n;
        ++n_fin;
        run0 = run1 = -__builtin_inff();
        if constexpr (!DEFER) {  // (A/B form: a barrier of its own per document, sums at once)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            MI355_BARRIER();
            flush_pending_now = true;
        }
    };
    auto publish_pending = [&]() {  // right behind a barrier: the maxima parked before it are visible to every wave
#pragma unroll
        for (int i = 0; i < BPS; ++i) {
            ready_doc[i] = pend_doc[i];
            ready_buf[i] = pend_buf[i];
        }
        ready_n = pend_n;
        pend_n = 0;
    };
    auto flush_ready = [&]() {
#pragma unroll
        for (int p = 0; p < BPS; ++p) {
            if (p >= ready_n) break;  // workgroup-uniform
            const float* cm = colmax + ready_buf[p] * 512;
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
            write_doc(ready_doc[p], out[0], out[1]);
        }
        ready_n = 0;
    };
    auto flush_pending = [&]() {  // (the unstaggered form: the immediate epilogue and the end of the range)
        flush_ready();
        publish_pending();
        flush_ready();
    };

    // ---- staging: wave w moves k-group fragment w of every block (1 KiB per instruction); past the range: the last block again
    const char* const tokbase = (const char*)a.tok16;
    const unsigned voff = (unsigned)lane * 16u;
    int64_t s_issue = 0;  // next stage to stage
    int slot_issue = 0;
    auto issue_stage = [&]() {
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
    auto read_block = [&](ms_bf16x8(&tf)[8], int slot, int u) {
        const char* p = smem + slot * kMwStageBytes + u * 8192 + lane * 16;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) tf[kk] = __builtin_bit_cast(ms_bf16x8, *(const uint4*)(p + kk * 1024));
    };
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto mul_block = [&](const ms_bf16x8(&tf)[8]) {
        f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[0], qf0[0], zero, 0, 0, 0);
#pragma unroll
        for (int i = 1; i < 8; ++i) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[i], qf0[i], acc0, 0, 0, 0);
        if (two) {  // wave-uniform
            f32x16 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[0], qf1[0], zero, 0, 0, 0);
#pragma unroll
            for (int i = 1; i < 8; ++i) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[i], qf1[i], acc1, 0, 0, 0);
            float m1 = acc1[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m1 = fmaxf(m1, acc1[r]);
            run1 = fmaxf(run1, m1);
        }
        float m0 = acc0[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m0 = fmaxf(m0, acc0[r]);
        run0 = fmaxf(run0, m0);
    };
    auto after_block = [&]() {  // one block of the stream has been multiplied
        ++pos;
        if (pos == end_cur) {  // workgroup-uniform: the document is complete
            finish_doc();
            if constexpr (!DEFER) {
                if (flush_pending_now) {
                    publish_pending();
                    flush_ready();
                    flush_pending_now = false;
                }
            }
            advance_doc();
        }
    };

    // stage 0 has landed (this wave's pieces: the BPS (stages - 1) youngest may still fly) and is visible
    static_assert(BPS * (kMwStages - 1) == 12, "the counted wait below");
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    MI355_BARRIER();
    if constexpr (PIPE) {
        // SOFTWARE-PIPELINED form: the maxima of block j are folded while the MFMAs of block j + 1 run.  A wave's 16 MFMAs
        // keep the SIMD's matrix pipe busy for 512 cycles; what follows them -- wait for the last result, two chains of 8
        // dependent v_max3, the bookkeeping, the fragment reads of the block after next -- is ~300 cycles in which THIS wave
        // issues no MFMA, and the two waves of a SIMD, walking the same stream behind the same barriers, do that at the same
        // time.  Here the fold sits in the same basic block as the next block's MFMAs (two accumulator sets, by block
        // parity) and is interleaved with them one VALU operation per MFMA.  Past the range the ring holds the last block
        // again: the MFMAs of those dummy blocks are issued unconditionally (no branch between MFMAs and fold), their maxima
        // fall into a running maximum nobody reads.
        auto stage_loop = [&](auto two_c) __attribute__((always_inline)) {
            constexpr bool TWO = decltype(two_c)::value;
            f32x16 x0, x1, y0, y1;
#pragma unroll
            for (int r = 0; r < 16; ++r) x0[r] = x1[r] = y0[r] = y1[r] = -__builtin_inff();
            bool primed = false;
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
                float m0 = p0[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m0 = fmaxf(m0, p0[r]);
                run0 = fmaxf(run0, m0);
                if constexpr (TWO) {
                    float m1 = p1[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) m1 = fmaxf(m1, p1[r]);
                    run1 = fmaxf(run1, m1);
                }
#pragma unroll
                for (int i = 0; i < (TWO ? 16 : 8); ++i) {  // one VALU operation in the shadow of every MFMA
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                }
                if (primed) after_block();
                primed = true;
            };
            int slot = 0;
            read_block(tfA, 0, 0);
            read_block(tfB, 0, 1);
            for (int64_t s = 0; s < n_stages; ++s) {
                const int slot_n = slot + 1 == kMwStages ? 0 : slot + 1;
#pragma unroll
                for (int j = 0; j < BPS; ++j) {
                    if (j & 1) slot_body(tfB, y0, y1, x0, x1);
                    else slot_body(tfA, x0, x1, y0, y1);
                    if (j == BPS - 2) {
                        if constexpr (BPS == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        MI355_BARRIER();
                        issue_stage();
                        if constexpr (DEFER) {
                            publish_pending();
                            if (wave < 4) flush_ready();
                        }
                    }
                    if constexpr (DEFER)
                        if (j == BPS - 1 && wave >= 4) flush_ready();
                    const int jn = j + 2;
                    if (j & 1) read_block(tfB, jn < BPS ? slot : slot_n, jn < BPS ? jn : jn - BPS);
                    else read_block(tfA, jn < BPS ? slot : slot_n, jn < BPS ? jn : jn - BPS);
                }
                slot = slot_n;
            }
            // the last block's products (BPS is even: they sit in the y set)
            float m0 = y0[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m0 = fmaxf(m0, y0[r]);
            run0 = fmaxf(run0, m0);
            if constexpr (TWO) {
                float m1 = y1[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m1 = fmaxf(m1, y1[r]);
                run1 = fmaxf(run1, m1);
            }
            after_block();
        };
        if (two) stage_loop(std::true_type{});
        else stage_loop(std::false_type{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        MI355_BARRIER();
        if constexpr (DEFER) flush_pending();
        return;
    }
    int slot = 0;
    // two fragment buffers: block j of a stage lives in buffer j & 1 and is read two blocks ahead of its MFMAs
    read_block(tfA, 0, 0);
    read_block(tfB, 0, 1);
    for (int64_t s = 0; s < n_stages; ++s) {
        const int slot_n = slot + 1 == kMwStages ? 0 : slot + 1;
#pragma unroll
        for (int j = 0; j < BPS; ++j) {
            if (j == 0 || pos < b_end) {  // workgroup-uniform (the range may end inside a stage)
                if (j & 1) mul_block(tfB);
                else mul_block(tfA);
                after_block();
            }
            if (j == BPS - 2) {
                // ---- hand-over: this wave's pieces of stage s + 1 have landed, its last fragments of stage s are in registers
                if constexpr (BPS == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                MI355_BARRIER();  // ... everybody's have, and everybody is done reading stage s: its slot takes the stage one ring ahead
                issue_stage();
                if constexpr (DEFER) {
                    publish_pending();  // the documents that ended since the previous barrier
                    if (wave < 4) flush_ready();
                }
            }
            if constexpr (DEFER)
                if (j == BPS - 1 && wave >= 4) flush_ready();  // (one block behind the other wave of this SIMD)
            // the block two ahead into the buffer just consumed: lands under the next block's MFMAs
            const int jn = j + 2;
            if (j & 1) read_block(tfB, jn < BPS ? slot : slot_n, jn < BPS ? jn : jn - BPS);
            else read_block(tfA, jn < BPS ? slot : slot_n, jn < BPS ? jn : jn - BPS);
        }
        slot = slot_n;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy stages must land before the LDS is freed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MI355_BARRIER();
    if constexpr (DEFER) flush_pending();  // the range's last document(s)
}

}  // namespace mi355
