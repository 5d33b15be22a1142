// mi355dr_maxsim_screen.hip -- the bf16 MFMA screens of the MaxSim path and their launch tables (split from mi355dr_maxsim.hip in
// round 6: same code, its own translation unit).  Reference: the `@#` scan of orm/repository/base.py:518-524 -- the screen only
// chooses which documents the exact kernel re-scores (rigorous bound, mi355dr_maxsim.hip header).
#include "maxsim_common.h"

using namespace mi355;

namespace mi355 {

// wave-wide fp32 sum by DPP, valid in LANE 63: four row_shr steps (inclusive prefix inside each row of 16 lanes), then
// row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 -- six VALU operations where the shuffle butterfly was six
// dependent ds_bpermute round trips through the LDS (a text document's two query sums: a tenth of the kernel)
template <int CTRL, int ROWMASK, bool BC>
__device__ __forceinline__ float mw_dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROWMASK, 0xF, BC));
}
__device__ __forceinline__ float mw_wave_sum_lane63(float v) {
    v += mw_dpp<0x111, 0xF, true>(v);   // row_shr:1
    v += mw_dpp<0x112, 0xF, true>(v);   // row_shr:2
    v += mw_dpp<0x114, 0xF, true>(v);   // row_shr:4
    v += mw_dpp<0x118, 0xF, true>(v);   // row_shr:8 -> lane 15 of every row holds the row's sum
    v += mw_dpp<0x142, 0xA, false>(v);  // row_bcast:15 -> rows 1, 3
    v += mw_dpp<0x143, 0xC, false>(v);  // row_bcast:31 -> rows 2, 3: lane 63 holds the wave's sum
    return v;
}

__device__ __forceinline__ void ms16_load_piece(uint4 (&a)[8], const uint4* blk, int piece, int nkk, int lane) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int kk = piece * 8 + i;
        a[i] = kk < nkk ? blk[(int64_t)kk * 64 + lane] : make_uint4(0u, 0u, 0u, 0u);
    }
}

__device__ __forceinline__ void ms16_compute_piece(f32x16 (&acc)[4], const uint4 (&a)[8], const uint4* qs, int ncb, int piece,
                                                   int nkk, int lane) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        if (cb >= ncb) break;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int kk = piece * 8 + i;
            if (kk >= nkk) break;
            const uint4 bv = qs[(cb * nkk + kk) * 64 + lane];
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ms_bf16x8, a[i]),
                                                              __builtin_bit_cast(ms_bf16x8, bv), acc[cb], 0, 0, 0);
        }
    }
}

__global__ __launch_bounds__(kMsThreads, 2) void k_maxsim16(Ms16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* qs = (uint4*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * a.nkk * 64; i += kMsThreads) qs[i] = a.qfrag[i];
    __syncthreads();
    int ncb = 0;
    for (int qi = 0; qi < a.nq_launch; ++qi) ncb = max(ncb, (a.q_col0[qi] + a.q_len[qi] + 31) / 32);
    const int npp = (a.nkk + 7) / 8;  // pieces of 8 fragments per block
    for (int dw = 0; dw < kMsDocsPerWave; ++dw) {
        const int64_t doc = ((int64_t)dw * gridDim.x + blockIdx.x) * 4 + wave;
        if (doc >= a.n_docs) break;
        const int64_t b0 = a.blk_off[doc], b1 = a.blk_off[doc + 1];
        float run[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) run[c] = -__builtin_inff();
        const int64_t npieces = (b1 - b0) * npp;
        f32x16 acc[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
        uint4 pa[8], pb[8];
        auto blk_of = [&](int64_t p) { return a.tok16 + (b0 + p / npp) * (int64_t)a.nkk * 64; };
        auto finish_block = [&]() {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                if (cb >= ncb) break;
                float m = acc[cb][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[cb][r]);
                m = fmaxf(m, __shfl_xor(m, 32, kWave));
                run[cb] = fmaxf(run[cb], m);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
            }
        };
        if (npieces > 0) ms16_load_piece(pa, blk_of(0), 0, a.nkk, lane);
        for (int64_t p = 0; p < npieces; p += 2) {
            if (p + 1 < npieces) ms16_load_piece(pb, blk_of(p + 1), (int)((p + 1) % npp), a.nkk, lane);
            ms16_compute_piece(acc, pa, qs, ncb, (int)(p % npp), a.nkk, lane);
            if ((p + 1) % npp == 0) finish_block();
            if (p + 1 < npieces) {
                if (p + 2 < npieces) ms16_load_piece(pa, blk_of(p + 2), (int)((p + 2) % npp), a.nkk, lane);
                ms16_compute_piece(acc, pb, qs, ncb, (int)((p + 1) % npp), a.nkk, lane);
                if ((p + 2) % npp == 0) finish_block();
            }
        }
        for (int qi = 0; qi < a.nq_launch; ++qi) {  // (masked butterfly sum over the query's column blocks: see k_maxsim16_d128)
            const int c0 = a.q_col0[qi], len = a.q_len[qi];  // (queries are packed column after column: any first column)
            float part = 0.0f;
#pragma unroll
            for (int cbi = 0; cbi < 4; ++cbi) {
                const int j = cbi * 32 + (lane & 31) - c0;  // this lane's column of block cbi as a token index of query qi
                if (j >= 0 && j < len) part += run[cbi];
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) part += __shfl_xor(part, o, kWave);
            if (lane == 0) a.dist[(int64_t)qi * a.n_docs + doc] = b1 > b0 ? -part : __uint_as_float(0x7FC00000u);
        }
    }
}

// ---- the same screen for dims <= 128 (8 fragments = ONE piece per 32-token block; the ColBERT / ColPali shape),
// with the number of column blocks a template parameter: every loop is unrolled at compile time, the query fragments
// stay in registers, LDS and global addresses are one base register + immediates, and the accumulator of a block
// starts from the MFMA's inline-zero C operand instead of 16 v_mov. ----
// NW = waves per workgroup: 4 while two workgroups fit a CU (<= 8 column blocks = 64 KiB of query fragments each), 8 beyond
// (one workgroup per CU by LDS: still two waves per SIMD).  A wave's documents do not depend on NW's siblings: no barrier
// after the staging.
// PK (round 6): the same walk over the GRANULE-PACKED copy (k_maxsim_wg8.h: documents rounded up to 8 tokens, the stream cut into 32-token
// blocks wherever they fall).  A document's first and last block may hold a neighbour's granules: their lanes are not loaded (a
// granule of one k-group half is one 128-byte line: the pass reads the document's own bytes only -- 10 % fewer on a store of
// passages, and up to ~8 column blocks this form is bound by exactly those bytes), and since register quad j of the accumulator IS
// granule j of the block (both wave halves), a boundary block's maximum is taken over the document's quads only.  The rows of
// lanes that were not loaded multiply whatever the registers held: an MFMA's rows do not mix, and their quads are never looked at.
template <int NCB, int NW, bool PK>
__global__ __launch_bounds__(NW * 64, 2) void k_maxsim16_d128(Ms16Args a, Ms16Pack pk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4* qs = (uint4*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    // (the wave index said to be uniform: the documents' block ranges -- and the packed form's boundary granules -- then live in
    // SGPRs.  That is what keeps PK from spilling next to the hoisted query fragments, and the padded form gains from it too:
    // 5 x 32-vector queries over 1 M passages 6.10 -> 5.74 ms, everything else within noise; profiles/r06_maxsim_pack8_ab.txt, table 11)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < NCB * 8 * 64; i += NW * 64) qs[i] = a.qfrag[i];
    __syncthreads();
    const uint4* const ql = qs + lane;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // The wave's kMsDocsPerWave documents are walked as ONE stream of 32-token blocks: the first block of the next document
    // is requested while the last block of the current one is multiplied (a text document is ~3 blocks: with the pipeline
    // restarted per document, every document paid one exposed HBM round trip).
    // The grid may be smaller than the store (option maxsim_persistent): the workgroups then walk the documents in rounds of 4
    // docs per wave and stage the query fragments once.  Measured (round 3, interleaved on one box): no gain on 1 M text docs,
    // 4 % slower on 100 k pages -- the default grid is one round.
    for (int64_t round = 0; round * ((int64_t)gridDim.x * NW * kMsDocsPerWave) < a.n_docs; ++round) {
    int64_t dq[kMsDocsPerWave], db0[kMsDocsPerWave], dnb[kMsDocsPerWave];
    int dlo[kMsDocsPerWave], dhi[kMsDocsPerWave];  // (PK) granules of the first block in front of the document, of the last block that are its own
#pragma unroll
    for (int dw = 0; dw < kMsDocsPerWave; ++dw) {
        dq[dw] = ((round * kMsDocsPerWave + dw) * (int64_t)gridDim.x + blockIdx.x) * NW + wave;
        const bool live = dq[dw] < a.n_docs;
        dlo[dw] = 0;
        dhi[dw] = 4;
        if constexpr (PK) {
            const int64_t g0 = live ? pk.goff[dq[dw]] : 0, g1 = live ? pk.goff[dq[dw] + 1] : 0;
            db0[dw] = g0 >> 2;
            dnb[dw] = g1 > g0 ? ((g1 + 3) >> 2) - (g0 >> 2) : 0;
            dlo[dw] = (int)(g0 & 3);
            dhi[dw] = (int)(g1 - (((g1 + 3) >> 2) - 1) * 4);  // 1..4 (unused when the document is empty)
        } else {
            db0[dw] = live ? a.blk_off[dq[dw]] : 0;
            dnb[dw] = live ? a.blk_off[dq[dw] + 1] - db0[dw] : 0;
        }
        if (!live) dq[dw] = -1;
    }
    uint4 pa[8], pb[8];
    if constexpr (PK) {
#pragma unroll
        for (int i = 0; i < 8; ++i) pa[i] = pb[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    // (PK) granule range [jlo, jhi) of a document's block p: all four but in its first and last block
    auto lo_of = [&](int dw, int64_t p) { return p == 0 ? dlo[dw] : 0; };
    auto hi_of = [&](int dw, int64_t p) { return p == dnb[dw] - 1 ? dhi[dw] : 4; };
    auto load = [&](uint4(&dst)[8], const uint4* src, int jlo, int jhi) {
        if constexpr (PK) {
            const int gr = (lane & 31) >> 3;  // this lane's granule of the block
            if (gr >= jlo && gr < jhi) {
#pragma unroll
                for (int i = 0; i < 8; ++i) dst[i] = src[i * 64];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) dst[i] = src[i * 64];
        }
    };
    float run[NCB];
    auto block = [&](const uint4(&fr)[8], int jlo, int jhi) {
        // Up to 8 column blocks the compiler keeps as many query fragments in registers as fit (they are loop-invariant) and
        // reads the rest per MFMA.  Beyond 8 that hoisting only costs: 16 blocks x 32 VGPRs cannot stay, and what it keeps anyway
        // pushes the kernel into scratch (10 spilled VGPRs at 16 blocks).  There the base pointer is made opaque once per token
        // block: every fragment is one ds_read_b128 in front of its MFMA, 128 B/clk per CU at the full matrix rate -- half of
        // what the LDS delivers.
        typedef const __attribute__((address_space(3))) uint4 lds_uint4;
        unsigned qoff = (unsigned)(unsigned long)((const __attribute__((address_space(3))) char*)(const char*)ql);
        // (PK at 4 blocks: what the compiler hoists there on top of the boundary logic's state spills)
        if constexpr (NCB > 8 || (PK && NCB == 4)) asm volatile("" : "+v"(qoff));  // (an LDS byte offset: the reads stay ds_read_b128, not FLAT)
        lds_uint4* const qlb = (lds_uint4*)(unsigned long)qoff;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ms_bf16x8, fr[0]),
                                                                 __builtin_bit_cast(ms_bf16x8, qlb[(cb * 8) * 64]), zero, 0, 0, 0);
#pragma unroll
            for (int i = 1; i < 8; ++i)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ms_bf16x8, fr[i]),
                                                              __builtin_bit_cast(ms_bf16x8, qlb[(cb * 8 + i) * 64]), acc, 0, 0, 0);
            float m;
            if (!PK || (jlo == 0 && jhi == 4)) {  // (wave-uniform)
                m = acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
            } else {  // a boundary block of the packed copy: the document's own quads
                m = -__builtin_inff();
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j >= jlo && j < jhi) m = fmaxf(m, fmaxf(fmaxf(acc[4 * j], acc[4 * j + 1]), fmaxf(acc[4 * j + 2], acc[4 * j + 3])));
            }
            run[cb] = fmaxf(run[cb], m);  // (the two halves of the wave are combined once per doc, below)
        }
    };
    auto blk_ptr = [&](int dw, int64_t p) { return (PK ? pk.tok16p : a.tok16) + (db0[dw] + p) * (8 * 64) + lane; };
    // first block of the first non-empty document
    int parity = 0;
    {
        const uint4* first = nullptr;
        int flo = 0, fhi = 4;
#pragma unroll
        for (int f = kMsDocsPerWave - 1; f >= 0; --f)
            if (dnb[f] > 0) {
                first = blk_ptr(f, 0);
                flo = lo_of(f, 0);
                fhi = hi_of(f, 0);
            }
        if (first) load(pa, first, flo, fhi);
    }
#pragma unroll
    for (int dw = 0; dw < kMsDocsPerWave; ++dw) {
        const int64_t doc = dq[dw];
        if (doc < 0) break;
        const uint4* first_next = nullptr;  // first block of the next non-empty document of this wave
        int fn_lo = 0, fn_hi = 4;
#pragma unroll
        for (int f = kMsDocsPerWave - 1; f > dw; --f)
            if (dnb[f] > 0) {
                first_next = blk_ptr(f, 0);
                fn_lo = lo_of(f, 0);
                fn_hi = hi_of(f, 0);
            }
        const int64_t nb = dnb[dw];
#pragma unroll
        for (int c = 0; c < NCB; ++c) run[c] = -__builtin_inff();
        for (int64_t p = 0; p < nb; ++p) {
            // the block after this one: the next of this document, or the first of the next non-empty one
            const uint4* nxt = p + 1 < nb ? blk_ptr(dw, p + 1) : first_next;
            const int n_lo = p + 1 < nb ? 0 : fn_lo, n_hi = p + 1 < nb ? hi_of(dw, p + 1) : fn_hi;
            if (parity == 0) {
                if (nxt) load(pb, nxt, n_lo, n_hi);
                block(pa, lo_of(dw, p), hi_of(dw, p));
            } else {
                if (nxt) load(pa, nxt, n_lo, n_hi);
                block(pb, lo_of(dw, p), hi_of(dw, p));
            }
            parity ^= 1;
        }
        // The two halves of the wave hold different token rows of the same column: one v_permlane32_swap joins the halves of
        // TWO column blocks at once (lanes 0..31: block 2 p, lanes 32..63: block 2 p + 1) -- round 4; was one ds_bpermute
        // round trip per block.
        float rr[(NCB + 1) / 2];
#pragma unroll
        for (int p2 = 0; p2 < (NCB + 1) / 2; ++p2) {
            float hi = 2 * p2 + 1 < NCB ? run[2 * p2 + 1] : -__builtin_inff();
            asm volatile("" : "+v"(hi));  // (two distinct registers: the swap of a register with itself is miscompiled)
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(run[2 * p2]), __float_as_uint(hi), false, false);
            rr[p2] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        // per query: the sum of its columns' maxima -- every column lives in exactly one lane now (block cbi in the wave half
        // cbi & 1), so a masked add per touched block and ONE wave-wide DPP sum (six VALU operations, result in lane 63) do it;
        // round 3 had five ds_bpermute butterfly steps per query here, and on 3-block text documents at 8 queries per pass that
        // epilogue was a third of the kernel.  The queries of a pass are packed column after column (eight 24-vector queries =
        // 6 column blocks, not 8 padded ones), so a query may start anywhere and span a block boundary.
        // The order of the fp32 additions differs from the exact kernel's; the screen's bound covers any order (e_acc).
        for (int qi = 0; qi < a.nq_launch; ++qi) {
            const int c0 = a.q_col0[qi], len = a.q_len[qi];
            float part = 0.0f;
#pragma unroll
            for (int cbi = 0; cbi < NCB; ++cbi) {
                // (wave-uniform skip of the blocks the query does not touch: a 32-token query touches one or two of the eight)
                if (cbi * 32 + 31 < c0 || cbi * 32 >= c0 + len) continue;
                const int j = cbi * 32 + (lane & 31) - c0;  // this lane's column of block cbi as a token index of query qi
                if (j >= 0 && j < len && (lane >> 5) == (cbi & 1)) part += rr[cbi >> 1];
            }
            part = mw_wave_sum_lane63(part);
            if (lane == 63) a.dist[(int64_t)qi * a.n_docs + doc] = nb > 0 ? -part : __uint_as_float(0x7FC00000u);
        }
    }
    }  // rounds
}

}  // namespace mi355
#include "k_maxsim_wg.h"
#include "k_maxsim_wg8.h"

// ---- k_maxsim16_d128<NCB, NW> by run-time NCB (1 .. kMsPassBlocks): NW = 4 up to 8 column blocks, 8 beyond ----
namespace mi355 {
namespace {
typedef void (*Ms16Kernel)(mi355::Ms16Args, mi355::Ms16Pack);
template <int NCB, bool PK>
constexpr Ms16Kernel ms16_kernel_of() {
    if constexpr (NCB <= 8) return mi355::k_maxsim16_d128<NCB, 4, PK>;
    else return mi355::k_maxsim16_d128<NCB, 8, PK>;
}
template <bool PK, int... I>
constexpr std::array<Ms16Kernel, sizeof...(I)> ms16_table(std::integer_sequence<int, I...>) {
    return {ms16_kernel_of<I + 1, PK>()...};
}
const std::array<Ms16Kernel, mi355::kMsPassBlocks> kMs16Kernels = ms16_table<false>(std::make_integer_sequence<int, mi355::kMsPassBlocks>{});
// ... over the granule-packed copy: up to FOUR column blocks -- the passes of one to four queries, which are bound by the token stream's
// bytes.  Measured, 1 M passages, 1 / 2 / 3 / 4 / 5 / 6 / 7 queries of 32 vectors per call, padded -> packed: 5.05 -> 4.53, 5.08 -> 4.55,
// 5.15 -> 4.85, 5.44 -> 5.36, 6.09 -> 6.07, 6.87 -> 6.90, 7.67 -> 9.16 ms (profiles/r06_maxsim_pack8_ab.txt, table 10): beyond four blocks
// the pass is no longer bound by bytes.  (At 4 and 5 blocks the query fragments are read per MFMA: hoisted on top of the boundary
// logic's state they spill.)
constexpr int kMs16PkMaxNcb = 4;
const std::array<Ms16Kernel, kMs16PkMaxNcb> kMs16PkKernels = ms16_table<true>(std::make_integer_sequence<int, kMs16PkMaxNcb>{});
inline int ms16_waves(int ncb) { return ncb <= 8 ? 4 : 8; }

// the workgroup-cooperative form (k_maxsim_wg.h) for 9 .. 16 column blocks
typedef void (*Ms16WgKernel)(mi355::Ms16Args, int64_t);
template <bool DEFER, int BPS, bool PIPE, int... I>
constexpr std::array<Ms16WgKernel, sizeof...(I)> ms16wg_table(std::integer_sequence<int, I...>) {
    return {mi355::k_maxsim16_wg<I + 9, DEFER, BPS, PIPE>...};
}
// [blocks per stage: 2, 4, 4 software-pipelined][epilogue: parked, at once][NCB - 9]
const std::array<Ms16WgKernel, 8> kMs16WgKernels[3][2] = {
    {ms16wg_table<true, 2, false>(std::make_integer_sequence<int, 8>{}), ms16wg_table<false, 2, false>(std::make_integer_sequence<int, 8>{})},
    {ms16wg_table<true, 4, false>(std::make_integer_sequence<int, 8>{}), ms16wg_table<false, 4, false>(std::make_integer_sequence<int, 8>{})},
    {ms16wg_table<true, 4, true>(std::make_integer_sequence<int, 8>{}), ms16wg_table<false, 4, true>(std::make_integer_sequence<int, 8>{})}};

// 8 column blocks (one per wave; the pipelined form only): [epilogue: parked, at once].  Measured (interleaved, round 4): with
// fewer MFMAs per block the workgroup form has a floor of ~880 cycles per block and CU (6.0 ms per pass over 100 k pages, 7.8 ms
// over 1 M text docs, whatever the column count) where one wave per document streams at 6.2 TB/s up to 4 column blocks:
// 8 blocks over short documents 8.17 against 8.97 ms; pages 6.04 against 5.15 ms and 5..7 blocks everywhere: one wave per document.
const Ms16WgKernel kMs16Wg8Kernels[2] = {mi355::k_maxsim16_wg<8, true, 4, true>, mi355::k_maxsim16_wg<8, false, 4, true>};

}  // namespace

int ms16_prepare(mi355dr_index* idx, size_t lds16) {
    if (lds16 <= 160 * 1024)
        HIPCHECK(idx, hipFuncSetAttribute((const void*)k_maxsim16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
    for (auto kfn : kMs16Wg8Kernels)
        HIPCHECK(idx, hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, mi355::mw_lds(4)));
    HIPCHECK(idx, hipFuncSetAttribute((const void*)mi355::k_maxsim16_wg8, hipFuncAttributeMaxDynamicSharedMemorySize, mi355::mw_lds(4)));
    for (int ncb = 1; ncb <= mi355::kMsPassBlocks; ++ncb)
        if (ncb * 8192 > 64 * 1024)
            HIPCHECK(idx, hipFuncSetAttribute((const void*)kMs16Kernels[ncb - 1], hipFuncAttributeMaxDynamicSharedMemorySize, ncb * 8192));
    for (int b = 0; b < 3; ++b)
        for (int e = 0; e < 2; ++e)
            for (auto kfn : kMs16WgKernels[b][e])
                HIPCHECK(idx, hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, mi355::mw_lds(b ? 4 : 2)));
    return MI355DR_OK;
}

// one screen launch over every doc: each wave walks kMsDocsPerWave docs; `persistent`: only as many workgroups as are resident
// at once, walking the docs in rounds (evened out: 100 k pages over 512 workgroups would be 12.2 rounds, a fifth of the chip
// idle in the last one)
bool ms16_takes_wg(const mi355dr_index* idx, int ncb, int64_t n_docs, int64_t n_blocks) {
    if (ncb == 8) return idx->maxsim_wg && idx->maxsim_wg_min <= 8 && (idx->maxsim_wg > 0 || n_blocks < 8 * n_docs);
    return ncb >= 9 && idx->maxsim_wg;
}

int ms16_d128_launch(mi355dr_index* idx, hipStream_t s, int ncb, int64_t n_docs, int64_t n_blocks, bool persistent,
                     const Ms16Args& sa, const Ms16Pack* pk) {
    if (pk && sa.aligned && ms16_takes_wg(idx, ncb, n_docs, n_blocks)) {
        // the granule-packed copy (k_maxsim_wg8.h): one workgroup per CU, each a contiguous range of documents with ~1/256 of the granules
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(256, (pk->n_pblocks + 31) / 32));
        hipLaunchKernelGGL(mi355::k_maxsim16_wg8, dim3(grid), dim3(512), (size_t)mi355::mw_lds(4), s, sa, *pk, ncb);
        HIPCHECK(idx, hipGetLastError());
        return MI355DR_OK;
    }
    if (ncb == 8 && idx->maxsim_wg && idx->maxsim_wg_min <= 8 && (idx->maxsim_wg > 0 || n_blocks < 8 * n_docs)) {
        const bool now = idx->maxsim_wg == 2;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(256, (n_blocks + 31) / 32));
        hipLaunchKernelGGL(kMs16Wg8Kernels[now ? 1 : 0], dim3(grid), dim3(512), (size_t)mi355::mw_lds(4), s, sa, n_blocks);
        HIPCHECK(idx, hipGetLastError());
        return MI355DR_OK;
    }
    if (ncb >= 9 && idx->maxsim_wg) {
        // per-document epilogue: parked behind the next stage barrier for stores of short documents (text: -3.5 % on the kernel),
        // at once for long ones (pages: the parked form's bookkeeping per stage costs 2 % there) -- interleaved A/B, round 4
        const bool now = idx->maxsim_wg == 2 || (idx->maxsim_wg < 0 && n_blocks >= 8 * n_docs);
        // one workgroup per CU, each a contiguous range of documents with ~1/256 of the token blocks (at least 32 blocks each)
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(256, (n_blocks + 31) / 32));
        const int bps = idx->maxsim_wg_bps == 2 ? 2 : 4;
        hipLaunchKernelGGL(kMs16WgKernels[bps == 4 ? (idx->maxsim_wg_pipe ? 2 : 1) : 0][now ? 1 : 0][ncb - 9], dim3(grid), dim3(512), (size_t)mi355::mw_lds(bps), s, sa,
                           n_blocks);
        HIPCHECK(idx, hipGetLastError());
        return MI355DR_OK;
    }
    const int nw = ms16_waves(ncb);
    const unsigned grid_docs = (unsigned)((n_docs + (int64_t)nw * mi355::kMsDocsPerWave - 1) / ((int64_t)nw * mi355::kMsDocsPerWave));
    const unsigned resident = ncb <= 10 && nw == 4 ? 512u : 256u;
    const unsigned rounds = persistent ? (grid_docs + resident - 1) / resident : 1u;
    const unsigned grid = std::max(1u, (grid_docs + rounds - 1) / std::max(rounds, 1u));
    const bool pk8 = pk && ncb <= kMs16PkMaxNcb;
    hipLaunchKernelGGL(pk8 ? kMs16PkKernels[ncb - 1] : kMs16Kernels[ncb - 1], dim3(grid), dim3(nw * 64), (size_t)ncb * 8 * 64 * sizeof(uint4), s, sa,
                       pk8 ? *pk : mi355::Ms16Pack{});
    HIPCHECK(idx, hipGetLastError());
    return MI355DR_OK;
}

int ms16_generic_launch(mi355dr_index* idx, hipStream_t s, unsigned grid, size_t lds16, const Ms16Args& sa) {
    hipLaunchKernelGGL(k_maxsim16, dim3(grid), dim3(kMsThreads), lds16, s, sa);
    HIPCHECK(idx, hipGetLastError());
    return MI355DR_OK;
}

}  // namespace mi355
