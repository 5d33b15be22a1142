// k_scan.h -- the guaranteed exact path: one k-ascending fp32 chain per (query,row), no screen.
// It is what the oracle does, written for a wave: each lane owns a row, rows are staged through LDS in
// coalesced 256-B pieces, up to kScanQ queries are scored against the staged tile at once.  A pair is
// appended to the query's candidate list iff its exact (distance,row) key sorts before the query's
// current k-th best; k_prune (exact mode) folds the list in.  Used (a) for every query the screen path
// flags (candidate overflow, irregular query norm), (b) for the inner-product metric, (c) when asked
// for explicitly (tests cross-check screen vs scan vs oracle).
// Reference semantics: base.py:409-415 (sequential scan + ORDER BY distance LIMIT k).
#pragma once
#include "dev_common.h"
#include "k_prep.h"

namespace mi355 {

constexpr int kScanQ = 32;         // queries per launch = one MFMA column block
constexpr int kScanThreads = 256;  // 4 waves x 64 rows

struct ScanArgs {
    const float* rows;
    const float* nrm2;
    const float* q;       // [B, d]
    QueryState st;
    int32_t* cand_row;
    float* cand_val;
    const int* qlist;     // [nq] query indices handled by this launch (nq <= kScanQ)
    int nq;
    int cap, d, metric;
    int64_t row0, row1;   // chunk
};

// per wave: a 64-row x 64-column piece of the corpus rows; per WORKGROUP: the matching 32-query x 64-column piece of the
// queries (round 3: one copy for the four waves instead of one each -- 77 KiB instead of 103 per workgroup, so TWO workgroups
// fit a CU and every SIMD holds two waves: one walks its LDS hand-off while the other one's MFMAs run)
constexpr int kScanQTileFloats = 32 * kStageLd;
__host__ __device__ inline size_t scan_lds_bytes(int /*d*/, int /*nq*/) {
    return (size_t)(4 * kStageFloats + kScanQTileFloats) * sizeof(float);
}

typedef float scan_f32x16 __attribute__((ext_vector_type(16)));

// Exact scan on the fp32 matrix pipe.  v_mfma_f32_32x32x2_f32 IS the oracle's chain: D = fma(a_k1, b_k1, fma(a_k0, b_k0,
// C)) per output element, k ascending, one rounding per product (MI355X guide: bitwise equal to a v_fmac_f32 loop) -- the
// same instruction the exact MaxSim kernel uses.  Each wave owns 64 rows (2 blocks of 32) x up to 32 queries; row pieces of
// 64 columns are pulled with coalesced 256-B loads two pieces ahead of use (StagePiece, dev_common.h), the query piece comes
// from L2 the same way; both are read back from a padded LDS image as float4 (k, k+1, k+2, k+3): the wave half h = lane>>5
// feeds k+h, then k+2+h.  Zero padding (columns >= d, idle rows, queries >= nq) adds fma(0, 0, acc) = acc: exact.
// At 32 queries per pass the MFMA time of a piece equals its HBM time (16 B/clk/CU): the first form of this kernel (one
// VALU chain per lane, 8 queries, operands fetched one scalar LDS read per fma, no prefetch) ran at ~1 TB/s.
__global__ __launch_bounds__(kScanThreads, 2) void k_scan(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* tile = (float*)smem + wave * kStageFloats;
    float* qtile = (float*)smem + 4 * kStageFloats;  // shared by the four waves
    const int64_t wrow0 = a.row0 + (int64_t)blockIdx.x * kScanThreads + wave * kWave;
    // (a wave past the end of the chunk keeps walking with idle slots: the query piece is handed over at block barriers)
    const int64_t myrow = wrow0 + lane;
    const float* rp = myrow < a.row1 ? a.rows + myrow * (int64_t)a.d : nullptr;
    const int lq = lane & 31;
    // the query piece of a K step is loaded ONCE per workgroup: wave w brings rows 8w .. 8w+7 of the 32 (its lanes' slots
    // 8w + (lane >> 4) ... are the StageRows groups g = 2w, 2w + 1)
    const float* qp = (lane < 32 && lq < a.nq) ? a.q + (int64_t)a.qlist[lq] * a.d : nullptr;  // lanes 32..63: idle slots
    const int d = a.d;
    const bool vec = (d & 3) == 0;

    scan_f32x16 acc[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;

    // this wave's quarter of the query piece: groups g = 2 wave, 2 wave + 1 (query rows 8 wave .. 8 wave + 7)
    struct QPiece {
        float4 v[2];
    };
    StageRows sr;
    StagePiece p0, p1;
    QPiece q0, q1;
    // idle slots read a live lane's row (their products are never looked at): pieces inside the rows need no per-load
    // predicate (dev_common.h, stage_rows_init_dense)
    bool dense = false;
    auto issue = [&](StagePiece& p, const StageRows& r, int k0) __attribute__((always_inline)) {
        if (dense && k0 + kStageCols <= d) stage_issue_dense(p, r, k0, lane);  // wave-uniform
        else stage_issue(p, r, k0, d, lane);
    };
    int qoff[2] = {-1, -1};  // this wave's two query rows as float offsets from a.q (picked once: holding all eight pointers spills)
    auto issue_q = [&](QPiece& p, int k0) __attribute__((always_inline)) {
        const int c4 = (lane & 15) * 4;
#pragma unroll
        for (int u = 0; u < 2; ++u)
            p.v[u] = (qoff[u] >= 0 && k0 + c4 < d) ? load_gmem_f4(a.q + qoff[u] + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    if (vec) {
        const bool dr = stage_rows_init_dense(sr, rp, lane);
        {
            StageRows sq;
            const bool dq = stage_rows_init_dense(sq, qp, lane);
            dense = dr && dq;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float* src = wave == 0 ? sq.r[u] : wave == 1 ? sq.r[2 + u] : wave == 2 ? sq.r[4 + u] : sq.r[6 + u];
                qoff[u] = src != nullptr ? (int)(src - a.q) : -1;  // (query blocks are far below 2^31 floats)
            }
        }
        issue(p0, sr, 0);
        issue_q(q0, 0);
        if (kStageCols < d) {
            issue(p1, sr, kStageCols);
            issue_q(q1, kStageCols);
        }
    }
    const int h = lane >> 5;
    const float* ta0 = tile + lq * kStageLd;
    const float* ta1 = tile + (32 + lq) * kStageLd;
    const float* tb = qtile + lq * kStageLd;
    auto mfma_piece = [&]() {
#pragma unroll 4
        for (int u = 0; u < kStageCols / 4; ++u) {
            const float4 a0 = *(const float4*)(ta0 + 4 * u), a1 = *(const float4*)(ta1 + 4 * u);
            const float4 b = *(const float4*)(tb + 4 * u);
            const float bx = h ? b.y : b.x, bz = h ? b.w : b.z;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a0.y : a0.x, bx, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a1.y : a1.x, bx, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a0.w : a0.z, bz, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a1.w : a1.z, bz, acc[1], 0, 0, 0);
        }
    };
    // the query tile holds 32 rows: the committing helper writes rows 4g + (lane>>4), g = 0..15 -> rows 32..63 of a
    // 64-row image; qtile only has 32, so the query piece is committed by hand (g = 0..7)
    auto commit_q = [&](const QPiece& p) {  // rows 4 g + (lane >> 4), g = 2 wave + u
        const int sub = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) *(float4*)(qtile + ((2 * wave + u) * 4 + sub) * kStageLd + c4) = p.v[u];
    };
    if (vec) {
        for (int k0 = 0; k0 < d; k0 += 2 * kStageCols) {
            stage_commit(tile, p0, lane);  // (wave_sync before and after: this wave's rows)
            __syncthreads();               // every wave is done reading the previous query piece
            commit_q(q0);
            __syncthreads();               // the query piece is complete
            if (k0 + 2 * kStageCols < d) {
                issue(p0, sr, k0 + 2 * kStageCols);
                issue_q(q0, k0 + 2 * kStageCols);
            }
            mfma_piece();
            if (k0 + kStageCols < d) {
                stage_commit(tile, p1, lane);
                __syncthreads();
                commit_q(q1);
                __syncthreads();
                if (k0 + 3 * kStageCols < d) {
                    issue(p1, sr, k0 + 3 * kStageCols);
                    issue_q(q1, k0 + 3 * kStageCols);
                }
                mfma_piece();
            }
        }
    } else {  // rows not 16-B aligned: scalar staging, no prefetch (rare dims)
        for (int k0 = 0; k0 < d; k0 += kStageCols) {
            stage_rows(tile, rp, k0, d, lane);
            __syncthreads();
            for (int s = wave * 8; s < wave * 8 + 8; ++s) {  // query rows, one scalar column per lane; 8 rows per wave
                const int qi = s < a.nq ? a.qlist[s] : -1;
                const int k = k0 + lane;
                qtile[s * kStageLd + lane] = (qi >= 0 && k < d) ? a.q[(int64_t)qi * d + k] : 0.0f;
            }
            __syncthreads();
            mfma_piece();
        }
    }

    // ---- epilogue.  C/D layout: column (query) = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) within each block of 32
    if (lq >= a.nq || wrow0 >= a.row1) return;
    const int q = a.qlist[lq];
    const uint64_t tk = a.st.thr_key[q];
    const int32_t tr = a.st.thr_row[q];
    const float qn = a.st.qn[q];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = wrow0 + 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= a.row1) continue;
            const float dot = acc[rb][r];
            const uint64_t key = dist_to_key(distance_from(a.metric, dot, qn, a.nrm2[row]));
            if (key < tk || (key == tk && (int32_t)row < tr)) {
                const int slot = atomicAdd(&a.st.cnt[q], 1);
                if (slot < a.cap) {
                    a.cand_row[(int64_t)q * a.cap + slot] = (int32_t)row;
                    a.cand_val[(int64_t)q * a.cap + slot] = dot;
                }
            }
        }
}

// ---- round 3: the same scan with LDS-DMA staging (dims that are a multiple of 32: every 128-byte piece of a row is whole)
// What bounded k_scan above: two 64-column pieces of 64 rows live in 128 VGPRs per wave, every piece walks a ds_write phase
// behind a wave-level hand-off, and at 256 VGPRs the allocator spills.  Here a piece is 32 columns (128 B of every row):
// a wave's 64 rows x 128 B = 8 KiB arrive by eight global_load_lds_dwordx4 (8 rows x 128 B each, full lines, the screens'
// lane-linear image with the XOR swizzle on the source chunk), the 32-query piece (4 KiB) by one such instruction per wave;
// double-buffered in LDS (2 x (4 x 8 + 4) KiB = 72 KiB per workgroup: two workgroups per CU), ONE block barrier per piece.
// No register pieces, no ds_write, fragments read back as 16-byte chunks (k .. k+3 of a row; the wave half picks k + h, then
// k + 2 + h).  Same MFMA chain, same epilogue: bit-identical to k_scan.
typedef __bf16 scan_bf16x8 __attribute__((ext_vector_type(8)));
typedef float scan_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kScan32PieceCols = 32;
constexpr int kScan32RowTile = 64 * 128;    // one wave's rows of a piece: 8 KiB
constexpr int kScan32QTile = 32 * 128;      // the query piece: 4 KiB
constexpr int kScan32Stage = 4 * kScan32RowTile + kScan32QTile;  // 36 KiB
constexpr int kScan32Lds = 2 * kScan32Stage;                       // 72 KiB

__device__ __forceinline__ void scan_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// 16-byte chunk c of row r of a 128-byte-pitch image lives in slot c ^ ((r >> 1) & 7)
__device__ __forceinline__ scan_f32x4 scan_frag(const char* tile, int row, int chunk) {
    // (typed bf16x8 on purpose: with a plain float4 / uint4 type the waitcnt insertion assumes the read may alias the LDS-DMA in
    // flight and puts s_waitcnt vmcnt(0) in front of it -- see the NOTE in k_screen.h)
    return __builtin_bit_cast(scan_f32x4, *(const scan_bf16x8*)(tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4)));
}

__global__ __launch_bounds__(kScanThreads, 2) void k_scan32(ScanArgs a, int64_t n_rows_valid) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t wrow0 = a.row0 + (int64_t)blockIdx.x * kScanThreads + wave * kWave;
    const int d = a.d, npieces = d / kScan32PieceCols;
    const int64_t row_bytes = (int64_t)d * 4;
    // ---- DMA sources.  Rows: instruction g (0..7) brings local rows 8 g + (lane >> 3), lane & 7 = the LDS slot, whose source
    // chunk is slot ^ ((row >> 1) & 7).  Rows past the end of the index read its last row (their results are never looked at).
    const char* srcA[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int r = 8 * g + (lane >> 3);
        const int64_t row = min(wrow0 + r, n_rows_valid - 1);
        srcA[g] = (const char*)a.rows + row * row_bytes + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    // Queries: wave w brings query slots 8 w + (lane >> 3); slots >= nq read query slot 0 (their columns are never looked at)
    const char* srcQ;
    {
        const int r = 8 * wave + (lane >> 3);
        const int qi = a.qlist[r < a.nq ? r : 0];
        srcQ = (const char*)a.q + (int64_t)qi * row_bytes + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    auto stage = [&](int piece, int buf) __attribute__((always_inline)) {
        char* base = smem + buf * kScan32Stage;
        const int64_t koff = (int64_t)piece * 128;
#pragma unroll
        for (int g = 0; g < 8; ++g) scan_glds16(srcA[g] + koff, base + wave * kScan32RowTile + g * 1024);
        scan_glds16(srcQ + koff, base + 4 * kScan32RowTile + wave * 1024);
    };
    scan_f32x16 acc[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
    const int h = lane >> 5, lq = lane & 31;
    stage(0, 0);
    for (int p = 0; p < npieces; ++p) {
        const int buf = p & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's parts of piece p have landed
        __syncthreads();                                   // ... everybody's; and everybody is done reading the other buffer
        if (p + 1 < npieces) stage(p + 1, buf ^ 1);
        const char* ta = smem + buf * kScan32Stage + wave * kScan32RowTile;
        const char* tq = smem + buf * kScan32Stage + 4 * kScan32RowTile;
        // all 24 fragments of the piece are requested up front (96 VGPRs; the kernel has room: no register pieces), so the
        // 32 MFMAs of the piece run without an LDS round trip between them
        scan_f32x4 fa0[8], fa1[8], fb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            fa0[u] = scan_frag(ta, lq, u);
            fa1[u] = scan_frag(ta, 32 + lq, u);
            fb[u] = scan_frag(tq, lq, u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {  // 4 columns per step: K pairs (4u, 4u+1) and (4u+2, 4u+3)
            const scan_f32x4 a0 = fa0[u], a1 = fa1[u], b = fb[u];
            const float bx = h ? b.y : b.x, bz = h ? b.w : b.z;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a0.y : a0.x, bx, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a1.y : a1.x, bx, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a0.w : a0.z, bz, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a1.w : a1.z, bz, acc[1], 0, 0, 0);
        }
    }
    // ---- epilogue (as k_scan).  C/D layout: column (query) = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) per block of 32
    if (lq >= a.nq || wrow0 >= a.row1) return;
    const int q = a.qlist[lq];
    const uint64_t tk = a.st.thr_key[q];
    const int32_t tr = a.st.thr_row[q];
    const float qn = a.st.qn[q];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = wrow0 + 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row >= a.row1) continue;
            const float dot = acc[rb][r];
            const uint64_t key = dist_to_key(distance_from(a.metric, dot, qn, a.nrm2[row]));
            if (key < tk || (key == tk && (int32_t)row < tr)) {
                const int slot = atomicAdd(&a.st.cnt[q], 1);
                if (slot < a.cap) {
                    a.cand_row[(int64_t)q * a.cap + slot] = (int32_t)row;
                    a.cand_val[(int64_t)q * a.cap + slot] = dot;
                }
            }
        }
}

// debug / test hook: exact dot + distance for explicit (query,row) pairs through the same staged chain.
// grid: ceil(n_pairs/64) blocks of 64 threads.  The query row is staged like a corpus row.
__global__ __launch_bounds__(64) void k_rescore_pairs(const float* __restrict__ rows, const float* __restrict__ nrm2,
                                                       const float* __restrict__ q, const float* __restrict__ qn,
                                                       const int32_t* __restrict__ pair_q,
                                                       const int64_t* __restrict__ pair_row, int64_t n_pairs, int d,
                                                       int metric, float* out_dot, double* out_dist) {
    __shared__ float tile_c[kStageFloats];
    __shared__ float tile_q[kStageFloats];
    const int lane = threadIdx.x;
    const int64_t p = (int64_t)blockIdx.x * kWave + lane;
    const bool live = p < n_pairs;
    const float* rp = live ? rows + pair_row[p] * (int64_t)d : nullptr;
    const float* qp = live ? q + (int64_t)pair_q[p] * d : nullptr;
    float acc = 0.0f;
    for (int k0 = 0; k0 < d; k0 += kStageCols) {
        stage_rows(tile_c, rp, k0, d, lane);
        stage_rows(tile_q, qp, k0, d, lane);
        const int kn = min(kStageCols, d - k0);
        const float* tc = tile_c + lane * kStageLd;
        const float* tq = tile_q + lane * kStageLd;
        for (int k = 0; k < kn; ++k) acc = __builtin_fmaf(tc[k], tq[k], acc);
    }
    if (live) {
        out_dot[p] = acc;
        out_dist[p] = distance_from(metric, acc, qn[pair_q[p]], nrm2[pair_row[p]]);
    }
}

}  // namespace mi355
