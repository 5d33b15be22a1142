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

constexpr int kScanQ = 8;        // queries per launch
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

__host__ __device__ inline size_t scan_lds_bytes(int d, int nq) {
    return (size_t)4 * kStageFloats * sizeof(float) + (size_t)nq * d * sizeof(float);
}

__global__ __launch_bounds__(kScanThreads) void k_scan(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tiles = (float*)smem;
    float* qs = (float*)(smem + (size_t)4 * kStageFloats * sizeof(float));  // [nq][d]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < a.nq * a.d; i += kScanThreads) {
        const int j = i / a.d, k = i - j * a.d;
        qs[i] = a.q[(int64_t)a.qlist[j] * a.d + k];
    }
    __syncthreads();
    const int64_t row = a.row0 + (int64_t)blockIdx.x * kScanThreads + wave * kWave + lane;
    const bool live = row < a.row1;
    const float* rp = live ? a.rows + row * (int64_t)a.d : nullptr;
    float acc[kScanQ];
#pragma unroll
    for (int j = 0; j < kScanQ; ++j) acc[j] = 0.0f;
    float* tile = tiles + wave * kStageFloats;
    if (a.row0 + (int64_t)blockIdx.x * kScanThreads + wave * kWave < a.row1) {  // wave-uniform
        for (int k0 = 0; k0 < a.d; k0 += kStageCols) {
            stage_rows(tile, rp, k0, a.d, lane);
            const int kn = min(kStageCols, a.d - k0);
            const float* t = tile + lane * kStageLd;
            for (int k = 0; k < kn; ++k) {
                const float cv = t[k];
#pragma unroll
                for (int j = 0; j < kScanQ; ++j)
                    if (j < a.nq) acc[j] = __builtin_fmaf(cv, qs[j * a.d + k0 + k], acc[j]);
            }
        }
    }
    if (!live) return;
    const float nc = a.nrm2[row];
#pragma unroll
    for (int j = 0; j < kScanQ; ++j) {
        if (j >= a.nq) break;
        const int q = a.qlist[j];
        const uint64_t key = dist_to_key(distance_from(a.metric, acc[j], a.st.qn[q], nc));
        const uint64_t tk = a.st.thr_key[q];
        const int32_t tr = a.st.thr_row[q];
        if (key < tk || (key == tk && (int32_t)row < tr)) {
            const int slot = atomicAdd(&a.st.cnt[q], 1);
            if (slot < a.cap) {
                a.cand_row[(int64_t)q * a.cap + slot] = (int32_t)row;
                a.cand_val[(int64_t)q * a.cap + slot] = acc[j];
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
