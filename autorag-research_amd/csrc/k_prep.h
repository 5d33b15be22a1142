// k_prep.h -- corpus-side and query-side preparation kernels.
//   k_row_nrm2      |c|^2 per stored row, exact k-ascending fp32 chain (pgvector's `normb`, oracle.c orc_dot)
//   k_build_shadow  bf16 shadow row  c_hat = bf16_rn(c / |c|)  streamed by the screen kernel
//   k_prep_queries  |q|^2, q_hat = bf16_rn(q / |q|), per-query search state reset
#pragma once
#include "dev_common.h"

namespace mi355 {

// grid: ceil(n/64) blocks of 64 threads (one wave, 64 rows)
__global__ __launch_bounds__(64) void k_row_nrm2(const float* __restrict__ rows, int64_t row0, int64_t n, int d,
                                                  float* __restrict__ nrm2) {
    __shared__ float tile[kStageFloats];
    const int lane = threadIdx.x;
    const int64_t i = row0 + (int64_t)blockIdx.x * kWave + lane;
    const bool live = i < row0 + n;
    const float* my_row = live ? rows + i * (int64_t)d : nullptr;
    float acc = 0.0f;
    for (int k0 = 0; k0 < d; k0 += kStageCols) {
        stage_rows(tile, my_row, k0, d, lane);
        __syncthreads();
        const int kn = min(kStageCols, d - k0);
        const float* t = tile + lane * kStageLd;
        for (int k = 0; k < kn; ++k) acc = __builtin_fmaf(t[k], t[k], acc);
        __syncthreads();
    }
    if (live) nrm2[i] = acc;
}

// grid: n blocks of 256 threads (one row each).  Irregular rows get an all-NaN shadow row (never
// emitted by the screen: NaN compares false) and are recorded in irr_rows for the exact side pass.
__global__ __launch_bounds__(256) void k_build_shadow(const float* __restrict__ rows, const float* __restrict__ nrm2,
                                                       int64_t row0, int64_t n, int d, int dpad,
                                                       uint16_t* __restrict__ shadow, int32_t* __restrict__ irr_rows,
                                                       int* __restrict__ irr_count) {
    const int64_t i = row0 + blockIdx.x;
    if (i >= row0 + n) return;
    const float n2 = nrm2[i];
    const bool regular = norm_is_regular(n2);
    const float rc = regular ? 1.0f / sqrtf(n2) : 0.0f;
    const float* r = rows + i * (int64_t)d;
    uint16_t* s = shadow + i * (int64_t)dpad;
    for (int k = threadIdx.x; k < dpad; k += blockDim.x) {
        uint16_t v = 0;
        if (k < d) v = regular ? f32_to_bf16_rn(r[k] * rc) : (uint16_t)0x7FC0;  // bf16 quiet NaN
        s[k] = v;
    }
    if (!regular && threadIdx.x == 0) {
        int slot = atomicAdd(irr_count, 1);
        if (slot < kIrrCap) irr_rows[slot] = (int32_t)i;
    }
}


// grid: Bpad blocks of 64 threads
__global__ __launch_bounds__(64) void k_prep_queries(const float* __restrict__ q, int B, int d, int dpad, int metric,
                                                      QueryState st) {
    extern __shared__ float qs[];  // [d]
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    uint16_t* qh = st.qhat + (int64_t)b * dpad;
    if (b >= B) {  // padding rows of the last 128-query tile
        for (int k = lane; k < dpad; k += kWave) qh[k] = 0;
        if (lane == 0) {
            st.qn[b] = 0.0f;
            st.thr[b] = __builtin_inff();
            st.cnt[b] = 0;
            st.best_n[b] = 0;
            st.thr_key[b] = 0;
            st.thr_row[b] = -1;
            st.status[b] = 0;
        }
        return;
    }
    const float* qr = q + (int64_t)b * d;
    for (int k = lane; k < d; k += kWave) qs[k] = qr[k];
    __syncthreads();
    float acc = 0.0f;
    // every lane runs the same chain (LDS broadcast reads): no divergence, lane 0's value is used
    for (int k = 0; k < d; ++k) acc = __builtin_fmaf(qs[k], qs[k], acc);
    const bool regular = norm_is_regular(acc);
    const float rq = regular ? 1.0f / sqrtf(acc) : 0.0f;
    for (int k = lane; k < dpad; k += kWave) qh[k] = (k < d && regular) ? f32_to_bf16_rn(qs[k] * rq) : (uint16_t)0;
    if (lane == 0) {
        st.qn[b] = acc;
        // cosine screen cannot rank an irregular query: park it (never emits) and flag it for the scan path
        st.thr[b] = (regular || metric != 0) ? -__builtin_inff() : __builtin_inff();
        st.cnt[b] = 0;
        st.best_n[b] = 0;
        st.thr_key[b] = kKeyNaN;
        st.thr_row[b] = 0x7FFFFFFF;
        st.status[b] = regular ? 0 : kStIrregular;
    }
}

}  // namespace mi355
