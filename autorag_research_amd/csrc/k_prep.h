// k_prep.h -- corpus-side and query-side preparation kernels.
//   k_row_nrm2      |c|^2 per stored row, exact k-ascending fp32 chain (pgvector's `normb`, oracle.c orc_dot)
//   k_build_shadow  bf16 shadow row  c_hat = bf16_rn(c / |c|)  streamed by the screen kernel
//   k_build_shadow8 int8 shadow rows round(c / |c| / S_g), one step per group of 32 rows (+ loose-row detection)
//   k_prep_queries  |q|^2, q_hat = bf16_rn(q / |q|) and its int8 form, per-query bound, search state reset
#pragma once
#include "dev_common.h"

namespace mi355 {

// grid: ceil(n/64) blocks of 64 threads (one wave, 64 rows)
// n2max: running maximum of the regular |c|^2 (float bits; inner-product thresholds need the largest row norm)
__global__ __launch_bounds__(64) void k_row_nrm2(const float* __restrict__ rows, int64_t row0, int64_t n, int d,
                                                  float* __restrict__ nrm2, unsigned* __restrict__ n2max) {
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
    float m = (live && norm_is_regular(acc)) ? acc : 0.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) {
        const unsigned v = __float_as_uint(m);
        if (v > *(volatile unsigned*)n2max) atomicMax(n2max, v);
    }
}

// grid: n blocks of 256 threads (one row each).  Irregular rows get an all-NaN shadow row (never
// emitted by the screen: NaN compares false) and are recorded in irr_rows for the exact side pass.  The residual norm
// |c_hat - bf16(c_hat)| of every regular row is measured; res2_max keeps the largest squared one (bit pattern of a
// non-negative float: unsigned max == float max) -- the corpus half of the bf16 screen bound.
// absolute != 0 (inner-product metric): the shadow holds bf16_rn(c) itself -- the screen then estimates <q_hat, c> =
// dot / |q| directly and no row norm enters the threshold (dev_common.h "inner product").
__global__ __launch_bounds__(256) void k_build_shadow(const float* __restrict__ rows, const float* __restrict__ nrm2,
                                                       int64_t row0, int64_t n, int d, int dpad,
                                                       uint16_t* __restrict__ shadow, int32_t* __restrict__ irr_rows,
                                                       int* __restrict__ irr_count, unsigned* __restrict__ res2_max,
                                                       int absolute) {
    __shared__ float sh[4];
    const int64_t i = row0 + blockIdx.x;
    if (i >= row0 + n) return;
    const float n2 = nrm2[i];
    const bool regular = norm_is_regular(n2);
    const float rc = regular ? (absolute ? 1.0f : 1.0f / sqrtf(n2)) : 0.0f;
    const float* r = rows + i * (int64_t)d;
    uint16_t* s = shadow + i * (int64_t)dpad;
    float e2 = 0.0f;
    for (int k = threadIdx.x; k < dpad; k += blockDim.x) {
        uint16_t v = 0;
        if (k < d) {
            if (regular) {
                const float ch = r[k] * rc;
                v = f32_to_bf16_rn(ch);
                const float e = ch - bf16_bits_to_f32(v);  // exact in fp32
                e2 = __builtin_fmaf(e, e, e2);
            } else {
                v = (uint16_t)0x7FC0;  // bf16 quiet NaN
            }
        }
        s[k] = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e2 += __shfl_xor(e2, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = e2;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (regular) {
            // (read first: the running maximum settles after a few thousand rows, and 250 k atomics on one address
            // per launch cost 2.5 ms)
            const unsigned v = __float_as_uint((sh[0] + sh[1]) + (sh[2] + sh[3]));
            if (v > *(volatile unsigned*)res2_max) atomicMax(res2_max, v);
        }
        else {
            int slot = atomicAdd(irr_count, 1);
            if (slot < kIrrCap) irr_rows[slot] = (int32_t)i;
        }
    }
}


// block-wide sum / max over 256 threads (4 waves)
__device__ __forceinline__ float block256_reduce(float v, bool is_max, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float w = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, w) : v + w;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])) : (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// grid: one block of 256 threads per GROUP of kI8GroupRows = 32 rows, groups [g0, g0 + gridDim.x) (dev_common.h: "int8
// screen quantisation").  Wave w owns rows 8w .. 8w+7 of the group.  Pass 1: every row's peak |c_hat_k|; a row is loose
// when it is irregular or its peak is above i8_row_peak_limit(d).  The group's step S_g = (largest peak of its tight rows)
// / 127.  Pass 2: c8 = round(c_hat / S_g) for the tight rows (all-zero rows for the loose ones and for rows >= n_total),
// the residual norm |c_hat - S_g c8| measured per row, e_g = the largest.  A group is always rebuilt whole: when a later
// add() lands in a partly filled group its old rows are requantised with the new step.  Loose rows are registered in
// irr8_rows once -- only rows >= first_new (the rows this add() brought) are appended.
__global__ __launch_bounds__(256) void k_build_shadow8(const float* __restrict__ rows, const float* __restrict__ nrm2,
                                                        int64_t g0, int64_t n_total, int64_t first_new, int d, int dpad8,
                                                        int8_t* __restrict__ shadow8, uint8_t* __restrict__ flag8,
                                                        I8Group* __restrict__ grp, int32_t* __restrict__ irr8_rows,
                                                        int* __restrict__ irr8_count, int absolute) {
    __shared__ float s_peak[kI8GroupRows];
    __shared__ float s_err[kI8GroupRows];
    __shared__ float s_step;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t g = g0 + blockIdx.x;
    const int64_t row_g = g * kI8GroupRows;
    const float peak_limit = i8_row_peak_limit(d);
    for (int j = 0; j < 8; ++j) {
        const int lr = wave * 8 + j;
        const int64_t i = row_g + lr;
        float peak = -1.0f;  // < 0: not a tight row
        if (i < n_total) {
            const float n2 = nrm2[i];
            if (norm_is_regular(n2)) {
                // absolute (inner product): the row itself is quantised; "outlier component" stays relative to its own norm
                const float rc = absolute ? 1.0f : 1.0f / sqrtf(n2);
                const float lim = absolute ? peak_limit * sqrtf(n2) : peak_limit;
                const float* r = rows + i * (int64_t)d;
                float mx = 0.0f;
                for (int k = lane; k < d; k += kWave) mx = fmaxf(mx, fabsf(r[k] * rc));
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
                if (mx <= lim) peak = mx;  // (NaN components compare false: loose)
            }
        }
        if (lane == 0) s_peak[lr] = peak;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float mx = 0.0f;
        for (int j = 0; j < kI8GroupRows; ++j) mx = fmaxf(mx, s_peak[j]);
        s_step = mx * (1.0f / 127.0f) * 1.0000005f;  // rounded up: peak / step <= 127 exactly
    }
    __syncthreads();
    const float step = s_step;
    const float inv_step = step > 0.0f ? 1.0f / step : 0.0f;
    for (int j = 0; j < 8; ++j) {
        const int lr = wave * 8 + j;
        const int64_t i = row_g + lr;
        if (i >= n_total) {  // (rows past the end of the index: their shadow rows stay as allocated, all zero)
            if (lane == 0) s_err[lr] = 0.0f;
            continue;
        }
        const bool tight = s_peak[lr] >= 0.0f && step > 0.0f;
        const float rc = tight ? (absolute ? 1.0f : 1.0f / sqrtf(nrm2[i])) : 0.0f;
        const float* r = rows + i * (int64_t)d;
        int8_t* s = shadow8 + i * (int64_t)dpad8;
        float e2 = 0.0f;
        for (int k = lane; k < dpad8; k += kWave) {
            float qv = 0.0f;
            if (k < d && tight) {
                const float ch = r[k] * rc;
                qv = fminf(fmaxf(rintf(ch * inv_step), -127.0f), 127.0f);
                const float e = __builtin_fmaf(-step, qv, ch);
                e2 = __builtin_fmaf(e, e, e2);
            }
            s[k] = (int8_t)qv;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e2 += __shfl_xor(e2, o);
        if (lane == 0) {
            s_err[lr] = tight ? sqrtf(e2) * 1.001f : 0.0f;
            flag8[i] = tight ? 0 : 1;
            if (!tight && i >= first_new) {
                const int slot = atomicAdd(irr8_count, 1);
                if (slot < kIrrCap) irr8_rows[slot] = (int32_t)i;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float mx = 0.0f;
        for (int j = 0; j < kI8GroupRows; ++j) mx = fmaxf(mx, s_err[j]);
        grp[g].step = step;
        grp[g].err = mx;
    }
}

// grid: Bpad blocks of 64 threads.  i8 != 0: also prepare the int8 screen (qhat8, sc, kq, E from the
// measured query residual); otherwise the bf16 screen: E from the measured query residual and bf16_ec, the largest
// residual norm of the stored rows.
// block 0 also re-arms the block's OR-ed status word and both hand-over counters of k_prune (two launches less per block)
__global__ __launch_bounds__(64) void k_prep_queries(const float* __restrict__ q, int B, int d, int dpad, int metric,
                                                      QueryState st, int dpad8, int i8, float bf16_ec, int* status_or,
                                                      int* prune_skip, int cnt0, float cscale) {
    extern __shared__ float qs[];  // [d]
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    if (b == 0 && lane == 0) {
        if (status_or) *status_or = 0;
        if (prune_skip) prune_skip[0] = prune_skip[1] = 0;
    }
    uint16_t* qh = st.qhat + (int64_t)b * dpad;
    if (b >= B) {  // padding rows of the last 128-query tile
        for (int k = lane; k < dpad; k += kWave) qh[k] = 0;
        if (lane == 0) {
            st.qn[b] = 0.0f;
            st.thr[b] = __builtin_inff();
            st.cnt[b] = 0;
            st.carry[b] = 0;
            st.best_n[b] = 0;
            st.thr_key[b] = 0;
            st.thr_row[b] = -1;
            st.status[b] = 0;
            st.E[b] = bf16_screen_bound(0.00390625f, bf16_ec, d, cscale);
            st.E16[b] = st.E[b];
            st.sc[b] = 0.0f;
            st.kq[b] = 1.0f;
        }
        if (i8)
            for (int k = lane; k < dpad8; k += kWave) st.qhat8[(int64_t)b * dpad8 + k] = 0;
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
    float eq2 = 0.0f;
    for (int k = lane; k < dpad; k += kWave) {
        uint16_t v = 0;
        if (k < d && regular) {
            const float qh_f = qs[k] * rq;
            v = f32_to_bf16_rn(qh_f);
            const float e = qh_f - bf16_bits_to_f32(v);
            eq2 = __builtin_fmaf(e, e, eq2);
        }
        qh[k] = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) eq2 += __shfl_xor(eq2, o);
    float E = bf16_screen_bound(sqrtf(eq2) * 1.001f, bf16_ec, d, cscale), sc = 1.0f, kq = 1.0f;
    const float E16 = E;
    if (i8) {
        // per-query step S_q = max|q_hat| / 127, residual norm measured like the corpus side
        float mx = 0.0f;
        for (int k = lane; k < d; k += kWave) mx = fmaxf(mx, fabsf(qs[k] * rq));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float sq = mx / 127.0f;
        const float inv = sq > 0.0f ? 1.0f / sq : 0.0f;
        float e2 = 0.0f;
        int8_t* q8 = st.qhat8 + (int64_t)b * dpad8;
        for (int k = lane; k < dpad8; k += kWave) {
            float qv = 0.0f;
            if (k < d && regular) {
                const float qh_f = qs[k] * rq;
                qv = fminf(fmaxf(rintf(qh_f * inv), -127.0f), 127.0f);
                const float e = __builtin_fmaf(-sq, qv, qh_f);
                e2 = __builtin_fmaf(e, e, e2);
            }
            q8[k] = (int8_t)qv;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e2 += __shfl_xor(e2, o);
        const float e_q = sqrtf(e2) * 1.001f;
        E = i8_query_bound(e_q, d) * cscale;
        kq = i8_pair_factor(e_q);
        sc = sq;
    }
    if (lane == 0) {
        st.qn[b] = acc;
        st.E[b] = E;
        st.E16[b] = E16;
        st.sc[b] = sc;
        st.kq[b] = kq;
        // cosine screen cannot rank an irregular query: park it (never emits) and flag it for the scan path
        // (metric 2 = test hook: every query screens with thresholds at -inf)
        st.thr[b] = (regular || metric == 2) ? -__builtin_inff() : __builtin_inff();
        st.cnt[b] = cnt0;  // (> 0: the starter writes one candidate per slab at fixed slots)
        st.carry[b] = 0;
        st.best_n[b] = 0;
        st.thr_key[b] = kKeyNaN;
        st.thr_row[b] = 0x7FFFFFFF;
        st.status[b] = regular ? 0 : kStIrregular;
    }
}

}  // namespace mi355
