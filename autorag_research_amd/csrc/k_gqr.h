// k_gqr.h -- Guided Query Refinement (GQR) of a candidate pool: the test-time optimisation loop of the reference's
// GQR hybrid pipeline (autorag_research/pipelines/retrieval/gqr_hybrid.py), one workgroup per query, float64.
//   k_gqr_single  _optimize_query_embedding        (:321-340)  cosine scores / gradients of `_cosine_scores` (:65-74),
//                                                              `_cosine_gradients` (:77-92) over fp32 corpus rows
//   k_gqr_multi   _optimize_query_multi_embedding  (:342-362)  `_maxsim_scores` (:95-110) and the argmax subgradient
//                                                              `_maxsim_gradients` (:113-127) over stored token rows
//   k_gqr_scores  _optimize_in_score_space         (:306-319)  the same consensus step on the score vector itself
// All three share the step  p = softmax(score / T),  target = (1-a) p + a p_comp,  g = (p - target) / T  (:43-57).
// The reference runs this in numpy float64 on the float32 vectors it fetched; here the stored fp32 values are widened
// to double on load, every sum is a float64 sum (different association than numpy's BLAS: agreement ~1e-15 relative).
// Pools are tens to hundreds of rows per query, so a query is one workgroup and a block of queries fills the chip.
#pragma once
#include "dev_common.h"

namespace mi355 {

constexpr int kGqrThreads = 256;
constexpr int kGqrPoolMax = 2048;  // candidates per query
constexpr double kGqrEps = 1e-8;   // _EPSILON (:36)

struct GqrParams {
    int n_steps;
    double lr, temperature, alpha;
};

__device__ __forceinline__ double gqr_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double gqr_wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
// block-wide (4 waves) reductions; red = 4 doubles of LDS; every thread gets the result
__device__ __forceinline__ double gqr_block_sum(double v, double* red) {
    v = gqr_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double gqr_block_max(double v, double* red) {
    v = gqr_wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// g[j] = (p_j - ((1-a) p_j + a comp_j)) / T  with  p = softmax(score / T)   (gqr_hybrid.py:43-57 and :313-316)
// score, comp, g: LDS arrays of n entries; g may alias neither.  Ends with a barrier.
__device__ __forceinline__ void gqr_logit_grad(const double* score, const double* comp, double* g, int n,
                                               const GqrParams& P, double* red) {
    const double T = fmax(P.temperature, kGqrEps);
    double m = -__builtin_inf();
    for (int j = threadIdx.x; j < n; j += kGqrThreads) m = fmax(m, score[j] / T);
    m = gqr_block_max(m, red);
    double s = 0.0;
    for (int j = threadIdx.x; j < n; j += kGqrThreads) {
        const double e = exp(score[j] / T - m);
        g[j] = e;
        s += e;
    }
    const double denom = gqr_block_sum(s, red);
    const bool uniform = !(fabs(denom) < __builtin_inf()) || denom <= kGqrEps;  // non-finite or vanishing: 1/n each
    for (int j = threadIdx.x; j < n; j += kGqrThreads) {
        const double p = uniform ? 1.0 / (double)n : g[j] / denom;
        const double target = (1.0 - P.alpha) * p + P.alpha * comp[j];
        g[j] = (p - target) / T;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
struct GqrSingleArgs {
    const float* rows;    // [n_rows, d] fp32 corpus
    int d;
    const double* q0;     // [B, d]
    const int32_t* cand;  // [B, P] local row, < 0 = padding (at the tail only)
    const double* comp;   // [B, P] complementary distribution
    double* out;          // [B, P] refined cosine scores (NaN at padding)
    int P;
    GqrParams prm;
};

// dynamic LDS (doubles): q[d] | cn[P] | cs[P] | comp[P] | g[P] | red[4]
__host__ __device__ inline size_t gqr_single_lds(int d, int P) { return ((size_t)d + 4 * (size_t)P + 4) * sizeof(double); }

__global__ __launch_bounds__(kGqrThreads) void k_gqr_single(GqrSingleArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* q = (double*)smem;
    double* cn = q + a.d;
    double* cs = cn + a.P;
    double* comp = cs + a.P;
    double* g = comp + a.P;
    double* red = g + a.P;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t* cand = a.cand + (int64_t)b * a.P;
    int n = 0;
    for (int j = tid; j < a.P; j += kGqrThreads) n += cand[j] >= 0;
    n = (int)(gqr_block_sum((double)n, red) + 0.5);
    for (int k = tid; k < a.d; k += kGqrThreads) q[k] = a.q0[(int64_t)b * a.d + k];
    for (int j = tid; j < n; j += kGqrThreads) comp[j] = a.comp[(int64_t)b * a.P + j];
    // candidate norms, once: max(|c|, eps)   (:71-73)
    for (int j = wave; j < n; j += kGqrThreads / 64) {
        const float* r = a.rows + (int64_t)cand[j] * a.d;
        double s = 0.0;
        for (int k = lane; k < a.d; k += 64) s = fma((double)r[k], (double)r[k], s);
        s = gqr_wave_sum(s);
        if (lane == 0) cn[j] = fmax(sqrt(s), kGqrEps);
    }
    __syncthreads();
    for (int step = 0; step <= a.prm.n_steps; ++step) {
        double s = 0.0;
        for (int k = tid; k < a.d; k += kGqrThreads) s = fma(q[k], q[k], s);
        const double qn = sqrt(gqr_block_sum(s, red));
        const bool dead = qn <= kGqrEps;  // zero query: scores and gradients are all zero (:68-69, :84-85)
        for (int j = wave; j < n; j += kGqrThreads / 64) {
            const float* r = a.rows + (int64_t)cand[j] * a.d;
            double dt = 0.0;
            for (int k = lane; k < a.d; k += 64) dt = fma((double)r[k], q[k], dt);
            dt = gqr_wave_sum(dt);
            if (lane == 0) cs[j] = dead ? 0.0 : dt / (cn[j] * qn);
        }
        __syncthreads();
        if (step == a.prm.n_steps) break;  // the last pass only scores the refined query (:340)
        gqr_logit_grad(cs, comp, g, n, a.prm, red);
        if (!dead) {
            const double qn2 = qn * qn;
            for (int k = tid; k < a.d; k += kGqrThreads) {
                const double qk = q[k];
                double acc = 0.0;
                for (int j = 0; j < n; ++j) {
                    const double left = (double)a.rows[(int64_t)cand[j] * a.d + k] / (cn[j] * qn);
                    const double right = (cs[j] * qk) / qn2;
                    acc += g[j] * (left - right);
                }
                q[k] = qk - a.prm.lr * acc;
            }
        }
        __syncthreads();
    }
    for (int j = tid; j < a.P; j += kGqrThreads) a.out[(int64_t)b * a.P + j] = j < n ? cs[j] : __builtin_nan("");
}

// ---------------------------------------------------------------------------------------------------------------
// score-space form: the logits themselves are the variables (:306-319)
struct GqrScoreArgs {
    const double* score0;  // [B, P] primary scores
    const int32_t* count;  // [B] live entries of each row
    const double* comp;    // [B, P]
    double* out;           // [B, P]
    int P;
    GqrParams prm;
};

// dynamic LDS (doubles): z[P] | comp[P] | g[P] | red[4]
__global__ __launch_bounds__(kGqrThreads) void k_gqr_scores(GqrScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* z = (double*)smem;
    double* comp = z + a.P;
    double* g = comp + a.P;
    double* red = g + a.P;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.count[b];
    for (int j = tid; j < n; j += kGqrThreads) {
        z[j] = a.score0[(int64_t)b * a.P + j];
        comp[j] = a.comp[(int64_t)b * a.P + j];
    }
    __syncthreads();
    for (int step = 0; step < a.prm.n_steps; ++step) {
        gqr_logit_grad(z, comp, g, n, a.prm, red);
        for (int j = tid; j < n; j += kGqrThreads) z[j] -= a.prm.lr * g[j];
        __syncthreads();
    }
    for (int j = tid; j < a.P; j += kGqrThreads) a.out[(int64_t)b * a.P + j] = j < n ? z[j] : __builtin_nan("");
}

// ---------------------------------------------------------------------------------------------------------------
// multi-vector form.  Token rows live in the MaxSim store: every doc owns whole 32-row blocks, the tail of its last
// block repeats the last token (so the FIRST maximum is always a real token, np.argmax's rule), columns are permuted
// inside groups of 8 and zero-padded to dpad -- the query matrix is handed over in the same column order, and since
// only scores leave the kernel the permutation never has to be undone.
constexpr int kGqrQChunk = 16;  // query vectors per MFMA tile (v_mfma_f64_16x16x4_f64: 16 query vectors x 16 tokens)

typedef double f64x4 __attribute__((ext_vector_type(4)));

struct GqrMultiArgs {
    const float* tok;        // [blocks*32, dpad]
    const int64_t* blk_off;  // [n_docs+1]
    int dpad;
    const double* q0;        // [sum n_q, dpad]
    const int32_t* q_off;    // [B+1]
    const int32_t* cand;     // [B, P] local doc, < 0 = padding (tail only)
    const double* comp;      // [B, P]
    double* out;             // [B, P]
    int32_t* arg_ws;         // [B, P, nq_pad] argmax token (store row) per (candidate, query vector)
    int P, nq_pad;           // nq_pad: largest n_q rounded up to kGqrQChunk
    GqrParams prm;
};

// Q rows sit dpad + 2 doubles apart in LDS: the 16 lanes of an MFMA operand read (ds_read_b128 of 16 different rows,
// same column) then fall into different banks instead of one
__host__ __device__ inline int gqr_multi_ld(int dpad) { return dpad + 2; }
// dynamic LDS (doubles): Q[nq_pad * ld] | sc[P] | comp[P] | g[P] | red[4]
__host__ __device__ inline size_t gqr_multi_lds(int nq_pad, int dpad, int P) {
    return ((size_t)nq_pad * gqr_multi_ld(dpad) + 3 * (size_t)P + 4) * sizeof(double);
}

// two 16-token tiles x 16 query vectors over the whole dimension; KSTEPS > 0: dpad == 16*KSTEPS, everything unrolled
// and all token loads issued before the first MFMA; KSTEPS == 0: any dpad (multiple of 8)
template <int KSTEPS>
__device__ __forceinline__ void gqr_tile_pair(const float* ta, const float* tb, const double* qrow, int kk, int dp,
                                              f64x4& acc_a, f64x4& acc_b) {
    if constexpr (KSTEPS > 0) {
        float4 xa[KSTEPS], xb[KSTEPS];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            xa[s] = *(const float4*)(ta + 16 * s + 4 * kk);
            xb[s] = *(const float4*)(tb + 16 * s + 4 * kk);
        }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const double* qp = qrow + 16 * s + 4 * kk;
            const double q0 = qp[0], q1 = qp[1], q2 = qp[2], q3 = qp[3];
            acc_a = __builtin_amdgcn_mfma_f64_16x16x4f64(q0, (double)xa[s].x, acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f64_16x16x4f64(q0, (double)xb[s].x, acc_b, 0, 0, 0);
            acc_a = __builtin_amdgcn_mfma_f64_16x16x4f64(q1, (double)xa[s].y, acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f64_16x16x4f64(q1, (double)xb[s].y, acc_b, 0, 0, 0);
            acc_a = __builtin_amdgcn_mfma_f64_16x16x4f64(q2, (double)xa[s].z, acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f64_16x16x4f64(q2, (double)xb[s].z, acc_b, 0, 0, 0);
            acc_a = __builtin_amdgcn_mfma_f64_16x16x4f64(q3, (double)xa[s].w, acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f64_16x16x4f64(q3, (double)xb[s].w, acc_b, 0, 0, 0);
        }
    } else {
        for (int K = 0; K < dp; K += 16) {
            const int k = K + 4 * kk;
            float4 xa = make_float4(0.f, 0.f, 0.f, 0.f), xb = xa;
            double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
            if (k < dp) {  // dpad is a multiple of 8: the last K-step may cover only half of the lanes
                xa = *(const float4*)(ta + k);
                xb = *(const float4*)(tb + k);
                q0 = qrow[k];
                q1 = qrow[k + 1];
                q2 = qrow[k + 2];
                q3 = qrow[k + 3];
            }
            acc_a = __builtin_amdgcn_mfma_f64_16x16x4f64(q0, (double)xa.x, acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f64_16x16x4f64(q0, (double)xb.x, acc_b, 0, 0, 0);
            acc_a = __builtin_amdgcn_mfma_f64_16x16x4f64(q1, (double)xa.y, acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f64_16x16x4f64(q1, (double)xb.y, acc_b, 0, 0, 0);
            acc_a = __builtin_amdgcn_mfma_f64_16x16x4f64(q2, (double)xa.z, acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f64_16x16x4f64(q2, (double)xb.z, acc_b, 0, 0, 0);
            acc_a = __builtin_amdgcn_mfma_f64_16x16x4f64(q3, (double)xa.w, acc_a, 0, 0, 0);
            acc_b = __builtin_amdgcn_mfma_f64_16x16x4f64(q3, (double)xb.w, acc_b, 0, 0, 0);
        }
    }
}

// Scores: S = Q D^T per candidate doc on the f64 matrix pipe, 16 query vectors x 16 tokens per accumulator tile.
// Operand maps of v_mfma_f64_16x16x4_f64 (MI355X guide): A[l&15][k = l>>4], B[k = l>>4][l&15], one f64 per lane;
// D: col = l&15, row = (l>>4) + 4*reg.  Lane l therefore loads 4 consecutive dims (k = K + 4*(l>>4) .. +3) of query
// vector i0 + (l&15) from LDS and of token t0 + (l&15) from HBM/L2 (a float4, widened to double) and feeds them to 4
// MFMAs; which k a lane holds does not matter as long as A and B agree.
__global__ __launch_bounds__(kGqrThreads) void k_gqr_multi(GqrMultiArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int dp = a.dpad, ld = gqr_multi_ld(dp);
    double* Q = (double*)smem;
    double* sc = Q + (size_t)a.nq_pad * ld;
    double* comp = sc + a.P;
    double* g = comp + a.P;
    double* red = g + a.P;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq = a.q_off[b + 1] - a.q_off[b];
    const int32_t* cand = a.cand + (int64_t)b * a.P;
    int32_t* arg = a.arg_ws + (int64_t)b * a.P * a.nq_pad;
    int n = 0;
    for (int j = tid; j < a.P; j += kGqrThreads) n += cand[j] >= 0;
    n = (int)(gqr_block_sum((double)n, red) + 0.5);
    for (int e = tid; e < a.nq_pad * dp; e += kGqrThreads) {
        const int i = e / dp, k = e - i * dp;
        Q[(size_t)i * ld + k] = i < nq ? a.q0[((int64_t)a.q_off[b] + i) * dp + k] : 0.0;
    }
    for (int j = tid; j < n; j += kGqrThreads) comp[j] = a.comp[(int64_t)b * a.P + j];
    __syncthreads();
    const double den = (double)max(nq, 1);
    const int col = lane & 15, kk = lane >> 4;
    for (int step = 0; step <= a.prm.n_steps; ++step) {
        // ---- scores: one wave per candidate doc
        for (int j = wave; j < n; j += kGqrThreads / 64) {
            const int64_t r0 = a.blk_off[cand[j]] * 32, r1 = a.blk_off[cand[j] + 1] * 32;
            double total = 0.0;
            for (int i0 = 0; i0 < nq; i0 += kGqrQChunk) {
                const double* qrow = Q + (size_t)(i0 + col) * ld;
                double best[4];
                int barg[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    best[r] = -__builtin_inf();
                    barg[r] = 0x7FFFFFFF;
                }
                for (int64_t t0 = r0; t0 < r1; t0 += 32) {  // docs own whole 32-row blocks: two 16-token tiles at a time
                    const float* ta = a.tok + (t0 + col) * dp;
                    const float* tb = ta + 16 * (int64_t)dp;
                    f64x4 acc_a = {0.0, 0.0, 0.0, 0.0}, acc_b = {0.0, 0.0, 0.0, 0.0};
                    if (dp == 128) gqr_tile_pair<8>(ta, tb, qrow, kk, dp, acc_a, acc_b);  // ColBERT / ColPali: unrolled
                    else gqr_tile_pair<0>(ta, tb, qrow, kk, dp, acc_a, acc_b);
                    // acc[r] = <q_{i0 + kk + 4r}, token t0 (+16) + col>; rows ascend per lane: strict > keeps the first maximum
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (acc_a[r] > best[r]) {
                            best[r] = acc_a[r];
                            barg[r] = (int)(t0 + col);
                        }
                        if (acc_b[r] > best[r]) {
                            best[r] = acc_b[r];
                            barg[r] = (int)(t0 + 16 + col);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double v = best[r];
                    int w = barg[r];
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) {  // over the 16 token columns of this lane's row group
                        const double ov = __shfl_xor(v, o);
                        const int ow = __shfl_xor(w, o);
                        if (ov > v || (ov == v && ow < w)) {
                            v = ov;
                            w = ow;
                        }
                    }
                    const int i = i0 + kk + 4 * r;
                    if (col == 0 && i < nq) {
                        total += v;
                        arg[(int64_t)j * a.nq_pad + i] = w;
                    }
                }
            }
            total += __shfl_xor(total, 16);  // the four lanes with col == 0 hold the partial sums; the others hold 0
            total += __shfl_xor(total, 32);
            if (lane == 0) sc[j] = total / den;
        }
        __syncthreads();
        if (step == a.prm.n_steps) break;
        gqr_logit_grad(sc, comp, g, n, a.prm, red);
        // ---- Q[i] -= lr * sum_j g_j * tok[argmax_ji] / n_q     (:113-127, :357-360)
        for (int e = tid; e < nq * dp; e += kGqrThreads) {
            const int i = e / dp, k = e - i * dp;
            double acc = 0.0;
            for (int j = 0; j < n; ++j)
                acc += g[j] * ((double)a.tok[(int64_t)arg[(int64_t)j * a.nq_pad + i] * dp + k] / den);
            Q[(size_t)i * ld + k] -= a.prm.lr * acc;
        }
        __syncthreads();
    }
    for (int j = tid; j < a.P; j += kGqrThreads) a.out[(int64_t)b * a.P + j] = j < n ? sc[j] : __builtin_nan("");
}

}  // namespace mi355
