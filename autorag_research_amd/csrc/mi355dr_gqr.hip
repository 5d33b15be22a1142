// mi355dr_gqr.hip -- C ABI of the Guided Query Refinement loops (kernels: k_gqr.h).
//
// Replaces the numpy loops of the reference's GQR hybrid pipeline, which run once per query on rows pulled out of
// PostgreSQL (autorag_research/pipelines/retrieval/gqr_hybrid.py:306-362 driven by `_run_gqr`, :415-470): here the
// candidate vectors never leave HBM -- the caller names them by row id -- and a block of queries is refined in one
// launch, one workgroup per query.
#include <vector>

#include "index.h"
#include "k_gqr.h"

using namespace mi355;

namespace {

struct DevBuf {  // per-call scratch: pools are a few KB per query
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};

int check_params(mi355dr_index* idx, int B, int P, int n_steps, double lr, double temperature, double alpha) {
    if (B < 0 || P < 0) return fail(idx, MI355DR_E_INVALID, "gqr: negative sizes");
    if (P > kGqrPoolMax) return fail(idx, MI355DR_E_UNSUPPORTED, "gqr: more than 2048 candidates per query");
    // same argument checks as GQRHybridRetrievalPipeline.__init__ (gqr_hybrid.py:202-222)
    if (n_steps <= 0) return fail(idx, MI355DR_E_INVALID, "n_steps must be positive");
    if (!(lr > 0)) return fail(idx, MI355DR_E_INVALID, "learning_rate must be positive");
    if (!(temperature > 0)) return fail(idx, MI355DR_E_INVALID, "temperature must be positive");
    if (!(alpha >= 0 && alpha <= 1)) return fail(idx, MI355DR_E_INVALID, "mixture_alpha must be between 0 and 1");
    return MI355DR_OK;
}

// ids -> local rows; live entries must form a prefix of every pool row (the host lists candidates, then pads with -1)
int localise(mi355dr_index* idx, const int64_t* ids, int B, int P, int64_t n_valid, std::vector<int32_t>& out,
             std::vector<int32_t>* counts) {
    out.resize((size_t)B * P);
    if (counts) counts->assign(B, 0);
    for (int b = 0; b < B; ++b) {
        bool tail = false;
        for (int j = 0; j < P; ++j) {
            const int64_t g = ids[(int64_t)b * P + j];
            if (g < 0) {
                tail = true;
                out[(size_t)b * P + j] = -1;
                continue;
            }
            const int64_t v = g - idx->row_offset;
            if (tail) return fail(idx, MI355DR_E_INVALID, "gqr: padding (-1) must come after the candidates");
            if (v < 0 || v >= n_valid) return fail(idx, MI355DR_E_INVALID, "gqr: candidate id is not a row of this index");
            out[(size_t)b * P + j] = (int32_t)v;
            if (counts) (*counts)[b]++;
        }
    }
    return MI355DR_OK;
}

}  // namespace

extern "C" {

int mi355dr_gqr_refine(mi355dr_index* idx, const double* queries, int B, const int64_t* cand_rows, int P,
                       const double* comp_dist, int n_steps, double learning_rate, double temperature,
                       double mixture_alpha, double* out_scores) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    CHECK(check_params(idx, B, P, n_steps, learning_rate, temperature, mixture_alpha));
    if (B == 0 || P == 0) return MI355DR_OK;
    if (!queries || !cand_rows || !comp_dist || !out_scores) return fail(idx, MI355DR_E_INVALID, "gqr: null buffer");
    std::vector<int32_t> local;
    CHECK(localise(idx, cand_rows, B, P, idx->n, local, nullptr));
    const int d = idx->dim;
    const size_t lds = gqr_single_lds(d, P);
    if (lds > 160 * 1024) return fail(idx, MI355DR_E_UNSUPPORTED, "gqr: dim + pool too large for one workgroup's LDS");
    HIPCHECK(idx, hipSetDevice(idx->device));
    hipStream_t s = idx->stream;
    DevBuf q, c, cp, o;
    const size_t nq = (size_t)B * d * sizeof(double), nc = (size_t)B * P * sizeof(int32_t), np = (size_t)B * P * sizeof(double);
    HIPCHECK(idx, hipMalloc(&q.p, nq));
    HIPCHECK(idx, hipMalloc(&c.p, nc));
    HIPCHECK(idx, hipMalloc(&cp.p, np));
    HIPCHECK(idx, hipMalloc(&o.p, np));
    HIPCHECK(idx, hipMemcpyAsync(q.p, queries, nq, hipMemcpyHostToDevice, s));
    HIPCHECK(idx, hipMemcpyAsync(c.p, local.data(), nc, hipMemcpyHostToDevice, s));
    HIPCHECK(idx, hipMemcpyAsync(cp.p, comp_dist, np, hipMemcpyHostToDevice, s));
    GqrSingleArgs a{};
    a.rows = idx->rows;
    a.d = d;
    a.q0 = (const double*)q.p;
    a.cand = (const int32_t*)c.p;
    a.comp = (const double*)cp.p;
    a.out = (double*)o.p;
    a.P = P;
    a.prm = GqrParams{n_steps, learning_rate, temperature, mixture_alpha};
    HIPCHECK(idx, hipFuncSetAttribute((const void*)k_gqr_single, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_gqr_single, dim3((unsigned)B), dim3(kGqrThreads), lds, s, a);
    HIPCHECK(idx, hipGetLastError());
    HIPCHECK(idx, hipMemcpyAsync(out_scores, o.p, np, hipMemcpyDeviceToHost, s));
    HIPCHECK(idx, hipStreamSynchronize(s));
    return MI355DR_OK;
}

int mi355dr_gqr_refine_scores(mi355dr_index* idx, const double* primary_scores, const int32_t* counts, int B, int P,
                              const double* comp_dist, int n_steps, double learning_rate, double temperature,
                              double mixture_alpha, double* out_scores) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    CHECK(check_params(idx, B, P, n_steps, learning_rate, temperature, mixture_alpha));
    if (B == 0 || P == 0) return MI355DR_OK;
    if (!primary_scores || !counts || !comp_dist || !out_scores) return fail(idx, MI355DR_E_INVALID, "gqr: null buffer");
    for (int b = 0; b < B; ++b)
        if (counts[b] < 0 || counts[b] > P) return fail(idx, MI355DR_E_INVALID, "gqr: counts[b] must be in [0, P]");
    HIPCHECK(idx, hipSetDevice(idx->device));
    hipStream_t s = idx->stream;
    DevBuf z, n, cp, o;
    const size_t np = (size_t)B * P * sizeof(double);
    HIPCHECK(idx, hipMalloc(&z.p, np));
    HIPCHECK(idx, hipMalloc(&n.p, (size_t)B * sizeof(int32_t)));
    HIPCHECK(idx, hipMalloc(&cp.p, np));
    HIPCHECK(idx, hipMalloc(&o.p, np));
    HIPCHECK(idx, hipMemcpyAsync(z.p, primary_scores, np, hipMemcpyHostToDevice, s));
    HIPCHECK(idx, hipMemcpyAsync(n.p, counts, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHECK(idx, hipMemcpyAsync(cp.p, comp_dist, np, hipMemcpyHostToDevice, s));
    GqrScoreArgs a{};
    a.score0 = (const double*)z.p;
    a.count = (const int32_t*)n.p;
    a.comp = (const double*)cp.p;
    a.out = (double*)o.p;
    a.P = P;
    a.prm = GqrParams{n_steps, learning_rate, temperature, mixture_alpha};
    const size_t lds = (3 * (size_t)P + 4) * sizeof(double);
    hipLaunchKernelGGL(k_gqr_scores, dim3((unsigned)B), dim3(kGqrThreads), lds, s, a);
    HIPCHECK(idx, hipGetLastError());
    HIPCHECK(idx, hipMemcpyAsync(out_scores, o.p, np, hipMemcpyDeviceToHost, s));
    HIPCHECK(idx, hipStreamSynchronize(s));
    return MI355DR_OK;
}

int mi355dr_gqr_refine_maxsim(mi355dr_index* idx, const double* qtok, const int32_t* q_offsets, int B,
                              const int64_t* doc_ids, int P, const double* comp_dist, int n_steps, double learning_rate,
                              double temperature, double mixture_alpha, double* out_scores) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    CHECK(check_params(idx, B, P, n_steps, learning_rate, temperature, mixture_alpha));
    if (B == 0 || P == 0) return MI355DR_OK;
    if (!qtok || !q_offsets || !doc_ids || !comp_dist || !out_scores) return fail(idx, MI355DR_E_INVALID, "gqr: null buffer");
    MultiVecView mv{};
    if (!multivec_view(idx, &mv)) return fail(idx, MI355DR_E_INVALID, "gqr: the index holds no multi-vector docs");
    std::vector<int32_t> local;
    CHECK(localise(idx, doc_ids, B, P, mv.n_docs, local, nullptr));
    for (size_t i = 0; i < local.size(); ++i)
        if (local[i] >= 0 && mv.blk_off_host[local[i] + 1] == mv.blk_off_host[local[i]])
            return fail(idx, MI355DR_E_INVALID, "gqr: a candidate doc has no vectors");
    int nq_max = 0;
    for (int b = 0; b < B; ++b) {
        const int nq = q_offsets[b + 1] - q_offsets[b];
        if (nq <= 0) return fail(idx, MI355DR_E_INVALID, "gqr: every query needs at least one vector");
        nq_max = std::max(nq_max, nq);
    }
    const int d = idx->dim, dp = mv.dpad;
    const int nq_pad = (nq_max + kGqrQChunk - 1) / kGqrQChunk * kGqrQChunk;
    const size_t lds = gqr_multi_lds(nq_pad, dp, P);
    if (lds > 160 * 1024) return fail(idx, MI355DR_E_UNSUPPORTED, "gqr: query matrix + pool too large for one workgroup's LDS");
    // the query matrix in the store's column order, zero-padded to dpad
    const int64_t nrow = q_offsets[B];
    std::vector<double> qimg((size_t)nrow * dp, 0.0);
    for (int64_t r = 0; r < nrow; ++r)
        for (int c = 0; c < dp; ++c) {
            const int oc = multivec_col_perm(c);
            if (oc < d) qimg[(size_t)r * dp + c] = qtok[r * d + oc];
        }
    HIPCHECK(idx, hipSetDevice(idx->device));
    hipStream_t s = idx->stream;
    DevBuf q, qo, c, cp, o, ws;
    const size_t np = (size_t)B * P * sizeof(double);
    HIPCHECK(idx, hipMalloc(&q.p, qimg.size() * sizeof(double)));
    HIPCHECK(idx, hipMalloc(&qo.p, (size_t)(B + 1) * sizeof(int32_t)));
    HIPCHECK(idx, hipMalloc(&c.p, local.size() * sizeof(int32_t)));
    HIPCHECK(idx, hipMalloc(&cp.p, np));
    HIPCHECK(idx, hipMalloc(&o.p, np));
    HIPCHECK(idx, hipMalloc(&ws.p, (size_t)B * P * nq_pad * sizeof(int32_t)));
    HIPCHECK(idx, hipMemcpyAsync(q.p, qimg.data(), qimg.size() * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHECK(idx, hipMemcpyAsync(qo.p, q_offsets, (size_t)(B + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHECK(idx, hipMemcpyAsync(c.p, local.data(), local.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHECK(idx, hipMemcpyAsync(cp.p, comp_dist, np, hipMemcpyHostToDevice, s));
    GqrMultiArgs a{};
    a.tok = mv.tok;
    a.blk_off = mv.blk_off;
    a.dpad = dp;
    a.q0 = (const double*)q.p;
    a.q_off = (const int32_t*)qo.p;
    a.cand = (const int32_t*)c.p;
    a.comp = (const double*)cp.p;
    a.out = (double*)o.p;
    a.arg_ws = (int32_t*)ws.p;
    a.P = P;
    a.nq_pad = nq_pad;
    a.prm = GqrParams{n_steps, learning_rate, temperature, mixture_alpha};
    HIPCHECK(idx, hipFuncSetAttribute((const void*)k_gqr_multi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_gqr_multi, dim3((unsigned)B), dim3(kGqrThreads), lds, s, a);
    HIPCHECK(idx, hipGetLastError());
    HIPCHECK(idx, hipMemcpyAsync(out_scores, o.p, np, hipMemcpyDeviceToHost, s));
    HIPCHECK(idx, hipStreamSynchronize(s));
    return MI355DR_OK;
}

}  // extern "C"
