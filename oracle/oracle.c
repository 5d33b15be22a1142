/*
 * oracle.c -- CPU restatement of the score-and-select arithmetic behind
 * AutoRAG-Research's VectorSearch hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (autorag-research_amd/)
 * may link, import or call this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / the timed CPU
 * baseline -- never as the thing shipped.
 *
 * What it restates (reference file:line, all under /root/reference):
 *   - autorag_research/orm/repository/base.py:409-415
 *       SELECT id, embedding <=> q AS distance ... ORDER BY distance LIMIT k
 *     `<=>` is pgvector's cosine_distance.  pgvector is NOT vendored in the
 *     reference (docker image tensorchord/vchord-suite:pg18-latest, unpinned;
 *     python client pin pgvector==0.4.2).  Its published algorithm (vector.c,
 *     VectorCosineSimilarity / cosine_distance) is restated here:
 *         float dot=0,na=0,nb=0; for i: dot+=a[i]*b[i]; na+=a[i]*a[i]; nb+=b[i]*b[i];
 *         double sim = (double)dot / sqrt((double)na * (double)nb);
 *         clamp sim to [-1,1];  return 1.0 - sim;        (float8; NaN if a norm is 0)
 *   - autorag_research/orm/repository/base.py:518-524, 562-568
 *       embeddings @# ARRAY[q_1..q_n]   (VectorChord MaxSim, also not vendored):
 *         f32 acc=0; for q in queries { m=+inf; for d in docvecs { m=min(m, -dot(d,q)) }; acc+=m }
 *   - `ORDER BY distance LIMIT k` with no ANN index = exact top-k.
 *
 * PARITY STATUS: "parity unpinned" at the SQL-operator boundary -- the reference's
 * own tests hold no numeric known-answer for `<=>` / `@#` (property tests on
 * unseeded vectors only, tests/autorag_research/orm/repository/
 * test_base_vector_repository.py:121-143,192-218,240-263,505-536).  What IS pinned
 * (tests/test_oracle_golden.py): the reference's in-process restatements of the
 * same math (gqr_hybrid._cosine_scores/_maxsim_scores, heaven._score_candidates),
 * its service-level score conversions and its nDCG known answers, via fixtures
 * generated from the imported reference by tests/golden/make_golden.py.
 *
 * Two things the reference leaves undefined are FIXED here so that a GPU
 * implementation can be compared bit-for-bit:
 *   (1) fp32 summation order.  pgvector's loop is auto-vectorised by whatever
 *       compiler built it (order is build-dependent).  The oracle's canonical
 *       order is the plain k-ascending chain with fused multiply-add,
 *           acc = fmaf(a[k], b[k], acc),  k = 0..d-1,
 *       i.e. the literal pgvector loop under FMA contraction.  orc_*_seq()
 *       variants (separate mul + add, no FMA) are kept for cross-checks.
 *   (2) tie order.  Postgres' top-N heapsort leaves ties unspecified; NaN sorts
 *       last in ASC.  Total order used everywhere: (distance asc, NaN last,
 *       row index asc).
 */
#define _GNU_SOURCE
#include <math.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_QB 64 /* queries scored together per corpus row (vector lanes) */

/*
 * Pin the calling OpenMP worker to one allowed CPU.  Some sandboxes/VMs keep all
 * unbound libgomp workers on a single core (measured: 8 threads = 1x without
 * binding, 7.5x with), and OMP_PROC_BIND cannot be relied on because another
 * library (torch) may have initialised libgomp before this one is loaded.
 */
static void orc_pin_thread(void) {
#ifdef _OPENMP
    static __thread int pinned = 0;
    if (pinned || omp_get_num_threads() <= 1) return;
    pinned = 1;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    int ncpu = CPU_COUNT(&allowed);
    if (ncpu <= 1 || omp_get_thread_num() == 0) return; /* leave the caller's own thread alone */
    int want = omp_get_thread_num() % ncpu, seen = 0;
    for (int c = 0; c < CPU_SETSIZE; ++c) {
        if (!CPU_ISSET(c, &allowed)) continue;
        if (seen++ == want) {
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(c, &one);
            sched_setaffinity(0, sizeof(one), &one);
            return;
        }
    }
#endif
}

/* ---- canonical fp32 chains ------------------------------------------------ */

float orc_dot(const float* a, const float* b, int d) {
    float acc = 0.0f;
    for (int k = 0; k < d; ++k) acc = fmaf(a[k], b[k], acc);
    return acc;
}

/* literal pgvector loop, no contraction (cross-check only) */
float orc_dot_seq(const float* a, const float* b, int d) {
    volatile float acc = 0.0f;
    for (int k = 0; k < d; ++k) {
        volatile float p = a[k] * b[k];
        acc = acc + p;
    }
    return acc;
}

/* 8 rows at a time: 8 independent k-ascending chains (ILP), each bit-identical to orc_dot(row,row) */
static void orc_nrm2_rows(const float* rows, int64_t lo, int64_t hi, int d, float* out) {
    int64_t i = lo;
    for (; i + 8 <= hi; i += 8) {
        const float* r = rows + i * (int64_t)d;
        float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < d; ++k)
            for (int u = 0; u < 8; ++u) {
                float v = r[(int64_t)u * d + k];
                a[u] = fmaf(v, v, a[u]);
            }
        for (int u = 0; u < 8; ++u) out[i + u] = a[u];
    }
    for (; i < hi; ++i) out[i] = orc_dot(rows + i * (int64_t)d, rows + i * (int64_t)d, d);
}

void orc_row_nrm2(const float* rows, int64_t n, int d, float* out) {
#pragma omp parallel
    {
        orc_pin_thread();
        int tid = 0, nt = 1;
#ifdef _OPENMP
        tid = omp_get_thread_num();
        nt = omp_get_num_threads();
#endif
        orc_nrm2_rows(rows, n * tid / nt, n * (tid + 1) / nt, d, out);
    }
}

/* distance from the three fp32 accumulators, pgvector cosine_distance semantics */
double orc_cosine_distance_from(float dot, float nq, float nc) {
    double sim = (double)dot / sqrt((double)nq * (double)nc);
    if (sim > 1.0) sim = 1.0;
    else if (sim < -1.0) sim = -1.0;
    return 1.0 - sim; /* NaN propagates */
}

double orc_cosine_distance(const float* q, const float* c, int d) {
    return orc_cosine_distance_from(orc_dot(c, q, d), orc_dot(q, q, d), orc_dot(c, c, d));
}

double orc_cosine_distance_seq(const float* q, const float* c, int d) {
    return orc_cosine_distance_from(orc_dot_seq(c, q, d), orc_dot_seq(q, q, d), orc_dot_seq(c, c, d));
}

/* ---- total order + bounded top-k list -------------------------------------- */

/* returns 1 if (d1,r1) sorts strictly before (d2,r2): distance asc, NaN last, row asc */
static inline int orc_before(double d1, int64_t r1, double d2, int64_t r2) {
    int n1 = isnan(d1), n2 = isnan(d2);
    if (n1 != n2) return n2; /* non-NaN first */
    if (!n1) {
        if (d1 < d2) return 1;
        if (d1 > d2) return 0;
    }
    return r1 < r2;
}

typedef struct {
    double* dist;
    int64_t* row;
    int k;
    int len;
} orc_topk;

/* sorted insertion (k is small); keeps best-first */
static inline void orc_topk_push(orc_topk* t, double d, int64_t r) {
    if (t->len == t->k && !orc_before(d, r, t->dist[t->k - 1], t->row[t->k - 1])) return;
    int pos = t->len < t->k ? t->len : t->k - 1;
    while (pos > 0 && orc_before(d, r, t->dist[pos - 1], t->row[pos - 1])) {
        t->dist[pos] = t->dist[pos - 1];
        t->row[pos] = t->row[pos - 1];
        --pos;
    }
    t->dist[pos] = d;
    t->row[pos] = r;
    if (t->len < t->k) t->len++;
}

/* ---- brute-force cosine / inner-product top-k ------------------------------ */

/*
 * metric 0: cosine distance (pgvector <=>), metric 1: negative inner product
 * (pgvector <#>: (double)dot * -1).  Outputs are [B,k]; unused tail slots hold
 * row -1 / distance NaN.  Returns 0, or <0 on bad arguments.
 */
int orc_topk_search(const float* C, int64_t n, int d, const float* Q, int B, int k, int metric, double* out_dist,
                    int64_t* out_rows, int threads) {
    if (d <= 0 || B < 0 || k <= 0 || n < 0) return -1;
    for (int64_t i = 0; i < (int64_t)B * k; ++i) {
        out_dist[i] = NAN;
        out_rows[i] = -1;
    }
    if (n == 0 || B == 0) return 0;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = threads > 0 ? threads : omp_get_max_threads();
#endif
    (void)threads;
    float* cn = (float*)malloc(sizeof(float) * (size_t)n);
    float* qn = (float*)malloc(sizeof(float) * (size_t)B);
    if (!cn || !qn) return -2;
#pragma omp parallel num_threads(nthreads)
    {
        orc_pin_thread();
        int tid = 0, nt = 1;
#ifdef _OPENMP
        tid = omp_get_thread_num();
        nt = omp_get_num_threads();
#endif
        orc_nrm2_rows(C, n * tid / nt, n * (tid + 1) / nt, d, cn);
    }
    for (int b = 0; b < B; ++b) qn[b] = orc_dot(Q + (int64_t)b * d, Q + (int64_t)b * d, d);

    /* per-thread partial top-k lists, merged at the end */
    double* pd = (double*)malloc(sizeof(double) * (size_t)nthreads * B * k);
    int64_t* pr = (int64_t*)malloc(sizeof(int64_t) * (size_t)nthreads * B * k);
    int* plen = (int*)calloc((size_t)nthreads * B, sizeof(int));
    if (!pd || !pr || !plen) return -2;

    for (int b0 = 0; b0 < B; b0 += ORC_QB) {
        int nb = B - b0 < ORC_QB ? B - b0 : ORC_QB;
        /* transposed query block qt[k][lane] so the inner loop vectorises over lanes;
           every lane is still its own k-ascending fmaf chain */
        float* qt = (float*)calloc((size_t)d * ORC_QB, sizeof(float));
        for (int j = 0; j < nb; ++j)
            for (int kk = 0; kk < d; ++kk) qt[(size_t)kk * ORC_QB + j] = Q[(int64_t)(b0 + j) * d + kk];
#pragma omp parallel num_threads(nthreads)
        {
            int tid = 0, nt = 1;
            orc_pin_thread();
#ifdef _OPENMP
            tid = omp_get_thread_num();
            nt = omp_get_num_threads();
#endif
            int64_t lo = n * tid / nt, hi = n * (tid + 1) / nt;
            orc_topk tk[ORC_QB];
            for (int j = 0; j < nb; ++j) {
                tk[j].dist = pd + ((size_t)tid * B + b0 + j) * k;
                tk[j].row = pr + ((size_t)tid * B + b0 + j) * k;
                tk[j].k = k;
                tk[j].len = 0;
            }
            float skip_below[ORC_QB], rqn[ORC_QB];
            for (int j = 0; j < ORC_QB; ++j) {
                skip_below[j] = -INFINITY;
                rqn[j] = (metric == 0 && j < nb) ? 1.0f / sqrtf(qn[b0 + j]) : 1.0f;
            }
            for (int64_t i = lo; i < hi; ++i) {
                const float* c = C + i * (int64_t)d;
                float acc[ORC_QB];
                for (int j = 0; j < ORC_QB; ++j) acc[j] = 0.0f;
                for (int kk = 0; kk < d; ++kk) {
                    const float cv = c[kk];
                    const float* qrow = qt + (size_t)kk * ORC_QB;
#pragma omp simd
                    for (int j = 0; j < ORC_QB; ++j) acc[j] = fmaf(cv, qrow[j], acc[j]);
                }
                /* cheap fp32 pre-screen (pure speed-up, never changes the result): a pair whose
                   approximate similarity is more than 1e-4 below the current k-th best cannot
                   enter the list; NaN / inf approximations fall through to the exact path */
                const float rci = metric == 0 ? 1.0f / sqrtf(cn[i]) : 1.0f;
                for (int j = 0; j < nb; ++j) {
                    if (acc[j] * rci * rqn[j] < skip_below[j]) continue;
                    double dist = metric == 0 ? orc_cosine_distance_from(acc[j], qn[b0 + j], cn[i])
                                              : (double)acc[j] * -1.0;
                    orc_topk_push(&tk[j], dist, i);
                    if (tk[j].len == k && !isnan(tk[j].dist[k - 1]))
                        skip_below[j] = metric == 0 ? (float)(1.0 - tk[j].dist[k - 1]) - 1e-4f
                                                    : nextafterf((float)(-tk[j].dist[k - 1]), -INFINITY);
                }
            }
            for (int j = 0; j < nb; ++j) plen[(size_t)tid * B + b0 + j] = tk[j].len;
        }
        free(qt);
    }
    for (int b = 0; b < B; ++b) {
        orc_topk fin = {out_dist + (size_t)b * k, out_rows + (size_t)b * k, k, 0};
        for (int t = 0; t < nthreads; ++t) {
            size_t base = ((size_t)t * B + b) * k;
            for (int s = 0; s < plen[(size_t)t * B + b]; ++s) orc_topk_push(&fin, pd[base + s], pr[base + s]);
        }
    }
    free(cn);
    free(qn);
    free(pd);
    free(pr);
    free(plen);
    return 0;
}

/* ---- MaxSim (VectorChord @#) ------------------------------------------------ */

/* distance of one document (T x d tokens) against one multi-vector query (nq x d) */
float orc_maxsim_distance(const float* doc, int64_t T, const float* q, int nq, int d) {
    float acc = 0.0f;
    for (int i = 0; i < nq; ++i) {
        float m = INFINITY;
        for (int64_t j = 0; j < T; ++j) {
            float v = -orc_dot(doc + j * (int64_t)d, q + (int64_t)i * d, d);
            m = fminf(m, v);
        }
        acc = acc + m;
    }
    return acc;
}

/*
 * tok: [sum_T, d] ragged doc token matrix, offsets: [n_docs+1].
 * qtok: [sum_nq, d], q_off: [B+1].  out: [B,k] f32 distances (= -sum max dot) and
 * doc indices; same total order.  Docs with zero tokens are skipped (NULL/empty
 * arrays never satisfy `embeddings IS NOT NULL` with a MaxSim distance).
 */
int orc_maxsim_topk(const float* tok, const int64_t* offsets, int64_t n_docs, int d, const float* qtok,
                    const int32_t* q_off, int B, int k, float* out_dist, int64_t* out_rows, int threads) {
    if (d <= 0 || B < 0 || k <= 0 || n_docs < 0) return -1;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = threads > 0 ? threads : omp_get_max_threads();
#endif
    (void)threads;
    /* per-thread partial lists over a slice of the documents, merged per query (same total order) */
    double* pd = (double*)malloc(sizeof(double) * (size_t)nthreads * k);
    int64_t* pr = (int64_t*)malloc(sizeof(int64_t) * (size_t)nthreads * k);
    int* plen = (int*)malloc(sizeof(int) * (size_t)nthreads);
    double* td = (double*)malloc(sizeof(double) * (size_t)k);
    if (!pd || !pr || !plen || !td) return -2;
    for (int b = 0; b < B; ++b) {
        for (int s = 0; s < k; ++s) {
            out_rows[(size_t)b * k + s] = -1;
            out_dist[(size_t)b * k + s] = NAN;
        }
        const float* q = qtok + (int64_t)q_off[b] * d;
        const int nq = q_off[b + 1] - q_off[b];
        for (int t = 0; t < nthreads; ++t) plen[t] = 0;
#pragma omp parallel num_threads(nthreads)
        {
            orc_pin_thread();
            int tid = 0, nt = 1;
#ifdef _OPENMP
            tid = omp_get_thread_num();
            nt = omp_get_num_threads();
#endif
            orc_topk part = {pd + (size_t)tid * k, pr + (size_t)tid * k, k, 0};
            for (int64_t doc = n_docs * tid / nt; doc < n_docs * (tid + 1) / nt; ++doc) {
                const int64_t T = offsets[doc + 1] - offsets[doc];
                if (T <= 0) continue;
                const float dist = orc_maxsim_distance(tok + offsets[doc] * (int64_t)d, T, q, nq, d);
                orc_topk_push(&part, (double)dist, doc);
            }
            plen[tid] = part.len;
        }
        orc_topk fin = {td, out_rows + (size_t)b * k, k, 0};
        for (int t = 0; t < nthreads; ++t)
            for (int s = 0; s < plen[t]; ++s) orc_topk_push(&fin, pd[(size_t)t * k + s], pr[(size_t)t * k + s]);
        for (int s = 0; s < fin.len; ++s) out_dist[(size_t)b * k + s] = (float)td[s];
    }
    free(pd);
    free(pr);
    free(plen);
    free(td);
    return 0;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
