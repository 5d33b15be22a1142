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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <xmmintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_QB 64 /* queries scored together per corpus row (vector lanes) */

/*
 * Default worker count: what OpenMP offers, capped by the container's CPU bandwidth quota (cgroup v2 cpu.max,
 * "quota period"): a box that shows 256 CPUs but grants 16 CPUs' worth of time runs 256 spinning threads far slower
 * than 16, and the throttling makes every barrier a scheduling lottery.
 */
static int orc_default_threads(void) {
    int nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
#endif
    FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char quota[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && period > 0 && strcmp(quota, "max") != 0) {
            long q = atol(quota);
            int cap = (int)((q + period - 1) / period);
            if (cap >= 1 && cap < nt) nt = cap;
        }
        fclose(f);
    }
    return nt;
}

/*
 * The oracle's arithmetic is IEEE round-to-nearest with denormals: make that a property of this file, not of
 * whatever floating-point environment a worker thread inherited (pthreads copy the creator's MXCSR; a thread created
 * while some other library had flush-to-zero or a directed rounding mode set keeps it for life, and e.g. float
 * overflow then yields FLT_MAX instead of +inf).  Called at the top of every parallel region and entry point;
 * orc_fpenv_fix_count() reports how often a non-default environment was found.
 */
static int orc_fpenv_fixes = 0;
static inline void orc_default_fpenv(void) {
    const unsigned csr = _mm_getcsr();
    if ((csr & 0xFFC0u) != 0x1F80u) { /* bits 6..15: DAZ, exception masks, rounding control, FTZ */
        __atomic_fetch_add(&orc_fpenv_fixes, 1, __ATOMIC_RELAXED);
        _mm_setcsr(0x1F80u | (csr & 0x3Fu));
    }
}
int orc_fpenv_fix_count(void) { return __atomic_load_n(&orc_fpenv_fixes, __ATOMIC_RELAXED); }

/*
 * Pin the calling OpenMP worker to one allowed CPU.  Some sandboxes/VMs keep all
 * unbound libgomp workers on a single core (measured: 8 threads = 1x without
 * binding, 7.5x with), and OMP_PROC_BIND cannot be relied on because another
 * library (torch) may have initialised libgomp before this one is loaded.
 */
static void orc_pin_thread(void) {
    orc_default_fpenv();
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
#pragma omp parallel num_threads(orc_default_threads())
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
    nthreads = threads > 0 ? threads : orc_default_threads();
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
    nthreads = threads > 0 ? threads : orc_default_threads();
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

/*
 * Self-check of the OpenMP runtime binding: inside a parallel region every thread must see a distinct
 * omp_get_thread_num() below omp_get_num_threads().  (If a process holds two OpenMP runtimes and this library's
 * GOMP_parallel and omp_get_* resolve to different ones, every thread reports tid 0 of 1 -- all of them would then
 * work on the same slice and the same partial lists.)  Returns threads that entered the region; *out_nt = the team
 * size they saw, *out_distinct = number of distinct thread numbers.
 */
int orc_debug_team(int* out_nt, int* out_distinct) {
    int entered = 0, nt_seen = 0;
    unsigned char seen[4096];
    memset(seen, 0, sizeof(seen));
#pragma omp parallel num_threads(orc_default_threads())
    {
        int tid = 0, nt = 1;
#ifdef _OPENMP
        tid = omp_get_thread_num();
        nt = omp_get_num_threads();
#endif
        __atomic_fetch_add(&entered, 1, __ATOMIC_RELAXED);
        __atomic_store_n(&nt_seen, nt, __ATOMIC_RELAXED);
        if (tid >= 0 && tid < 4096) __atomic_store_n(&seen[tid], 1, __ATOMIC_RELAXED);
    }
    int distinct = 0;
    for (int i = 0; i < 4096; ++i) distinct += seen[i];
    *out_nt = nt_seen;
    *out_distinct = distinct;
    return entered;
}

int orc_num_threads(void) { return orc_default_threads(); }

/*
 * Independent single-threaded check of a top-k result (any producer): every returned (query,row) pair must carry the
 * exact distance of that pair, rows must be distinct and the list must follow the total order.  Does not prove
 * completeness.  Returns the number of offending entries.
 */
int64_t orc_verify_topk(const float* C, int64_t n, int d, const float* Q, int B, int k, int metric, const double* dist,
                        const int64_t* rows) {
    orc_default_fpenv();
    int64_t bad = 0;
    for (int b = 0; b < B; ++b) {
        const float* q = Q + (int64_t)b * d;
        const float nq = orc_dot(q, q, d);
        for (int s = 0; s < k; ++s) {
            const int64_t r = rows[(int64_t)b * k + s];
            const double dv = dist[(int64_t)b * k + s];
            if (r < 0) continue; /* padding */
            if (r >= n) {
                ++bad;
                continue;
            }
            const float* c = C + r * (int64_t)d;
            const float dot = orc_dot(c, q, d);
            const double want = metric == 0 ? orc_cosine_distance_from(dot, nq, orc_dot(c, c, d)) : (double)dot * -1.0;
            if (!((isnan(want) && isnan(dv)) || want == dv)) ++bad;
            if (s > 0 && rows[(int64_t)b * k + s - 1] >= 0 &&
                !orc_before(dist[(int64_t)b * k + s - 1], rows[(int64_t)b * k + s - 1], dv, r))
                ++bad;
        }
    }
    return bad;
}
