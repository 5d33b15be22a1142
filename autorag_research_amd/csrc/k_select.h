// k_select.h -- exact re-score + select kernels (the part of `ORDER BY distance LIMIT k` that decides).
//   k_prune        per query: fold the candidates appended since the last prune into the kept exact
//                  top-k.  Screen candidates are first cut with a rigorous bound, then re-scored with the
//                  exact fp32 chain (bit-identical to oracle.c), converted to pgvector's double distance
//                  and sorted by the total order (distance asc, NaN last, row asc).
//   k_emit_irregular  append the (rare) rows the bf16 screen cannot see (zero / non-finite / extreme norm)
//   k_finalize     write [B,k] outputs, OR the per-query status bits into one word
//   k_merge_topk   multi-GPU: [world,B,k] gathered shard results -> global [B,k] under the same order
// Reference: replaces the top-N heapsort behind base.py:409-415 and the score conversion inputs of
// orm/service/retrieval_pipeline.py:504-524.
#pragma once
#include "dev_common.h"

namespace mi355 {

struct PruneArgs {
    const float* rows;   // [n, d] fp32 corpus
    const float* nrm2;   // [n]
    const float* q;      // [B, d] fp32 queries
    QueryState st;
    int32_t* cand_row;   // [Bpad, cap]
    float* cand_val;     // [Bpad, cap]  screen: t (NaN = "no bound, always re-score"); exact: the fp32 dot
    const int* qlist;    // optional compact list of query indices (nullptr: blockIdx.x)
    unsigned long long* stat;  // [2*kQBlockMax] per-query counters (candidates, re-scored): no shared atomics
    int cap, d, k, metric;
    int exact;           // 0: screen candidates (re-score), 1: cand_val already holds the exact dot
    const uint8_t* flag8;  // int8 screen only (else nullptr): rows outside the int8 shadow
    float cscale;          // 1 (cosine) or the largest stored row norm, inflated (inner product): scale of the absolute slacks
    const uint16_t* shadow16;  // optional [n, dpad] bf16 shadow: second screen of round-B candidates (int8 screen, cosine)
    int dpad;
    int round_a;           // rows re-scored before the cut is known (0: max(32, 2k)); always at least k, at most 64
    // hand-over from the one-wave instantiation to the general one: [0], [1] = number of queries the one-wave form left
    // (two counters, used alternately by successive prunes), [2 + p * kQBlockMax ...] = their indices.  With it the general
    // form runs on a small grid that walks the list (usually empty: ~3 us instead of a 1024-workgroup launch); nullptr or a
    // `qlist` (exact path) = one workgroup per query as before.
    int* skip_list;
    int skip_parity;
    // starter (run_screen): the candidates are a SAMPLE (one per slab of the first rows) -- re-score the best-looking ones,
    // publish the threshold their k-th best exact score gives, and keep NOTHING: the rows are screened again by the first
    // regular chunk.  One-wave form only.
    int thr_only;
    // no general-form companion launch behind this one: a query the one-wave form cannot hold (more than its 1024 entries,
    // a candidate list beyond its capacity) is flagged kStOverflow -- re-screened by the host with the tighter bound --
    // instead of being handed over.
    int one_wave_only;
    // not the last prune of the pass: the candidates that survive the cut are NOT re-scored now -- they move to the head of
    // the query's candidate list (count left in st.cnt, st.carry) and meet the next chunk's candidates in the next prune,
    // under its tighter cut; only the pass's last prune walks round B.  A prune before the last one then costs one gather
    // round instead of two (the starter's 33 us against 57-67), and a survivor is re-scored at most once, under the final
    // cut.  The threshold a deferring prune publishes comes from kept U round A alone: any k exact scores bound the k-th
    // best from below.  One-wave form only (the general form re-scores everything it is handed).
    int defer_b;
};                         // (the screen bound is per query: st.E[q])

// Similarity of an exact key in the SCREEN's units: the cosine itself, or dot / |q| (the inner-product shadows hold the rows
// themselves and the queries normalised: dev_common.h "Inner product").  inv_qn = 1 / sqrt(|q|^2) in fp32; callers subtract
// |u| 4e-6 for its roundings wherever the value is used as a lower bound.
__device__ __forceinline__ float unit_sim(int metric, double dist, float inv_qn) {
    return metric == 0 ? (float)(1.0 - dist) : (float)(-dist) * inv_qn;
}
// the screen threshold the k-th best exact key `dist_k` allows: every row of the final top-k has a screen value >= this
__device__ __forceinline__ float screen_threshold(int metric, double dist_k, float E, float nq) {
    if (metric == 0) return float_below((float)((1.0 - dist_k) - (double)E));
    const double u = -dist_k / sqrt((double)nq);
    return float_below((float)(u - fabs(u) * 4e-6 - (double)E));
}

// Two instantiations share the code: a small one (1 wave, <= 1024 entries, ~36 KiB LDS, 4 workgroups
// per CU so that a whole 1024-query block is resident at once and the re-score latency overlaps across
// queries) handles the common case; the large one (4 waves, 4096 entries) is launched right after it
// and only finds work for queries the small one skipped (first chunk, k > ~400, adversarial data).
constexpr int kPruneSmallThreads = 64, kPruneSmallSort = 1024;
constexpr int kPruneBigThreads = 256, kPruneBigSort = kSortMax;

// dynamic LDS: SK[SORT] u64 | SR[SORT] i32 | X = max(Lf[SORT] f32, stage tiles) | R[SORT] i32 | qs[d] f32 | 2 scalars
//              | q16[dpad16] f32 (one-wave form only: the bf16 query of the second screen, expanded)
__host__ __device__ inline size_t prune_qs_floats(int d) { return ((size_t)d + 3) / 4 * 4 + 16; }  // qs + scalars, 16-B multiple
__host__ __device__ inline size_t prune_lds_bytes(int d, int threads, int sortmax, int dpad16) {
    size_t x = (size_t)(threads / kWave) * kStageFloats * sizeof(float);
    size_t lf = (size_t)sortmax * sizeof(float);
    if (lf > x) x = lf;
    return (size_t)sortmax * 8 + (size_t)sortmax * 4 + x + (size_t)sortmax * 4 + prune_qs_floats(d) * 4 + (size_t)dpad16 * 4;
}

// The candidates' screen values as order keys, entry e in slot e / T of thread e % T: 0 = no candidate (below every real key),
// 0xFFFFFFFF = "no bound" (NaN value: always re-scored); int8 screen (flag8 != nullptr): a finite value on a row outside the
// int8 shadow is a stale zero -> dropped (that row comes through k_emit_irregular with NaN instead).
// All loads of a phase are issued before the first is waited for -- values, then (flag8 only) rows, then flags: three memory
// round trips per prune.  (Round 5 loaded slot by slot inside `if (e < n_new)`: one to three round trips PER SLOT, ~0.7 us each
// -- a third of a 45 us prune at the headline's ~6 slots with loose rows in the corpus.)  Indices are clamped, not predicated.
template <int PER, int T, bool ROWS = false>
__device__ __forceinline__ void load_candidate_keys(uint32_t (&key)[PER], const float* __restrict__ cval,
                                                    const int32_t* __restrict__ crow, const uint8_t* __restrict__ flag8,
                                                    int n_new, int tid, int32_t* rows_out = nullptr) {
    float v[PER];
    int32_t r[PER];
    const int last = n_new - 1;  // (callers return before this when n_new == 0)
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        v[j] = 0.0f;
        r[j] = 0;
        if (j * T >= n_new) continue;  // uniform: the list ends before this slot
        v[j] = cval[min(j * T + tid, last)];
        if (ROWS) r[j] = crow[min(j * T + tid, last)];
    }
    uint8_t f[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) f[j] = 0;
    if (flag8 != nullptr) {
        if (!ROWS) {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                if (j * T >= n_new) continue;
                r[j] = crow[min(j * T + tid, last)];
            }
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (j * T >= n_new) continue;
            f[j] = flag8[r[j]];
        }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        uint32_t kk = 0;
        if (j * T + tid < n_new) {
            if (v[j] != v[j]) kk = 0xFFFFFFFFu;
            else if (!f[j]) kk = f32_order_key(v[j]);
        }
        key[j] = kk;
        if (ROWS) rows_out[j] = r[j];
    }
}

template <int THREADS, int SORT>
__device__ __forceinline__ void prune_body(const PruneArgs& a, char* smem);

template <int THREADS, int SORT>
__global__ __launch_bounds__(THREADS) void k_prune(PruneArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // (LDS carve: prune_body; two scalars sit behind the query -- no static LDS keeps the carve 16-B aligned)
    const bool list_mode = a.skip_list != nullptr && a.qlist == nullptr;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    int q = a.qlist ? a.qlist[blockIdx.x] : blockIdx.x;
    if (SORT == kPruneBigSort && list_mode) {
        // general form on a small grid: each workgroup takes the entries blockIdx.x, +gridDim.x, ... of the hand-over list.
        // Workgroup 0 re-arms the OTHER counter for the next prune (nothing touches it during this launch).
        if (blockIdx.x == 0 && tid == 0) a.skip_list[a.skip_parity ^ 1] = 0;
        const int n_left = a.skip_list[a.skip_parity];
        for (int i = blockIdx.x; i < n_left; i += gridDim.x) {
            PruneArgs b = a;
            b.skip_list = nullptr;
            b.qlist = a.skip_list + 2 + a.skip_parity * kQBlockMax + i - 0;  // read as qlist[blockIdx.x]: shift below
            b.qlist -= blockIdx.x;
            prune_body<THREADS, SORT>(b, smem);
            __syncthreads();
        }
        return;
    }
    prune_body<THREADS, SORT>(a, smem);
}

template <int THREADS, int SORT>
__device__ __forceinline__ void prune_body(const PruneArgs& a, char* smem) {
    constexpr int kWaves = THREADS / kWave;
    uint64_t* SK = (uint64_t*)smem;
    int32_t* SR = (int32_t*)(smem + (size_t)SORT * 8);
    char* X = smem + (size_t)SORT * 12;
    constexpr size_t xa = (size_t)kWaves * kStageFloats * sizeof(float), xb = (size_t)SORT * sizeof(float);
    constexpr size_t xbytes = xa > xb ? xa : xb;
    float* Lf = (float*)X;
    float* tiles = (float*)X;
    int32_t* R = (int32_t*)(X + xbytes);
    float* qs = (float*)(X + xbytes + (size_t)SORT * 4);
    int& s_cnt = *(int*)(X + xbytes + (size_t)SORT * 4 + (prune_qs_floats(a.d) - 16) * 4);
    float* q16 = qs + prune_qs_floats(a.d);
    const bool list_mode = a.skip_list != nullptr && a.qlist == nullptr;
    const int q = a.qlist ? a.qlist[blockIdx.x] : blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int raw_cnt = a.st.cnt[q];
    const int n_best = a.st.best_n[q];
    if (raw_cnt == 0) return;  // nothing new; kept list and thresholds stay as they are
    // exact keys (scan path) only need the total-order sort: the four-wave form does it ~4x faster than one wave would, and
    // is launched right behind this one for every listed query
    if (a.exact && SORT < kPruneBigSort) return;
    auto leave_for_general = [&]() {  // one-wave form: hand the query over
        if (a.one_wave_only) {
            if (tid == 0) {
                a.st.status[q] |= kStOverflow;
                a.st.cnt[q] = 0;
            }
            return;
        }
        if (list_mode && tid == 0) {
            const int i = atomicAdd(&a.skip_list[a.skip_parity], 1);
            a.skip_list[2 + a.skip_parity * kQBlockMax + i] = q;
        }
    };
    if (raw_cnt > a.cap) {     // overflow: do not commit; the query is recomputed by the guaranteed path
        if (SORT < kPruneBigSort) {  // let the large instantiation record it
            leave_for_general();
            return;
        }
        if (tid == 0) {
            a.st.status[q] |= kStOverflow;
            a.st.cnt[q] = 0;
        }
        return;
    }
    if (n_best + raw_cnt > SORT) {  // too many for this instantiation: left for the large one
        if (SORT < kPruneBigSort) leave_for_general();
        return;
    }
    const int n_new = raw_cnt;
    const int32_t* crow = a.cand_row + (int64_t)q * a.cap;
    const float* cval = a.cand_val + (int64_t)q * a.cap;
    uint64_t* bkey = a.st.best_key + (int64_t)q * kKMax;
    int32_t* brow = a.st.best_row + (int64_t)q * kKMax;
    const float nq = a.st.qn[q];
    const float inv_qn = 1.0f / sqrtf(nq);  // (inner product: exact dots -> the screen's units; an irregular |q| never screens)
    const float E = a.st.E[q];
    if (tid == 0) {
        a.stat[2 * q] += (unsigned long long)(n_new - a.st.carry[q]);  // (carried entries were counted when they were appended)
        a.st.carry[q] = 0;
    }

    int n_res;
    int n_base = n_best;  // kept entries carried into the final sort (truncated to k after round A)
    bool kept_loaded = false;
    if constexpr (THREADS == kWave) {
        if (!a.exact) {
            // ============ one-wave form: selections instead of sorts ============
            // Every lane keeps 16 candidates' screen values (as order keys) in registers; "the n best" are found by
            // bisection on the key (32 rounds of ballots), lists are built with ballot + mbcnt.  The three LDS
            // bitonic sorts of the general form cost 60-100 us per launch at one wave per SIMD; this costs a few.
            float* tile = tiles;
            uint32_t key[kSelPerLane];
            unsigned in_a = 0;  // bit j: candidate j*64+lane went through round A
            load_candidate_keys<kSelPerLane, kWave>(key, cval, crow, a.flag8, n_new, lane);
            for (int k = lane; k < a.d; k += kWave) qs[k] = a.q[(int64_t)q * a.d + k];
            if (a.shadow16 != nullptr)
                for (int k = lane; k < a.dpad; k += kWave) q16[k] = bf16_bits_to_f32(a.st.qhat[(int64_t)q * a.dpad + k]);
            const int n_cand = wave_count_ge(key, 1u, n_new);
            // wave-local exact re-score of the candidates listed in R[0..n) -> SK/SR[dst..]
            auto rescore_list = [&](int n, int dst) __attribute__((always_inline)) {
                for (int base = 0; base < n; base += kWave) {
                    const int e = base + lane;
                    const bool live = e < n;
                    const int32_t row = live ? crow[R[e]] : -1;
                    const float* rp = live ? a.rows + (int64_t)row * a.d : nullptr;
                    const float acc = staged_dot(tile, rp, qs, a.d, lane);
                    if (live) {
                        SK[dst + e] = dist_to_key(distance_from(a.metric, acc, nq, a.nrm2[row]));
                        SR[dst + e] = row;
                    }
                }
            };
            // list the candidates with (key >= x) && !(in_a), at most `room` of them, into R[0..); marks them in_a if mark
            auto compact = [&](uint32_t x, int room, bool mark) -> int {
                int n = 0;
#pragma unroll
                for (int j = 0; j < kSelPerLane; ++j) {
                    if (j * kWave >= n_new) break;  // wave-uniform: the list ends before this slot
                    const bool want = key[j] >= x && key[j] != 0 && !((in_a >> j) & 1u);
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64(want);
                    const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32),
                                                                       __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
                    if (want && pos < room) {
                        R[pos] = j * kWave + lane;
                        if (mark) in_a |= 1u << j;
                    }
                    n += __builtin_popcountll(bal);
                }
                return min(n, room);
            };
            wave_sync();  // qs visible
            // ---- round A: the best-looking candidates (see the general form below for the reasoning)
            const int wantA = min(n_cand, min(kWave, max(a.k, a.round_a > 0 ? a.round_a : max(32, 2 * a.k))));
            int nA = 0;
            if (wantA > 0) {
                const uint32_t xA = wave_nth_largest(key, wantA, n_new);
                nA = compact(xA, kWave, true);
                wave_sync();
                rescore_list(nA, n_best);
            }
            for (int i = lane; i < n_best; i += kWave) {
                SK[i] = bkey[i];
                SR[i] = brow[i];
            }
            kept_loaded = true;
            wave_sync();
            int n1 = n_best + nA;
            int nB = 0, carried = 0;
            if (n_cand > nA && !a.thr_only) {
                // ---- cut = (k-th largest exact similarity over kept U round A) - E: as float, rounded down
                float cut = -__builtin_inff(), cut16 = -__builtin_inff();
                if (n1 >= a.k) {
                    uint32_t sk[kSelPerLane];
#pragma unroll
                    for (int j = 0; j < kSelPerLane; ++j) {
                        const int e = j * kWave + lane;
                        uint32_t kk = 0;
                        if (e < n1 && SK[e] != kKeyNaN) kk = f32_order_key(sim_of_dist(a.metric, key_to_dist(SK[e])));
                        sk[j] = kk;
                    }
                    if (wave_count_ge(sk, 1u, n1) >= a.k) {
                        const uint32_t xs = wave_nth_largest(sk, a.k, n1);
                        const float kth = __uint_as_float((xs & 0x80000000u) ? (xs & 0x7FFFFFFFu) : ~xs);  // invert the key
                        // cosine: candidates carry v, exact <= v + E.  inner product: they carry the upper bound itself.
                        // candidates carry v with exact <= v + E, in the screen's units (cosine; dot / |q|)
                        const float ku = a.metric == 0 ? kth : kth * inv_qn;
                        cut = ku - fabsf(ku) * (a.metric == 0 ? 0.0f : 4e-6f) - E * 1.001f - 2e-6f * a.cscale;
                        if (a.shadow16 != nullptr && a.metric == 0) cut16 = kth - a.st.E16[q] * 1.001f - 2e-6f;
                    }
                }
                // ---- round B: everything that can still reach the top-k (v + E >= exact k-th best)
                const uint32_t xB = cut == -__builtin_inff() ? 1u : f32_order_key(cut);
                nB = compact(xB, SORT, false);
                wave_sync();
                if (a.defer_b && cut != -__builtin_inff()) {  // (no cut yet -- fewer than k exact scores, k > 64 --: full round B)
                    // carry the survivors: (row, value) of entry R[i] -> slot i of the list.  All loads, then all stores:
                    // a slot may be another survivor's source.
                    int32_t* crow_w = a.cand_row + (int64_t)q * a.cap;
                    float* cval_w = a.cand_val + (int64_t)q * a.cap;
                    int32_t cr[kSelPerLane];
                    float cv[kSelPerLane];
#pragma unroll
                    for (int j = 0; j < kSelPerLane; ++j) {
                        if (j * kWave >= nB) break;  // wave-uniform
                        const int i = j * kWave + lane;
                        if (i < nB) {
                            cr[j] = crow[R[i]];
                            cv[j] = cval[R[i]];
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int j = 0; j < kSelPerLane; ++j) {
                        if (j * kWave >= nB) break;
                        const int i = j * kWave + lane;
                        if (i < nB) {
                            crow_w[i] = cr[j];
                            cval_w[i] = cv[j];
                        }
                    }
                    carried = nB;
                    nB = 0;
                }
                if (a.shadow16 != nullptr && cut16 != -__builtin_inff() && nB > 0) {
                    // ---- second screen: the bf16 image of every survivor (half the bytes of its fp32 row) under the
                    // bf16 bound (~5x tighter than the int8 one): t16 + E16 < exact k-th best -> cannot reach the top-k.
                    // NaN (irregular row: no bf16 image) compares false and stays.  Survivors are compacted in place.
                    const int words = a.dpad / 2;
                    int nS = 0;
                    for (int base = 0; base < nB; base += kWave) {
                        const int e = base + lane;
                        const bool live = e < nB;
                        const int ci = live ? R[e] : 0;
                        const float* rp = live ? (const float*)(a.shadow16 + (int64_t)crow[ci] * a.dpad) : nullptr;
                        const float t16 = staged_dot16(tile, rp, q16, words, lane);
                        const bool keep = live && !(t16 < cut16);
                        const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
                        const int pos = nS + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32),
                                                                           __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
                        wave_sync();  // every lane holds its entry of this batch: slots below base + 64 may be rewritten
                        if (keep) R[pos] = ci;
                        nS += __builtin_popcountll(bal);
                    }
                    nB = nS;
                    wave_sync();
                }
                rescore_list(nB, n1);
            }
            if (lane == 0) a.stat[2 * q + 1] += (unsigned long long)(nA + nB);
            wave_sync();
            // ---- final: the k best of kept U A U B under (key,row).  Select by similarity (float image of the key,
            // monotone), then sort only the selected few.
            const int n_tot = n1 + nB;
            uint64_t* K2 = (uint64_t*)X;                      // the stage tile is dead now
            int32_t* R2 = (int32_t*)(X + (size_t)SORT * 8);
            int n_sel = n_tot;
            if (n_tot > a.k) {
                uint32_t sk[kSelPerLane];
#pragma unroll
                for (int j = 0; j < kSelPerLane; ++j) {
                    const int e = j * kWave + lane;
                    uint32_t kk = 0;
                    if (e < n_tot) kk = SK[e] == kKeyNaN ? 1u : f32_order_key(sim_of_dist(a.metric, key_to_dist(SK[e])));
                    sk[j] = kk;  // every real similarity (order key >= 0x007FFFFF) ranks above the NaN class 1 and "absent" 0
                }
                const uint32_t xs = wave_nth_largest(sk, a.k, n_tot);
                n_sel = 0;
#pragma unroll
                for (int j = 0; j < kSelPerLane; ++j) {
                    if (j * kWave >= n_tot) break;  // wave-uniform
                    const bool want = sk[j] >= xs && sk[j] != 0;
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64(want);
                    const int pos = n_sel + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32),
                                                                           __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
                    if (want) {
                        K2[pos] = SK[j * kWave + lane];
                        R2[pos] = SR[j * kWave + lane];
                    }
                    n_sel += __builtin_popcountll(bal);
                }
            } else {
                for (int i = lane; i < n_tot; i += kWave) {
                    K2[i] = SK[i];
                    R2[i] = SR[i];
                }
            }
            const int np = next_pow2(max(n_sel, 1));
            for (int i = n_sel + lane; i < np; i += kWave) {
                K2[i] = kKeyNaN;
                R2[i] = 0x7FFFFFFF;
            }
            __syncthreads();
            bitonic_asc_key_row(K2, R2, np);
            const int n_keep = min(a.k, n_sel);
            if (!a.thr_only)
                for (int i = lane; i < n_keep; i += kWave) {
                    bkey[i] = K2[i];
                    brow[i] = R2[i];
                }
            if (lane == 0) {
                if (!a.thr_only) a.st.best_n[q] = n_keep;
                a.st.cnt[q] = carried;  // (the next chunk's appends go behind the carried survivors)
                a.st.carry[q] = carried;
                if (n_keep >= a.k) {
                    const uint64_t wk = K2[a.k - 1];
                    if (!a.thr_only) {  // (the exact path's threshold speaks for KEPT rows)
                        a.st.thr_key[q] = wk;
                        a.st.thr_row[q] = R2[a.k - 1];
                    }
                    if (wk != kKeyNaN && !(a.st.status[q] & kStIrregular))
                        a.st.thr[q] = screen_threshold(a.metric, key_to_dist(wk), E, nq);
                }
            }
            return;
        }
    }
    if (!a.exact) {
        // ---- phase 1: order the new candidates by their screen value, best first.  NaN = "no bound" sorts first;
        // int8 screen: a finite value on a row outside the int8 shadow is a stale zero -> dropped (that row comes
        // through k_emit_irregular with NaN instead).
        const int nLp = next_pow2(n_new);
        for (int i = tid; i < nLp; i += THREADS) {
            float v = -__builtin_inff();
            if (i < n_new) {
                v = cval[i];
                if (v != v) v = __builtin_inff();
                else if (a.flag8 && a.flag8[crow[i]]) v = -__builtin_inff();
            }
            Lf[i] = v;
            R[i] = i;
        }
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        bitonic_desc_f32_i32(Lf, R, nLp);
        for (int i = tid; i < nLp; i += THREADS)
            if (Lf[i] > -__builtin_inff() && (i + 1 == nLp || !(Lf[i + 1] > -__builtin_inff()))) s_cnt = i + 1;
        for (int k = tid; k < a.d; k += THREADS) qs[k] = a.q[(int64_t)q * a.d + k];
        __syncthreads();
        const int n_cand = s_cnt;
        __syncthreads();  // everyone has read s_cnt (it is reused below) and Lf (overwritten by the stage tiles)
        float* tile = tiles + wave * kStageFloats;
        // exact fp32 chain for sorted entries [e0, e1) -> SK/SR slots [dst, dst + e1 - e0)
        auto rescore = [&](int e0, int e1, int dst) {
            for (int base = e0; base < e1; base += THREADS) {
                const int e = base + wave * kWave + lane;
                const bool live = e < e1;
                const int32_t row = live ? crow[R[e]] : -1;
                const float* rp = live ? a.rows + (int64_t)row * a.d : nullptr;
                float acc = 0.0f;
                if (base + wave * kWave < e1) acc = staged_dot(tile, rp, qs, a.d, lane);  // wave-uniform condition
                if (live) {
                    const double dist = distance_from(a.metric, acc, nq, a.nrm2[row]);
                    SK[dst + (e - e0)] = dist_to_key(dist);
                    SR[dst + (e - e0)] = row;
                }
            }
        };
        // ---- phase 2 (round A): re-score the best-looking candidates -- enough of them to contain the new top-k with
        // near certainty (the screen's ACTUAL error is ~10x below its bound, so its order is almost the exact order),
        // but no more: every row costs d*4 gathered bytes, and the gather is what bounds this kernel
        const int nA = min(n_cand, min(THREADS, max(32, 2 * a.k)));
        rescore(0, nA, n_best);
        int n_pass = nA;
        if (n_cand > nA) {
            // ---- phase 3: exact k-th best over (kept U round A) -> cut for the rest.  A candidate with
            // v + E < cut cannot reach the top-k: its exact similarity is <= v + E.
            for (int i = tid; i < n_best; i += THREADS) {
                SK[i] = bkey[i];
                SR[i] = brow[i];
            }
            kept_loaded = true;
            const int n1 = n_best + nA;
            const int np1 = next_pow2(n1);
            for (int i = n1 + tid; i < np1; i += THREADS) {
                SK[i] = kKeyNaN;
                SR[i] = 0x7FFFFFFF;
            }
            __syncthreads();
            bitonic_asc_key_row(SK, SR, np1);
            n_base = min(a.k, n1);
            float cut = -__builtin_inff();
            if (n1 >= a.k) {
                const uint64_t wk = SK[a.k - 1];
                if (wk != kKeyNaN) {
                    const float ku = unit_sim(a.metric, key_to_dist(wk), inv_qn);
                    cut = ku - fabsf(ku) * (a.metric == 0 ? 0.0f : 4e-6f) - E * 1.001f - 2e-6f * a.cscale;
                }
            }
            if (tid == 0) s_cnt = nA;
            __syncthreads();
            // the list is sorted by v: the survivors are a prefix
            for (int i = nA + tid; i < n_cand; i += THREADS) {
                const float v = cval[R[i]];
                if (!(v < cut)) atomicMax(&s_cnt, i + 1);  // NaN passes
            }
            __syncthreads();
            n_pass = s_cnt;
            // ---- phase 4 (round B): re-score the survivors
            rescore(nA, n_pass, n_base);
            n_res = n_pass - nA;
        } else {
            n_res = nA;
        }
        if (tid == 0) a.stat[2 * q + 1] += (unsigned long long)n_pass;
    } else {
        n_res = n_new;
        for (int i = tid; i < n_new; i += THREADS) {
            const int32_t row = crow[i];
            const double dist = distance_from(a.metric, cval[i], nq, a.nrm2[row]);
            SK[n_best + i] = dist_to_key(dist);
            SR[n_best + i] = row;
        }
    }
    // ---- final: total-order sort of kept U re-scored, keep k
    if (!kept_loaded)
        for (int i = tid; i < n_best; i += THREADS) {
            SK[i] = bkey[i];
            SR[i] = brow[i];
        }
    const int n_tot = n_base + n_res;
    const int np = next_pow2(n_tot);
    for (int i = n_tot + tid; i < np; i += THREADS) {
        SK[i] = kKeyNaN;
        SR[i] = 0x7FFFFFFF;
    }
    __syncthreads();
    bitonic_asc_key_row(SK, SR, np);
    const int n_keep = min(a.k, n_tot);
    for (int i = tid; i < n_keep; i += THREADS) {
        bkey[i] = SK[i];
        brow[i] = SR[i];
    }
    if (tid == 0) {
        a.st.best_n[q] = n_keep;
        a.st.cnt[q] = 0;
        if (n_keep >= a.k) {
            const uint64_t wk = SK[a.k - 1];
            a.st.thr_key[q] = wk;
            a.st.thr_row[q] = SR[a.k - 1];
            if (!a.exact && wk != kKeyNaN && !(a.st.status[q] & kStIrregular))
                a.st.thr[q] = screen_threshold(a.metric, key_to_dist(wk), E, nq);
        }
    }
}

// grid: B blocks of 64 threads; appends every irregular row as a "no bound" candidate (val = NaN)
// rows below `skip_below` were already kept by the emit-all first chunk (their NaN screen value = "no bound")
__global__ __launch_bounds__(64) void k_emit_irregular(const int32_t* __restrict__ irr_rows, int irr_n, QueryState st,
                                                        int32_t* cand_row, float* cand_val, int cap, int skip_below) {
    const int q = blockIdx.x;
    if (st.status[q] & kStIrregular) return;
    for (int i = threadIdx.x; i < irr_n; i += blockDim.x) {
        if (irr_rows[i] < skip_below) continue;
        const int slot = atomicAdd(&st.cnt[q], 1);
        if (slot < cap) {
            cand_row[(int64_t)q * cap + slot] = irr_rows[i];
            cand_val[(int64_t)q * cap + slot] = __builtin_nanf("");
        }
    }
}

// grid: B blocks of 64 threads
__global__ __launch_bounds__(64) void k_finalize(QueryState st, int k, int64_t row_offset, double* out_dist,
                                                  int64_t* out_rows, int* status_or) {
    const int q = blockIdx.x;
    const int n = st.best_n[q];
    for (int s = threadIdx.x; s < k; s += blockDim.x) {
        if (s < n) {
            out_dist[(int64_t)q * k + s] = key_to_dist(st.best_key[(int64_t)q * kKMax + s]);
            out_rows[(int64_t)q * k + s] = (int64_t)st.best_row[(int64_t)q * kKMax + s] + row_offset;
        } else {
            out_dist[(int64_t)q * k + s] = __longlong_as_double(0x7FF8000000000000ll);
            out_rows[(int64_t)q * k + s] = -1;
        }
    }
    if (threadIdx.x == 0 && st.status[q] != 0) atomicOr(status_or, st.status[q]);
}

// grid: B blocks of 256 threads.  Shard lists are already in total order and carry global rows; the
// merge is one more sort under the same order, so the result equals the single-GPU result.
// rank w's lists start at dist_all + w*rank_stride and rows_all + w*rank_stride (elements), each [B][k]
__global__ __launch_bounds__(256) void k_merge_topk(const double* __restrict__ dist_all,
                                                     const int64_t* __restrict__ rows_all, int64_t rank_stride, int world,
                                                     int B, int k, double* out_dist, int64_t* out_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* SK = (uint64_t*)smem;
    const int np_ = next_pow2(world * k);
    int32_t* SI = (int32_t*)(smem + (size_t)np_ * 8);  // index into the gathered lists (row may exceed int32)
    const int q = blockIdx.x;
    const int n = world * k;
    const int np = next_pow2(n);
    // sort by (key, global row): rows are compared through a second key pass, so pack (key,row) order
    // as: primary key in SK, tie-break resolved after the sort window by a stable fix-up (rows differ
    // across shards, and equal keys are rare) -- done exactly below with a 2-level compare.
    for (int i = threadIdx.x; i < np; i += blockDim.x) {
        if (i < n) {
            const int w = i / k, s = i % k;
            const int64_t src = (int64_t)w * rank_stride + (int64_t)q * k + s;
            const int64_t r = rows_all[src];
            SK[i] = r < 0 ? kKeyNaN : dist_to_key(dist_all[src]);
            SI[i] = r < 0 ? 0x7FFFFFFF : i;
        } else {
            SK[i] = kKeyNaN;
            SI[i] = 0x7FFFFFFF;
        }
    }
    __syncthreads();
    // bitonic sort with compare (key asc, global row asc); padding (SI = INT_MAX) sorts last
    for (int kk = 2; kk <= np; kk <<= 1) {
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np; i += blockDim.x) {
                const int p = i ^ j;
                if (p > i) {
                    const bool asc = ((i & kk) == 0);
                    const uint64_t ka = SK[i], kb = SK[p];
                    const int32_t ia = SI[i], ib = SI[p];
                    auto grow = [&](int32_t ix) -> int64_t {
                        if (ix == 0x7FFFFFFF) return INT64_MAX;
                        const int w = ix / k, s = ix % k;
                        return rows_all[(int64_t)w * rank_stride + (int64_t)q * k + s];
                    };
                    bool a_before_b, b_before_a;
                    if (ka != kb) {
                        a_before_b = ka < kb;
                        b_before_a = !a_before_b;
                    } else {
                        const int64_t ra = grow(ia), rb = grow(ib);
                        a_before_b = ra < rb;
                        b_before_a = rb < ra;
                    }
                    if (asc ? b_before_a : a_before_b) {
                        SK[i] = kb;
                        SK[p] = ka;
                        SI[i] = ib;
                        SI[p] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int s = threadIdx.x; s < k; s += blockDim.x) {
        const int32_t ix = s < np ? SI[s] : 0x7FFFFFFF;
        if (ix != 0x7FFFFFFF) {
            const int w = ix / k, ss = ix % k;
            const int64_t src = (int64_t)w * rank_stride + (int64_t)q * k + ss;
            out_dist[(int64_t)q * k + s] = dist_all[src];
            out_rows[(int64_t)q * k + s] = rows_all[src];
        } else {
            out_dist[(int64_t)q * k + s] = __longlong_as_double(0x7FF8000000000000ll);
            out_rows[(int64_t)q * k + s] = -1;
        }
    }
}

}  // namespace mi355
