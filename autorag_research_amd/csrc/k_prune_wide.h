// k_prune_wide.h -- exact re-score + select for 32 < k <= 128 (round 6): one workgroup of TWO waves per query, up to 4096
// candidates per prune, selections instead of sorts.
//
// Why a second form.  The one-wave k_prune (k_select.h) selects among at most 1024 entries and re-scores at most 64 rows
// before its cut is known -- fewer than k from k = 65 up.  A chunk appends ~ k * inflation * (growth of the rows seen)
// candidates per query, so at k = 100 its 1024 entries held the chunk ratio at 1.35: THIRTY chunks for N = 10 M
// (profiles/r05_timeline_k100.txt), each with a prune whose round A pulled 64 rows per query through the exact chain --
// 3 359 re-scored rows per query and pass, 10 GB of gathered fp32 rows, 3 ms of the 10.5 ms pass.  Here:
//   * 4096 entries (32 order keys per lane in registers), so the schedule may triple the rows seen per chunk: 5 chunks
//     behind a 64 k-row starter at N = 10 M, k = 100;
//   * round A = min(128, max(k, 2k)) rows in ONE gather round (one batch per wave, side by side): a cut always exists;
//   * every selection is a bisection on the order key with one LDS exchange per round (two waves: one s_barrier); the
//     round-A selection stops after the 20 leading bits (any >= k exact scores give a valid cut: it need not be THE best);
//   * round B walks the survivors in windows of 1024 rows, keeps only exact scores that can still enter the top-k and
//     re-selects ("shrinks") its 512-entry exact buffer whenever a batch might overflow it -- no 4096-entry sort anywhere;
//   * a deferring prune (every prune of a pass but its last) moves the survivors to the head of the list straight from
//     registers (the value is the inverse image of the order key);
//   * what the form cannot hold (more candidates than its entries or the list's capacity) is flagged kStOverflow for the
//     host's re-screen; there is no companion launch.
// Results do not depend on any of this: the kept top-k is decided by the exact (key, row) order alone, and every cut is the
// rigorous bound of DESIGN.md "Screen bounds" (same expressions as k_select.h).
// Reference: the same `ORDER BY distance LIMIT k` of orm/repository/base.py:409-415, at limit = 33 ... 128 (BASELINE config 2).
#pragma once
#include <type_traits>

#include "k_select.h"

namespace mi355 {

constexpr int kWideEntries = 4096;   // candidates per prune: WAVES x 64 threads x PER keys (2 x 32 or 1 x 64)
constexpr int kWideKeep = 512;   // exact (key, row) pairs in LDS: kept (<= 128) + round A (<= 128) + one batch, then shrunk
constexpr int kWideList = 1024;  // rows listed per re-score window
constexpr int kWideKMin = 33, kWideKMax = 128;
constexpr int kWideSortWhole = 256;  // kept (<= 128) + round A (<= 128)

// dynamic LDS: SK[kWideKeep] u64 | SR[kWideKeep] i32 | stage tiles (one per wave; the final sort's K2 / R2 alias them) |
//              RL[kWideList] i32 | qs[d] f32 + 16 scalars
__host__ __device__ inline size_t prune_wide_lds_bytes(int d, int waves) {
    return (size_t)kWideKeep * 12 + (size_t)waves * kStageFloats * sizeof(float) + (size_t)kWideList * 4 +
           prune_qs_floats(d) * 4;
}

// inverse of f32_order_key (0xFFFFFFFF, the "no bound" class, comes back as a NaN)
__device__ __forceinline__ float f32_from_order_key(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// WAVES x PER = 2 x 32 is the form the library launches.  1 x 64 (ONE wave per query, 64 keys per lane, 30 KiB of LDS: all 1024
// queries of a block resident at once instead of 512, round A in two batches one after the other) was built and measured,
// bit-identical: its deferring prunes take 164-184 us against 128-155 (N = 10 M, k = 100; profiles/r06_prune_wide_ab.txt) --
// what bounds a prune is the chain of dependent steps inside one query, not how many queries are resident.
template <int WAVES, int PER>
__global__ __launch_bounds__(WAVES * kWave) void k_prune_wide(PruneArgs a) {
    constexpr int T = WAVES * kWave;
    static_assert(T * PER == kWideEntries && (PER == 32 || PER == 64), "entries");
    typedef std::conditional_t<PER == 64, unsigned long long, uint32_t> amask_t;
    constexpr int KS = kWideKeep / T;  // slots per thread over the exact buffer
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* SK = (uint64_t*)smem;
    int32_t* SR = (int32_t*)(smem + (size_t)kWideKeep * 8);
    float* tiles = (float*)(smem + (size_t)kWideKeep * 12);
    uint64_t* K2 = (uint64_t*)tiles;  // final sort (the stage tiles are dead by then)
    int32_t* R2 = (int32_t*)((char*)tiles + (size_t)kWideKeep * 8);
    int32_t* RL = (int32_t*)((char*)tiles + (size_t)WAVES * kStageFloats * sizeof(float));
    float* qs = (float*)((char*)RL + (size_t)kWideList * 4);
    int* scal = (int*)(qs + prune_qs_floats(a.d) - 16);  // [0..3] exchange words (two parities x two waves), [4] exact-buffer fill
    const int q = a.qlist ? a.qlist[blockIdx.x] : blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int raw_cnt = a.st.cnt[q];
    const int n_best = a.st.best_n[q];
    if (raw_cnt == 0) return;  // nothing new; kept list and thresholds stay as they are
    if (raw_cnt > a.cap || raw_cnt > kWideEntries) {  // more than the list or this form holds: re-screened by the host
        if (tid == 0) {
            a.st.status[q] |= kStOverflow;
            a.st.cnt[q] = 0;
            a.st.carry[q] = 0;
        }
        return;
    }
    const int n_new = raw_cnt;
    const int32_t* crow = a.cand_row + (int64_t)q * a.cap;
    const float* cval = a.cand_val + (int64_t)q * a.cap;
    uint64_t* bkey = a.st.best_key + (int64_t)q * kKMax;
    int32_t* brow = a.st.best_row + (int64_t)q * kKMax;
    const float nq = a.st.qn[q];
    const float inv_qn = 1.0f / sqrtf(nq);
    const float E = a.st.E[q];
    if (tid == 0) {
        a.stat[2 * q] += (unsigned long long)(n_new - a.st.carry[q]);  // (carried entries were counted when they were appended)
        scal[4] = 0;
    }

    // ---- workgroup-wide helpers ---------------------------------------------------------------------------------------------
    int par = 0;
    // sum of a wave-uniform value over the waves; one barrier (two exchange parities: a wave may enter the next exchange
    // while its sibling still reads this one's words)
    auto wg_sum = [&](int v) __attribute__((always_inline)) -> int {
        if constexpr (WAVES == 1) return v;
        if (lane == 0) scal[par * WAVES + wave] = v;
        __syncthreads();
        int c = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) c += scal[par * WAVES + w];
        par ^= 1;
        return __builtin_amdgcn_readfirstlane(c);
    };
    // the same, plus the sum over the waves before this one
    auto wg_scan = [&](int v, int& before) __attribute__((always_inline)) -> int {
        if constexpr (WAVES == 1) {
            before = 0;
            return v;
        }
        if (lane == 0) scal[par * WAVES + wave] = v;
        __syncthreads();
        int c = 0, b = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const int x = scal[par * WAVES + w];
            c += x;
            if (w < wave) b += x;
        }
        par ^= 1;
        before = __builtin_amdgcn_readfirstlane(b);
        return __builtin_amdgcn_readfirstlane(c);
    };
    auto mbcnt = [&](unsigned long long bal) __attribute__((always_inline)) -> int {
        return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
    };

    // ---- the candidates' screen values as order keys: entry e lives in slot e / T of thread e % T ---------------------------
    uint32_t key[PER];
    int32_t rowv[PER];  // ... and their rows (lists and the carried survivors are written from registers: no second trip to memory)
    amask_t in_a = 0;  // bit j: the candidate of slot j went through round A
    load_candidate_keys<PER, T, true>(key, cval, crow, a.flag8, n_new, tid, rowv);
    for (int k = tid; k < a.d; k += T) qs[k] = a.q[(int64_t)q * a.d + k];
    auto count_ge = [&](uint32_t x) __attribute__((always_inline)) -> int {
        int c = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (j * T >= n_new) break;  // uniform: the list ends before this slot
            c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[j] >= x));
        }
        return wg_sum(c);
    };
    const int n_cand = count_ge(1u);  // (its barrier also publishes qs and scal[4])

    // list the rows of the candidates with want(j), positions [off, off + room) of them, into RL[0 ...); returns their number
    auto list_rows = [&](auto&& want, int off, int room, bool mark) __attribute__((always_inline)) -> int {
        int c = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (j * T >= n_new) break;
            c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(want(j)));
        }
        int n = 0;
        const int tot = wg_scan(c, n);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (j * T >= n_new) break;
            const bool w = want(j);
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(w);
            const int pos = n + mbcnt(bal) - off;
            if (w && pos >= 0 && pos < room) {
                RL[pos] = rowv[j];
                if (mark) in_a |= (amask_t)1 << j;
            }
            n += __builtin_popcountll(bal);
        }
        return tot;
    };

    float* tile = tiles + (size_t)wave * kStageFloats;
    // k-th largest float image over the exact buffer's first n entries (NaN distances: class 1 when with_nan, else absent);
    // returns 0 when fewer than `kth` entries carry a key
    auto exact_kth_image = [&](uint32_t (&sk)[KS], int n, int kth, bool with_nan) __attribute__((always_inline)) -> uint32_t {
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const int e = j * T + tid;
            uint32_t kk = 0;
            if (e < n) {
                const uint64_t k64 = SK[e];
                if (k64 != kKeyNaN) kk = f32_order_key(sim_of_dist(a.metric, key_to_dist(k64)));
                else if (with_nan) kk = 1u;
            }
            sk[j] = kk;
        }
        uint32_t x = 0;
        for (int b = 31; b >= 0; --b) {
            const uint32_t t = x | (1u << b);
            int c = 0;
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                if (j * T >= n) break;
                c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(sk[j] >= t));
            }
            if (wg_sum(c) >= kth) x = t;
        }
        return x;
    };
    // the k best of SK / SR[0, n) under (key, row), sorted, in K2 / R2[0 ...); returns how many (<= max(k, ties)) were sorted.
    // Up to kWideSortWhole entries (kept U round A: every prune but a pass's last) are sorted whole -- no selection in front of the sort.
    // Callers have passed a barrier since SK / SR were last written; K2 / R2 are complete (barrier) on return.
    auto select_sort = [&](int n) __attribute__((always_inline)) -> int {
        int n_sel = n;
        if (n > a.k && n > kWideSortWhole) {
            uint32_t sk[KS];
            const uint32_t xs = exact_kth_image(sk, n, a.k, true);  // (every real similarity ranks above the NaN class 1 and "absent" 0)
            int c = 0;
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                if (j * T >= n) break;
                c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(sk[j] >= xs && sk[j] != 0));
            }
            int pos0 = 0;
            n_sel = wg_scan(c, pos0);
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                if (j * T >= n) break;
                const bool w = sk[j] >= xs && sk[j] != 0;
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(w);
                const int pos = pos0 + mbcnt(bal);
                if (w) {
                    K2[pos] = SK[j * T + tid];
                    R2[pos] = SR[j * T + tid];
                }
                pos0 += __builtin_popcountll(bal);
            }
        } else {
            for (int i = tid; i < n; i += T) {
                K2[i] = SK[i];
                R2[i] = SR[i];
            }
        }
        const int np = next_pow2(max(n_sel, 1));
        for (int i = n_sel + tid; i < np; i += T) {
            K2[i] = kKeyNaN;
            R2[i] = 0x7FFFFFFF;
        }
        __syncthreads();
        bitonic_asc_key_row(K2, R2, np);
        return n_sel;
    };

    // ---- round A: the best-looking candidates, one gather round (a batch per wave) -------------------------------------------
    const int wantA = min(n_cand, min(2 * kWave, max(a.k, a.round_a > 0 ? a.round_a : max(32, 2 * a.k))));
    int nA = 0;
    if (wantA > 0) {
        // largest x (20 leading bits) with count(key >= x) >= wantA: a few more than wantA may pass, the list takes the first 128
        uint32_t x = 0;
        for (int b = 31; b >= 12; --b) {
            const uint32_t t = x | (1u << b);
            if (count_ge(t) >= wantA) x = t;
        }
        if (x == 0) x = 1u;
        constexpr int kAMax = 2 * kWave;  // rows of round A (one batch per wave of the two-wave form, two batches of the one-wave form)
        nA = min(kAMax, list_rows([&](int j) { return key[j] >= x; }, 0, kAMax, true));
        __syncthreads();  // RL complete
        for (int base = 0; base < nA; base += T) {  // uniform
            const int e = base + tid;
            const bool live = e < nA;
            const int32_t row = live ? RL[e] : -1;
            const float* rp = live ? a.rows + (int64_t)row * a.d : nullptr;
            if (base + wave * kWave < nA) {  // wave-uniform
                const float acc = staged_dot(tile, rp, qs, a.d, lane);
                if (live) {
                    SK[n_best + e] = dist_to_key(distance_from(a.metric, acc, nq, a.nrm2[row]));
                    SR[n_best + e] = row;
                }
            }
        }
    }
    for (int i = tid; i < n_best; i += T) {
        SK[i] = bkey[i];
        SR[i] = brow[i];
    }
    __syncthreads();
    const int n1 = n_best + nA;  // <= 2 T: sorted whole, once -- the cut reads its k-th entry, a deferring prune publishes it as it is
    int n_sel = select_sort(n1);
    int carried = 0, nB_rescored = 0;
    if (n_cand > nA && !a.thr_only) {
        // ---- cut = (k-th best exact similarity over kept U round A) - E: as float, rounded down (k_select.h, same expressions)
        float cut = -__builtin_inff();
        uint32_t keep_x = 0;  // an exact score must reach this float image to enter the exact buffer (0: anything does)
        if (n1 >= a.k && K2[a.k - 1] != kKeyNaN) {  // (NaN distances sort last: a NaN here = fewer than k real scores)
            const float kth = sim_of_dist(a.metric, key_to_dist(K2[a.k - 1]));
            const float ku = a.metric == 0 ? kth : kth * inv_qn;
            cut = ku - fabsf(ku) * (a.metric == 0 ? 0.0f : 4e-6f) - E * 1.001f - 2e-6f * a.cscale;
            keep_x = f32_order_key(kth);
        }
        const uint32_t xB = cut == -__builtin_inff() ? 1u : f32_order_key(cut);
        auto wantB = [&](int j) { return key[j] >= xB && key[j] != 0 && !((in_a >> j) & (amask_t)1); };
        if (a.defer_b && cut != -__builtin_inff()) {  // (no cut yet -- fewer than k exact scores --: full round B)
            // carry the survivors to the head of the list, from registers.  A slot may be another survivor's source, but every
            // source was read when the keys were loaded: only the barrier between the counting pass and the stores remains.
            int32_t* crow_w = a.cand_row + (int64_t)q * a.cap;
            float* cval_w = a.cand_val + (int64_t)q * a.cap;
            int c = 0;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                if (j * T >= n_new) break;
                c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(wantB(j)));
            }
            int pos0 = 0;
            carried = wg_scan(c, pos0);
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                if (j * T >= n_new) break;
                const bool w = wantB(j);
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(w);
                const int pos = pos0 + mbcnt(bal);
                if (w) {
                    crow_w[pos] = rowv[j];
                    cval_w[pos] = f32_from_order_key(key[j]);
                }
                pos0 += __builtin_popcountll(bal);
            }
        } else {
            // ---- round B: everything that can still reach the top-k, in windows of kWideList rows; an exact score enters the
            // buffer only if it reaches the k-th best known so far, and the buffer is re-selected before a batch could overflow it
            const int n_keepA = min(a.k, n_sel);
            for (int i = tid; i < n_keepA; i += T) {  // the sorted best of kept U A back into the exact buffer (K2 / R2 are the stage tiles)
                SK[i] = K2[i];
                SR[i] = R2[i];
            }
            if (tid == 0) scal[4] = n_keepA;
            int totB = 0;
            for (int off = 0; off == 0 || off < totB; off += kWideList) {  // uniform
                totB = list_rows(wantB, off, kWideList, false);
                __syncthreads();  // RL complete, scal[4] visible, K2 / R2 read
                const int m = min(kWideList, totB - off);
                for (int base = 0; base < m; base += T) {  // uniform
                    const int e = base + tid;
                    const bool live = e < m;
                    const int32_t row = live ? RL[e] : -1;
                    const float* rp = live ? a.rows + (int64_t)row * a.d : nullptr;
                    if (base + wave * kWave < m) {  // wave-uniform
                        const float acc = staged_dot(tile, rp, qs, a.d, lane);
                        uint64_t k64 = kKeyNaN;
                        bool pass = false;
                        if (live) {
                            k64 = dist_to_key(distance_from(a.metric, acc, nq, a.nrm2[row]));
                            // (a NaN distance only matters while fewer than k real ones are known: keep_x == 0)
                            pass = keep_x == 0 ||
                                   (k64 != kKeyNaN && f32_order_key(sim_of_dist(a.metric, key_to_dist(k64))) >= keep_x);
                        }
                        const unsigned long long bal = __builtin_amdgcn_ballot_w64(pass);
                        int at = 0;
                        if (lane == 0 && bal != 0) at = atomicAdd(&scal[4], __builtin_popcountll(bal));
                        at = __shfl(at, 0, kWave) + mbcnt(bal);
                        if (pass) {
                            SK[at] = k64;
                            SR[at] = row;
                        }
                    }
                    __syncthreads();
                    const int fill = __builtin_amdgcn_readfirstlane(scal[4]);
                    if (fill + T > kWideKeep) {  // uniform: the next batch might not fit -- keep the k best, tighten the filter
                        const int ns = select_sort(fill);
                        const int nk = min(a.k, ns);
                        for (int i = tid; i < nk; i += T) {
                            SK[i] = K2[i];
                            SR[i] = R2[i];
                        }
                        if (nk >= a.k && K2[a.k - 1] != kKeyNaN)
                            keep_x = max(keep_x, f32_order_key(sim_of_dist(a.metric, key_to_dist(K2[a.k - 1]))));
                        __syncthreads();  // K2 / R2 (the stage tiles) read, SK / SR rewritten
                        if (tid == 0) scal[4] = nk;
                        __syncthreads();
                    }
                }
                nB_rescored += m;
            }
            // ---- final: the k best of kept U A U B under (key, row)
            n_sel = select_sort(__builtin_amdgcn_readfirstlane(scal[4]));
        }
    }
    if (tid == 0) a.stat[2 * q + 1] += (unsigned long long)(nA + nB_rescored);
    const int n_keep = min(a.k, n_sel);
    if (!a.thr_only)
        for (int i = tid; i < n_keep; i += T) {
            bkey[i] = K2[i];
            brow[i] = R2[i];
        }
    if (tid == 0) {
        if (!a.thr_only) a.st.best_n[q] = n_keep;
        a.st.cnt[q] = carried;  // (the next chunk's appends go behind the carried survivors)
        a.st.carry[q] = carried;
        if (n_keep >= a.k) {
            const uint64_t wk = K2[a.k - 1];
            if (!a.thr_only) {  // (the exact path's threshold speaks for KEPT rows)
                a.st.thr_key[q] = wk;
                a.st.thr_row[q] = R2[a.k - 1];
            }
            if (wk != kKeyNaN && !(a.st.status[q] & kStIrregular)) a.st.thr[q] = screen_threshold(a.metric, key_to_dist(wk), E, nq);
        }
    }
}

}  // namespace mi355
