// dev_common.h -- device-side helpers shared by all kernels of libmi355dr (gfx950 only).
//
// Exact arithmetic contract (must stay bit-identical to oracle/oracle.c):
//   dot(a,b)  = k-ascending chain  acc = fmaf(a[k], b[k], acc)            (fp32)
//   cosine distance = 1 - clamp((double)dot / sqrt((double)nq * (double)nc), -1, 1)   (double)
//   total order = (distance asc, NaN last, row asc)
// The translation unit is compiled with -ffp-contract=off so nothing here is re-associated or fused
// behind our back; every fused multiply-add is an explicit __builtin_fmaf.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi355 {

constexpr int kWave = 64;
constexpr int kQBlockMax = 1024;  // queries scored per corpus pass (one internal block)
constexpr int kKMax = 1024;       // largest k served
constexpr int kCandCap = 2048;    // candidate slots per query between two prunes
constexpr int kCandCapWide = 4096;  // ... of a pass whose prunes take the two-wave form (k_prune_wide.h: 33 <= k <= 128); the lists are allocated for it
constexpr int kSortMax = 4096;    // LDS sort capacity (>= kKMax + kCandCap, power of two)
constexpr int kIrrCap = 1024;     // irregular (zero / non-finite / extreme-norm) rows the screen path tolerates
constexpr uint64_t kKeyNaN = 0xFFFFFFFFFFFFFFFFull;

// status bits per query
constexpr int kStOverflow = 1;   // candidate buffer overflowed in some chunk -> result must be recomputed
constexpr int kStIrregular = 2;  // query norm is zero / non-finite / out of the screen's range

// per-query search state (device arrays, one slot per query of the current block)
struct QueryState {
    float* qn;          // [Bpad] |q|^2
    uint16_t* qhat;     // [Bpad, dpad] bf16 normalised queries (rows >= B are zero)
    float* thr;         // [Bpad] screen threshold (emit iff t >= thr)
    int* cnt;           // [Bpad] candidates appended since the last prune
    int* best_n;        // [Bpad]
    uint64_t* best_key; // [Bpad, kKMax]
    int32_t* best_row;  // [Bpad, kKMax]
    uint64_t* thr_key;  // [Bpad] exact-path threshold (worst kept key) ...
    int32_t* thr_row;   // [Bpad] ... and its row
    int* status;        // [Bpad]
    // screen bound and int8-screen state
    float* E;           // [Bpad] exact cosine <= candidate value + E.  bf16 screen: the whole bound; int8: its per-query part
    float* E16;         // [Bpad] the bf16 bound of this query whatever the active screen (second screen inside k_prune)
    float* sc;          // [Bpad] int8 screen: the query's step S_q  (1 for the bf16 screen)
    float* kq;          // [Bpad] int8 screen: factor on a row group's residual norm, 1.0001 + 3 e_q
    int8_t* qhat8;      // [Bpad, dpad8] int8 quantised normalised queries
    int* carry;         // [Bpad] candidates at the head of the list that an earlier prune of this pass carried over (k_prune, defer_b)
};

// bf16 value (upper 16 bits) back to fp32
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// bf16 screen bound from MEASURED residual norms (same algebra as the int8 bound below): q_hat = q16 + e_q, c_hat =
// c16 + e_c exactly, so |q_hat.c_hat - q16.c16| <= |c16||e_q| + |q16||e_c| + |e_q||e_c| <= 1.0001 (e_q + e_c) + 3 e_q e_c;
// + 8 d 2^-24 + 2^-16 for the fp32 normalisation, the MFMA's fp32 accumulation and the exact key's own rounding.
// e_c = the largest residual norm over the stored rows.  Never above the a-priori 2^-7 + 2^-15 + 8 d 2^-24.
// Inner product (round 3): the shadows hold the rows THEMSELVES (not normalised), so the screen estimates <q_hat, c> =
// dot / |q| and its threshold is dot_k / |q| - E: no row norm, no "largest row norm" in the threshold (round 2 screened the
// cosine and asked for cos >= dot_k / (|q| cmax) - E: rows 2 % shorter than the longest one loosened it by 2 % OF THE COSINE,
// an order of magnitude more than E -- every query of the C2 stand-in overflowed its candidate list).  The algebra is the
// same with |c_hat| = 1 replaced by |c| <= cscale (the largest stored row norm, inflated): the residual norms e_c / e_g are
// measured in the rows' own units, the query-side and rounding terms scale with cscale.  cscale = 1 for the cosine metric.
__host__ __device__ inline float bf16_screen_bound(float e_q, float e_c, int d, float cscale = 1.0f) {
    return 1.0001f * (e_q * cscale + e_c) + 3.0f * e_q * e_c + (8.0f * (float)d * 5.9604645e-8f + 1.5258789e-5f) * cscale;
}

// ---- int8 screen quantisation (DESIGN.md "int8 screen bound") ----
// Normalised rows c_hat are quantised per GROUP of kI8GroupRows = 32 consecutive rows (one 32x32 MFMA accumulator block is
// 32 rows x 32 queries, so a block has ONE step and the epilogue reads it with a scalar load): S_g = (largest |component|
// over the group's rows) / 127, c8 = round(c_hat / S_g) -- nothing is clipped, and the residual norm |c_hat - S_g c8| of
// every row is MEASURED at build time; the group keeps e_g = the largest of them (~ S_g sqrt(d/12)).  A unit Gaussian row of
// d = 768 peaks at ~3.5 sigma, a group of 32 at ~4.4 sigma: e_g ~ 0.010 -- against 0.0136 (and the 0.0150 limit the bound
// had to use) for the one corpus-wide 6-sigma step of round 1.  Rows with an outlier component (|component| sqrt(d) > kI8Z)
// would coarsen their whole group: they are "loose" -- left out of the int8 shadow (all-zero row, flag 1) and re-scored for
// every query like irregular rows; with more than kIrrCap of them the index keeps the bf16 screen.
// Queries: one step per query, S_q = max|q_hat| / 127, residual norm e_q measured.
// Bound: q_hat.c_hat - (S_q q8).(S_g c8) = q~.e_c + e_q.c~ + e_q.e_c  ->  |..| <= e_q + e_c + 3 e_q e_c, split into
//   a per-query part   E_q  = 1.0001 e_q + 4 d 2^-24 + 2^-16 + 4e-6        (st.E: what k_prune adds to a candidate's value)
//   a per-pair part    e_g * kq,  kq = 1.0001 + 3 e_q                       (added to the screen value by the epilogue)
// so a candidate carries v = S_q S_g acc + e_g kq and  exact cosine <= v + E_q;  it is emitted iff v >= thr = tau - E_q.
// (4e-6: the epilogue's three fp32 roundings -- |acc| < 2^24 converts exactly below d = 1040, the product S_q S_g, the fma.)
constexpr float kI8Z = 6.0f;
constexpr int kI8GroupRows = 32;
struct I8Group {  // [ceil(rows / 32)]
    float step;   // S_g
    float err;    // e_g (inflated by 1e-3 for its own rounding)
};
__host__ __device__ inline float i8_row_peak_limit(int d) { return kI8Z / sqrtf((float)d); }
__host__ __device__ inline float i8_query_bound(float e_q, int d) {
    return 1.0001f * e_q + 4.0f * (float)d * 5.9604645e-8f + 1.5258789e-5f + 4.0e-6f;
}
__host__ __device__ inline float i8_pair_factor(float e_q) { return 1.0001f + 3.0f * e_q; }
// conservative integer threshold of one accumulator block: every acc with fl(fma((float)acc, m, ek)) >= th has acc >= the
// result (m = S_q S_g >= 0, ek = e_g kq).  Hit path only, for lanes whose own float test passed (callers AND the lane's
// `any` into every compare: a NaN threshold converts to 0 here).  Branch-free: v_cvt_i32_f32 saturates, so -inf (emit
// everything; an all-zero block, m = 0, whose ek passes) becomes INT_MIN.
//   fl(fma) >= th  only if  acc m + ek >= th - |th| 2^-24;   rcp: 1 ulp, the subtraction and the products 1/2 ulp each.
__device__ __forceinline__ int i8_block_threshold(float th, float m, float ek) {
    const float r = __builtin_amdgcn_rcpf(m);
    const float x = (th - ek) * r;
    const float y = floorf(x - fabsf(x) * 4.8e-7f - fabsf(th) * (1.2e-7f * r) - 2.0f);
    int out;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(out) : "v"(y));
    return out;
}

__device__ __forceinline__ float bits_f(uint32_t u) { return __uint_as_float(u); }

// round-to-nearest-even fp32 -> bf16 (finite inputs)
__device__ __forceinline__ uint16_t f32_to_bf16_rn(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// pgvector cosine_distance from the three fp32 accumulators (see oracle.c: orc_cosine_distance_from)
__device__ __forceinline__ double cosine_distance_from(float dot, float nq, float nc) {
    double sim = (double)dot / sqrt((double)nq * (double)nc);
    if (sim > 1.0) sim = 1.0;
    else if (sim < -1.0) sim = -1.0;
    return 1.0 - sim;
}

__device__ __forceinline__ double distance_from(int metric, float dot, float nq, float nc) {
    return metric == 0 ? cosine_distance_from(dot, nq, nc) : (double)dot * -1.0;
}

// "higher is better" image of a distance key, as float (monotone; used only to SELECT, the exact keys decide order):
// cosine: 1 - distance = the similarity; inner product: -distance = the dot product
__device__ __forceinline__ float sim_of_dist(int metric, double dist) {
    return metric == 0 ? (float)(1.0 - dist) : (float)(-dist);
}

// monotone map double -> uint64 so that unsigned compare == (distance asc, NaN last)
__device__ __forceinline__ uint64_t dist_to_key(double d) {
    if (d != d) return kKeyNaN;
    uint64_t b = (uint64_t)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_to_dist(uint64_t k) {
    if (k == kKeyNaN) return __longlong_as_double(0x7FF8000000000000ll);
    uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}

// next representable fp32 below x (x finite or +inf)
__device__ __forceinline__ float float_below(float x) {
    if (x != x || x == -__builtin_inff()) return x;
    uint32_t u = __float_as_uint(x);
    if (x > 0.0f) u -= 1u;
    else if (x < 0.0f) u += 1u;
    else u = 0x80000001u;
    return __uint_as_float(u);
}

// a norm the bf16 screen can normalise safely
__device__ __forceinline__ bool norm_is_regular(float n2) { return n2 >= 1e-30f && n2 <= 1e30f; }

__device__ __forceinline__ int next_pow2(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

// ---- bitonic sorts over LDS arrays (n = power of two, all threads of the block participate) ----
__device__ __forceinline__ void bitonic_desc_f32(float* v, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int p = i ^ j;
                if (p > i) {
                    const bool desc = ((i & k) == 0);
                    const float a = v[i], b = v[p];
                    if (desc ? (a < b) : (a > b)) {
                        v[i] = b;
                        v[p] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ---- one-wave selection helpers (k_prune's small form: every lane holds kSelPerLane values in registers) ----
constexpr int kSelPerLane = 16;

// monotone map float -> uint32 (unsigned compare == float compare, -inf lowest finite image, NaN not expected)
__device__ __forceinline__ uint32_t f32_order_key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// number of keys >= x over the wave's first NS slots per lane (wave-uniform result): NS ballots + scalar popcounts, no
// shuffles.  Entry e of a list lives in slot e / 64 of lane e % 64, so a list of n entries only occupies its first
// ceil(n / 64) slots -- the others hold 0 ("absent", below every real key) and need not be looked at.
template <int NS>
__device__ __forceinline__ int wave_count_ge_n(const uint32_t (&key)[kSelPerLane], uint32_t x) {
    int c = 0;
#pragma unroll
    for (int j = 0; j < NS; ++j) c += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[j] >= x));
    return c;
}
// the n-th largest key among the first NS slots (1 <= n <= number of keys): bitwise construction of the largest
// x with count(key >= x) >= n.  32 rounds of wave_count_ge_n.
template <int NS>
__device__ __forceinline__ uint32_t wave_nth_largest_n(const uint32_t (&key)[kSelPerLane], int n) {
    uint32_t x = 0;
    for (int b = 31; b >= 0; --b) {
        const uint32_t t = x | (1u << b);
        if (wave_count_ge_n<NS>(key, t) >= n) x = t;
    }
    return x;
}
// the same for a list of `entries` entries (wave-uniform): the smallest instantiation that covers its slots.  A prune of
// the headline pass selects among ~250 new candidates, then among ~40 and ~70 exact scores: 4, 1 and 2 slots, not 16.
__device__ __forceinline__ int wave_count_ge(const uint32_t (&key)[kSelPerLane], uint32_t x, int entries) {
    if (entries <= kWave) return wave_count_ge_n<1>(key, x);
    if (entries <= 2 * kWave) return wave_count_ge_n<2>(key, x);
    if (entries <= 4 * kWave) return wave_count_ge_n<4>(key, x);
    if (entries <= 8 * kWave) return wave_count_ge_n<8>(key, x);
    return wave_count_ge_n<kSelPerLane>(key, x);
}
__device__ __forceinline__ uint32_t wave_nth_largest(const uint32_t (&key)[kSelPerLane], int n, int entries) {
    if (entries <= kWave) return wave_nth_largest_n<1>(key, n);
    if (entries <= 2 * kWave) return wave_nth_largest_n<2>(key, n);
    if (entries <= 4 * kWave) return wave_nth_largest_n<4>(key, n);
    if (entries <= 8 * kWave) return wave_nth_largest_n<8>(key, n);
    return wave_nth_largest_n<kSelPerLane>(key, n);
}

__device__ __forceinline__ int wave_count_ge(const uint32_t (&key)[kSelPerLane], uint32_t x) {
    return wave_count_ge_n<kSelPerLane>(key, x);
}
__device__ __forceinline__ uint32_t wave_nth_largest(const uint32_t (&key)[kSelPerLane], int n) {
    return wave_nth_largest_n<kSelPerLane>(key, n);
}

// descending sort of (value, payload) pairs
__device__ __forceinline__ void bitonic_desc_f32_i32(float* v, int32_t* w, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int p = i ^ j;
                if (p > i) {
                    const bool desc = ((i & k) == 0);
                    const float a = v[i], b = v[p];
                    if (desc ? (a < b) : (a > b)) {
                        v[i] = b;
                        v[p] = a;
                        const int32_t t = w[i];
                        w[i] = w[p];
                        w[p] = t;
                    }
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ bool key_before(uint64_t k1, int32_t r1, uint64_t k2, int32_t r2) {
    return k1 < k2 || (k1 == k2 && r1 < r2);
}

__device__ __forceinline__ void bitonic_asc_key_row(uint64_t* key, int32_t* row, int n) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int p = i ^ j;
                if (p > i) {
                    const bool asc = ((i & k) == 0);
                    const uint64_t ka = key[i], kb = key[p];
                    const int32_t ra = row[i], rb = row[p];
                    const bool a_after_b = key_before(kb, rb, ka, ra);
                    if (asc ? a_after_b : key_before(ka, ra, kb, rb)) {
                        key[i] = kb;
                        key[p] = ka;
                        row[i] = rb;
                        row[p] = ra;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ---- one-wave staged fp32 chains ---------------------------------------------------------------
// 64 lanes own 64 "slots" (rows).  Rows are pulled from global memory 64 columns at a time with
// row-contiguous (coalesced) loads into a padded LDS tile, then every lane walks its own row of the
// tile in k order, so each lane's accumulator is exactly the k-ascending fmaf chain.
constexpr int kStageCols = 64;
constexpr int kStageLd = 68;  // row stride in floats: rows stay 16-B aligned and lane l's ds_read_b128 of row l
                              // lands on 16-B slot (17*l + k/4) mod 16 -> conflict-free within the b128 lane groups
constexpr int kStageFloats = kWave * kStageLd;

// wave-level LDS hand-off (writes by some lanes -> reads by others of the SAME wave)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The row pointers of a batch travel between lanes as integers (__shfl), so the compiler no longer knows their address
// space and would gather with FLAT loads -- which count on lgkmcnt as well as vmcnt: every LDS wait of the chain then also
// waits for the pieces in flight and the prefetch is worth nothing.  Rows live in global memory: say so.
typedef float f32x4_native __attribute__((ext_vector_type(4)));
typedef const f32x4_native __attribute__((address_space(1)))* gmem_f4;
__device__ __forceinline__ float4 load_gmem_f4(const float* p) {
    const f32x4_native v = *(gmem_f4)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

// One 64-column piece of the 64 rows, held in registers between issue and commit so the HBM/L2 latency
// of piece p+1 (and p+2) overlaps the chain over piece p.
struct StagePiece {
    float4 v[16];
};
// The 16 row pointers this lane loads from (rows 4g + (lane>>4), g = 0..15): fetched from the owning lanes
// once per batch of 64 rows, reused for every 64-column piece.
struct StageRows {
    const float* r[16];
};
__device__ __forceinline__ void stage_rows_init(StageRows& sr, const float* my_row, int lane) {
    const unsigned long long a = (unsigned long long)my_row;
    const unsigned plo = (unsigned)(a & 0xFFFFFFFFull), phi = (unsigned)(a >> 32);
    const int sub = lane >> 4;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const unsigned lo = __shfl(plo, g * 4 + sub, kWave), hi = __shfl(phi, g * 4 + sub, kWave);
        sr.r[g] = (const float*)(((unsigned long long)hi << 32) | lo);
    }
}

// The same, with idle slots (nullptr) pointed at a live lane's row: every lane then loads from a valid address and a piece
// that lies inside the rows (k0 + 64 <= d) needs no per-load predicate -- 16 x (address add + load) instead of 16 x
// (compare, exec mask, branch, add, load, restore).  At one wave per SIMD the issue of these instructions, not the memory,
// is what a piece costs.  Returns false when no lane has a row (the caller keeps the predicated path).
__device__ __forceinline__ bool stage_rows_init_dense(StageRows& sr, const float* my_row, int lane) {
    const unsigned long long live = __builtin_amdgcn_ballot_w64(my_row != nullptr);
    if (live == 0) {
        stage_rows_init(sr, my_row, lane);
        return false;
    }
    const int src = __builtin_ctzll(live);
    unsigned long long a = (unsigned long long)my_row;
    unsigned plo = (unsigned)(a & 0xFFFFFFFFull), phi = (unsigned)(a >> 32);
    const unsigned flo = __shfl(plo, src, kWave), fhi = __shfl(phi, src, kWave);
    if (my_row == nullptr) {
        plo = flo;
        phi = fhi;
    }
    const int sub = lane >> 4;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const unsigned lo = __shfl(plo, g * 4 + sub, kWave), hi = __shfl(phi, g * 4 + sub, kWave);
        sr.r[g] = (const float*)(((unsigned long long)hi << 32) | lo);
    }
    return true;
}
__device__ __forceinline__ void stage_issue_dense(StagePiece& p, const StageRows& sr, int k0, int lane) {
    const int c4 = (lane & 15) * 4;
#pragma unroll
    for (int g = 0; g < 16; ++g) p.v[g] = load_gmem_f4(sr.r[g] + k0 + c4);
}

// d % 4 == 0: lane (sub = lane>>4, c4 = 4*(lane&15)) loads 16 B of row 4g+sub for g = 0..15 -- one
// global_load_dwordx4 covers 4 rows x 256 B, all 16 loads are independent and issued back to back.
__device__ __forceinline__ void stage_issue(StagePiece& p, const StageRows& sr, int k0, int d, int lane) {
    const int c4 = (lane & 15) * 4;
#pragma unroll
    for (int g = 0; g < 16; ++g)
        p.v[g] = (sr.r[g] != nullptr && k0 + c4 < d) ? load_gmem_f4(sr.r[g] + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ void stage_commit(float* tile, const StagePiece& p, int lane) {
    wave_sync();  // previous readers of the tile are done
    const int sub = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
    for (int g = 0; g < 16; ++g) *(float4*)(tile + (g * 4 + sub) * kStageLd + c4) = p.v[g];
    wave_sync();  // tile visible to every lane of this wave
}

// generic (any d) one-shot staging: scalar loads when rows are not 16-B aligned
__device__ __forceinline__ void stage_rows(float* tile, const float* my_row, int k0, int d, int lane) {
    if ((d & 3) == 0) {
        StageRows sr;
        StagePiece p;
        stage_rows_init(sr, my_row, lane);
        stage_issue(p, sr, k0, d, lane);
        stage_commit(tile, p, lane);
        return;
    }
    wave_sync();
    const unsigned long long a = (unsigned long long)my_row;
    const unsigned plo = (unsigned)(a & 0xFFFFFFFFull), phi = (unsigned)(a >> 32);
#pragma unroll 8
    for (int s = 0; s < kWave; ++s) {
        const unsigned lo = __shfl(plo, s, kWave), hi = __shfl(phi, s, kWave);
        const float* r = (const float*)(((unsigned long long)hi << 32) | lo);
        const int k = k0 + lane;
        tile[s * kStageLd + lane] = (r != nullptr && k < d) ? r[k] : 0.0f;
    }
    wave_sync();
}

// acc = k-ascending fmaf chain over columns [0,kn) of this lane's tile row against qv[0..kn) (qv in LDS, 16-B aligned)
__device__ __forceinline__ float chain_piece(const float* tile, int lane, const float* qv, int kn, float acc) {
    const float* t = tile + lane * kStageLd;
    if (kn == kStageCols) {  // full piece: all 32 ds_read_b128 issued up front, then the 64-step chain
        float4 tv[16], q4[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            tv[i] = *(const float4*)(t + 4 * i);
            q4[i] = *(const float4*)(qv + 4 * i);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc = __builtin_fmaf(tv[i].x, q4[i].x, acc);
            acc = __builtin_fmaf(tv[i].y, q4[i].y, acc);
            acc = __builtin_fmaf(tv[i].z, q4[i].z, acc);
            acc = __builtin_fmaf(tv[i].w, q4[i].w, acc);
        }
        return acc;
    }
    int k = 0;
    for (; k + 4 <= kn; k += 4) {
        const float4 tv = *(const float4*)(t + k);
        const float4 q4 = *(const float4*)(qv + k);
        acc = __builtin_fmaf(tv.x, q4.x, acc);
        acc = __builtin_fmaf(tv.y, q4.y, acc);
        acc = __builtin_fmaf(tv.z, q4.z, acc);
        acc = __builtin_fmaf(tv.w, q4.w, acc);
    }
    for (; k < kn; ++k) acc = __builtin_fmaf(t[k], qv[k], acc);
    return acc;
}

// full chain <row, qv> for this lane's row (nullptr = idle slot); rows prefetched DEPTH pieces ahead when d%4==0 (2: 128
// registers of pieces in flight; 3: 192).  Measured round 6 (-DMI355_STAGE_DEPTH=3, profiles/r06_stage_depth_ab.txt): a third
// piece in flight changes neither prune (k = 10: 6.246 / 6.241 ms per pass, k = 100: 7.68 / 7.69) -- a prune is a chain of ~25
// dependent steps of ~1 us each (counts, keys, rows, the pieces, norms, kept list, sort, publish), not a wait for the pieces.
#ifndef MI355_STAGE_DEPTH
#define MI355_STAGE_DEPTH 2
#endif
template <int DEPTH = MI355_STAGE_DEPTH>
__device__ __forceinline__ float staged_dot(float* tile, const float* my_row, const float* qv, int d, int lane) {
    static_assert(DEPTH == 2 || DEPTH == 3, "pieces in flight");
    float acc = 0.0f;
    if ((d & 3) == 0) {
        StageRows sr;
        StagePiece p0, p1, p2;
        const bool dense = stage_rows_init_dense(sr, my_row, lane);
        auto issue = [&](StagePiece& p, int k0) __attribute__((always_inline)) {
            if (dense && k0 + kStageCols <= d) stage_issue_dense(p, sr, k0, lane);  // wave-uniform
            else stage_issue(p, sr, k0, d, lane);
        };
        issue(p0, 0);
        if (kStageCols < d) issue(p1, kStageCols);
        if constexpr (DEPTH == 3) {
            if (2 * kStageCols < d) issue(p2, 2 * kStageCols);
        }
        for (int k0 = 0; k0 < d; k0 += DEPTH * kStageCols) {
            stage_commit(tile, p0, lane);
            if (k0 + DEPTH * kStageCols < d) issue(p0, k0 + DEPTH * kStageCols);
            acc = chain_piece(tile, lane, qv + k0, min(kStageCols, d - k0), acc);
            if (k0 + kStageCols < d) {
                stage_commit(tile, p1, lane);
                if (k0 + (DEPTH + 1) * kStageCols < d) issue(p1, k0 + (DEPTH + 1) * kStageCols);
                acc = chain_piece(tile, lane, qv + k0 + kStageCols, min(kStageCols, d - k0 - kStageCols), acc);
            }
            if constexpr (DEPTH == 3) {
                if (k0 + 2 * kStageCols < d) {
                    stage_commit(tile, p2, lane);
                    if (k0 + 5 * kStageCols < d) issue(p2, k0 + 5 * kStageCols);
                    acc = chain_piece(tile, lane, qv + k0 + 2 * kStageCols, min(kStageCols, d - k0 - 2 * kStageCols), acc);
                }
            }
        }
    } else {
        for (int k0 = 0; k0 < d; k0 += kStageCols) {
            stage_rows(tile, my_row, k0, d, lane);
            acc = chain_piece(tile, lane, qv + k0, min(kStageCols, d - k0), acc);
        }
    }
    return acc;
}

// Same staging for bf16 rows: a row of 2*words bf16 values is moved as `words` 32-bit words; word w holds elements
// 2w (low half) and 2w+1 (high half).  qv: the bf16 query expanded to fp32 in LDS, element order.  fp32 accumulation of
// exact products in k order (what the bf16 MFMA screen computes up to the order of its fp32 sums).
__device__ __forceinline__ float chain_piece16(const float* tile, int lane, const float* qv, int kn_words, float acc) {
    const float* t = tile + lane * kStageLd;
    for (int w = 0; w + 4 <= kn_words; w += 4) {
        const uint4 tv = *(const uint4*)(t + w);
        const float4 qa = *(const float4*)(qv + 2 * w), qb = *(const float4*)(qv + 2 * w + 4);
        acc = __builtin_fmaf(__uint_as_float(tv.x << 16), qa.x, acc);
        acc = __builtin_fmaf(__uint_as_float(tv.x & 0xFFFF0000u), qa.y, acc);
        acc = __builtin_fmaf(__uint_as_float(tv.y << 16), qa.z, acc);
        acc = __builtin_fmaf(__uint_as_float(tv.y & 0xFFFF0000u), qa.w, acc);
        acc = __builtin_fmaf(__uint_as_float(tv.z << 16), qb.x, acc);
        acc = __builtin_fmaf(__uint_as_float(tv.z & 0xFFFF0000u), qb.y, acc);
        acc = __builtin_fmaf(__uint_as_float(tv.w << 16), qb.z, acc);
        acc = __builtin_fmaf(__uint_as_float(tv.w & 0xFFFF0000u), qb.w, acc);
    }
    return acc;
}
// words must be a multiple of 4 (shadow rows are padded to 64 elements)
__device__ __forceinline__ float staged_dot16(float* tile, const float* my_row_words, const float* qv, int words, int lane) {
    float acc = 0.0f;
    StageRows sr;
    StagePiece p0, p1;
    stage_rows_init(sr, my_row_words, lane);
    stage_issue(p0, sr, 0, words, lane);
    if (kStageCols < words) stage_issue(p1, sr, kStageCols, words, lane);
    for (int k0 = 0; k0 < words; k0 += 2 * kStageCols) {
        stage_commit(tile, p0, lane);
        if (k0 + 2 * kStageCols < words) stage_issue(p0, sr, k0 + 2 * kStageCols, words, lane);
        acc = chain_piece16(tile, lane, qv + 2 * k0, min(kStageCols, words - k0), acc);
        if (k0 + kStageCols < words) {
            stage_commit(tile, p1, lane);
            if (k0 + 3 * kStageCols < words) stage_issue(p1, sr, k0 + 3 * kStageCols, words, lane);
            acc = chain_piece16(tile, lane, qv + 2 * (k0 + kStageCols), min(kStageCols, words - k0 - kStageCols), acc);
        }
    }
    return acc;
}

}  // namespace mi355
