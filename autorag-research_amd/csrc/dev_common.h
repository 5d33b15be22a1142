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
};

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
constexpr int kStageLd = 65;  // +1 pad: lane l reads word l*65+k -> conflict-free for ds_read_b32
constexpr int kStageFloats = kWave * kStageLd;

// wave-level LDS hand-off (writes by some lanes -> reads by others of the SAME wave)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// stage columns [k0, k0+64) of the 64 rows `my_row(lane)` (nullptr = slot unused -> zeros).
// d % 4 == 0 (16-B aligned row pieces): one global_load_dwordx4 per lane covers 4 rows x 256 B per
// wave instruction, 16 independent loads are issued back to back before the first LDS write, so the
// HBM latency is paid once per 64-column piece instead of once per row.
__device__ __forceinline__ void stage_rows(float* tile, const float* my_row, int k0, int d, int lane) {
    wave_sync();  // previous readers of the tile are done
    const unsigned long long p = (unsigned long long)my_row;
    const unsigned plo = (unsigned)(p & 0xFFFFFFFFull), phi = (unsigned)(p >> 32);
    if ((d & 3) == 0) {
        const int sub = lane >> 4;        // row within the group of 4
        const int c4 = (lane & 15) * 4;   // first of this lane's 4 columns
        float4 v[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int s = g * 4 + sub;
            const unsigned lo = __shfl(plo, s, kWave), hi = __shfl(phi, s, kWave);
            const float* r = (const float*)(((unsigned long long)hi << 32) | lo);
            v[g] = (r != nullptr && k0 + c4 < d) ? *(const float4*)(r + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            float* t = tile + (g * 4 + sub) * kStageLd + c4;
            t[0] = v[g].x;
            t[1] = v[g].y;
            t[2] = v[g].z;
            t[3] = v[g].w;
        }
    } else {
#pragma unroll 8
        for (int s = 0; s < kWave; ++s) {
            const unsigned lo = __shfl(plo, s, kWave), hi = __shfl(phi, s, kWave);
            const float* r = (const float*)(((unsigned long long)hi << 32) | lo);
            const int k = k0 + lane;
            tile[s * kStageLd + lane] = (r != nullptr && k < d) ? r[k] : 0.0f;
        }
    }
    wave_sync();  // tile visible to every lane of this wave
}

}  // namespace mi355
