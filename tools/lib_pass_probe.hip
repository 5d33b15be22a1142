// lib_pass_probe.hip -- the library's pass in a PLAIN process (no Python, no torch): what k_screen_rq's launches cost when
// libmi355dr.so is driven by a 100-line C++ program, next to what bench.py measures and to what tools/screen_ab measures for the
// same kernel on its own operands.  Corpus: N x d unit Gaussian rows generated on the device (hash + Box-Muller) and handed
// over by pointer; queries: B Gaussian rows.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/lib_pass_probe.hip -Lautorag_research_amd -lmi355dr
//              -Wl,-rpath,'$ORIGIN/../../autorag_research_amd' -o tools/bin/lib_pass_probe
// run:   tools/bin/lib_pass_probe <rows> <queries> <dim> <steps> [park_rows] [extra_GiB]
//        park_rows > 0: k_screen_rq launches of at least that many rows run with thresholds at +inf (results wrong: timing only)
//        extra_GiB: device memory allocated AND touched before the index is built (the footprint of a larger process)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mi355dr.h"

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e = (x);                                                                \
        if (e != hipSuccess) {                                                             \
            fprintf(stderr, "%s failed: %s (%d)\n", #x, hipGetErrorString(e), __LINE__);   \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)
#define LK(x)                                                                              \
    do {                                                                                   \
        int rc = (x);                                                                      \
        if (rc != 0) {                                                                     \
            fprintf(stderr, "%s failed: %d %s (%d)\n", #x, rc, idx ? mi355dr_last_error(idx) : "", __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

__device__ inline uint32_t hash32(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return (uint32_t)x;
}
__global__ void k_gauss(float* p, int64_t n, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = hash32(i * 2 + seed), h2 = hash32(i * 2 + 1 + seed);
    const float u1 = ((float)h + 1.0f) * (1.0f / 4294967296.0f), u2 = (float)h2 * (1.0f / 4294967296.0f);
    p[i] = sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
}
__global__ void k_touch(char* p, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4096;
    if (i < n) p[i] = 1;
}

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 10000000;
    const int B = argc > 2 ? atoi(argv[2]) : 1024;
    const int d = argc > 3 ? atoi(argv[3]) : 768;
    const int steps = argc > 4 ? atoi(argv[4]) : 20;
    const int64_t park = argc > 5 ? atoll(argv[5]) : 0;
    const double extra_gib = argc > 6 ? atof(argv[6]) : 0.0;
    mi355dr_index* idx = nullptr;
    if (extra_gib > 0) {
        char* extra = nullptr;
        const size_t bytes = (size_t)(extra_gib * 1073741824.0);
        CK(hipMalloc(&extra, bytes));
        hipLaunchKernelGGL(k_touch, dim3((unsigned)((bytes / 4096 + 255) / 256)), dim3(256), 0, 0, extra, bytes);
        CK(hipDeviceSynchronize());
        printf("extra %.1f GiB allocated and touched\n", extra_gib);
    }
    LK(mi355dr_create(&idx, 0, d, MI355DR_METRIC_COSINE));
    LK(mi355dr_reserve(idx, N));
    const int64_t chunk = 500000;
    float* buf = nullptr;
    CK(hipMalloc(&buf, (size_t)chunk * d * sizeof(float)));
    for (int64_t r0 = 0; r0 < N; r0 += chunk) {
        const int64_t n = std::min(chunk, N - r0);
        hipLaunchKernelGGL(k_gauss, dim3((unsigned)((n * d + 255) / 256)), dim3(256), 0, 0, buf, n * d, (uint64_t)r0 * d * 2 + 17);
        CK(hipDeviceSynchronize());
        LK(mi355dr_add_rows_device(idx, buf, n));
    }
    float* q = nullptr;
    double* od = nullptr;
    int64_t* orow = nullptr;
    const int k = 10;
    CK(hipMalloc(&q, (size_t)B * d * sizeof(float)));
    CK(hipMalloc(&od, (size_t)B * k * sizeof(double)));
    CK(hipMalloc(&orow, (size_t)B * k * sizeof(int64_t)));
    hipLaunchKernelGGL(k_gauss, dim3((unsigned)(((size_t)B * d + 255) / 256)), dim3(256), 0, 0, q, (int64_t)B * d, 987654321ull);
    CK(hipDeviceSynchronize());
    LK(mi355dr_set_option(idx, "profile", 1));
    if (park > 0) LK(mi355dr_set_option(idx, "debug_park_thresholds", park));
    for (int w = 0; w < 3; ++w) LK(mi355dr_search_device(idx, q, B, k, od, orow, nullptr));
    LK(mi355dr_synchronize(idx));
    LK(mi355dr_reset_stats(idx));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int s = 0; s < steps; ++s) LK(mi355dr_search_device(idx, q, B, k, od, orow, nullptr));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    LK(mi355dr_synchronize(idx));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    int64_t ns = 0, launches = 0, rows = 0, rq = 0, fb = 0;
    LK(mi355dr_get_stat(idx, "screen256_ns", &ns));
    LK(mi355dr_get_stat(idx, "screen256_launches", &launches));
    LK(mi355dr_get_stat(idx, "screen256_rows", &rows));
    LK(mi355dr_get_stat(idx, "screen_rq_launches", &rq));
    LK(mi355dr_get_stat(idx, "fallback_queries", &fb));
    printf("rows %lld queries %d dim %d steps %d park %lld: %.3f ms per step (wall, events) | large-block screen launches: %lld (%lld k_screen_rq), "
           "%.3f ms per step, %.4f ns per row x 1024 queries, %.0f TOP/s | fallback queries %lld\n",
           (long long)N, B, d, steps, (long long)park, ms / steps, (long long)launches, (long long)rq, ns * 1e-6 / steps,
           (double)ns / (double)rows, 2.0 * (double)rows * B * d / ((double)ns * 1e-9) / 1e12, (long long)fb);
    return 0;
}
