// tools/screen_bench.hip -- developer micro-benchmark of the screen kernels (not part of the library).
// Builds: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Iautorag_research_amd/csrc -Itools/forms tools/screen_bench.hip -o gpurun_out/screen_bench
// Runs each variant on synthetic bf16 rows: (1) pure compute (thresholds = +inf), (2) with a finite
// threshold, comparing the appended candidate sets between variants.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dev_common.h"
#include "k_screen.h"
#include "k_screen256.h"
#include "k_screen256d.h"

using namespace mi355;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e = (x);                                                                \
        if (e != hipSuccess) {                                                             \
            fprintf(stderr, "%s failed: %s (%d)\n", #x, hipGetErrorString(e), __LINE__);   \
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
// rows ~ N(0, 1/d) per element (sum of 4 uniforms), bf16
__global__ void k_fill(uint16_t* p, int64_t rows, int dpad, int d, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * dpad) return;
    const int k = (int)(i % dpad);
    float v = 0.f;
    if (k < d) {
        uint32_t h = hash32(i * 4 + seed), h2 = hash32(i * 4 + 1 + seed);
        float u = ((h & 0xFFFF) + (h >> 16) + (h2 & 0xFFFF) + (h2 >> 16)) * (1.0f / 65536.0f) - 2.0f;  // var 1/3
        v = u * sqrtf(3.0f / d);
    }
    p[i] = f32_to_bf16_rn(v);
}
// int8 data patterns (DATA env): does the matrix pipe's power -- the chip runs this kernel AT its power cap -- depend on the
// operand values?  1 = Gaussian sigma 29 clipped to +-127 (what the library's shadows hold), 2 = zeros, 3 = Gaussian sigma 4,
// 4 = |Gaussian| sigma 29 (non-negative), 5 = uniform full range, 6 = Gaussian sigma 29 with the low 2 bits cleared
__global__ void k_fill8(int8_t* p, int64_t n, int mode, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h = hash32(i * 4 + seed), h2 = hash32(i * 4 + 1 + seed);
    const float u = ((h & 0xFFFF) + (h >> 16) + (h2 & 0xFFFF) + (h2 >> 16)) * (1.0f / 65536.0f) - 2.0f;  // var 1/3
    const float g = u * sqrtf(3.0f);
    int v = 0;
    if (mode == 1) v = (int)rintf(fminf(fmaxf(g * 29.0f, -127.f), 127.f));
    else if (mode == 3) v = (int)rintf(g * 4.0f);
    else if (mode == 4) v = (int)rintf(fminf(fabsf(g) * 29.0f, 127.f));
    else if (mode == 5) v = (int)(h & 0xFF) - 128;
    else if (mode == 6) v = ((int)rintf(fminf(fmaxf(g * 29.0f, -127.f), 127.f))) & ~3;
    p[i] = (int8_t)v;
}
__global__ void k_fillf(float* p, int n, float v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

struct Cand {
    int q, row;
    float v;
    bool operator<(const Cand& o) const { return q != o.q ? q < o.q : row < o.row; }
};

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : (1 << 20);
    const int B = argc > 2 ? atoi(argv[2]) : 1024;
    const int d = argc > 3 ? atoi(argv[3]) : 768;
    const int reps = argc > 4 ? atoi(argv[4]) : 5;
    const int only = argc > 5 ? atoi(argv[5]) : 0;  // run a single variant (profiling)
    const int dpad = (d + 63) / 64 * 64;
    const int64_t Npad = (N + 255) / 256 * 256;
    const int Bpad = (B + 255) / 256 * 256;
    const int cap = 2048;
    uint16_t *shadow, *qhat;
    float *thr, *cval;
    int *cnt;
    int32_t* crow;
    CK(hipMalloc(&shadow, (size_t)Npad * dpad * 2));
    CK(hipMalloc(&qhat, (size_t)Bpad * dpad * 2));
    CK(hipMalloc(&thr, Bpad * 4));
    CK(hipMalloc(&cnt, Bpad * 4));
    CK(hipMalloc(&crow, (size_t)Bpad * cap * 4));
    CK(hipMalloc(&cval, (size_t)Bpad * cap * 4));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((Npad * dpad + 255) / 256)), dim3(256), 0, 0, shadow, Npad, dpad, d, 1234ull);
    hipLaunchKernelGGL(k_fill, dim3((unsigned)(((int64_t)Bpad * dpad + 255) / 256)), dim3(256), 0, 0, qhat, (int64_t)Bpad, dpad, d, 99ull);
    if (getenv("DATA")) {
        const int mode = atoi(getenv("DATA"));
        const int64_t nb = (int64_t)Npad * dpad * 2, nq = (int64_t)Bpad * dpad * 2;
        hipLaunchKernelGGL(k_fill8, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, 0, (int8_t*)shadow, nb, mode, 1234ull);
        hipLaunchKernelGGL(k_fill8, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, (int8_t*)qhat, nq, getenv("QDATA") ? atoi(getenv("QDATA")) : mode, 99ull);
    }
    CK(hipDeviceSynchronize());
    CK(hipFuncSetAttribute((const void*)k_screen<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreenLds));
    CK(hipFuncSetAttribute((const void*)k_screen<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreenLds));
    CK(hipFuncSetAttribute((const void*)k_screen256<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreen256Lds));
    CK(hipFuncSetAttribute((const void*)k_screen256<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreen256Lds));
    CK(hipFuncSetAttribute((const void*)k_screen256<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreen256Lds));
    CK(hipFuncSetAttribute((const void*)k_screen256<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreen256Lds));
    CK(hipFuncSetAttribute((const void*)k_screen256<3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreen256Lds));
    CK(hipFuncSetAttribute((const void*)k_screen256<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreen256Lds));
    CK(hipFuncSetAttribute((const void*)k_screen256b<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreen256Lds));
    CK(hipFuncSetAttribute((const void*)k_screen256b<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreen256Lds));
#define SB_FORMS(X) X(3136) X(7232)
#define SB_ATTR(A) CK(hipFuncSetAttribute((const void*)k_screen256b<A, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreen256Lds));
    SB_FORMS(SB_ATTR)
#define SC_FORMS(X) X(1024) X(1152) X(1280) X(1408) X(3072) X(1040) X(1041) X(1032) X(1056) X(1064) X(1049) X(5120)
#define SC_ATTR(A) CK(hipFuncSetAttribute((const void*)k_screen256c<A, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreen256Lds));
    SC_FORMS(SC_ATTR)
#define SD_FORMS(X) X(0) X(16) X(32)
#define SD_ATTR(A) CK(hipFuncSetAttribute((const void*)k_screen256d<A, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreen256Lds));
    SD_FORMS(SD_ATTR)
    CK(hipFuncSetAttribute((const void*)k_screen256d<kScreen256dAbl, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreen256Lds));
    CK(hipFuncSetAttribute((const void*)k_screen256c<kScreen256cAbl, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreen256Lds));
    CK(hipFuncSetAttribute((const void*)k_screen256c<kScreen256cAbl, false>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreen256Lds));
    int* status;
    CK(hipMalloc(&status, Bpad * 4));
    CK(hipMemset(status, 0, Bpad * 4));
    // int8 variants (1128 / 1256) reuse the same buffers as raw bytes: rows of dpad8 = round_up(d,128) int8.  The
    // harness bytes are uniform in [-128,127] -> acc ~ N(0, d * 5461^2): with S_q = 1 and one group step 1 / (d * 5461)
    // for every row group the int8 screen value is N(0, 1/d) like the bf16 one, so both kinds share the float thresholds
    const int dpad8 = (d + 127) / 128 * 128;
    float *scv, *kqv;
    uint8_t* flag8;
    I8Group* grp;
    CK(hipMalloc(&scv, Bpad * 4));
    CK(hipMalloc(&kqv, Bpad * 4));
    CK(hipMalloc(&flag8, Npad));
    CK(hipMemset(flag8, 0, Npad));
    CK(hipMalloc(&grp, (size_t)(Npad / 32) * sizeof(I8Group)));
    {
        std::vector<float> one(Bpad, 1.0f);
        CK(hipMemcpy(scv, one.data(), Bpad * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(kqv, one.data(), Bpad * 4, hipMemcpyHostToDevice));
        std::vector<I8Group> hg((size_t)(Npad / 32), I8Group{1.0f / ((float)d * 5461.0f), 0.0f});
        CK(hipMemcpy(grp, hg.data(), hg.size() * sizeof(I8Group), hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    auto launch = [&](int variant) {
        ScreenArgs2 sa{};
        sa.status = status;
        sa.shadow = shadow;
        sa.qhat = qhat;
        sa.thr = thr;
        sa.cnt = cnt;
        sa.cand_row = crow;
        sa.cand_val = cval;
        const bool i8 = variant >= 1000 && variant != 200000 && variant != 300000;  // (x00000 = bf16, x01000 + ABL = int8)
        if (i8) variant -= 1000;
        sa.sc = scv;
        sa.kq = kqv;
        sa.grp = grp;
        sa.flag8 = flag8;
        sa.row_bytes = i8 ? dpad8 : dpad * 2;
        sa.ksteps = sa.row_bytes / 128;
        sa.cap = cap;
        sa.ct0 = 0;
        sa.row_end = N;
        if (variant == 128) {  // (ScreenArgs2 slices to ScreenArgs for the first-form kernels)
            sa.n_ctiles = (int)((N + 127) / 128);
            sa.n_qtiles = (B + 127) / 128;
            const int64_t grid = (int64_t)((sa.n_ctiles + 7) / 8 * 8) * sa.n_qtiles;
            if (i8) hipLaunchKernelGGL(k_screen<true>, dim3((unsigned)grid), dim3(256), kScreenLds, 0, (ScreenArgs)sa);
            else hipLaunchKernelGGL(k_screen<false>, dim3((unsigned)grid), dim3(256), kScreenLds, 0, (ScreenArgs)sa);
        } else {
            sa.n_ctiles = (int)((N + 255) / 256);
            sa.n_qtiles = (B + 255) / 256;
            int64_t grid = screen256_grid(sa.n_ctiles, sa.n_qtiles);
            if (getenv("GRIDDIV")) grid = grid / atoi(getenv("GRIDDIV")) / 8 / sa.n_qtiles * 8 * sa.n_qtiles;  // fewer CUs busy
            if (variant >= 300000) {  // fourth form (k_screen256d): 300000 + ABL (+1000 for int8)
                const int abl = variant - 300000;
                if (!i8) hipLaunchKernelGGL((k_screen256d<kScreen256dAbl, false>), dim3((unsigned)grid), dim3(512), kScreen256Lds, 0, sa);
#define SD_LAUNCH(A) else if (abl == A) hipLaunchKernelGGL((k_screen256d<A, true>), dim3((unsigned)grid), dim3(512), kScreen256Lds, 0, sa);
                SD_FORMS(SD_LAUNCH)
                else hipLaunchKernelGGL((k_screen256d<kScreen256dAbl, true>), dim3((unsigned)grid), dim3(512), kScreen256Lds, 0, sa);
            } else if (variant >= 200000) {  // third form (k_screen256c): 200000 + ABL (+1000 for int8)
                const int abl = variant - 200000;
                if (!i8) hipLaunchKernelGGL((k_screen256c<kScreen256cAbl, false>), dim3((unsigned)grid), dim3(512), kScreen256Lds, 0, sa);
#define SC_LAUNCH(A) else if (abl == A) hipLaunchKernelGGL((k_screen256c<A, true>), dim3((unsigned)grid), dim3(512), kScreen256Lds, 0, sa);
                SC_FORMS(SC_LAUNCH)
                else hipLaunchKernelGGL((k_screen256c<kScreen256cAbl, true>), dim3((unsigned)grid), dim3(512), kScreen256Lds, 0, sa);
            } else if (variant >= 300) {  // second form (k_screen256b): 300 + ABL
                const int abl = variant - 300;
                if (!i8) hipLaunchKernelGGL((k_screen256b<0, false>), dim3((unsigned)grid), dim3(512), kScreen256Lds, 0, sa);
#define SB_LAUNCH(A) else if (abl == A) hipLaunchKernelGGL((k_screen256b<A, true>), dim3((unsigned)grid), dim3(512), kScreen256Lds, 0, sa);
                SB_FORMS(SB_LAUNCH)
                else hipLaunchKernelGGL((k_screen256b<0, true>), dim3((unsigned)grid), dim3(512), kScreen256Lds, 0, sa);
            } else if (i8) {
                hipLaunchKernelGGL((k_screen256<0, true>), dim3((unsigned)grid), dim3(512), kScreen256Lds, 0, (ScreenArgs)sa);
            } else switch (variant - 256) {
                case 0: hipLaunchKernelGGL((k_screen256<0, false>), dim3((unsigned)grid), dim3(512), kScreen256Lds, 0, (ScreenArgs)sa); break;
                case 1: hipLaunchKernelGGL((k_screen256<1, false>), dim3((unsigned)grid), dim3(512), kScreen256Lds, 0, (ScreenArgs)sa); break;
                case 2: hipLaunchKernelGGL((k_screen256<2, false>), dim3((unsigned)grid), dim3(512), kScreen256Lds, 0, (ScreenArgs)sa); break;
                case 3: hipLaunchKernelGGL((k_screen256<3, false>), dim3((unsigned)grid), dim3(512), kScreen256Lds, 0, (ScreenArgs)sa); break;
                default: hipLaunchKernelGGL((k_screen256<4, false>), dim3((unsigned)grid), dim3(512), kScreen256Lds, 0, (ScreenArgs)sa); break;
            }
        }
        CK(hipGetLastError());
    };

    std::vector<int> variants = {1256, 4372, 4404, 4436, 4440};
    if (getenv("VARIANTS")) {
        variants.clear();
        for (char* tok = strtok(getenv("VARIANTS"), ","); tok; tok = strtok(nullptr, ",")) variants.push_back(atoi(tok));
    }
    const double flops = 2.0 * B * (double)N * d;
    if (getenv("ROUNDS")) {
        // interleaved A/B (the chip is power-limited: the first launches after idle run ~10 % faster than sustained ones, and
        // boxes differ): ROUNDS rounds, every variant once per round back to back, thresholds parked; median / mean per variant
        const int rounds = atoi(getenv("ROUNDS"));
        hipLaunchKernelGGL(k_fillf, dim3((Bpad + 255) / 256), dim3(256), 0, 0, thr, Bpad, INFINITY);
        std::vector<std::vector<float>> ms(variants.size());
        for (int w = 0; w < 3; ++w)
            for (int variant : variants) launch(variant);  // warm-up
        CK(hipDeviceSynchronize());
        for (int r = 0; r < rounds; ++r)
            for (size_t v = 0; v < variants.size(); ++v) {
                CK(hipEventRecord(e0));
                launch(variants[v]);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float t;
                CK(hipEventElapsedTime(&t, e0, e1));
                ms[v].push_back(t);
            }
        for (size_t v = 0; v < variants.size(); ++v) {
            std::vector<float> x = ms[v];
            std::sort(x.begin(), x.end());
            double mean = 0;
            for (float t : x) mean += t;
            mean /= x.size();
            printf("variant %4d: median %.3f ms (%.0f TOP/s)  mean %.3f ms (%.0f)  min %.3f  max %.3f   [%d interleaved rounds]\n", variants[v],
                   x[x.size() / 2], flops / x[x.size() / 2] / 1e9, mean, flops / mean / 1e9, x.front(), x.back(), rounds);
        }
        return 0;
    }
    std::vector<std::vector<Cand>> sets;
    std::vector<int> set_kind;  // 0 = bf16 data, 1 = int8 data (candidate sets are compared within a kind)
    for (int variant : variants) {
        if (only && variant != only) continue;
        // (1) pure compute
        hipLaunchKernelGGL(k_fillf, dim3((Bpad + 255) / 256), dim3(256), 0, 0, thr, Bpad, INFINITY);
        launch(variant);
        CK(hipDeviceSynchronize());
        float best = 1e30f, sum = 0;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0));
            launch(variant);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
            sum += ms;
        }
        printf("variant %d: N=%lld B=%d d=%d  best %.3f ms  avg %.3f ms  -> %.1f TFLOP/s (best) %.1f (avg)\n", variant,
               (long long)N, B, d, best, sum / reps, flops / best / 1e9, flops / (sum / reps) / 1e9);
        // (1b) cost of the hit path: same launch at decreasing thresholds (z sigmas of the score distribution)
        if (getenv("SWEEP")) {
            for (float z : {4.6f, 4.2f, 3.9f, 3.5f, 3.0f}) {
                hipLaunchKernelGGL(k_fillf, dim3((Bpad + 255) / 256), dim3(256), 0, 0, thr, B, z / sqrtf((float)d));
                float bz = 1e30f;
                for (int r = 0; r < 3; ++r) {
                    CK(hipMemset(cnt, 0, Bpad * 4));
                    CK(hipEventRecord(e0));
                    launch(variant);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    bz = std::min(bz, ms);
                }
                std::vector<int> hc(B);
                CK(hipMemcpy(hc.data(), cnt, B * 4, hipMemcpyDeviceToHost));
                long long tot = 0;
                for (int q = 0; q < B; ++q) tot += hc[q];
                printf("   z=%.1f: %.3f ms, %lld hits (%.2e of pairs)\n", z, bz, tot, (double)tot / ((double)N * B));
            }
        }
        // (2) finite threshold: collect candidates
        const float T0 = 4.6f / sqrtf((float)d);  // ~4.6 sigma
        hipLaunchKernelGGL(k_fillf, dim3((Bpad + 255) / 256), dim3(256), 0, 0, thr, B, T0);
        CK(hipMemset(cnt, 0, Bpad * 4));
        launch(variant);
        CK(hipDeviceSynchronize());
        std::vector<int> hc(B);
        std::vector<int32_t> hr((size_t)B * cap);
        std::vector<float> hv((size_t)B * cap);
        CK(hipMemcpy(hc.data(), cnt, B * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr.data(), crow, hr.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hv.data(), cval, hv.size() * 4, hipMemcpyDeviceToHost));
        std::vector<Cand> s;
        long long over = 0;
        for (int q = 0; q < B; ++q) {
            if (hc[q] > cap) over++;
            for (int j = 0; j < std::min(hc[q], cap); ++j) s.push_back({q, hr[(size_t)q * cap + j], hv[(size_t)q * cap + j]});
        }
        std::sort(s.begin(), s.end());
        printf("   threshold %.4f: %zu candidates, %lld overflowed queries\n", T0, s.size(), over);
        sets.push_back(s);
        set_kind.push_back(variant >= 1000 ? 1 : 0);
        {
            std::vector<int> hs(Bpad);
            CK(hipMemcpy(hs.data(), status, Bpad * 4, hipMemcpyDeviceToHost));
            long long fl = 0;
            for (int q = 0; q < B; ++q) fl += hs[q] != 0;
            if (fl) printf("   %lld queries flagged overflow by the kernel\n", fl);
            CK(hipMemset(status, 0, Bpad * 4));
        }
    }
    bool all_same = true;
    if (only) return 0;
    for (size_t v = 1; v < sets.size(); ++v) {
        size_t ref = 0;
        while (set_kind[ref] != set_kind[v]) ++ref;
        if (ref == v) continue;
        const std::vector<Cand>& s0 = sets[ref];
        bool same = s0.size() == sets[v].size();
        double maxdiff = 0;
        if (same)
            for (size_t i = 0; i < s0.size(); ++i) {
                if (s0[i].q != sets[v][i].q || s0[i].row != sets[v][i].row) {
                    same = false;
                    break;
                }
                maxdiff = std::max(maxdiff, (double)fabsf(s0[i].v - sets[v][i].v));
            }
        printf("candidate set %zu (variant %d) vs %zu: %s (max |dv| %.3g)\n", v, variants[v], ref, same ? "IDENTICAL" : "DIFFER", maxdiff);
        all_same = all_same && same;
    }
    return all_same ? 0 : 2;
}
