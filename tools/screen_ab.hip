// tools/screen_ab.hip -- developer A/B of the large-block int8 screens (not part of the library):
//   variant 0          k_screen256c<int8>   (256 rows x 256 queries, both operands through the LDS)
//   variant 100 + ABL  k_screen_rq<KS, ABL, int8>  (128 rows x 256 queries, query operand resident in registers);
//                      ABL = timing builds (1 no fragment reads, 4 no tests, 8 no barrier, 16 no LDS-DMA)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Iautorag_research_amd/csrc -Itools tools/screen_ab.hip -o tools/bin/screen_ab
// Run:   screen_ab [rows] [queries] [dim] ;  env ROUNDS (interleaved timing rounds, default 12), VARIANTS=0,100,..., DATA (1 =
//        Gaussian sigma 29 -- what the library's shadows hold --, 2 = zeros), SECONDS (sustained run of the FIRST variant, for power)
// Output: per variant median / mean ms and TOP/s with thresholds parked (+inf), then the candidate sets of every non-timing
// variant at a 4.6-sigma threshold compared entry by entry with variant 0's.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dev_common.h"
#include "k_screen256c_abl.h"  // (tools/: the kernels WITH their timing forms; the library's headers carry the kernels alone)
#include "k_screen_rq_abl.h"
#include "k_screen_rq1.h"  // (tools/: the one-wave-per-SIMD experiment, not part of the library)

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
// mode 1: sum of four uniforms, sigma 29 (light tails); mode 2: zeros; mode 3: a TRUE Gaussian (Box-Muller) of sigma `sigma` -- what
// the library's shadows hold for unit Gaussian rows: rows sigma 31.4 (127 at the largest of a 32-row group's 24 576 components),
// queries sigma 40 (127 at the largest of a query's 768)
// (grid-stride: a grid of one thread per byte exceeds 2^32 work-items above 5.59 M rows of 768 B -- the launch is then refused and,
// unchecked, left the operands of every larger run partly unwritten: the absolute times of rounds 4 / 5 at 7.2 M and 10 M rows
// were too fast for that reason; profiles/r05_harness_fill_fix.txt)
__global__ void k_fill8(int8_t* p, int64_t n, int mode, uint64_t seed, float sigma) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = hash32(i * 4 + seed), h2 = hash32(i * 4 + 1 + seed);
    const float u = ((h & 0xFFFF) + (h >> 16) + (h2 & 0xFFFF) + (h2 >> 16)) * (1.0f / 65536.0f) - 2.0f;  // var 1/3
    const float g = u * sqrtf(3.0f);
    int v = 0;
    if (mode == 1) v = (int)rintf(fminf(fmaxf(g * 29.0f, -127.f), 127.f));
    if (mode == 3) {
        const float u1 = ((float)h + 1.0f) * (1.0f / 4294967296.0f), u2 = (float)h2 * (1.0f / 4294967296.0f);
        const float z = sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
        v = (int)rintf(fminf(fmaxf(z * sigma, -127.f), 127.f));
    }
    p[i] = (int8_t)v;
  }
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

#define RQ_FORMS(X) X(0) X(1) X(4) X(8) X(16) X(17) X(21) X(32) X(64) X(256) X(512) X(768) X(2048) X(4096) X(8192)

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : (1 << 21);
    const int B = argc > 2 ? atoi(argv[2]) : 1024;
    const int d = argc > 3 ? atoi(argv[3]) : 768;
    const int rounds = getenv("ROUNDS") ? atoi(getenv("ROUNDS")) : 12;
    const int mode = getenv("DATA") ? atoi(getenv("DATA")) : 1;
    const int dpad8 = (d + 127) / 128 * 128;
    const int ks = dpad8 / 128;
    const int64_t Npad = (N + 255) / 256 * 256;
    const int Bpad = (B + 255) / 256 * 256;
    const int cap = 2048;
    int8_t *shadow, *qhat;
    float *thr, *cval, *scv, *kqv;
    int *cnt, *status;
    int32_t* crow;
    uint8_t* flag8;
    I8Group* grp;
    CK(hipMalloc(&shadow, (size_t)Npad * dpad8));
    CK(hipMalloc(&qhat, (size_t)Bpad * dpad8));
    CK(hipMalloc(&thr, Bpad * 4));
    CK(hipMalloc(&cnt, Bpad * 4));
    CK(hipMalloc(&status, Bpad * 4));
    CK(hipMemset(status, 0, Bpad * 4));
    CK(hipMalloc(&crow, (size_t)Bpad * cap * 4));
    CK(hipMalloc(&cval, (size_t)Bpad * cap * 4));
    CK(hipMalloc(&scv, Bpad * 4));
    CK(hipMalloc(&kqv, Bpad * 4));
    CK(hipMalloc(&flag8, Npad));
    CK(hipMemset(flag8, 0, Npad));
    CK(hipMalloc(&grp, (size_t)(Npad / 32) * sizeof(I8Group)));
    {
        const int64_t nb = (int64_t)Npad * dpad8, nq = (int64_t)Bpad * dpad8;
        const float sig_r = mode == 3 ? 31.4f : 29.0f, sig_q = mode == 3 ? 40.0f : 29.0f;
        CK(hipMemset(shadow, 0, (size_t)nb));
        hipLaunchKernelGGL(k_fill8, dim3(1 << 16), dim3(256), 0, 0, shadow, nb, mode, 1234ull, sig_r);
        CK(hipGetLastError());
        hipLaunchKernelGGL(k_fill8, dim3(1 << 12), dim3(256), 0, 0, qhat, nq, mode, 99ull, sig_q);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        if (mode != 2) {  // the LAST row must hold data (a refused or truncated fill leaves it zero)
            std::vector<int8_t> tail(dpad8);
            CK(hipMemcpy(tail.data(), shadow + (size_t)(N - 1) * dpad8, dpad8, hipMemcpyDeviceToHost));
            int nz = 0;
            for (int i = 0; i < d; ++i) nz += tail[i] != 0;
            if (nz < d / 2) { fprintf(stderr, "operand fill incomplete: last row has %d non-zero bytes\n", nz); exit(1); }
        }
        std::vector<float> one(Bpad, 1.0f);
        CK(hipMemcpy(scv, one.data(), Bpad * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(kqv, one.data(), Bpad * 4, hipMemcpyHostToDevice));
        // acc ~ N(0, d 29^4): with S_q = 1 and every group's step 1 / (d 841) the screen value is N(0, 1/d); the group's
        // residual term differs from group to group so that a wrong record shows up in the values
        std::vector<I8Group> hg((size_t)(Npad / 32));
        for (size_t g = 0; g < hg.size(); ++g) hg[g] = I8Group{1.0f / ((float)d * sig_r * sig_q), 1e-4f * (float)(g % 7)};
        CK(hipMemcpy(grp, hg.data(), hg.size() * sizeof(I8Group), hipMemcpyHostToDevice));
    }
    CK(hipDeviceSynchronize());
    CK(hipFuncSetAttribute((const void*)k_screen256c<kScreen256cAbl, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kScreen256Lds));
#define RQ_ATTR(A)                                                                                                            \
    CK(hipFuncSetAttribute((const void*)k_screen_rq<6, A, true>, hipFuncAttributeMaxDynamicSharedMemorySize, rq_lds(6)));     \
    CK(hipFuncSetAttribute((const void*)k_screen_rq<3, A, true>, hipFuncAttributeMaxDynamicSharedMemorySize, rq_lds(3)));
    RQ_FORMS(RQ_ATTR)
#define RQ1_FORMS(X) X(0) X(1) X(4) X(16) X(4096)
#define RQ1_ATTR(A) CK(hipFuncSetAttribute((const void*)k_screen_rq1<6, A, true>, hipFuncAttributeMaxDynamicSharedMemorySize, rq_lds(6)));
    RQ1_FORMS(RQ1_ATTR)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    int* progress = nullptr;  // drift limiter of k_screen_rq (env DRIFT = tiles, 0 = off; variant 100 + 4096 = built without it)
    CK(hipMalloc(&progress, kRqProgressWords * 4));
    CK(hipMemset(progress, 0, kRqProgressWords * 4));
    const int drift = getenv("DRIFT") ? atoi(getenv("DRIFT")) : 3;
    int epoch = 0;
    auto launch = [&](int variant) {
        ScreenArgs2 sa{};
        sa.status = status;
        sa.progress = progress;
        sa.drift = drift;
        sa.epoch = epoch = epoch % 4095 + 1;
        sa.shadow = shadow;
        sa.qhat = qhat;
        sa.thr = thr;
        sa.cnt = cnt;
        sa.cand_row = crow;
        sa.cand_val = cval;
        sa.sc = scv;
        sa.kq = kqv;
        sa.grp = grp;
        sa.flag8 = flag8;
        sa.row_bytes = dpad8;
        sa.ksteps = ks;
        sa.cap = cap;
        sa.ct0 = 0;
        sa.row_end = N;
        sa.n_qtiles = (B + 255) / 256;
        if (variant == 0) {
            sa.n_ctiles = (int)((N + 255) / 256);
            hipLaunchKernelGGL((k_screen256c<kScreen256cAbl, true>), dim3(screen256_grid(sa.n_ctiles, sa.n_qtiles)), dim3(512),
                               kScreen256Lds, 0, sa);
        } else if (variant >= 200 && variant < 100000 && (variant - 200 == 0 || variant - 200 == 1 || variant - 200 == 4 || variant - 200 == 16 || variant - 200 == 4096)) {
            // k_screen_rq1 (one wave per SIMD, 4 waves x 64 queries; d = 768 only)
            const int abl = variant - 200;
            sa.n_ctiles = (int)((N + 127) / 128);
            const unsigned grid = screen_rq_grid(sa.n_ctiles, sa.n_qtiles);
            if (ks != 6) { fprintf(stderr, "rq1: d = 768 only\n"); exit(1); }
#define RQ1_LAUNCH(A) if (abl == A) hipLaunchKernelGGL((k_screen_rq1<6, A, true>), dim3(grid), dim3(256), rq_lds(6), 0, sa);
            RQ1_FORMS(RQ1_LAUNCH)
        } else {
            const int abl = variant - 100;
            sa.n_ctiles = (int)((N + 127) / 128);
            const unsigned grid = screen_rq_grid(sa.n_ctiles, sa.n_qtiles);
            bool done = false;
#define RQ_LAUNCH(A)                                                                                              \
    if (!done && abl == A) {                                                                                      \
        done = true;                                                                                              \
        if (ks == 6) hipLaunchKernelGGL((k_screen_rq<6, A, true>), dim3(grid), dim3(512), rq_lds(6), 0, sa);      \
        else if (ks == 3) hipLaunchKernelGGL((k_screen_rq<3, A, true>), dim3(grid), dim3(512), rq_lds(3), 0, sa); \
        else { fprintf(stderr, "rq: dim not instantiated\n"); exit(1); }                                          \
    }
            RQ_FORMS(RQ_LAUNCH)
            if (!done) { fprintf(stderr, "unknown variant %d\n", variant); exit(1); }
        }
        CK(hipGetLastError());
    };

    std::vector<int> variants = {0, 100};
    if (getenv("VARIANTS")) {
        variants.clear();
        for (char* tok = strtok(getenv("VARIANTS"), ","); tok; tok = strtok(nullptr, ",")) variants.push_back(atoi(tok));
    }
    const double ops = 2.0 * B * (double)N * d;
    printf("rows %lld  queries %d  dim %d  data mode %d\n", (long long)N, B, d, mode);
    if (getenv("SECONDS_RUN")) {  // sustained run of the first variant (read the power next to it)
        const double secs = atof(getenv("SECONDS_RUN"));
        hipLaunchKernelGGL(k_fillf, dim3((Bpad + 255) / 256), dim3(256), 0, 0, thr, Bpad, INFINITY);
        const bool hits = getenv("THRZ") != nullptr;  // (thresholds at THRZ sigma instead of parked; counters reset before every launch)
        if (hits) hipLaunchKernelGGL(k_fillf, dim3((Bpad + 255) / 256), dim3(256), 0, 0, thr, B, (float)atof(getenv("THRZ")) / sqrtf((float)d));
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        long n = 0;
        double el = 0;
        do {
            for (int i = 0; i < 20; ++i) {
                if (hits) CK(hipMemsetAsync(cnt, 0, Bpad * 4, 0));
                launch(variants[0]);
            }
            CK(hipDeviceSynchronize());
            n += 20;
            el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        } while (el < secs);
        printf("variant %d sustained: %ld launches in %.2f s: %.3f ms per launch (%.0f TOP/s)\n", variants[0], n, el, el / n * 1e3,
               ops / (el / n) / 1e12);
        return 0;
    }
    // ---- interleaved timing, thresholds parked (or, THRZ=<z>: at z sigma of the score distribution -- the cost of the hit path;
    // the candidate counters are reset before every launch, outside the timed events)
    {
        hipLaunchKernelGGL(k_fillf, dim3((Bpad + 255) / 256), dim3(256), 0, 0, thr, Bpad, INFINITY);
        const bool hits = getenv("THRZ") != nullptr;
        if (hits) {
            hipLaunchKernelGGL(k_fillf, dim3((Bpad + 255) / 256), dim3(256), 0, 0, thr, B, (float)atof(getenv("THRZ")) / sqrtf((float)d));
            printf("thresholds at %.2f sigma\n", atof(getenv("THRZ")));
        }
        std::vector<std::vector<float>> ms(variants.size());
        for (int w = 0; w < 3; ++w)
            for (int variant : variants) launch(variant);
        CK(hipDeviceSynchronize());
        if (hits) {
            std::vector<int> hc(B);
            CK(hipMemset(cnt, 0, Bpad * 4));
            launch(variants[0]);
            CK(hipMemcpy(hc.data(), cnt, B * 4, hipMemcpyDeviceToHost));
            long long tot = 0;
            for (int q = 0; q < B; ++q) tot += hc[q];
            printf("%lld hits per launch (%.1f per query, %.2f per 32 x 32 block)\n", tot, (double)tot / B, (double)tot / ((double)N / 32 * B / 32));
            CK(hipMemset(status, 0, Bpad * 4));
        }
        for (int r = 0; r < rounds; ++r)
            for (size_t v = 0; v < variants.size(); ++v) {
                if (hits) CK(hipMemsetAsync(cnt, 0, Bpad * 4, 0));
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
            printf("variant %4d: median %.3f ms (%.0f TOP/s)  mean %.3f ms  min %.3f  max %.3f   [%d interleaved rounds]\n", variants[v],
                   x[x.size() / 2], ops / x[x.size() / 2] / 1e9, mean, x.front(), x.back(), rounds);
        }
    }
    // ---- candidate sets at a finite threshold (only the variants that are real kernels)
    std::vector<std::vector<Cand>> sets;
    std::vector<int> set_variant;
    for (int variant : variants) {
        if (variant != 0 && variant != 100 && variant != 164 && variant != 612 && variant != 868 && variant != 2148 && variant != 4196 && variant != 8292 && variant != 200) continue;
        const float T0 = 4.6f / sqrtf((float)d);
        hipLaunchKernelGGL(k_fillf, dim3((Bpad + 255) / 256), dim3(256), 0, 0, thr, Bpad, INFINITY);
        hipLaunchKernelGGL(k_fillf, dim3((Bpad + 255) / 256), dim3(256), 0, 0, thr, B, T0);
        CK(hipMemset(cnt, 0, Bpad * 4));
        CK(hipMemset(status, 0, Bpad * 4));
        launch(variant);
        CK(hipDeviceSynchronize());
        std::vector<int> hc(B), hs(B);
        std::vector<int32_t> hr((size_t)B * cap);
        std::vector<float> hv((size_t)B * cap);
        CK(hipMemcpy(hc.data(), cnt, B * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs.data(), status, B * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hr.data(), crow, hr.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hv.data(), cval, hv.size() * 4, hipMemcpyDeviceToHost));
        std::vector<Cand> s;
        long long over = 0, fl = 0;
        for (int q = 0; q < B; ++q) {
            if (hc[q] > cap) over++;
            fl += hs[q] != 0;
            for (int j = 0; j < std::min(hc[q], cap); ++j) s.push_back({q, hr[(size_t)q * cap + j], hv[(size_t)q * cap + j]});
        }
        std::sort(s.begin(), s.end());
        printf("variant %d at threshold %.4f: %zu candidates, %lld overflowed lists, %lld queries flagged\n", variant, T0, s.size(), over, fl);
        sets.push_back(s);
        set_variant.push_back(variant);
    }
    bool all_same = true;
    for (size_t v = 1; v < sets.size(); ++v) {
        const std::vector<Cand>& s0 = sets[0];
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
        printf("candidate set of variant %d vs variant %d: %s (max |dv| %.3g)\n", set_variant[v], set_variant[0], same ? "IDENTICAL" : "DIFFER", maxdiff);
        all_same = all_same && same;
    }
    return all_same ? 0 : 2;
}
