// tools/screen_trace.hip -- developer probe: per-phase timeline of the 256x256 screen kernels (s_memtime stamps of
// waves 0 and 4 of workgroup 0, K-steps 24..29), first form vs second form, int8.  Not part of the library.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Iautorag_research_amd/csrc -Itools/forms tools/screen_trace.hip -o tools/bin/screen_trace
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "k_screen256b.h"

using namespace mi355;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s (%d)\n", #x, hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_fill8(uint32_t* p, size_t nwords, uint64_t seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwords) return;
    uint64_t x = i * 0x9E3779B97F4A7C15ull + seed;
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    p[i] = (uint32_t)x;
}

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 4194304;
    const int B = 1024, d = 768, dpad8 = 768;
    char *shadow, *qhat;
    float *thr, *scv, *cval;
    int *cnt, *status;
    float* kqv;
    I8Group* grp;
    int32_t* crow;
    uint8_t* flag8;
    CK(hipMalloc(&shadow, (size_t)N * dpad8)); CK(hipMalloc(&qhat, (size_t)B * dpad8));
    CK(hipMalloc(&thr, B * 4)); CK(hipMalloc(&scv, B * 4)); CK(hipMalloc(&kqv, B * 4)); CK(hipMalloc(&cnt, B * 4));
    CK(hipMalloc(&grp, (size_t)(N / 32 + 8) * sizeof(I8Group))); CK(hipMemset(grp, 0, (size_t)(N / 32 + 8) * sizeof(I8Group)));
    CK(hipMemset(scv, 0, B * 4)); CK(hipMemset(kqv, 0, B * 4));
    CK(hipMalloc(&status, B * 4)); CK(hipMalloc(&crow, (size_t)B * 2048 * 4)); CK(hipMalloc(&cval, (size_t)B * 2048 * 4));
    CK(hipMalloc(&flag8, N)); CK(hipMemset(flag8, 0, N)); CK(hipMemset(cnt, 0, B * 4)); CK(hipMemset(status, 0, B * 4));
    hipLaunchKernelGGL(k_fill8, dim3((unsigned)(((size_t)N * dpad8 / 4 + 255) / 256)), dim3(256), 0, 0, (uint32_t*)shadow, (size_t)N * dpad8 / 4, 1ull);
    hipLaunchKernelGGL(k_fill8, dim3((unsigned)(((size_t)B * dpad8 / 4 + 255) / 256)), dim3(256), 0, 0, (uint32_t*)qhat, (size_t)B * dpad8 / 4, 2ull);
    std::vector<float> inf(B, INFINITY);
    CK(hipMemcpy(thr, inf.data(), B * 4, hipMemcpyHostToDevice));
    unsigned long long* trace;
    CK(hipMalloc(&trace, 2 * kTraceStamps * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_out), &trace, sizeof(trace)));
    const int lds = 163840;
    CK(hipFuncSetAttribute((const void*)k_screen256<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)k_screen256b<1040, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)k_screen256b<3152, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)k_screen256b<1108, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)k_screen256b<1112, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    ScreenArgs2 sa{};
    sa.status = status; sa.shadow = shadow; sa.qhat = qhat; sa.thr = thr; sa.sc = scv; sa.kq = kqv; sa.grp = grp; sa.flag8 = flag8;
    sa.cnt = cnt; sa.cand_row = crow; sa.cand_val = cval; sa.row_bytes = dpad8; sa.ksteps = dpad8 / 128; sa.cap = 2048;
    sa.ct0 = 0; sa.row_end = N; sa.n_ctiles = (int)(N / 256); sa.n_qtiles = B / 256;
    const unsigned grid = screen256_grid(sa.n_ctiles, sa.n_qtiles);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const bool brief = argc > 2;
    for (int form = 3; form < 4; ++form) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(trace, 0, 2 * kTraceStamps * 8));
            CK(hipEventRecord(e0));
            if (form == 0) hipLaunchKernelGGL((k_screen256<16, true>), dim3(grid), dim3(512), lds, 0, (ScreenArgs)sa);
            else if (form == 1) hipLaunchKernelGGL((k_screen256b<16, true>), dim3(grid), dim3(512), lds, 0, sa);
            else if (form == 2) hipLaunchKernelGGL((k_screen256b<1040, true>), dim3(grid), dim3(512), lds, 0, sa);   // NM = 2, TAIL = 1
            else if (form == 3) hipLaunchKernelGGL((k_screen256b<3152, true>), dim3(grid), dim3(512), lds, 0, sa);   // NM = 2
            else if (form == 4) hipLaunchKernelGGL((k_screen256b<1108, true>), dim3(grid), dim3(512), lds, 0, sa);  // NM = 2, TAIL = 2
            else hipLaunchKernelGGL((k_screen256b<1112, true>), dim3(grid), dim3(512), lds, 0, sa);                 // NM = 2, no setprio
            CK(hipGetLastError());
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        std::vector<unsigned long long> t(2 * kTraceStamps);
        CK(hipMemcpy(t.data(), trace, t.size() * 8, hipMemcpyDeviceToHost));
        printf("=== form %d (0 first form, 1 second form, 2 NM=0 saddr, 3 NM=2 saddr, 4 NM=2 saddr no setprio, 5 NM=2 saddr lgkm-before-barrier): %.3f ms for N=%lld (%.0f TOP/s); stamps per phase [P=phase start, L=before barrier 1, M0=MFMA start, M1=MFMA issued]\n", form,
               ms, (long long)N, 2.0 * B * N * d / ms / 1e9);
        const unsigned long long t0 = t[0];
        for (int g = 0; g < 2; ++g) {
            printf("group %d (wave %d): per phase: load=L-P  bar1+lgkm=M0-L  mfma=M1-M0  bar2=P'-M1   | P rel. to group0's first stamp\n", g, 4 * g);
            for (int ph = (brief ? 8 : 0); ph < (brief ? 12 : kTraceSteps * 4); ++ph) {
                const unsigned long long* s = &t[g * kTraceStamps + ph * 4];  // memory order: P, M0, L, M1
                const unsigned long long P = s[0], M0 = s[1], L = s[2], M1 = s[3];
                const unsigned long long Pn = ph + 1 < kTraceSteps * 4 ? s[4] : M1;
                printf("  k%d.p%d  load %5lld  bar1 %5lld  mfma %5lld  bar2 %5lld   | P=%lld\n", ph / 4, ph % 4, (long long)(L - P), (long long)(M0 - L),
                       (long long)(M1 - M0), (long long)(Pn - M1), (long long)(P - t0));
            }
        }
    }
    return 0;
}
