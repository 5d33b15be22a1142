// tools/bw_bench.hip -- developer probe: L2->CU delivery rate of the screen kernels' access pattern,
// LDS-DMA (global_load_lds) vs plain global_load_dwordx4, no compute.  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE, int ROWB>
__global__ __launch_bounds__(512, 2) void k_bw(const char* A, const char* Bq, int dpad, int n_qtiles, int n_ctiles, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long row_bytes = (long)dpad * 2;
    const int T = dpad * 2 / ROWB;
    const int b = blockIdx.x, xcd = b & 7, lb = b >> 3;
    const int qt = lb % n_qtiles, ctl = (lb / n_qtiles) * 8 + xcd;
    if (ctl >= n_ctiles) return;
    const char* baseA = A + (long)ctl * 256 * row_bytes;
    const char* baseB = Bq + (long)qt * 256 * row_bytes;
    constexpr int NU = ROWB / 32;  // pieces per operand per wave per step
    unsigned loff[NU];
    for (int u = 0; u < NU; ++u) {
        const int rows_per = 1024 / ROWB;  // rows covered by one wave instruction
        const int r = (NU * wave + u) * rows_per + lane / (ROWB / 16);
        loff[u] = (unsigned)(r * row_bytes) + (lane % (ROWB / 16)) * 16;
    }
    unsigned acc = 0;
    for (int g = 0; g < T; ++g) {
        const char* sa = baseA + (long)g * ROWB;
        const char* sb = baseB + (long)g * ROWB;
        if (MODE == 0) {
            char* d = smem + (g % (ROWB == 64 ? 5 : (ROWB == 128 ? 2 : 1))) * (512 * (ROWB > 256 ? 256 : ROWB)) + ((NU * wave) * 1024) % 65536;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sa + loff[u]), (__attribute__((address_space(3))) void*)(d + u * 1024), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + loff[u]), (__attribute__((address_space(3))) void*)(d + (256 * ROWB) % 65536 + u * 1024), 16, 0, 0);
            }
            if ((g & 1) == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                uint4 v0 = *(const uint4*)(sa + loff[u]);
                uint4 v2 = *(const uint4*)(sb + loff[u]);
                acc ^= v0.x ^ v2.z;
            }
        }
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const long N = argc > 1 ? atol(argv[1]) : 4194304;
    const int B = 1024, dpad = argc > 2 ? atoi(argv[2]) : 768;
    char *A, *Bq; unsigned* sink;
    CK(hipMalloc(&A, (size_t)N * dpad * 2)); CK(hipMalloc(&Bq, (size_t)B * dpad * 2)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(A, 1, (size_t)N * dpad * 2)); CK(hipMemset(Bq, 1, (size_t)B * dpad * 2));
    CK(hipFuncSetAttribute((const void*)k_bw<0, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    CK(hipFuncSetAttribute((const void*)k_bw<0, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    CK(hipFuncSetAttribute((const void*)k_bw<0, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    CK(hipFuncSetAttribute((const void*)k_bw<0, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    const int n_ctiles = (int)(N / 256), n_qtiles = B / 256;
    const long slots = (long)((n_ctiles + 7) / 8) * n_qtiles;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)slots * 8 * (double)dpad * 2 * 512;
    for (int rowb : {64, 128, 256, 512})
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            const dim3 gr((unsigned)(slots * 8));
            if (mode == 0 && rowb == 64) hipLaunchKernelGGL((k_bw<0, 64>), gr, dim3(512), 163840, 0, A, Bq, dpad, n_qtiles, n_ctiles, sink);
            else if (mode == 0 && rowb == 128) hipLaunchKernelGGL((k_bw<0, 128>), gr, dim3(512), 163840, 0, A, Bq, dpad, n_qtiles, n_ctiles, sink);
            else if (mode == 0 && rowb == 256) hipLaunchKernelGGL((k_bw<0, 256>), gr, dim3(512), 163840, 0, A, Bq, dpad, n_qtiles, n_ctiles, sink);
            else if (mode == 0) hipLaunchKernelGGL((k_bw<0, 512>), gr, dim3(512), 163840, 0, A, Bq, dpad, n_qtiles, n_ctiles, sink);
            else if (rowb == 64) hipLaunchKernelGGL((k_bw<1, 64>), gr, dim3(512), 0, 0, A, Bq, dpad, n_qtiles, n_ctiles, sink);
            else if (rowb == 128) hipLaunchKernelGGL((k_bw<1, 128>), gr, dim3(512), 0, 0, A, Bq, dpad, n_qtiles, n_ctiles, sink);
            else if (rowb == 256) hipLaunchKernelGGL((k_bw<1, 256>), gr, dim3(512), 0, 0, A, Bq, dpad, n_qtiles, n_ctiles, sink);
            else hipLaunchKernelGGL((k_bw<1, 512>), gr, dim3(512), 0, 0, A, Bq, dpad, n_qtiles, n_ctiles, sink);
            CK(hipGetLastError());
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("row piece %3d B, %s: %.3f ms  %.2f TB/s delivered to CUs\n", rowb, mode ? "global_load_dwordx4" : "global_load_lds    ", ms, bytes / ms / 1e9);
        }
    }
    return 0;
}
