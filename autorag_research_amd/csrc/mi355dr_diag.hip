// mi355dr_diag.hip -- measurement support, not part of the search path: the BARE matrix-pipe rate of the
// instruction a screen kernel issues, on operands distributed like the shadows it multiplies.
//
// Why it is in the library: the large-block screens run AT the socket power cap, where the rate of the matrix
// pipe is set by the energy of its multiply-adds on real operand bits, not by its nominal peak.  `roofline.frac` keeps the
// nominal peak; `bench.py` additionally runs this stream IN THE SAME RUN, on the same chip, and reports the screen's rate
// as a fraction of it -- so that the power-cap argument rests on something the driver observes, not on a constant measured
// on another box.
//
// The stream: 256 workgroups x 8 waves (two per SIMD on every CU), four operand register sets cycled so that consecutive
// instructions see different bit patterns, four independent accumulators, no memory traffic inside the loop.  Operands:
// Gaussian, sigma 29 clipped to +-127 for int8 (what k_build_shadow8 produces for unit Gaussian rows), unit-norm-scale
// Gaussian for bf16.  tools/mfma_power_probe.hip is the stand-alone form (more formats, encodings, instruction shapes).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <vector>

#include "mi355dr.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kSets = 4;

template <bool I8>
__global__ __launch_bounds__(512, 2) void k_diag_stream(const v8i* __restrict__ ops, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    // (operand sets of 8 dwords per lane, of which these two formats use 4: the register footprint of
    // tools/mfma_power_probe.hip's stream -- 128 + VGPRs, one workgroup per CU -- so that both report the same machine state;
    // with 4-dword sets the kernel fits twice per CU and the bf16 stream measured 1.32 PF where the probe measures 1.80)
    v8i a8[kSets], b8[kSets];
    v4i a[kSets], b[kSets];
#pragma unroll
    for (int s = 0; s < kSets; ++s) {
        a8[s] = ops[((wave * 2 * kSets + 2 * s) % 4096) * 64 + lane];
        b8[s] = ops[((wave * 2 * kSets + 2 * s + 1) % 4096) * 64 + lane];
        a[s] = v4i{a8[s][0], a8[s][1], a8[s][2], a8[s][3]};
        b[s] = v4i{b8[s][0], b8[s][1], b8[s][2], b8[s][3]};
    }
    v16f acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < kSets; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (I8)
                    acc[i] = __builtin_bit_cast(v16f, __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[(s + i) % kSets],
                                                                                            __builtin_bit_cast(v16i, acc[i]), 0, 0, 0));
                else
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[s]),
                                                                     __builtin_bit_cast(bf16x8, b[(s + i) % kSets]), acc[i], 0, 0, 0);
            }
        if constexpr (I8) {  // keep the int32 accumulators from saturating into one stuck pattern
            if ((it & 255) == 255)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] = __builtin_bit_cast(float, __builtin_bit_cast(int, acc[i][r]) >> 8);
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[i][r];
#pragma unroll
    for (int s = 0; s < kSets; ++s) sum += (float)(a8[s][4] ^ a8[s][7] ^ b8[s][5] ^ b8[s][6]);  // (keeps the upper halves live)
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

// unit Gaussian from a 64-bit LCG (sum of 12 uniforms - 6): deterministic, no <random> state per call
struct Gauss {
    uint64_t s;
    float next() {
        float u = 0.0f;
        for (int i = 0; i < 12; ++i) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            u += (float)((s >> 40) & 0xFFFFFF) * (1.0f / 16777216.0f);
        }
        return u - 6.0f;
    }
};

}  // namespace

extern "C" int mi355dr_diag_mfma_stream(int device, int format, double seconds, double* out_tops) {
    if (!out_tops || (format != 0 && format != 1) || !(seconds > 0.0) || seconds > 60.0) return MI355DR_E_INVALID;
    *out_tops = 0.0;
    if (hipSetDevice(device) != hipSuccess) return MI355DR_E_HIP;
    std::vector<uint32_t> words((size_t)4096 * 64 * 8);
    Gauss g{0x9E3779B97F4A7C15ull};
    for (size_t w = 0; w < words.size(); ++w) {
        uint32_t v = 0;
        if ((w & 7) >= 4) {  // the unused upper half of an 8-dword set
            words[w] = (uint32_t)w * 2654435761u;
            continue;
        }
        if (format == 0) {
            for (int e = 0; e < 4; ++e) {
                float x = g.next() * 29.0f;
                x = x > 127.f ? 127.f : (x < -127.f ? -127.f : x);
                v |= (uint32_t)(uint8_t)(int8_t)std::nearbyint(x) << (8 * e);
            }
        } else {
            for (int e = 0; e < 2; ++e) {
                const float x = g.next() * 0.036f;
                uint32_t u;
                __builtin_memcpy(&u, &x, 4);
                v |= ((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16) << (16 * e);
            }
        }
        words[w] = v;
    }
    v8i* ops = nullptr;
    float* out = nullptr;
    const int grid = 256, block = 512;
    if (hipMalloc(&ops, words.size() * 4) != hipSuccess) return MI355DR_E_NOMEM;
    if (hipMalloc(&out, (size_t)grid * block * 4) != hipSuccess) {
        (void)hipFree(ops);
        return MI355DR_E_NOMEM;
    }
    int rc = MI355DR_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto launch = [&](int iters) {
        if (format == 0) hipLaunchKernelGGL(k_diag_stream<true>, dim3(grid), dim3(block), 0, 0, ops, out, iters);
        else hipLaunchKernelGGL(k_diag_stream<false>, dim3(grid), dim3(block), 0, 0, ops, out, iters);
    };
    do {
        if (hipMemcpy(ops, words.data(), words.size() * 4, hipMemcpyHostToDevice) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
            hipEventCreate(&e1) != hipSuccess) {
            rc = MI355DR_E_HIP;
            break;
        }
        launch(2000);
        if (hipDeviceSynchronize() != hipSuccess) {
            rc = MI355DR_E_HIP;
            break;
        }
        float ms = 0.0f;
        (void)hipEventRecord(e0, 0);
        launch(20000);
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.0f)) {
            rc = MI355DR_E_HIP;
            break;
        }
        const int iters = std::max(1000, (int)(20000 * 50.0 / ms));  // ~50 ms per launch
        std::vector<float> per;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
            (void)hipEventRecord(e0, 0);
            launch(iters);
            (void)hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) {
                rc = MI355DR_E_HIP;
                break;
            }
            per.push_back(ms);
        }
        if (rc != MI355DR_OK || per.empty()) break;
        // the settled rate: the second half of the launches (the governor ramps for ~0.4 s)
        double tail_ms = 0.0;
        for (size_t i = per.size() / 2; i < per.size(); ++i) tail_ms += per[i];
        const double insts = (double)(per.size() - per.size() / 2) * iters * kSets * 4 * (double)grid * (block / 64);
        const double ops_per_inst = format == 0 ? 2.0 * 32 * 32 * 32 : 2.0 * 32 * 32 * 16;
        *out_tops = insts * ops_per_inst / (tail_ms * 1e-3) / 1e12;
    } while (false);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(ops);
    (void)hipFree(out);
    return rc;
}
