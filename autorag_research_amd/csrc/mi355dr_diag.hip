// mi355dr_diag.hip -- measurement support, not part of the search path: the BARE matrix-pipe rate of the
// instruction a screen kernel issues, on operands distributed like the shadows it multiplies.
//
// Why it is in the library: the large-block screens run AT the socket power cap, where the rate of the matrix
// pipe is set by the energy of its multiply-adds on real operand bits, not by its nominal peak.  `roofline.frac` keeps the
// nominal peak; `bench.py` additionally runs this stream IN THE SAME RUN, on the same chip, and reports the screen's rate
// as a fraction of it -- so that the power-cap argument rests on something the driver observes, not on a constant measured
// on another box.
//
// The stream (mfma_stream.h: the kernel of tools/mfma_power_probe.hip, ONE source): 256 workgroups x 8 waves (two per SIMD on
// every CU), four operand register sets cycled so that consecutive instructions see different bit patterns, four independent
// accumulators, no memory traffic inside the loop.  Operands: Gaussian, sigma 29 clipped to +-127 for int8 (what
// k_build_shadow8 produces for unit Gaussian rows), unit-norm-scale Gaussian for bf16.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <vector>

#include "mfma_stream.h"
#include "mi355dr.h"

namespace {

using mfma_stream::k_stream;
using mfma_stream::kSets;
using mfma_stream::v8i;

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
        if ((w & 7) >= 4) continue;  // (these two formats fill the lower half of an 8-dword operand set)
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
        if (format == 0) hipLaunchKernelGGL(k_stream<100>, dim3(grid), dim3(block), 0, 0, ops, out, iters);
        else hipLaunchKernelGGL(k_stream<101>, dim3(grid), dim3(block), 0, 0, ops, out, iters);
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
