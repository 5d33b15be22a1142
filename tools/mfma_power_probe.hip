// mfma_power_probe.hip -- what the matrix pipe DRAWS per operation, by operand format.
//
// The headline screen kernel runs at the socket power cap (DESIGN "power"), so its time is energy / 1.3 kW: a format with
// twice the MFMA rate only helps if its energy per multiply-add is lower.  This probe runs a bare MFMA stream -- operands in
// registers (Gaussian data quantised to the format, or zeros), four independent accumulators, two waves per SIMD on every CU,
// no memory traffic inside the loop -- for a few seconds per format, so that a `rocm-smi` poll next to it (tools/r4_mx_probe.sh)
// reads the power and the clock the chip settles at.  Prints per format: ops per instruction, achieved TOP/s.
//
//   formats: i8 (v_mfma_i32_32x32x32_i8: what the screen issues), bf16 (v_mfma_f32_32x32x16_bf16), and the block-scaled
//   v_mfma_scale_f32_32x32x64_f8f6f4 with fp8-e4m3 / fp6-e2m3 / fp4-e2m1 operands (scale bytes 127 = 1.0).
//
// build: hipcc --offload-arch=gfx950 -O3 -Iautorag_research_amd/csrc tools/mfma_power_probe.hip -o /tmp/mfma_power_probe
// run:   /tmp/mfma_power_probe <format: i8|bf16|fp8|fp6|fp4> <data: gauss|zero> <seconds>
//
// Second mode (item 2c): `gather <rows> <queries> <candidates per query>` times an int8 second stage that GATHERS candidate rows
// from an int8 shadow (768 B per row, random rows) and scores them with v_dot4_i32_i8 -- what a loose pre-screen would have to
// be followed by.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e__ = (x);                                                                   \
        if (e__ != hipSuccess) {                                                                \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #x, hipGetErrorString(e__), __FILE__, __LINE__); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

#include "mfma_stream.h"
using namespace mfma_stream;

// ---- host-side quantisers (element codes; what matters here is a realistic bit pattern stream, not a GEMM result) ----
static uint8_t to_e4m3(float x) {  // OCP e4m3fn, round to nearest even, saturating at 448
    const uint8_t sign = x < 0 ? 0x80 : 0;
    float a = std::fabs(x);
    if (!(a == a)) return 0x7F;
    if (a >= 448.0f) return sign | 0x7E;
    if (a < std::ldexp(1.0f, -10)) return sign;
    int e;
    float m = std::frexp(a, &e);  // a = m * 2^e, m in [0.5, 1)
    int E = e - 1 + 7;            // biased exponent of 1.f form
    float frac;
    if (E <= 0) {  // subnormal: value = f/8 * 2^-6
        frac = a / std::ldexp(1.0f, -9);
        int q = (int)std::nearbyint(frac);
        if (q >= 8) return sign | 0x08;
        return sign | (uint8_t)q;
    }
    frac = (m * 2.0f - 1.0f) * 8.0f;
    int q = (int)std::nearbyint(frac);
    if (q == 8) {
        q = 0;
        ++E;
    }
    if (E > 15 || (E == 15 && q > 6)) return sign | 0x7E;
    return sign | (uint8_t)((E << 3) | q);
}
static uint8_t to_grid(float x, const float* grid, int n, int signbit) {  // nearest of a non-negative grid + sign bit
    const float a = std::fabs(x);
    int best = 0;
    for (int i = 1; i < n; ++i)
        if (std::fabs(grid[i] - a) < std::fabs(grid[best] - a)) best = i;
    return (uint8_t)((x < 0 ? signbit : 0) | best);
}

static void pack_bits(std::vector<uint32_t>& words, size_t lane_base_word, int elem, int bits, uint32_t code) {
    const size_t bit = (size_t)elem * bits;
    const size_t w = lane_base_word + bit / 32;
    const int sh = (int)(bit % 32);
    words[w] |= code << sh;
    if (sh + bits > 32) words[w + 1] |= code >> (32 - sh);
}

static int run_stream(const std::string& fmt, const std::string& data, double seconds) {
    // 4096 operand fragments x 64 lanes x 8 dwords
    std::vector<uint32_t> words((size_t)4096 * 64 * 8, 0u);
    std::mt19937_64 rng(1234);
    std::normal_distribution<float> g(0.0f, 1.0f);
    const bool zero = data == "zero";
    static const float e2m3[32] = {0, 0.125f, 0.25f, 0.375f, 0.5f, 0.625f, 0.75f, 0.875f, 1, 1.125f, 1.25f, 1.375f, 1.5f, 1.625f, 1.75f, 1.875f,
                                   2, 2.25f, 2.5f, 2.75f, 3, 3.25f, 3.5f, 3.75f, 4, 4.5f, 5, 5.5f, 6, 6.5f, 7, 7.5f};
    static const float e2m1[8] = {0, 0.5f, 1, 1.5f, 2, 3, 4, 6};
    int elems = 32, bits = 8;  // elements per lane and bits per element of one operand fragment
    const bool is_i8 = fmt.rfind("i8", 0) == 0;  // i8, i8_nos (no operand shared), i8_ab (both kept), i8_16 (16x16x64)
    if (is_i8) elems = 16, bits = 8;
    else if (fmt == "bf16") elems = 8, bits = 16;
    else if (fmt == "fp6") bits = 6;
    else if (fmt == "fp4") bits = 4;
    if (!zero) {
        for (size_t f = 0; f < 4096; ++f)
            for (int lane = 0; lane < 64; ++lane) {
                // one block of 32 consecutive k-values shares a scale: unit-variance Gaussian, block peak mapped to the format's top
                float blk[32];
                float peak = 0.f;
                for (int e = 0; e < elems; ++e) {
                    blk[e] = g(rng);
                    peak = std::fmax(peak, std::fabs(blk[e]));
                }
                const size_t base = (f * 64 + lane) * 8;
                for (int e = 0; e < elems; ++e) {
                    uint32_t code;
                    if (is_i8) {
                        // data modes beyond gauss / zero (what the operand bit patterns cost at the power cap): "abs" every operand
                        // |x|; "absA" the A side only (even fragments); "offA" the A side as 7-bit values + 64 (all in 1..127: what
                        // an offset-encoded corpus shadow would hold), "off" both sides
                        const bool a_side = (f & 1) == 0;
                        float v = blk[e] * (127.0f / 4.4f);
                        // "g7A" / "g6A" / "g5A": the A side at 7 / 6 / 5 bits (sigma 14.5 / 7.2 / 3.6), "g7" ... both sides
                        if (data.rfind("g", 0) == 0 && data != "gauss" && (a_side || data.back() != 'A'))
                            v = blk[e] * (127.0f / 4.4f) / (float)(1 << (8 - (data[1] - '0')));
                        if (data == "abs" || (data == "absA" && a_side)) v = std::fabs(v);
                        if (data == "off" || (data == "offA" && a_side)) v = blk[e] * (63.0f / 4.4f) + 64.0f;
                        v = v > 127 ? 127 : (v < -127 ? -127 : v);
                        code = (uint8_t)(int8_t)std::nearbyint(v);
                    }
                    else if (fmt == "bf16") {
                        uint32_t u;
                        const float v = blk[e] * 0.036f;
                        memcpy(&u, &v, 4);
                        code = (u + 0x7FFF + ((u >> 16) & 1)) >> 16;
                    } else if (fmt == "fp8") code = to_e4m3(blk[e] * 64.0f);
                    else if (fmt == "fp6") code = to_grid(blk[e] * (7.5f / std::exp2(std::ceil(std::log2(peak / 7.5f))) / 7.5f), e2m3, 32, 0x20);
                    else code = to_grid(blk[e] * (6.0f / std::exp2(std::ceil(std::log2(peak / 6.0f))) / 6.0f), e2m1, 8, 0x8);
                    pack_bits(words, base, e, bits, code);
                }
            }
    }
    v8i* ops = nullptr;
    float* out = nullptr;
    CK(hipMalloc(&ops, words.size() * 4));
    CK(hipMemcpy(ops, words.data(), words.size() * 4, hipMemcpyHostToDevice));
    const int grid = 256, block = 512;
    CK(hipMalloc(&out, (size_t)grid * block * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto launch = [&](int iters) {
        if (fmt == "i8") hipLaunchKernelGGL(k_stream<100>, dim3(grid), dim3(block), 0, 0, ops, out, iters);
        else if (fmt == "i8_nos") hipLaunchKernelGGL(k_stream<102>, dim3(grid), dim3(block), 0, 0, ops, out, iters);
        else if (fmt == "i8_ab") hipLaunchKernelGGL(k_stream<103>, dim3(grid), dim3(block), 0, 0, ops, out, iters);
        else if (fmt == "i8_dep") hipLaunchKernelGGL(k_stream<105>, dim3(grid), dim3(block), 0, 0, ops, out, iters);
        else if (fmt == "i8_dep2") hipLaunchKernelGGL(k_stream<106>, dim3(grid), dim3(block), 0, 0, ops, out, iters);
        else if (fmt == "i8_16") hipLaunchKernelGGL(k_stream<104>, dim3(grid), dim3(block), 0, 0, ops, out, iters);
        else if (fmt == "bf16") hipLaunchKernelGGL(k_stream<101>, dim3(grid), dim3(block), 0, 0, ops, out, iters);
        else if (fmt == "fp8") hipLaunchKernelGGL(k_stream<0>, dim3(grid), dim3(block), 0, 0, ops, out, iters);
        else if (fmt == "fp6") hipLaunchKernelGGL(k_stream<2>, dim3(grid), dim3(block), 0, 0, ops, out, iters);
        else hipLaunchKernelGGL(k_stream<4>, dim3(grid), dim3(block), 0, 0, ops, out, iters);
    };
    const double ops_per_inst = fmt == "i8_16" ? 2.0 * 16 * 16 * 64 : is_i8 ? 2.0 * 32 * 32 * 32 : fmt == "bf16" ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 64;
    launch(2000);
    CK(hipDeviceSynchronize());
    // calibrate to ~100 ms per launch
    CK(hipEventRecord(e0));
    launch(20000);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int iters = std::max(1000, (int)(20000 * 100.0 / ms));
    const auto t0 = std::chrono::steady_clock::now();
    double total_ms = 0;
    long launches = 0;
    std::vector<float> per;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0));
        launch(iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
        per.push_back(ms);
        ++launches;
    }
    const double insts = (double)launches * iters * kSets * 4 * (double)grid * (block / 64);
    // the settled rate: the second half of the launches
    double tail_ms = 0;
    for (size_t i = per.size() / 2; i < per.size(); ++i) tail_ms += per[i];
    const double tail_insts = (double)(per.size() - per.size() / 2) * iters * kSets * 4 * (double)grid * (block / 64);
    printf("format %-4s data %-5s  %ld launches x %d iters  %.2f s  %.0f TOP/s overall, %.0f TOP/s settled (second half)  [%g ops per instruction]\n",
           fmt.c_str(), data.c_str(), launches, iters, total_ms * 1e-3, insts * ops_per_inst / (total_ms * 1e-3) / 1e12,
           tail_insts * ops_per_inst / (tail_ms * 1e-3) / 1e12, ops_per_inst);
    return 0;
}

// ---- item 2c: gather-based int8 second stage ----
// One wave scores 4 candidate rows at a time: 16 lanes per 768-byte row (3 x 16 B per lane), v_dot4_i32_i8 against the query's
// int8 row held in registers (same 48 bytes per lane), 4-step reduction inside each group of 16 lanes.
__global__ __launch_bounds__(256) void k_gather_i8(const int8_t* __restrict__ shadow, const int8_t* __restrict__ qrows,
                                                   const int* __restrict__ cand, int cand_per_q, int row_bytes, int* __restrict__ out) {
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const int q = blockIdx.x;
    const int wave = threadIdx.x >> 6;
    const uint4* qp = (const uint4*)(qrows + (size_t)q * row_bytes) + sub * 3;
    const uint4 q0 = qp[0], q1 = qp[1], q2 = qp[2];
    for (int c0 = wave * 4; c0 < cand_per_q; c0 += 16) {
        const int c = c0 + grp;
        int acc = 0;
        if (c < cand_per_q) {
            const int row = cand[(size_t)q * cand_per_q + c];
            const uint4* rp = (const uint4*)(shadow + (size_t)row * row_bytes) + sub * 3;
            const uint4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
            acc = __builtin_amdgcn_sdot4(r0.x, q0.x, acc, false);
            acc = __builtin_amdgcn_sdot4(r0.y, q0.y, acc, false);
            acc = __builtin_amdgcn_sdot4(r0.z, q0.z, acc, false);
            acc = __builtin_amdgcn_sdot4(r0.w, q0.w, acc, false);
            acc = __builtin_amdgcn_sdot4(r1.x, q1.x, acc, false);
            acc = __builtin_amdgcn_sdot4(r1.y, q1.y, acc, false);
            acc = __builtin_amdgcn_sdot4(r1.z, q1.z, acc, false);
            acc = __builtin_amdgcn_sdot4(r1.w, q1.w, acc, false);
            acc = __builtin_amdgcn_sdot4(r2.x, q2.x, acc, false);
            acc = __builtin_amdgcn_sdot4(r2.y, q2.y, acc, false);
            acc = __builtin_amdgcn_sdot4(r2.z, q2.z, acc, false);
            acc = __builtin_amdgcn_sdot4(r2.w, q2.w, acc, false);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (sub == 0 && c < cand_per_q) out[(size_t)q * cand_per_q + c] = acc;
    }
}

static int run_gather(long rows, int nq, int cand_per_q) {
    const int row_bytes = 768;
    int8_t *shadow = nullptr, *qrows = nullptr;
    int *cand = nullptr, *out = nullptr;
    CK(hipMalloc(&shadow, (size_t)rows * row_bytes));
    CK(hipMemset(shadow, 3, (size_t)rows * row_bytes));
    CK(hipMalloc(&qrows, (size_t)nq * row_bytes));
    CK(hipMemset(qrows, 5, (size_t)nq * row_bytes));
    std::vector<int> h((size_t)nq * cand_per_q);
    std::mt19937_64 rng(7);
    for (auto& v : h) v = (int)(rng() % (uint64_t)rows);
    CK(hipMalloc(&cand, h.size() * 4));
    CK(hipMemcpy(cand, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&out, h.size() * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_gather_i8, dim3(nq), dim3(256), 0, 0, shadow, qrows, cand, cand_per_q, row_bytes, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double pairs = (double)nq * cand_per_q;
        printf("gather int8 second stage: %ld rows, %d queries x %d candidates = %.2f M pairs: %.3f ms, %.1f GB/s of gathered rows, %.2f ns per pair\n",
               rows, nq, cand_per_q, pairs / 1e6, ms, pairs * row_bytes / (ms * 1e-3) / 1e9, ms * 1e6 / pairs);
    }
    return 0;
}

#ifdef WITH_LIB_DIAG  // (built together with csrc/mi355dr_diag.hip: the library's own stream inside THIS process, for comparison)
extern "C" int mi355dr_diag_mfma_stream(int device, int format, double seconds, double* out_tops);
#endif
int main(int argc, char** argv) {
#ifdef WITH_LIB_DIAG
    if (argc >= 4 && std::string(argv[1]).rfind("lib_", 0) == 0) {
        double tops = 0;
        const int rc = mi355dr_diag_mfma_stream(0, std::string(argv[1]) == "lib_bf16" ? 1 : 0, atof(argv[3]), &tops);
        printf("library stream %s: rc %d, %.0f TOP/s settled\n", argv[1], rc, tops);
        return rc;
    }
#endif
    if (argc >= 5 && std::string(argv[1]) == "gather") return run_gather(atol(argv[2]), atoi(argv[3]), atoi(argv[4]));
    if (argc < 4) {
        fprintf(stderr, "usage: %s <i8|bf16|fp8|fp6|fp4> <gauss|zero> <seconds>   |   %s gather <rows> <queries> <cand per query>\n", argv[0], argv[0]);
        return 2;
    }
    return run_stream(argv[1], argv[2], atof(argv[3]));
}
