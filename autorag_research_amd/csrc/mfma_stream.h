// mfma_stream.h -- the BARE MFMA stream kernel: operands in registers, four independent accumulators, no memory traffic in the
// loop.  ONE source for the library's diagnostic entry point (mi355dr_diag.hip: `mi355dr_diag_mfma_stream`, instantiates i8 and
// bf16) and for the stand-alone probe (tools/mfma_power_probe.hip: every format, encoding and instruction shape) -- a second,
// "equivalent" copy of this loop in the library measured 1.29 PF on bf16 where this one measures 1.76 on the same box in the same
// process (profiles/r05_diag_vs_probe.txt), so there is no second copy.
#pragma once
#include <hip/hip_runtime.h>

namespace mfma_stream {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kSets = 4;  // operand register sets a wave cycles through (different bit patterns from one MFMA to the next)

// FMT: 0 fp8 e4m3, 2 fp6 e2m3, 4 fp4 e2m1 (the instruction's cbsz / blgp codes); 100 = i8, 101 = bf16
template <int FMT>
__global__ __launch_bounds__(512, 2) void k_stream(const v8i* __restrict__ ops, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    v8i a[kSets], b[kSets];
#pragma unroll
    for (int s = 0; s < kSets; ++s) {
        a[s] = ops[((wave * 2 * kSets + 2 * s) % 4096) * 64 + lane];
        b[s] = ops[((wave * 2 * kSets + 2 * s + 1) % 4096) * 64 + lane];
    }
    v16f acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    const int sc = 0x7F7F7F7F;  // E8M0 scale bytes: 2^0
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < kSets; ++s) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (FMT == 100) {
                    const v4i aa = {a[s][0], a[s][1], a[s][2], a[s][3]}, bb = {b[(s + i) % kSets][0], b[(s + i) % kSets][1], b[(s + i) % kSets][2], b[(s + i) % kSets][3]};
                    acc[i] = __builtin_bit_cast(v16f, __builtin_amdgcn_mfma_i32_32x32x32_i8(aa, bb, __builtin_bit_cast(v16i, acc[i]), 0, 0, 0));
                } else if constexpr (FMT == 102) {  // i8, NO operand shared by consecutive instructions (100 keeps A for four)
                    const int sa = (s + i) % kSets, sb = (s + 3 * i + 1) % kSets;
                    const v4i aa = {a[sa][0], a[sa][1], a[sa][2], a[sa][3]}, bb = {b[sb][0], b[sb][1], b[sb][2], b[sb][3]};
                    acc[i] = __builtin_bit_cast(v16f, __builtin_amdgcn_mfma_i32_32x32x32_i8(aa, bb, __builtin_bit_cast(v16i, acc[i]), 0, 0, 0));
                } else if constexpr (FMT == 103) {  // i8, BOTH operands kept for four consecutive instructions
                    const v4i aa = {a[s][0], a[s][1], a[s][2], a[s][3]}, bb = {b[s][0], b[s][1], b[s][2], b[s][3]};
                    acc[i] = __builtin_bit_cast(v16f, __builtin_amdgcn_mfma_i32_32x32x32_i8(aa, bb, __builtin_bit_cast(v16i, acc[i]), 0, 0, 0));
                } else if constexpr (FMT == 105 || FMT == 106) {  // i8, dependent chains: ONE accumulator for all 16 instructions of an
                    // iteration (105), or two used in runs of eight (106) -- does accumulating in place cost less than cycling four?
                    const int ai = FMT == 105 ? 0 : (s >> 1);
                    const v4i aa = {a[s][0], a[s][1], a[s][2], a[s][3]}, bb = {b[(s + i) % kSets][0], b[(s + i) % kSets][1], b[(s + i) % kSets][2], b[(s + i) % kSets][3]};
                    acc[ai] = __builtin_bit_cast(v16f, __builtin_amdgcn_mfma_i32_32x32x32_i8(aa, bb, __builtin_bit_cast(v16i, acc[ai]), 0, 0, 0));
                } else if constexpr (FMT == 104) {  // i8 as v_mfma_i32_16x16x64_i8 (same operand bytes, half the multiply-adds, 4 accumulator registers)
                    const v4i aa = {a[s][0], a[s][1], a[s][2], a[s][3]}, bb = {b[(s + i) % kSets][0], b[(s + i) % kSets][1], b[(s + i) % kSets][2], b[(s + i) % kSets][3]};
                    typedef int v4acc __attribute__((ext_vector_type(4)));
                    v4acc c4 = {__builtin_bit_cast(int, acc[i][0]), __builtin_bit_cast(int, acc[i][1]), __builtin_bit_cast(int, acc[i][2]), __builtin_bit_cast(int, acc[i][3])};
                    c4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(aa, bb, c4, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][r] = __builtin_bit_cast(float, c4[r]);
                } else if constexpr (FMT == 101) {
                    const v4i aa = {a[s][0], a[s][1], a[s][2], a[s][3]}, bb = {b[(s + i) % kSets][0], b[(s + i) % kSets][1], b[(s + i) % kSets][2], b[(s + i) % kSets][3]};
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aa), __builtin_bit_cast(bf16x8, bb), acc[i], 0, 0, 0);
                } else {
                    acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[s], b[(s + i) % kSets], acc[i], FMT, FMT, 0, sc, 0, sc);
                }
            }
        }
        if constexpr (FMT == 100 || FMT == 102 || FMT == 103 || FMT == 104 || FMT == 105 || FMT == 106) {  // keep the int32 accumulators from saturating into one stuck pattern
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
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

}  // namespace mfma_stream
