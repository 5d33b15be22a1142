// k_screen256.h -- the large-block form of the screen: 256 corpus rows x 256 queries per workgroup,
// 8 waves, LDS ring of 8 half-tiles (128 KiB), ping-pong between two wave groups.
//
// Same contract as k_screen (k_screen.h): t = <q_hat, c_hat> in fp32 via v_mfma_f32_32x32x16_bf16, fused
// threshold epilogue, candidates appended with one atomic per hit.  What changes is the pipeline:
//
//  * Tile 256x256, K step 64.  8 waves = 2 groups of 4; wave (wr = group, wc = wave&3) owns rows
//    [128wr,+128) x queries [64wc,+64) = 4x2 MFMA blocks = 128 accumulator VGPRs.
//  * The K-tile is cut into 4 half-tiles of 128 rows x 128 B (16 KiB): A0/A1 = the first/second 64 rows of
//    every wave-row, B0/B1 = the first/second 32 queries of every wave-column.  A wave's work on a K-tile is
//    4 phases = the 4 (row-half i, query-half j) quadrants in the order (0,0) (0,1) (1,1) (1,0), so the
//    half-tiles are first needed in the order A0,B0 | B1 | A1 -- which is also the order they are staged in.
//  * Each phase = LOAD (issue the DMA of ONE half-tile of the next K-tile, ds_read this quadrant's new
//    operands, counted s_waitcnt vmcnt) | s_barrier | MFMA (8 MFMAs, s_setprio 1) | s_barrier.  Group 1 runs
//    one barrier behind group 0, so on every SIMD one wave is in its MFMA half while its partner is in
//    its LOAD half: the matrix pipe sees back-to-back clusters and the LDS/DMA traffic hides under them.
//  * DMA runs 4 half-tiles ahead of use (2 phases of flight, 4 wave-instructions outstanding at every wait:
//    vmcnt(4), never 0 in the main loop).  Hazards (derivation in DESIGN.md section 4.1b): a half-tile is
//    read only >= 2 barriers after every wave retired its part of it, and a ring slot is re-staged >= 2
//    barriers after its last reader's lgkmcnt(0).
#pragma once
#include "k_screen.h"

namespace mi355 {

constexpr int kT2 = 256;                        // tile edge (rows and queries)
constexpr int kHalfBytes = 128 * kRowB;         // 16 KiB
constexpr int kScreen256Lds = 8 * kHalfBytes;   // ring of 8 half-tiles

#define MI355_BARRIER()                      \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)

// ABL: developer ablation switches for tools/screen_bench (0 in the library): bit0 = skip the ds_reads after
// the first K-tile, bit1 = skip the DMA after the prologue, bit2 = no s_setprio.
template <int ABL, bool I8>
__global__ __launch_bounds__(512, 2) void k_screen256(ScreenArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2;  // 0 leads, 1 runs one barrier behind
    const int wr = group, wc = wave & 3;

    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int lb = b >> 3;
    const int qt = lb % a.n_qtiles;
    const int ctl = (lb / a.n_qtiles) * 8 + xcd;
    if (ctl >= a.n_ctiles) return;
    const int64_t tile_row0 = (int64_t)(a.ct0 + ctl) * kT2;
    const int q0 = qt * kT2;
    const int64_t row_bytes = a.row_bytes;

    // ---- DMA source pointers: this wave stages local rows [16*wave + 8u, +8) of every half-tile, u = 0,1
    // half-tile types: 0 = A0, 1 = B0, 2 = B1, 3 = A1
    const char* gsrc[4][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = (2 * wave + u) * 8 + (lane >> 3);      // local row 0..127
        const int c = (lane & 7) ^ ((r >> 1) & 7);           // source chunk for this LDS slot (swizzle)
        const int arow0 = 128 * (r >> 6) + (r & 63);         // + 64*i
        const int bcol0 = 64 * (r >> 5) + (r & 31);          // + 32*j
        gsrc[0][u] = (const char*)a.shadow + (tile_row0 + arow0) * row_bytes + c * 16;
        gsrc[3][u] = (const char*)a.shadow + (tile_row0 + arow0 + 64) * row_bytes + c * 16;
        gsrc[1][u] = (const char*)a.qhat + (int64_t)(q0 + bcol0) * row_bytes + c * 16;
        gsrc[2][u] = (const char*)a.qhat + (int64_t)(q0 + bcol0 + 32) * row_bytes + c * 16;
    }
    // ---- fragment read offsets inside a half-tile
    int offA[2], offB;
    {
        const int g = lane >> 5;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int r = wr * 64 + rb * 32 + (lane & 31);
            offA[rb] = r * kRowB + ((g ^ ((r >> 1) & 7)) << 4);
        }
        const int r = wc * 32 + (lane & 31);
        offB = r * kRowB + ((g ^ ((r >> 1) & 7)) << 4);
    }

    f32x16 acc[2][2][2];  // [row half i][row block rb][query half j]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][rb][j][r] = 0.0f;
    bf16x8 fa[2][4], fb[4];  // typed bf16x8 also for int8 data: see the NOTE in k_screen.h (waitcnt insertion)

    const int T = a.ksteps;

// stage half-tile type S of K-tile TT into its ring slot (slot = 4*(TT&1) + S)
#define MI355_STAGE(S, TT)                                                                            \
    if (!(ABL & 2)) do {                                                                              \
        char* dst__ = smem + (4 * ((TT) & 1) + (S)) * kHalfBytes + (2 * wave) * 1024;                 \
        const int64_t ko__ = (int64_t)(TT) * kRowB;                                                   \
        glds16(gsrc[S][0] + ko__, dst__);                                                             \
        glds16(gsrc[S][1] + ko__, dst__ + 1024);                                                      \
    } while (0)
#define MI355_LOAD_A(I, TT)                                                                           \
    if (!(ABL & 1) || (TT) == 0) do {                                                                                              \
        const char* s__ = smem + (4 * ((TT) & 1) + ((I) ? 3 : 0)) * kHalfBytes;                       \
        _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                              \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                          \
                fa[rb][kk] = __builtin_bit_cast(bf16x8, *(const uint4*)(s__ + (offA[rb] ^ (kk * 32))));                          \
    } while (0)
#define MI355_LOAD_B(J, TT)                                                                           \
    if (!(ABL & 1) || (TT) == 0) do {                                                                                              \
        const char* s__ = smem + (4 * ((TT) & 1) + 1 + (J)) * kHalfBytes;                             \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                              \
            fb[kk] = __builtin_bit_cast(bf16x8, *(const uint4*)(s__ + (offB ^ (kk * 32))));                                    \
    } while (0)
#define MI355_MFMA(I, J)                                                                              \
    do {                                                                                              \
        if (!(ABL & 4)) __builtin_amdgcn_s_setprio(1);                                                \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                              \
            _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                          \
                acc[I][rb][J] = screen_mfma<I8>(fa[rb][kk], fb[kk], acc[I][rb][J]);                   \
        if (!(ABL & 4)) __builtin_amdgcn_s_setprio(0);                                                \
    } while (0)
#define MI355_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

    // ---- prologue: stage K-tile 0 completely
    MI355_STAGE(0, 0);
    MI355_STAGE(1, 0);
    MI355_STAGE(2, 0);
    MI355_STAGE(3, 0);
    MI355_WAIT_VM(0);
    __syncthreads();
    if (group == 1) MI355_BARRIER();  // stagger: group 1's LOAD halves line up with group 0's MFMA halves

    for (int t = 0; t + 1 < T; ++t) {
        // phase 0: quadrant (0,0) -- new A0 and B0
        MI355_STAGE(0, t + 1);
        MI355_LOAD_A(0, t);
        MI355_LOAD_B(0, t);
        MI355_WAIT_VM(4);
        MI355_BARRIER();
        MI355_MFMA(0, 0);
        MI355_BARRIER();
        // phase 1: quadrant (0,1) -- new B1
        MI355_STAGE(1, t + 1);
        MI355_LOAD_B(1, t);
        MI355_WAIT_VM(4);
        MI355_BARRIER();
        MI355_MFMA(0, 1);
        MI355_BARRIER();
        // phase 2: quadrant (1,1) -- new A1
        MI355_STAGE(2, t + 1);
        MI355_LOAD_A(1, t);
        MI355_WAIT_VM(4);
        MI355_BARRIER();
        MI355_MFMA(1, 1);
        MI355_BARRIER();
        // phase 3: quadrant (1,0) -- B0 again
        MI355_STAGE(3, t + 1);
        MI355_LOAD_B(0, t);
        MI355_WAIT_VM(4);
        MI355_BARRIER();
        MI355_MFMA(1, 0);
        MI355_BARRIER();
    }
    {
        // last K-tile: nothing left to stage; drain the DMA queue as the half-tiles are needed
        const int t = T - 1;
        MI355_LOAD_A(0, t);
        MI355_LOAD_B(0, t);
        MI355_WAIT_VM(2);
        MI355_BARRIER();
        MI355_MFMA(0, 0);
        MI355_BARRIER();
        MI355_LOAD_B(1, t);
        MI355_WAIT_VM(0);
        MI355_BARRIER();
        MI355_MFMA(0, 1);
        MI355_BARRIER();
        MI355_LOAD_A(1, t);
        MI355_BARRIER();
        MI355_MFMA(1, 1);
        MI355_BARRIER();
        MI355_LOAD_B(0, t);
        MI355_BARRIER();
        MI355_MFMA(1, 0);
        MI355_BARRIER();
    }
    if (group == 0) MI355_BARRIER();  // balance the stagger barrier

#undef MI355_STAGE
#undef MI355_LOAD_A
#undef MI355_LOAD_B
#undef MI355_MFMA
#undef MI355_WAIT_VM

    // ---- fused epilogue (same rule as k_screen): column (query) = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = q0 + 64 * wc + 32 * j + (lane & 31);
        const float th = a.thr[q];
        const int thi = I8 ? a.thr_i[q] : 0;
        const float sc = (I8 && a.emit_all) ? a.sc[q] : 1.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const int64_t rbase = tile_row0 + 128 * wr + 64 * i + 32 * rb + 4 * (lane >> 5);
                if (a.emit_all) screen_emit_all_block<I8>(a, acc[i][rb][j], q, rbase, sc);  // wave-uniform branch
                else screen_emit_block<I8>(a, acc[i][rb][j], q, rbase, th, thi);
            }
    }
}

}  // namespace mi355
