// k_screen.h -- the dominant kernel: bf16 MFMA screen of a query block against a corpus chunk.
//
// What it computes: t[q, r] = <q_hat, c_hat_r>  (fp32 accumulate) for a 128-row x 128-query tile per
// workgroup, where q_hat / c_hat are the L2-normalised bf16 shadows; then, fused in the epilogue and
// without ever writing t to memory, it appends (r, t) to query q's candidate list iff t >= thr[q].
// thr[q] = (exact k-th best cosine so far) - E, with E a rigorous bound on |t - exact cosine|
// (DESIGN.md "Screen bound"), so the exact top-k is always a subset of the candidates; the exact
// fp32 re-score + select happens in k_select.h.  This replaces pgvector's per-row cosine_distance +
// top-N heap (reference: autorag_research/orm/repository/base.py:409-415, executed once per query).
//
// Roofline: per launch the kernel streams the chunk's shadow rows once from HBM (dpad*2 B per row;
// the 8 query-tile workgroups of one corpus tile are adjacent on one XCD so 7 of 8 reads hit L2) and
// does 2*128*128*dpad flop per tile on v_mfma_f32_32x32x16_bf16.  At 1024 queries it is MFMA-bound,
// below ~256 queries HBM-bound.
//
// Layout notes (gfx950):
//   * 256 threads = 4 waves as 2(row) x 2(query); each wave owns a 64x64 sub-tile = 2x2 MFMA 32x32 blocks.
//   * K is walked in steps of 64 bf16 (128 B per row).  Both operand tiles are staged with
//     global_load_lds_dwordx4 (16 B/lane, 8 rows x 128 B per wave instruction = full 128-B lines),
//     double-buffered: the loads of step t+1 are issued before the MFMAs of step t.
//   * LDS image is lane-linear (DMA constraint), so the bank swizzle is applied on the per-lane SOURCE
//     address and again on the ds_read_b128 address: 16-B chunk c of row r lives in slot c ^ ((r>>1)&7).
//     With that key the 16-lane groups of ds_read_b128 touch 16 distinct slots (conflict-free).
//   * blockIdx -> (corpus tile, query tile) is XCD-aware: blocks b, b+8, b+16.. run on one XCD, and
//     consecutive ones there share the corpus tile.
#pragma once
#include "dev_common.h"

namespace mi355 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTileM = 128;  // corpus rows per workgroup tile
constexpr int kTileN = 128;  // queries per workgroup tile
constexpr int kStepK = 64;   // bf16 elements per K step (128 B)
constexpr int kRowB = 128;   // bytes per staged row
constexpr int kTileBytes = kTileM * kRowB;             // 16 KiB per operand per buffer
constexpr int kScreenLds = 2 * 2 * kTileBytes;         // 64 KiB: 2 buffers x (A,B)

struct ScreenArgs {
    const uint16_t* shadow;  // [rows_pad, dpad] bf16
    const uint16_t* qhat;    // [Bpad, dpad] bf16
    const float* thr;        // [Bpad]
    int* cnt;                // [Bpad]
    int32_t* cand_row;       // [Bpad, cap]
    float* cand_val;         // [Bpad, cap]
    int dpad;
    int cap;
    int ct0;        // first corpus tile of this chunk
    int n_ctiles;   // corpus tiles in this chunk
    int n_qtiles;   // query tiles
    int64_t row_end;  // rows >= row_end are not part of this chunk (tile padding)
    int64_t row0;     // first row of this chunk
    int emit_all;     // first chunk: every (query,row) is a candidate -> direct store at slot row-row0, no atomics
};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__global__ __launch_bounds__(256, 2) void k_screen(ScreenArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware tile mapping
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int lb = b >> 3;
    const int qt = lb % a.n_qtiles;
    const int ctl = (lb / a.n_qtiles) * 8 + xcd;
    if (ctl >= a.n_ctiles) return;
    const int64_t tile_row0 = (int64_t)(a.ct0 + ctl) * kTileM;
    const int q0 = qt * kTileN;

    const int wr = wave >> 1, wc = wave & 1;
    const int64_t row_bytes = (int64_t)a.dpad * 2;

    // ---- staging addresses: this wave issues A-instructions ii = wave*4..+3 and the same B ones
    const char* gA[4];
    const char* gB[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int ii = wave * 4 + u;
        const int r = ii * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        gA[u] = (const char*)a.shadow + (tile_row0 + r) * row_bytes + c * 16;
        gB[u] = (const char*)a.qhat + (int64_t)(q0 + r) * row_bytes + c * 16;
    }
    // ---- fragment read offsets (bytes inside an operand tile), per MFMA block and K sub-step
    int offA[2], offB[2];
    {
        const int g = lane >> 5;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            const int ia = 64 * wr + 32 * blk + (lane & 31);
            const int ib = 64 * wc + 32 * blk + (lane & 31);
            // chunk for K sub-step kk is (2*kk + g); the swizzle key only touches bits 0..2 -> fold kk in later
            offA[blk] = ia * kRowB + ((g ^ ((ia >> 1) & 7)) << 4);
            offB[blk] = ib * kRowB + ((g ^ ((ib >> 1) & 7)) << 4);
        }
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int T = a.dpad / kStepK;
    // prologue: stage step 0 into buffer 0
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        glds16(gA[u], smem + (wave * 4 + u) * 1024);
        glds16(gB[u], smem + kTileBytes + (wave * 4 + u) * 1024);
    }
    for (int t = 0; t < T; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cur = t & 1;
        if (t + 1 < T) {
            char* nb = smem + (cur ^ 1) * (2 * kTileBytes);
            const int64_t koff = (int64_t)(t + 1) * kRowB;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                glds16(gA[u] + koff, nb + (wave * 4 + u) * 1024);
                glds16(gB[u] + koff, nb + kTileBytes + (wave * 4 + u) * 1024);
            }
        }
        const char* bufA = smem + cur * (2 * kTileBytes);
        const char* bufB = bufA + kTileBytes;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            // chunk index (2*kk+g) ^ key == ((g ^ key) ^ (2*kk)) because 2*kk only sets bits 1..2
            const int kx = (2 * kk) << 4;
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                fa[blk] = __builtin_bit_cast(bf16x8, *(const uint4*)(bufA + (offA[blk] ^ kx)));
                fb[blk] = __builtin_bit_cast(bf16x8, *(const uint4*)(bufB + (offB[blk] ^ kx)));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- fused epilogue: threshold test, rare append
    // C/D layout of the 32x32 MFMA: column (query) = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (a.emit_all) {  // wave-uniform: the first chunk keeps everything, slot = row - row0 (counts set by the host)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = q0 + 64 * wc + 32 * j + (lane & 31);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int64_t rbase = tile_row0 + 64 * wr + 32 * i + 4 * (lane >> 5);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row < a.row_end) {
                        a.cand_row[(int64_t)q * a.cap + (row - a.row0)] = (int32_t)row;
                        a.cand_val[(int64_t)q * a.cap + (row - a.row0)] = acc[i][j][r];
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = q0 + 64 * wc + 32 * j + (lane & 31);
        const float th = a.thr[q];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float m = acc[i][j][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[i][j][r]);
            if (m >= th) {
                const int64_t rbase = tile_row0 + 64 * wr + 32 * i + 4 * (lane >> 5);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][j][r];
                    const int64_t row = rbase + (r & 3) + 8 * (r >> 2);
                    if (v >= th && row < a.row_end) {
                        const int slot = atomicAdd(&a.cnt[q], 1);
                        if (slot < a.cap) {
                            a.cand_row[(int64_t)q * a.cap + slot] = (int32_t)row;
                            a.cand_val[(int64_t)q * a.cap + slot] = v;
                        }
                    }
                }
            }
        }
    }
}

}  // namespace mi355
