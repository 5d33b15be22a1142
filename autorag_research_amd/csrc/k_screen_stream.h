// k_screen_stream.h -- the screen for SMALL query blocks (at most 64 queries per pass: the reference's call shape is ONE
// query per call, pipelines/retrieval/vector_search.py:157-169).
//
// With so few queries the MFMA work of a corpus tile is nothing and a pass is the time to stream the shadow rows once.
// k_screen (128 x 128 tiles, one tile per workgroup, double-buffered) keeps one 16-KiB stage of rows in flight per
// workgroup, two workgroups per CU: 32 KiB per CU against a ~2 us HBM round trip is ~4-4.7 TB/s chip-wide.  Here:
//   * persistent workgroups (one per CU), each walking corpus tiles b, b + G, ...;
//   * the query block is RESIDENT in LDS for the whole launch (NQ x row_bytes <= 48 KiB, staged once);
//   * the rows stream through a ring of kStreamStages x 16 KiB (128 rows x 128 B of K per stage) filled by LDS-DMA in the
//     SGPR-base form, kStreamStages - 1 stages = 80 KiB per CU in flight, counted vmcnt waits, one barrier per stage;
//   * 4 waves, wave w owns rows 32 w .. 32 w + 31 of the tile x all NQ queries: NQ / 32 accumulator blocks;
//   * same LDS image, swizzle, fragment typing, threshold test and append path (screen_emit_block) as k_screen: the
//     candidate sets are those of k_screen on the same thresholds.
// Past the last stage of its last tile a workgroup keeps re-staging that tile (valid memory) so that the number of pieces
// in flight -- what the counted wait relies on -- stays constant; drained before the exit.
#pragma once
#include "k_screen256_common.h"

namespace mi355 {

constexpr int kStreamStages = 6;
constexpr int kStreamStageBytes = kTileM * kRowB;  // 16 KiB
constexpr int kStreamQueryBytesMax = 48 * 1024;    // resident query image
__host__ __device__ inline size_t screen_stream_lds(int nq, int row_bytes) {
    return (size_t)nq * row_bytes + (size_t)kStreamStages * kStreamStageBytes;
}
static_assert(kStreamQueryBytesMax + kStreamStages * kStreamStageBytes <= 160 * 1024, "LDS per workgroup");

template <bool I8, int NQ>
__global__ __launch_bounds__(256, 1) void k_screen_stream(ScreenArgs a) {
    static_assert(NQ == 32 || NQ == 64, "one or two query blocks of 32");
    constexpr int NJ = NQ / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.ksteps;
    const int G = (int)gridDim.x;
    int ctl_c = (int)blockIdx.x;
    if (ctl_c >= a.n_ctiles) return;
    const int n_my = (a.n_ctiles - ctl_c + G - 1) / G;
    const int N = n_my * T;
    const int64_t row_bytes = a.row_bytes;
    char* const bimg = smem;                            // [T][NQ rows x 128 B]
    char* const ring = smem + (size_t)T * NQ * kRowB;   // [kStreamStages][128 rows x 128 B]

    // ---- DMA sources (swizzled like k_screen: 16-B chunk c of row r lives in slot c ^ ((r>>1)&7))
    unsigned voffA[4], voffB[NJ];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = (wave * 4 + u) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        voffA[u] = (unsigned)(r * (int)row_bytes + c * 16);
    }
#pragma unroll
    for (int u = 0; u < NJ; ++u) {
        const int r = (wave * NJ + u) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        voffB[u] = (unsigned)(r * (int)row_bytes + c * 16);
    }
    // ---- fragment read offsets
    int offA, offB[NJ];
    {
        const int g = lane >> 5;
        const int ia = 32 * wave + (lane & 31);
        offA = ia * kRowB + ((g ^ ((ia >> 1) & 7)) << 4);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int ib = 32 * j + (lane & 31);
            offB[j] = ib * kRowB + ((g ^ ((ib >> 1) & 7)) << 4);
        }
    }
    float th[NJ], sq[NJ], kq[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int q = 32 * j + (lane & 31);
        th[j] = a.thr[q];
        sq[j] = I8 ? a.sc[q] : 1.0f;
        kq[j] = I8 ? a.kq[q] : 1.0f;
    }
    // a use here makes the compiler wait for these loads NOW: left to their first use -- the test at the end of a tile -- its
    // vmcnt(0) would sit inside the loop and drain the ring once per tile
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(th[j]), "v"(sq[j]), "v"(kq[j]));

    // ---- the query block, once: K-step t of query row r at bimg + t * NQ*128 (+ swizzled chunk)
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int u = 0; u < NJ; ++u)
            glds16_saddr((const char*)a.qhat + (int64_t)t * kRowB, voffB[u],
                         lds_addr(bimg + (size_t)t * NQ * kRowB + (wave * NJ + u) * 1024));

    // ---- staging cursor (the next stage to issue)
    const int64_t tile_bytes = (int64_t)kTileM * row_bytes;
    const char* i_base = (const char*)a.shadow + (int64_t)(a.ct0 + ctl_c) * tile_bytes;
    int i_t = 0, i_ctl = ctl_c, i_slot = 0;
    auto issue = [&]() __attribute__((always_inline)) {
        char* const dst = ring + (size_t)i_slot * kStreamStageBytes + wave * 4096;
#pragma unroll
        for (int u = 0; u < 4; ++u) glds16_saddr(i_base + (int64_t)i_t * kRowB, voffA[u], lds_addr(dst + u * 1024));
        if (++i_t == T) {
            i_t = 0;
            if (i_ctl + G < a.n_ctiles) {  // (else: stay on the last tile -- dummy re-stage of valid memory)
                i_ctl += G;
                i_base += (int64_t)G * tile_bytes;
            }
        }
        if (++i_slot == kStreamStages) i_slot = 0;
    };
#pragma unroll
    for (int s = 0; s < kStreamStages - 1; ++s) issue();

    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    int t_c = 0, slot_c = 0;
    for (int s = 0; s < N; ++s) {
        // stage s has landed (this wave's pieces: the 4 (kStreamStages - 2) youngest may still fly; the query block went first)
        static_assert(kStreamStages == 6, "the counted wait below is 4 * (kStreamStages - 2)");
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        MI355_BARRIER();  // ... everybody's have; and everybody is done reading stage s - 1, whose slot is refilled now
        issue();
        const char* const bufA = ring + (size_t)slot_c * kStreamStageBytes;
        const char* const bufB = bimg + (size_t)t_c * NQ * kRowB;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int kx = (2 * kk) << 4;
            const bf16x8 fa = __builtin_bit_cast(bf16x8, *(const uint4*)(bufA + (offA ^ kx)));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const bf16x8 fb = __builtin_bit_cast(bf16x8, *(const uint4*)(bufB + (offB[j] ^ kx)));
                acc[j] = screen_mfma<I8>(fa, fb, acc[j]);
            }
        }
        if (++slot_c == kStreamStages) slot_c = 0;
        if (++t_c == T) {  // the tile is complete: threshold test, rare append, restart
            t_c = 0;
            const int64_t row0 = (int64_t)(a.ct0 + ctl_c) * kTileM + 32 * wave;  // wave-uniform: one row group
            const int64_t rbase = row0 + 4 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                I8Blk blk{1.0f, 0.0f};
                if constexpr (I8) blk = i8_blk(i8_group_of(a.grp, row0), sq[j], kq[j]);
                screen_emit_block<I8>(a, acc[j], 32 * j + (lane & 31), rbase, th[j], blk);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
            }
            ctl_c += G;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy stages must land before the LDS is freed
}

}  // namespace mi355
