// k_screen256_common.h -- what the large-block screen (k_screen256c.h) and the streaming screen (k_screen_stream.h) share:
// the 256 x 256 tile geometry, the LDS carve, the persistent grid, the LDS-DMA staging primitives.
//
//  * Tile 256 corpus rows x 256 queries, K step = 128 B of every row.  8 waves; wave (wr, wc) owns rows [128 wr, +128) x
//    queries [64 wc, +64) = 4 x 2 MFMA blocks = 128 accumulator VGPRs.
//  * A K-step is cut into 4 half-tiles of 128 rows x 128 B (16 KiB): A0 / A1 = the first / second 64 rows of every wave-row,
//    B0 / B1 = the first / second 32 queries of every wave-column.  The LDS ring holds 8 half-tile slots (2 K-steps).
//  * Persistent: the grid is 8 XCDs x L workgroups (one per CU, the ring fills the LDS); a workgroup keeps its query tile and
//    walks corpus tiles ctl, ctl + 8 L / n_qtiles, ...  Workgroups that run together on an XCD are the query tiles of
//    neighbouring corpus tiles, so the shadow is fetched from HBM once per pass and re-read from that XCD's L2.
//  * Hits go to a per-wave LDS queue (k_screen.h: screen_queue_hits) that is flushed to the global candidate lists when it
//    fills up and when the workgroup is done: the epilogue never touches the vector-memory counter while DMA is in flight.
// (The first, second and fourth forms of the kernel that were built on this geometry -- docs/LAB_NOTES_r1_r3.md 4.1, 4.1b,
// 4.1c -- are in the history only.  Round 5: k_screen_rq.h keeps the QUERY operand in registers and serves int8 shadows of at
// most 768 B per row; k_screen256c stays for the bf16 shadow and wider rows.  A/B harness: tools/screen_ab.hip.)
#pragma once
#include "k_screen.h"

namespace mi355 {

constexpr int kT2 = 256;                        // tile edge (rows and queries)
constexpr int kHalfBytes = 128 * kRowB;         // 16 KiB
constexpr int kRingBytes = 8 * kHalfBytes;      // ring of 8 half-tiles
constexpr int kRecOff = kRingBytes + 8 * kWaveQueueCap * 12;  // + one candidate queue per wave
constexpr int kRecBytes = 4 * 256;  // + 4 slots x 256 B of int8 row-group records; 128 + 30 + 1 KiB of 160
constexpr int kScreen256Lds = kRecOff + kRecBytes;
static_assert(kScreen256Lds <= 160 * 1024, "LDS per workgroup");

// persistent grid: 8 XCDs x L workgroups, L = the largest multiple of n_qtiles that fits the 32 CUs of an XCD
// (fewer when the chunk has fewer tiles)
__host__ __device__ inline unsigned screen256_grid(int n_ctiles, int n_qtiles) {
    const int lmax = (32 / n_qtiles) * n_qtiles;
    const int need = ((n_ctiles + 7) / 8) * n_qtiles;
    return 8u * (unsigned)(need < lmax ? need : lmax);
}

#define MI355_BARRIER()                      \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)

struct ScreenArgs2 : ScreenArgs {
    int* status;  // [Bpad] per-query status bits (kStOverflow is set when a wave's queue overflows)
    // k_screen_rq's sibling drift limiter (k_screen_rq.h); drift = 0 or progress = nullptr: off
    int* progress = nullptr;  // [kRqProgressWords] tile counters of the persistent workgroups, 8 words per row-tile slot
    int epoch = 0;            // launch stamp (12 bits) in the words' high bits: words of other launches are ignored
    int drift = 0;            // tiles a workgroup may run ahead of the slowest workgroup on the same row tiles
    // k_screen_rq, hit-lane queues: >= 0 = the eight waves of a workgroup flush their queues TOGETHER, at every tile whose
    // number has no bit of this mask set (period = mask + 1 tiles, chosen by the host from the hit density it expects);
    // < 0 = every wave on its own when its queue passes kLaneQueueFlushAt (round 5)
    int flush_mask = -1;
    int flush_alone = 40;     // ... under that schedule a wave still flushes by itself once its queue holds more than this many entries
};

// ---- one 1-KiB piece (U = 0,1) of half-tile type S into ring parity `par`; src = the half-tile's first row + K offset
// LDS-DMA with the source address split as the hardware takes it: a wave-uniform 64-bit base in SGPRs + a 32-bit
// per-lane offset (the "saddr" form of global_load).  The builtin form adds the two into a 64-bit VGPR pair per lane
// (one v_lshl_add_u64 per piece, and twice the address payload from the register file to the texture addresser).  M0 =
// LDS destination of the wave's 1-KiB piece; it is written in the same statement (the compiler does not preserve it).
__device__ __forceinline__ void glds16_saddr(const char* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
// 256 B (one dword per lane) through the same path: the int8 row-group records of a tile (k_screen256c)
__device__ __forceinline__ void glds4_saddr(const char* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
// ... the same at agent scope (sc1: not served from this CU's vector L1): words that other workgroups keep writing
__device__ __forceinline__ void glds4_saddr_sc1(const char* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1 sc1" ::"v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
template <int S, bool SADDR>
__device__ __forceinline__ void kb_stage(char* smem, int wave, int par, const char* src, const unsigned (&voff)[2], int u) {
    char* const dst = smem + (4 * par + S) * kHalfBytes + (2 * wave + u) * 1024;
    if constexpr (SADDR) glds16_saddr(src, voff[u], lds_addr(dst));
    else glds16(src + voff[u], dst);
}

}  // namespace mi355
