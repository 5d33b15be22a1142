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
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int kTileM = 128;  // corpus rows per workgroup tile
constexpr int kTileN = 128;  // queries per workgroup tile
constexpr int kStepK = 64;   // bf16 elements per K step (128 B)
constexpr int kRowB = 128;   // bytes per staged row
constexpr int kTileBytes = kTileM * kRowB;             // 16 KiB per operand per buffer
constexpr int kScreenLds = 2 * 2 * kTileBytes;         // 64 KiB: 2 buffers x (A,B)

struct ScreenArgs {
    const void* shadow;      // [rows_pad, row_bytes]  bf16 (2 B/element) or int8 shadow rows
    const void* qhat;        // [Bpad, row_bytes]      same element type
    const float* thr;        // [Bpad]  emit iff v >= thr  (bf16: v = t; int8: v = S_q S_g acc + e_g kq, dev_common.h)
    const float* sc;         // [Bpad]  int8 screen: the query's step S_q
    const float* kq;         // [Bpad]  int8 screen: factor on the row group's residual norm
    const I8Group* grp;      // [rows_pad / 32]  int8 screen: step and residual norm of every group of 32 rows
    const uint8_t* flag8;    // [rows]  int8 screen: 1 = row is not in the int8 shadow (read by the emit-all epilogue only)
    int* cnt;                // [Bpad]
    int32_t* cand_row;       // [Bpad, cap]
    float* cand_val;         // [Bpad, cap]
    int row_bytes;           // bytes per shadow row (a multiple of 128)
    int ksteps;              // row_bytes / 128: K steps of 64 bf16 or 128 int8
    int cap;
    int ct0;        // first corpus tile of this chunk
    int n_ctiles;   // corpus tiles in this chunk
    int n_qtiles;   // query tiles
    int64_t row_end;  // rows >= row_end are not part of this chunk (tile padding)
    int64_t row0;     // first row of this chunk
    int emit_all;     // 1 = first chunk: every (query,row) is a candidate -> direct store at slot row-row0, no atomics;
                      // 2 = starter (k_screen only): per query and 64-row slab ONLY the largest value, at slot (slab index)
};
constexpr int kEmitAll = 1, kEmitSlabMax = 2;
constexpr int kSlabRows = 64;  // rows of one wave's sub-tile in k_screen: the starter keeps one candidate per slab and query

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// one 32x32 MFMA step on 16-byte operand fragments (carried as bf16x8 registers): 16 bf16 k-values (I8 = false)
// or 32 int8 k-values (I8 = true).
// The int8 form accumulates exact int32; its accumulator travels in the same f32x16 registers (bit pattern).
template <bool I8>
__device__ __forceinline__ f32x16 screen_mfma(bf16x8 fa, bf16x8 fb, f32x16 acc) {
    if constexpr (I8) {
        return __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, fa),
                                                                                __builtin_bit_cast(i32x4, fb),
                                                                                __builtin_bit_cast(i32x16, acc), 0, 0, 0));
    } else {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    }
}

// int8 screen: the two per-lane constants of one 32x32 accumulator block (32 rows = one I8Group, 32 queries = the lanes):
// v = fma((float)acc, m, ek).  `g` is wave-uniform (a scalar load), sq / kq are the lane's query.
struct I8Blk {
    float m, ek;
};
__device__ __forceinline__ I8Blk i8_blk(const I8Group g, float sq, float kq) { return I8Blk{g.step * sq, g.err * kq}; }
__device__ __forceinline__ float i8_value(int acc, const I8Blk& b) { return __builtin_fmaf((float)acc, b.m, b.ek); }
// the group record of the block whose first row is row0 (a multiple of 32), through the scalar data cache: the address is
// wave-uniform, and the constant address space tells the compiler that nothing in this kernel writes it
__device__ __forceinline__ I8Group i8_group_of(const I8Group* grp, int64_t row0) {
    typedef const __attribute__((address_space(4))) float cfloat;
    cfloat* p = (cfloat*)(const float*)(grp + (row0 >> 5));
    return I8Group{p[0], p[1]};
}

// Fused epilogue of one 32x32 accumulator block, shared by both screen kernels.  C/D layout: column (query) =
// lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5); rbase = first row of the block + 4*(lane>>5).
template <bool I8>
__device__ __forceinline__ void screen_emit_block(const ScreenArgs& a, f32x16 acc, int q, int64_t rbase, float th,
                                                  I8Blk blk) {
    // fast path (almost always): one max over the lane's 16 rows and one compare
    bool any;
    if constexpr (I8) {
        const i32x16 v = __builtin_bit_cast(i32x16, acc);
        int m = v[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = max(m, v[r]);
        any = i8_value(m, blk) >= th;
    } else {
        float m = acc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
        any = m >= th;
    }
    if (!any) return;
    // hit path: collect the lane's hits in a mask, reserve all their slots with ONE atomic, then store.
    // (int8: rows outside the int8 shadow may show up here with a stale 0 -- k_prune drops them.)
    unsigned mask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = rbase + (r & 3) + 8 * (r >> 2);
        bool hit;
        if constexpr (I8) hit = i8_value(__builtin_bit_cast(i32x16, acc)[r], blk) >= th;
        else hit = acc[r] >= th;
        if (hit && row < a.row_end) mask |= 1u << r;
    }
    if (mask == 0) return;
    int slot = atomicAdd(&a.cnt[q], __builtin_popcount(mask));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if ((mask >> r) & 1u) {
            if (slot < a.cap) {
                float val;
                if constexpr (I8) val = i8_value(__builtin_bit_cast(i32x16, acc)[r], blk);
                else val = acc[r];
                a.cand_row[(int64_t)q * a.cap + slot] = (int32_t)(rbase + (r & 3) + 8 * (r >> 2));
                a.cand_val[(int64_t)q * a.cap + slot] = val;
            }
            ++slot;
        }
    }
}

// 32-bit LDS address of a pointer into dynamic shared memory
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(unsigned long)((const __attribute__((address_space(3))) char*)p);
}

// ---- k_screen256's form of the hit path: a per-WAVE candidate queue in LDS ----
// Hits are appended with wave-level bookkeeping only (ballot + mbcnt, count in an SGPR): no atomics, and no
// vector-memory instruction, so the in-flight LDS-DMA prefetch of the next tile is never waited for (vmcnt
// completes in order: waiting for a global atomic's return would wait for the whole prefetch).  A wave flushes its
// queue to the global candidate lists between two tiles when it is more than half full, and when the workgroup is done.
constexpr int kWaveQueueCap = 320;  // entries (q, row, value) per wave; 8 waves x 320 x 12 B = 30 KiB

// flush a wave's queue: one global atomic per entry.  Inlined at ONE site per tile (k_screen256) -- a call would
// make the register allocator spill the accumulators around it.
__device__ __forceinline__ void wave_queue_flush(const ScreenArgs& a, const int32_t* que, int n) {
    const int lane = threadIdx.x & 63;
    for (int e = lane; e < n; e += kWave) {
        const int q = que[e];
        const int slot = atomicAdd(&a.cnt[q], 1);
        if (slot < a.cap) {
            a.cand_row[(int64_t)q * a.cap + slot] = que[kWaveQueueCap + e];
            a.cand_val[(int64_t)q * a.cap + slot] = __int_as_float(que[2 * kWaveQueueCap + e]);
        }
    }
}

// The append path of k_screen256c, OUT OF LINE: one copy per kernel instead of one per test site.  Inlined at the 12 test
// sites of that kernel's K-step it made the loop body ~60 KB of code -- the hot path hopping over a dozen cold blocks, more
// than the instruction cache holds -- and cost 20 % of the kernel although it almost never runs.  A call spills the caller's
// live registers around the call site only, i.e. on the rare path.  Returns the wave's new queue fill.
template <bool I8>
__device__ __forceinline__ int screen_queue_hits_body(f32x16 acc, int any_i, int q, int rbase, int row_end, float th,
                                                     float m, float ek, unsigned a_q, int que_n, int* status) {
    const bool any = any_i != 0;
    bool gany[4];
    int thi = 0;
    if constexpr (I8) {
        const i32x16 v = __builtin_bit_cast(i32x16, acc);
        thi = i8_block_threshold(th, m, ek);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            gany[i] = any && max(max(v[4 * i], v[4 * i + 1]), max(v[4 * i + 2], v[4 * i + 3])) >= thi;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            gany[i] = fmaxf(fmaxf(acc[4 * i], acc[4 * i + 1]), fmaxf(acc[4 * i + 2], acc[4 * i + 3])) >= th;
    }
    const unsigned a_r = a_q + 4u * kWaveQueueCap, a_v = a_q + 8u * kWaveQueueCap;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
        if (__builtin_amdgcn_ballot_w64(gany[gi]) == 0) continue;  // wave-uniform: no hit in this group of four
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
            const int r = 4 * gi + ri;
            bool hit;
            if constexpr (I8) hit = any && __builtin_bit_cast(i32x16, acc)[r] >= thi;
            else hit = acc[r] >= th;
            if (__builtin_amdgcn_ballot_w64(hit) == 0) continue;  // wave-uniform
            const int row = rbase + (r & 3) + 8 * (r >> 2);
            hit = hit && row < row_end;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
            if (hit) {
                const unsigned e = (unsigned)que_n +
                                   __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
                float val;
                if constexpr (I8) val = __builtin_fmaf((float)__builtin_bit_cast(i32x16, acc)[r], m, ek);
                else val = acc[r];
                if (e < (unsigned)kWaveQueueCap) {
                    asm volatile("ds_write_b32 %0, %1" ::"v"(a_q + 4u * e), "v"(q) : "memory");
                    asm volatile("ds_write_b32 %0, %1" ::"v"(a_r + 4u * e), "v"(row) : "memory");
                    asm volatile("ds_write_b32 %0, %1" ::"v"(a_v + 4u * e), "v"(val) : "memory");
                } else {  // queue full: the query is re-screened by the host
                    __hip_atomic_fetch_or(&status[q], kStOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            que_n += __builtin_popcountll(bal);  // may run past the capacity: the flush clamps
        }
    }
    return que_n;
}
template <bool I8>
__device__ __attribute__((noinline)) int screen_queue_hits(f32x16 acc, int any_i, int q, int rbase, int row_end, float th,
                                                            float m, float ek, unsigned a_q, int que_n, int* status) {
    return screen_queue_hits_body<I8>(acc, any_i, q, rbase, row_end, th, m, ek, a_q, que_n, status);
}
// the test itself, inline: 15 max + the compare; the call only when some lane passes
template <bool I8>
__device__ __forceinline__ void screen_test_block(int* status, f32x16 acc, int q, int rbase, int row_end, float th, I8Blk blk,
                                                  int32_t* que, int& que_n) {
    bool any;
    if constexpr (I8) {
        const i32x16 v = __builtin_bit_cast(i32x16, acc);
        int g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = max(max(v[4 * i], v[4 * i + 1]), max(v[4 * i + 2], v[4 * i + 3]));
        any = i8_value(max(max(g[0], g[1]), max(g[2], g[3])), blk) >= th;
    } else {
        float g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = fmaxf(fmaxf(acc[4 * i], acc[4 * i + 1]), fmaxf(acc[4 * i + 2], acc[4 * i + 3]));
        any = fmaxf(fmaxf(g[0], g[1]), fmaxf(g[2], g[3])) >= th;
    }
    if (__builtin_amdgcn_ballot_w64(any) == 0) return;  // wave-uniform: almost always taken
    que_n = screen_queue_hits<I8>(acc, any ? 1 : 0, q, rbase, row_end, th, blk.m, blk.ek, lds_addr(que), que_n, status);
}
// A/B form (k_screen256c ABL bit 12): the append path INLINE at every test site but marked unlikely, so that block placement
// moves the twelve copies behind the loop -- no call, no argument moves, and above all no function entry: the calling
// convention opens every device function with s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0), which makes a wave with a hit wait for
// every LDS-DMA piece it has in flight.
template <bool I8>
__device__ __forceinline__ void screen_test_block_cold(int* status, f32x16 acc, int q, int rbase, int row_end, float th, I8Blk blk,
                                                       int32_t* que, int& que_n) {
    bool any;
    if constexpr (I8) {
        const i32x16 v = __builtin_bit_cast(i32x16, acc);
        int g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = max(max(v[4 * i], v[4 * i + 1]), max(v[4 * i + 2], v[4 * i + 3]));
        any = i8_value(max(max(g[0], g[1]), max(g[2], g[3])), blk) >= th;
    } else {
        float g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = fmaxf(fmaxf(acc[4 * i], acc[4 * i + 1]), fmaxf(acc[4 * i + 2], acc[4 * i + 3]));
        any = fmaxf(fmaxf(g[0], g[1]), fmaxf(g[2], g[3])) >= th;
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(any) != 0, 0))
        que_n = screen_queue_hits_body<I8>(acc, any ? 1 : 0, q, rbase, row_end, th, blk.m, blk.ek, lds_addr(que), que_n, status);
}

// ---- k_screen_rq's form of the hit path (round 5): a per-wave queue of HIT LANES -----------------------------------------
// What a hit cost before (tools/screen_ab THRZ, profiles/r05_hit_path.txt): the out-of-line append is entered through the
// calling convention's `s_waitcnt vmcnt(0)` -- the wave waits for every LDS-DMA piece it has in flight, the youngest issued a
// few hundred cycles earlier -- and then walks its sixteen registers with a scalar branch each, while the other seven waves of
// the workgroup wait at the next K-step barrier: 0.22 ns per hit chip-wide in a hit-dense chunk (3 hits per block), 0.67 ns
// in the last chunks (one hit per ten blocks: one call per hit) -- ~0.55 ms of a 6.9 ms pass at N = 10 M.
// Now a block with a hit costs the hot loop five LDS stores and no call: every lane whose float test passed writes ITS OWN
// sixteen accumulators + (query, first row, m, ek) -- 80 bytes -- to entry `rank` of the wave's queue (ballot + mbcnt; inline
// asm stores, no vector memory, no wait).  The queue is expanded into candidates at a tile's start once it holds more than 24
// entries, and at the kernel's end: one LANE per entry, all entries in parallel.
constexpr int kLaneQueueCap = 64;          // entries per wave: a block's hit lanes always fit an empty queue (one lane = one entry at the flush)
constexpr int kLaneQueueEntryBytes = 80;   // 16 accumulators (64 B, entry e at 64 e) + (q, rbase, m, ek) (16 B, at 64 cap + 16 e)
constexpr int kLaneQueueBytes = kLaneQueueCap * kLaneQueueEntryBytes;  // 5 KiB per wave

__device__ __forceinline__ void lds_store16(unsigned addr, i32x4 v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// expand entries [0, n) of the wave's lane queue into the global candidate lists (lane e = entry e).  INLINED at one site per
// tile (a call there made the allocator park query fragments in scratch and reload them inside the hot loop; at a tile's
// start two accumulator blocks are dead, which is the room this body lives in).  The caller has waited for its LDS-DMA before.
template <bool I8>
__device__ __forceinline__ void lane_queue_flush(const ScreenArgs& a, unsigned lq_addr, int n, int row_end) {
    const int lane = threadIdx.x & 63;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the entries were written with inline-asm stores the compiler does not track)
    if (lane < n) {
        const __attribute__((address_space(3))) int* e =
            (const __attribute__((address_space(3))) int*)(unsigned long)(lq_addr + (unsigned)lane * 64u);
        const __attribute__((address_space(3))) int* em =
            (const __attribute__((address_space(3))) int*)(unsigned long)(lq_addr + (unsigned)(kLaneQueueCap * 64) + (unsigned)lane * 16u);
        const int q = em[0], rbase = em[1];
        const float m = __int_as_float(em[2]), ek = __int_as_float(em[3]);
        const float th = a.thr[q];
        int thi = 0;
        if constexpr (I8) thi = i8_block_threshold(th, m, ek);
        // sixteen values in four 16-byte reads, ONE returning atomic per entry (it reserves the slots of all its hits: the
        // atomic's round trip is the flush's longest step), then the stores
        typedef __attribute__((address_space(3))) const i32x4 lds_i32x4;
        i32x4 w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = ((lds_i32x4*)e)[i];
        unsigned mask = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int v = w[r >> 2][r & 3];
            bool hit;
            if constexpr (I8) hit = v >= thi;
            else hit = __int_as_float(v) >= th;
            if (hit && rbase + (r & 3) + 8 * (r >> 2) < row_end) mask |= 1u << r;
        }
        if (mask != 0) {
            int slot = atomicAdd(&a.cnt[q], (int)__builtin_popcount(mask));
            int32_t* const cr = a.cand_row + (int64_t)q * a.cap;
            float* const cv = a.cand_val + (int64_t)q * a.cap;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if ((mask >> r) & 1u) {
                    if (slot < a.cap) {
                        const int v = w[r >> 2][r & 3];
                        cr[slot] = rbase + (r & 3) + 8 * (r >> 2);
                        cv[slot] = I8 ? __builtin_fmaf((float)v, m, ek) : __int_as_float(v);
                    }
                    ++slot;
                }
            }
        }
    }
}

// The same expansion with a small register footprint (the sixteen values read back one at a time in a rolled loop, one atomic
// per hit): for the test sites, where every accumulator is live and the unrolled body above would spill.  It runs when a
// block's hit lanes do not fit the queue any more -- the queue is flushed at a tile's start whenever it holds more than
// kLaneQueueFlushAt entries, so ONE tile has to bring more than 64 - 24 hit lanes to one wave: thresholds still loose (small k
// over few rows, the chunks right behind an emit-all ladder) or a burst of near-duplicate rows.
template <bool I8>
__device__ __forceinline__ void lane_queue_flush_small(const ScreenArgs& a, unsigned lq_addr, int n, int row_end) {
    const int lane = threadIdx.x & 63;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < n) {
        const __attribute__((address_space(3))) int* e =
            (const __attribute__((address_space(3))) int*)(unsigned long)(lq_addr + (unsigned)lane * 64u);
        const __attribute__((address_space(3))) int* em =
            (const __attribute__((address_space(3))) int*)(unsigned long)(lq_addr + (unsigned)(kLaneQueueCap * 64) + (unsigned)lane * 16u);
        const int q = em[0], rbase = em[1];
        const float m = __int_as_float(em[2]), ek = __int_as_float(em[3]);
        const float th = a.thr[q];
        int thi = 0;
        if constexpr (I8) thi = i8_block_threshold(th, m, ek);
#pragma unroll 1
        for (int r = 0; r < 16; ++r) {
            const int v = e[r];
            bool hit;
            if constexpr (I8) hit = v >= thi;
            else hit = __int_as_float(v) >= th;
            const int row = rbase + (r & 3) + 8 * (r >> 2);
            if (hit && row < row_end) {
                const int slot = atomicAdd(&a.cnt[q], 1);
                if (slot < a.cap) {
                    a.cand_row[(int64_t)q * a.cap + slot] = row;
                    a.cand_val[(int64_t)q * a.cap + slot] = I8 ? __builtin_fmaf((float)v, m, ek) : __int_as_float(v);
                }
            }
        }
    }
}
constexpr int kLaneQueueFlushAt = 24;

// the test of one 32 x 32 block + the enqueue of its hit lanes (k_screen_rq).  `lq_n` = entries in the wave's queue (wave-uniform).
// What a hit costs is NOT its instructions but the barrier: the eight waves of a workgroup meet every K-step, so whatever delays
// ONE wave -- even a taken branch alone, measured -- is paid by all eight, and with one hit per 3 ... 30 blocks some wave of the
// eight has one at most test sites (profiles/r05_hit_path.txt: this queue behind a branch costs the same as the out-of-line
// append it replaced; k_screen_rq's answer is its hand-over schedule -- all tests of a tile between two barriers).
// MODE (timing builds / A-B): 0 = the stores behind a wave-uniform branch (the kernel), 1 = bookkeeping without the stores,
// 2 = BRANCH-FREE: the five stores always issued under EXEC = hit lanes (uniform cost for every wave; measured +15 % with the
// thresholds parked: stores under an empty EXEC are not free -- not adopted), 3 = as 0 with the block laid out as fall-through.
// The block's largest value, in four PARTS of two v_max3 each (k_screen_rq issues one part behind each of four MFMAs: a test in
// one piece is ~14 dependent vector instructions during which its wave feeds the matrix pipe nothing -- and the other wave of
// the SIMD, in lockstep behind the same barriers, is at its own test).  `g` = the running maximum (int32 bits / float bits).
template <bool I8, int PART>
__device__ __forceinline__ int screen_block_max_part(const f32x16& acc, int g) {
    if constexpr (I8) {
        const i32x16 v = __builtin_bit_cast(i32x16, acc);
        if constexpr (PART == 0) return max(max(max(v[0], v[1]), v[2]), v[3]);
        else return max(max(max(max(g, v[4 * PART]), v[4 * PART + 1]), v[4 * PART + 2]), v[4 * PART + 3]);  // two v_max3_i32
    } else {
        if constexpr (PART == 0) return __float_as_int(fmaxf(fmaxf(fmaxf(acc[0], acc[1]), acc[2]), acc[3]));
        else return __float_as_int(fmaxf(fmaxf(fmaxf(fmaxf(__int_as_float(g), acc[4 * PART]), acc[4 * PART + 1]), acc[4 * PART + 2]), acc[4 * PART + 3]));
    }
}
template <bool I8, int MODE = 0>
__device__ __forceinline__ void screen_test_block_lq_max(const ScreenArgs& a, int* status, int row_end, f32x16 acc, int gmax, int q, int rbase,
                                                         float th, I8Blk blk, unsigned lq_addr, int& lq_n, int& lq_ovf);
template <bool I8, int MODE = 0>
__device__ __forceinline__ void screen_test_block_lq(const ScreenArgs& a, int* status, int row_end, f32x16 acc, int q, int rbase, float th,
                                                     I8Blk blk, unsigned lq_addr, int& lq_n, int& lq_ovf) {
    int g = screen_block_max_part<I8, 0>(acc, 0);
    g = screen_block_max_part<I8, 1>(acc, g);
    g = screen_block_max_part<I8, 2>(acc, g);
    g = screen_block_max_part<I8, 3>(acc, g);
    screen_test_block_lq_max<I8, MODE>(a, status, row_end, acc, g, q, rbase, th, blk, lq_addr, lq_n, lq_ovf);
}
// ... the rest of the test, given the block's largest value
template <bool I8, int MODE>
__device__ __forceinline__ void screen_test_block_lq_max(const ScreenArgs& a, int* status, int row_end, f32x16 acc, int gmax, int q, int rbase,
                                                         float th, I8Blk blk, unsigned lq_addr, int& lq_n, int& lq_ovf) {
    bool any;
    if constexpr (I8) any = i8_value(gmax, blk) >= th;
    else any = __int_as_float(gmax) >= th;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(any);
    const int n = __builtin_popcountll(bal);
    const i32x16 v = __builtin_bit_cast(i32x16, acc);
    if constexpr (MODE == 0 || MODE == 3) {
        // (MODE 3: the block laid out as the FALL-THROUGH path -- the common case takes one short forward branch over it and a
        // hit never leaves the loop's code for a cold block at the kernel's end and back)
        if (__builtin_expect(bal != 0, MODE == 3 ? 1 : 0)) {  // wave-uniform, rare
            if (__builtin_expect(lq_n + n > kLaneQueueCap, 0)) {  // a burst the per-tile flush did not foresee: make room now
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lane_queue_flush_small<I8>(a, lq_addr, lq_n, row_end);
                lq_n = 0;
            }
            if (any) {
                const unsigned e = (unsigned)lq_n + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
                const unsigned addr = lq_addr + (e << 6);
#pragma unroll
                for (int i = 0; i < 4; ++i) lds_store16(addr + 16u * i, i32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]});
                lds_store16(lq_addr + (unsigned)(kLaneQueueCap * 64) + (e << 4), i32x4{q, rbase, (int)__float_as_uint(blk.m), (int)__float_as_uint(blk.ek)});
            }
            lq_n += n;
        }
        return;
    }
    // does the block's hit lanes fit?  (Never false in practice -- see kLaneQueueFlushAt.)  If not, nothing is stored and the
    // wave remembers it in `lq_ovf`: at the next tile start it flags its 32 queries kStOverflow (the host re-screens them).
    const bool fits = lq_n + n <= kLaneQueueCap;
    const unsigned long long mask = fits ? bal : 0ull;   // s_cselect: no branch
    lq_ovf |= fits ? 0 : 1;
    const unsigned e = (unsigned)lq_n + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
    if constexpr (MODE == 2) {
        const unsigned addr = lq_addr + (e << 6), addr_m = lq_addr + (unsigned)(kLaneQueueCap * 64) + (e << 4);
        const i32x4 meta{q, rbase, (int)__float_as_uint(blk.m), (int)__float_as_uint(blk.ek)};
        unsigned long long saved;
        asm volatile(
            "s_mov_b64 %[sv], exec\n\t"
            "s_and_b64 exec, exec, %[mask]\n\t"
            "ds_write_b128 %[ad], %[v0]\n\t"
            "ds_write_b128 %[ad], %[v1] offset:16\n\t"
            "ds_write_b128 %[ad], %[v2] offset:32\n\t"
            "ds_write_b128 %[ad], %[v3] offset:48\n\t"
            "ds_write_b128 %[am], %[mt]\n\t"
            "s_mov_b64 exec, %[sv]"
            : [sv] "=&s"(saved)
            : [mask] "s"(mask), [ad] "v"(addr), [am] "v"(addr_m), [v0] "v"(i32x4{v[0], v[1], v[2], v[3]}),
              [v1] "v"(i32x4{v[4], v[5], v[6], v[7]}), [v2] "v"(i32x4{v[8], v[9], v[10], v[11]}),
              [v3] "v"(i32x4{v[12], v[13], v[14], v[15]}), [mt] "v"(meta)
            : "memory");
    }
    lq_n += fits ? n : 0;
}

// FLAG = false: a full queue falls back to a direct global append (a returning atomic: k_screen256, first form);
// FLAG = true: it only flags the query in `status` (no returning atomic inside the K loop: k_screen256b) -- the host
// re-screens flagged queries with the tighter bound.
template <bool I8, bool FLAG>
__device__ __forceinline__ void screen_queue_block(const ScreenArgs& a, int* status, f32x16 acc, int q, int rbase,
                                                   int row_end, float th, I8Blk blk, int32_t* que, int& que_n) {
    // maxima of the four groups of four registers first (same 15 max operations as one flat reduction): the append
    // path below skips a whole group with one compare
    bool any, gany[4];
    int thi = 0;
    if constexpr (I8) {
        const i32x16 v = __builtin_bit_cast(i32x16, acc);
        int g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = max(max(v[4 * i], v[4 * i + 1]), max(v[4 * i + 2], v[4 * i + 3]));
        any = i8_value(max(max(g[0], g[1]), max(g[2], g[3])), blk) >= th;
        if (__builtin_amdgcn_ballot_w64(any) == 0) return;  // wave-uniform: almost always taken
        // hit path: the block's integer threshold (conservative image of the float test above), then integer compares
        thi = i8_block_threshold(th, blk.m, blk.ek);
#pragma unroll
        for (int i = 0; i < 4; ++i) gany[i] = any && g[i] >= thi;
    } else {
        float g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = fmaxf(fmaxf(acc[4 * i], acc[4 * i + 1]), fmaxf(acc[4 * i + 2], acc[4 * i + 3]));
        any = fmaxf(fmaxf(g[0], g[1]), fmaxf(g[2], g[3])) >= th;
        if (__builtin_amdgcn_ballot_w64(any) == 0) return;  // wave-uniform: almost always taken
#pragma unroll
        for (int i = 0; i < 4; ++i) gany[i] = g[i] >= th;
    }
    // A hit costs the whole workgroup this traversal (the other waves wait at the next K-step barrier), so it is kept
    // short: one compare + one scalar branch per group of four accumulator registers, then per register of a group with
    // a hit; the row bound is only tested in the hit branch (rows past row_end exist in the last tile of a chunk only,
    // and their values are finite garbage at worst).
    // The queue is written with inline-asm LDS stores on purpose: for compiler-visible LDS accesses the waitcnt
    // insertion assumes they may alias the in-flight LDS-DMA and puts s_waitcnt vmcnt(0) in front
    // (tests/test_build_pipeline.py checks the generated code).  No wait after the stores: LDS operations of one wave
    // execute in order, the flush's reads come later in the same wave.
    const unsigned a_q = lds_addr(que), a_r = a_q + 4u * kWaveQueueCap, a_v = a_q + 8u * kWaveQueueCap;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
        if (__builtin_amdgcn_ballot_w64(gany[gi]) == 0) continue;  // wave-uniform: no hit in this group of four
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
            const int r = 4 * gi + ri;
            bool hit;
            if constexpr (I8) hit = any && __builtin_bit_cast(i32x16, acc)[r] >= thi;
            else hit = acc[r] >= th;
            if (__builtin_amdgcn_ballot_w64(hit) == 0) continue;  // wave-uniform
            const int row = rbase + (r & 3) + 8 * (r >> 2);
            hit = hit && row < row_end;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
            if (hit) {
                const unsigned e = (unsigned)que_n +
                                   __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
                float val;
                if constexpr (I8) val = i8_value(__builtin_bit_cast(i32x16, acc)[r], blk);
                else val = acc[r];
                if (e < (unsigned)kWaveQueueCap) {
                    asm volatile("ds_write_b32 %0, %1" ::"v"(a_q + 4u * e), "v"(q) : "memory");
                    asm volatile("ds_write_b32 %0, %1" ::"v"(a_r + 4u * e), "v"(row) : "memory");
                    asm volatile("ds_write_b32 %0, %1" ::"v"(a_v + 4u * e), "v"(val) : "memory");
                } else if constexpr (FLAG) {  // queue full: the query is re-screened by the host
                    __hip_atomic_fetch_or(&status[q], kStOverflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {  // queue full (a burst of hits inside one tile): direct global append, slow but correct
                    const int slot = atomicAdd(&a.cnt[q], 1);
                    if (slot < a.cap) {
                        a.cand_row[(int64_t)q * a.cap + slot] = row;
                        a.cand_val[(int64_t)q * a.cap + slot] = val;
                    }
                }
            }
            que_n += __builtin_popcountll(bal);  // may run past the capacity: the flush clamps
        }
    }
}

// first chunk: every (query,row) becomes a candidate at slot row-row0 (counts are set by the host)
template <bool I8>
__device__ __forceinline__ void screen_emit_all_block(const ScreenArgs& a, f32x16 acc, int q, int64_t rbase, I8Blk blk) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = rbase + (r & 3) + 8 * (r >> 2);
        if (row < a.row_end) {
            float val;
            if constexpr (I8) {
                val = a.flag8[row] ? __builtin_nanf("") : i8_value(__builtin_bit_cast(i32x16, acc)[r], blk);
            } else {
                val = acc[r];
            }
            a.cand_row[(int64_t)q * a.cap + (row - a.row0)] = (int32_t)row;
            a.cand_val[(int64_t)q * a.cap + (row - a.row0)] = val;
        }
    }
}

// Starter (run_screen: "sampled threshold estimator"): the wave's 64 rows x 64 queries sub-tile -> for each of its queries
// the LARGEST value over the 64 rows and its row, stored at slot (slab index) of the query's list: no thresholds, no
// atomics, S / 64 candidates per query from a sample of S rows.  The exact re-score of the best of them (k_prune,
// thr_only) gives a first threshold that is valid whatever the sample missed: any k exact scores bound the k-th best
// from below.  int8: v = fma((float)acc, m, ek) is monotone in acc (m >= 0), so a block's largest value comes from its
// largest accumulator.  acc[i] = the two row blocks of query block j.
template <bool I8>
__device__ __forceinline__ void screen_emit_slab_max(const ScreenArgs& a, const f32x16 (&acc)[2], int q, int64_t slab_row0,
                                                     int lane, const I8Blk (&blk)[2]) {
    float best = -__builtin_inff();
    int64_t best_row = slab_row0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t rbase = slab_row0 + 32 * i + 4 * (lane >> 5);
        int br = 0;
        float bv;
        if constexpr (I8) {
            const i32x16 v = __builtin_bit_cast(i32x16, acc[i]);
            int m = v[0];
#pragma unroll
            for (int r = 1; r < 16; ++r)
                if (v[r] > m) {
                    m = v[r];
                    br = r;
                }
            bv = i8_value(m, blk[i]);
        } else {
            float m = acc[i][0];
            if (!(m == m)) m = -__builtin_inff();  // (NaN image of an irregular row: never a maximum)
#pragma unroll
            for (int r = 1; r < 16; ++r)
                if (acc[i][r] > m) {
                    m = acc[i][r];
                    br = r;
                }
            bv = m;
        }
        const int64_t row = rbase + (br & 3) + 8 * (br >> 2);
        if (row < a.row_end && bv > best) {
            best = bv;
            best_row = row;
        }
    }
    // the other half of the wave holds the other 32 rows of the same query column
    const float ov = __shfl_xor(best, 32, kWave);
    const int orow = __shfl_xor((int)best_row, 32, kWave);
    if (ov > best || (ov == best && orow < (int)best_row)) {
        best = ov;
        best_row = orow;
    }
    if (lane < 32) {
        const int64_t slot = (slab_row0 - a.row0) / kSlabRows;
        a.cand_row[(int64_t)q * a.cap + slot] = (int32_t)best_row;
        a.cand_val[(int64_t)q * a.cap + slot] = best;
    }
}

template <bool I8>
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
    const int64_t row_bytes = a.row_bytes;

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

    const int T = a.ksteps;
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
            // NOTE: the fragments are typed bf16x8 on purpose (also for int8 data).  With this type the compiler's
            // waitcnt insertion keeps the LDS-DMA of the NEXT step in flight across these ds_reads; with a plain
            // uint4 load it conservatively adds s_waitcnt vmcnt(0) here and the double buffering is lost
            // (tests/test_build_pipeline.py checks the generated code).
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                fa[blk] = __builtin_bit_cast(bf16x8, *(const uint4*)(bufA + (offA[blk] ^ kx)));
                fb[blk] = __builtin_bit_cast(bf16x8, *(const uint4*)(bufB + (offB[blk] ^ kx)));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = screen_mfma<I8>(fa[i], fb[j], acc[i][j]);
        }
    }

    // ---- fused epilogue: threshold test, rare append
    if (a.emit_all == kEmitSlabMax) {  // wave-uniform: the starter's form
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = q0 + 64 * wc + 32 * j + (lane & 31);
            const float sq = I8 ? a.sc[q] : 1.0f, kq = I8 ? a.kq[q] : 1.0f;
            const int64_t slab_row0 = tile_row0 + 64 * wr;
            I8Blk blk[2] = {{1.0f, 0.0f}, {1.0f, 0.0f}};
            if constexpr (I8) {
                blk[0] = i8_blk(i8_group_of(a.grp, slab_row0), sq, kq);
                blk[1] = i8_blk(i8_group_of(a.grp, slab_row0 + 32), sq, kq);
            }
            const f32x16 col[2] = {acc[0][j], acc[1][j]};
            screen_emit_slab_max<I8>(a, col, q, slab_row0, lane, blk);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = q0 + 64 * wc + 32 * j + (lane & 31);
        const float th = a.thr[q];
        const float sq = I8 ? a.sc[q] : 1.0f, kq = I8 ? a.kq[q] : 1.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t row0 = tile_row0 + 64 * wr + 32 * i;  // wave-uniform: one row group per block
            const int64_t rbase = row0 + 4 * (lane >> 5);
            I8Blk blk{1.0f, 0.0f};
            if constexpr (I8) blk = i8_blk(i8_group_of(a.grp, row0), sq, kq);
            if (a.emit_all == kEmitAll) screen_emit_all_block<I8>(a, acc[i][j], q, rbase, blk);  // wave-uniform branch
            else screen_emit_block<I8>(a, acc[i][j], q, rbase, th, blk);
        }
    }
}

}  // namespace mi355
