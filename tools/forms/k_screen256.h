// k_screen256.h -- the large-block form of the screen: 256 corpus rows x 256 queries per tile, 8 waves,
// LDS ring of 8 half-tiles (128 KiB), ping-pong between two wave groups, PERSISTENT workgroups.
//
// Same contract as k_screen (k_screen.h): t = <q_hat, c_hat> via v_mfma_f32_32x32x16_bf16 (bf16 shadow) or
// v_mfma_i32_32x32x32_i8 (int8 shadow), fused threshold epilogue.  What changes is the pipeline:
//
//  * Tile 256x256, K step = 128 B of every row.  8 waves = 2 groups of 4; wave (wr = group, wc = wave&3) owns
//    rows [128wr,+128) x queries [64wc,+64) = 4x2 MFMA blocks = 128 accumulator VGPRs.
//  * The K-tile is cut into 4 half-tiles of 128 rows x 128 B (16 KiB): A0/A1 = the first/second 64 rows of
//    every wave-row, B0/B1 = the first/second 32 queries of every wave-column.  A wave's work on a K-tile is
//    4 phases = the 4 (row-half i, query-half j) quadrants in the order (0,0) (0,1) (1,1) (1,0), so the
//    half-tiles are first needed in the order A0,B0 | B1 | A1 -- which is also the order they are staged in.
//  * Each phase = LOAD (issue the DMA of ONE half-tile of the next K-tile, ds_read this quadrant's new
//    operands, counted s_waitcnt vmcnt) | s_barrier | MFMA (8 MFMAs, s_setprio 1) | s_barrier.  Group 1 runs
//    one barrier behind group 0, so on every SIMD one wave is in its MFMA half while its partner is in
//    its LOAD half: the matrix pipe sees back-to-back clusters and the LDS/DMA traffic hides under them.
//  * DMA runs 4 half-tiles ahead of use (4 wave-instructions outstanding at every wait: vmcnt(4), never 0 in
//    steady state).  Hazards: a half-tile is read only >= 2 barriers after every wave retired its part of it,
//    and a ring slot is re-staged >= 2 barriers after its last reader's lgkmcnt(0).
//  * Persistent: the grid is 8 XCDs x L workgroups (one per CU, the ring fills the LDS); a workgroup keeps its
//    query tile and walks corpus tiles ctl, ctl + 8*L/n_qtiles, ...  The LAST K-step of a tile already stages
//    the FIRST K-step of the next tile, so the DMA pipeline never drains between tiles (with 6 K-steps per tile
//    at d=768 int8, the per-tile fill/drain was ~1/4 of the time).  Workgroups that run together on an XCD
//    share corpus tiles (the n_qtiles query tiles of one corpus tile are neighbours), as in k_screen.
//  * The epilogue must not touch the vector-memory counter while that DMA is in flight (vmcnt completes in
//    order: waiting for an atomic's return would wait for the whole prefetch).  Thresholds are loaded once per
//    workgroup, and hits go to a per-wave LDS queue (k_screen.h: screen_queue_block) that is flushed to the
//    global candidate lists when it fills up and when the workgroup is done.
//  * The emit-all first chunk is not handled here (the host runs it through k_screen).
#pragma once
#include "k_screen256_common.h"

namespace mi355 {

// ---- developer timeline trace (tools/screen_bench, ABL bit 4): waves 0 and 4 of workgroup 0 stamp s_memtime at four
// points of every phase of K-steps [kTraceG0, kTraceG0 + kTraceSteps) into the 2 KiB of LDS behind the queues.
constexpr int kTraceG0 = 24, kTraceSteps = 6, kTraceStamps = kTraceSteps * 16;
constexpr int kTraceOff = kRingBytes;  // (developer trace: over the first wave's queue -- traced launches park the thresholds)
__device__ unsigned long long* g_trace_out;  // [2][kTraceStamps], set with hipMemcpyToSymbol
// stamp with the scalar-memory wait (only where no LDS read is outstanding or a full lgkmcnt(0) is due anyway)
#define MI355_TR_STAMP(ON, ADDR)                                                                       \
    do {                                                                                               \
        if (ON) {                                                                                      \
            unsigned long long t__;                                                                    \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                \
            asm volatile("ds_write_b64 %0, %1" ::"v"(ADDR), "v"(t__) : "memory");                      \
            ADDR += 8;                                                                                 \
        }                                                                                              \
    } while (0)
// stamp whose value is picked up later (in front of a barrier, with ds_reads in flight): issue only
#define MI355_TR_ISSUE(ON, T) do { if (ON) asm volatile("s_memtime %0" : "=s"(T)::"memory"); } while (0)
#define MI355_TR_STORE(ON, ADDR, T)                                                                    \
    do {                                                                                               \
        if (ON) {                                                                                      \
            asm volatile("ds_write_b64 %0, %1" ::"v"(ADDR), "v"(T) : "memory");                        \
            ADDR += 8;                                                                                 \
        }                                                                                              \
    } while (0)

// ABL: developer ablation switches for tools/screen_bench (0 in the library): bit0 = skip the ds_reads after
// the first K-tile, bit1 = skip the DMA after the prologue, bit2 = no s_setprio.
// grid: 8 * L workgroups of 512 threads, L a multiple of n_qtiles (host: screen256_grid()).
template <int ABL, bool I8>
__global__ __launch_bounds__(512, 2) void k_screen256(ScreenArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2;  // 0 leads, 1 runs one barrier behind
    const int wr = group, wc = wave & 3;
    int32_t* const que = (int32_t*)(smem + kRingBytes + wave * (kWaveQueueCap * 12));  // [q | row | value bits]
    int que_n = 0;                                                                       // wave-uniform

    // persistent, XCD-aware walk: workgroups b, b+8, ... share an XCD; there, l = b>>3 splits into
    // (corpus slot, query tile) so that the query tiles of one corpus tile run side by side
    const int b = blockIdx.x;
    const int xcd = b & 7;
    const int l = b >> 3;
    const int cslot = l / a.n_qtiles;
    const int qt = l - cslot * a.n_qtiles;
    const int cstep = ((int)(gridDim.x >> 3) / a.n_qtiles) * 8;  // corpus tiles between two visits
    int ctl = cslot * 8 + xcd;
    if (ctl >= a.n_ctiles) return;
    const int q0 = qt * kT2;
    const int64_t row_bytes = a.row_bytes;

    // ---- DMA sources: this wave stages local rows [16*wave + 8u, +8) of every half-tile, u = 0,1.
    // half-tile types: 0 = A0, 1 = B0, 2 = B1, 3 = A1.  Address = wave-uniform 64-bit base (SGPRs: tile / query
    // tile / half) + a 32-bit per-lane offset (one VGPR per u and operand), so the 8 sources cost 4 VGPRs.
    unsigned voffA[2], voffB[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int r = (2 * wave + u) * 8 + (lane >> 3);      // local row 0..127
        const int c = (lane & 7) ^ ((r >> 1) & 7);           // source chunk for this LDS slot (swizzle)
        const int arow0 = 128 * (r >> 6) + (r & 63);         // + 64*i
        const int bcol0 = 64 * (r >> 5) + (r & 31);          // + 32*j
        voffA[u] = (unsigned)(arow0 * (int)row_bytes + c * 16);
        voffB[u] = (unsigned)(bcol0 * (int)row_bytes + c * 16);
    }
    const char* baseA = (const char*)a.shadow + (int64_t)(a.ct0 + ctl) * kT2 * row_bytes;  // tile's first row
    const char* const baseB = (const char*)a.qhat + (int64_t)q0 * row_bytes;
    const int64_t tile_stride_bytes = (int64_t)cstep * kT2 * row_bytes;
    const int64_t half_A = 64 * row_bytes, half_B = 32 * row_bytes;
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
    // ---- per-workgroup constants of the epilogue (frozen for the launch): no vector-memory loads later
    float th[2], scq[2], kqq[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = q0 + 64 * wc + 32 * j + (lane & 31);
        th[j] = a.thr[q];
        scq[j] = I8 ? a.sc[q] : 1.0f;
        kqq[j] = I8 ? a.kq[q] : 1.0f;
    }

    f32x16 acc[2][2][2];  // [row half i][row block rb][query half j]
    bf16x8 fa[2][4], fb[4];  // typed bf16x8 also for int8 data: see the NOTE in k_screen.h (waitcnt insertion)
    const int T = a.ksteps;

// stage half-tile type S, K offset KO (bytes), into ring parity PAR
#define MI355_STAGE(S, PAR, SA, KO)                                                                   \
    if (!(ABL & 2)) do {                                                                              \
        char* dst__ = smem + (4 * (PAR) + (S)) * kHalfBytes + (2 * wave) * 1024;                      \
        const char* sb__ = ((S) == 0 || (S) == 3) ? (SA) + ((S) == 3 ? half_A : 0) + (KO)             \
                                                  : baseB + ((S) == 2 ? half_B : 0) + (KO);           \
        const unsigned* vo__ = ((S) == 0 || (S) == 3) ? voffA : voffB;                                \
        glds16(sb__ + vo__[0], dst__);                                                                \
        glds16(sb__ + vo__[1], dst__ + 1024);                                                         \
    } while (0)
#define MI355_LOAD_A(I, PAR)                                                                          \
    if (!(ABL & 1) || first_k) do {                                                                   \
        const char* s__ = smem + (4 * (PAR) + ((I) ? 3 : 0)) * kHalfBytes;                            \
        _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                              \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                          \
                fa[rb][kk] = __builtin_bit_cast(bf16x8, *(const uint4*)(s__ + (offA[rb] ^ (kk * 32)))); \
    } while (0)
#define MI355_LOAD_B(J, PAR)                                                                          \
    if (!(ABL & 1) || first_k) do {                                                                   \
        const char* s__ = smem + (4 * (PAR) + 1 + (J)) * kHalfBytes;                                  \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                              \
            fb[kk] = __builtin_bit_cast(bf16x8, *(const uint4*)(s__ + (offB ^ (kk * 32))));           \
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
// one K-step that also stages the next K-step (of this tile or of the next one): PAR = ring parity being read
#define MI355_PHASE(SS, LOADS, I, J, PAR, SA, KO)                                                      \
    do {                                                                                              \
        MI355_TR_STAMP(tr_on, tr_addr);                                                               \
        MI355_STAGE(SS, (PAR) ^ 1, SA, KO);                                                            \
        LOADS;                                                                                        \
        MI355_WAIT_VM(4);                                                                             \
        MI355_TR_ISSUE(tr_on, tr_l);                                                                  \
        MI355_BARRIER();                                                                              \
        MI355_TR_STAMP(tr_on, tr_addr);                                                               \
        MI355_TR_STORE(tr_on, tr_addr, tr_l);                                                         \
        MI355_MFMA(I, J);                                                                             \
        MI355_TR_STAMP(tr_on, tr_addr);                                                               \
        MI355_BARRIER();                                                                              \
    } while (0)
#define MI355_KSTEP_STAGING(PAR, SA, KO)                                                                \
    do {                                                                                              \
        MI355_PHASE(0, MI355_LOAD_A(0, PAR); MI355_LOAD_B(0, PAR), 0, 0, PAR, SA, KO);                 \
        MI355_PHASE(1, MI355_LOAD_B(1, PAR), 0, 1, PAR, SA, KO);                                       \
        MI355_PHASE(2, MI355_LOAD_A(1, PAR), 1, 1, PAR, SA, KO);                                       \
        MI355_PHASE(3, MI355_LOAD_B(0, PAR), 1, 0, PAR, SA, KO);                                       \
    } while (0)

    // ---- prologue: stage K-step 0 of the first tile completely
    MI355_STAGE(0, 0, baseA, 0);
    MI355_STAGE(1, 0, baseA, 0);
    MI355_STAGE(2, 0, baseA, 0);
    MI355_STAGE(3, 0, baseA, 0);
    MI355_WAIT_VM(0);
    __syncthreads();
    if (group == 1) MI355_BARRIER();  // stagger: group 1's LOAD halves line up with group 0's MFMA halves

    int par = 0;  // ring parity of the K-step being consumed
    bool first_k = true;
    // developer trace (ABL bit 4): see MI355_TR_* above
    int gk = 0;
    bool tr_on = false;
    unsigned tr_addr = lds_addr(smem + kTraceOff + group * (kTraceStamps * 8));
    unsigned long long tr_l = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][rb][j][r] = 0.0f;
        // Every K-step stages the one after it: the next K-step of this tile, or K-step 0 of the next tile (A base
        // moves on, B stays).  One code path for all K-steps -- no branch ever merges two versions of the accumulators.
        // The workgroup's very last K-step stages a dummy (this tile's K-step 0 again), drained before the exit.
        const bool has_next = ctl + cstep < a.n_ctiles;
        const char* const baseA_next = has_next ? baseA + tile_stride_bytes : baseA;
        for (int t = 0; t < T; ++t) {
            const bool last = t + 1 == T;
            const char* const sA = last ? baseA_next : baseA;
            const int64_t ko = last ? 0 : (int64_t)(t + 1) * kRowB;
            if (ABL & 16) tr_on = blockIdx.x == 0 && (wave & 3) == 0 && gk >= kTraceG0 && gk < kTraceG0 + kTraceSteps;
            MI355_KSTEP_STAGING(par, sA, ko);
            par ^= 1;
            first_k = false;
            ++gk;
        }
        baseA = baseA_next;

        // ---- fused epilogue of this tile: column (query) = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5).
        // The lane id is made opaque per tile so that nothing of the (cold) hit path is hoisted out of the
        // persistent loop into registers that the K loop needs.
        {
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int tile_row0 = (a.ct0 + ctl) * kT2;  // rows < 2^31 (checked by the host)
            const int row_end = (int)a.row_end;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = q0 + 64 * wc + 32 * j + (lane_e & 31);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) {
                        const int row0 = tile_row0 + 128 * wr + 64 * i + 32 * rb;  // wave-uniform: one row group
                        const int rbase = row0 + 4 * (lane_e >> 5);
                        I8Blk blk{1.0f, 0.0f};
                        if constexpr (I8) blk = i8_blk(i8_group_of(a.grp, row0), scq[j], kqq[j]);
                        screen_queue_block<I8, false>(a, nullptr, acc[i][rb][j], q, rbase, row_end, th[j], blk, que, que_n);
                    }
            }
        }
        if (!has_next) break;
        ctl += cstep;
        if (que_n > kWaveQueueCap / 2) {  // wave-uniform, rare: make room (this wave stalls on vector memory once)
            wave_queue_flush(a, que, min(que_n, kWaveQueueCap));
            que_n = 0;
        }
    }
    if (group == 0) MI355_BARRIER();  // balance the stagger barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy prefetch of the last K-step must land before the LDS is freed

#undef MI355_STAGE
#undef MI355_LOAD_A
#undef MI355_LOAD_B
#undef MI355_MFMA
#undef MI355_WAIT_VM
#undef MI355_KSTEP_STAGING
#undef MI355_PHASE
    if ((ABL & 16) && blockIdx.x == 0 && (wave & 3) == 0) {  // dump the trace
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long* src = (const unsigned long long*)(smem + kTraceOff + group * (kTraceStamps * 8));
        for (int i = lane; i < kTraceStamps; i += 64) g_trace_out[group * kTraceStamps + i] = src[i];
    }

    // ---- flush this wave's candidate queue: one global atomic per entry
    wave_queue_flush(a, que, min(que_n, kWaveQueueCap));
}

}  // namespace mi355
