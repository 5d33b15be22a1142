// mi355dr_maxsim.hip -- multi-vector (late interaction) store and exact MaxSim top-k.
//
// Replaces VectorChord's `embeddings @# ARRAY[q_1..q_n]` + ORDER BY distance LIMIT k
// (reference autorag_research/orm/repository/base.py:487-535, :537-571):
//     distance(doc) = sum_i min_j ( -<q_i, d_j> )      fp32; score = -distance / n_q on the host
// with every dot product the k-ascending fp32 fmaf chain (oracle.c orc_maxsim_distance), the sum over
// query vectors in query order.  Bit-exact by construction: v_mfma_f32_32x32x2_f32 IS a k-ordered
// fmaf chain per output element (MI355X guide: "bit-for-bit a k-ordered f32 fmaf chain"), max is
// exact, and the final sum is done sequentially in j.
//
// HBM layout: doc token rows are stored padded so that every doc owns whole 32-row blocks; the tail of
// the last block repeats the doc's last token (max over a multiset with repeats is unchanged), and the
// vector dimension is zero-padded to a multiple of 8 (fma(0,0,acc) == acc).  One wave owns one doc:
// each 32-row block x 32 query tokens is one chain of d/2 MFMAs; operands go global -> VGPR as float4
// (A, doc tokens) and LDS -> VGPR (B, query tokens), and two v_permlane32_swap per 4 MFMAs put k in
// ascending order.  fp32 MFMA runs at the vector rate (157 TF peak): at one 32-token query per pass the
// kernel is at the HBM/MFMA balance point (16 flop/B), with more queries per pass it is MFMA-bound.
//
// Search = bf16 MFMA screen over every doc (k_maxsim16, HBM-bound on a bf16 copy of the tokens laid out in MFMA
// fragment order) -> candidates under a rigorous bound -> exact kernel (k_maxsim, doc list) on the candidates ->
// exact top-k.  Same results as running the exact kernel over every doc (option "maxsim_screen" = 0), which stays
// the path for stores / queries with non-finite values and for candidate lists that overflow.
//   bound (a priori): every token pair |t_ij - s_ij| <= eps |q_i||d_j|, eps = 2^-7 + 2^-15 + 3 d 2^-24 (bf16 rounding of both
//   sides + fp32 accumulation + the exact chain's own rounding), so |max_j t_ij - max_j s_ij| <= eps |q_i| Dmax
//   (Dmax = largest token norm in the store) and |T - S| <= E = (eps + 2 n_q 2^-24) Dmax sum_i |q_i| for the
//   per-doc sums.  k docs have T >= x_k (k-th best screen score) hence S >= x_k - E; any doc of the exact top-k
//   (ties included) has S >= that, hence T >= x_k - 2E: the candidate set.
#include <chrono>

#include "maxsim_common.h"

using namespace mi355;

namespace mi355 {

struct MultiVecStore {
    int64_t n_docs = 0;
    int64_t n_blocks = 0, cap_blocks = 0;  // 32-row blocks stored / allocated
    int64_t cap_docs = 0;
    int dpad = 0;                  // dim rounded up to 8
    float* tok = nullptr;          // [cap_blocks*32, dpad]
    // bf16 copy for the screen, MFMA fragment order: [block][kk][lane][8] with lane = (row = lane&31, half = lane>>5)
    // holding dims kk*16 + half*8 + 0..7 of token `row` (original column order)
    int nkk = 0;                   // dim rounded up to 16, / 16
    uint4* tok16 = nullptr;        // [cap_blocks * nkk * 64]
    double tok_norm_max = 0.0;     // largest token norm (double, from the fp32 values)
    double tok16_norm_max = 0.0;   // largest norm of a bf16-rounded token
    double tok_res_max = 0.0;      // largest residual norm |d - bf16(d)| of a token
    bool finite = true;            // every stored value is finite (else: no screen)
    uint4* qfrag = nullptr;        // [kMsPassBlocks * nkk * 64] query fragments of one screen launch (up to four groups of <= 4 queries)
    float* dist16 = nullptr;       // [kMsPassQueries, cap_docs] screen distances (rows 4 g ..: the groups screened ahead)
    // candidate lists of a pass, per query (row stride kMsCandCap): section 0 = the WIDE list (screen distance within 2E of the
    // k-th best), 1 = the STARTER (the screen's own top-k), 2 = the FINAL list (within E of the starter's k-th best exact distance)
    int32_t* cand_list = nullptr;  // [3][kMsPassQueries][kMsCandCap]
    float* cand_dist = nullptr;    // [kMsPassQueries][kMsCandCap] exact distances of the list being re-scored (starter, then final)
    float* cand_sd = nullptr;      // [kMsPassQueries][kMsCandCap] screen distances of the wide list's entries
    int* cand_ctl = nullptr;       // [3][kMsPassQueries][2]: count, overflow flag
    int* cand_ctl_host = nullptr;  // pinned [2 * kMsPassQueries]: the final list's
    uint32_t* sel[2] = {nullptr, nullptr};  // fast path: per-segment k best screen keys, [4, ceil(cap_docs/1024) * 64]
    float* two_e_dev = nullptr;    // [4]
    char* stage_host = nullptr;    // pinned: query image | query fragments | 2E (H2D), results (D2H)
    size_t stage_bytes = 0;
    int64_t* blk_off = nullptr;    // [cap_docs+1] first block of each doc (device)
    std::vector<int64_t> blk_off_host;
    std::vector<int32_t> tok_cnt_host;  // [n_docs] token vectors of each doc (the padded copies do not keep it)
    // the granule-packed bf16 copy of k_maxsim_wg8.h: a second shadow, built on first use (ms_pack8_ensure), stale after an add
    uint4* tok16p = nullptr;       // [pack_cap_blocks * nkk * 64]
    int64_t* goff = nullptr;       // [pack_cap_docs + 1] first 8-token granule of each doc (device)
    int64_t pack_docs = -1;        // n_docs the copy was built (or judged) for; -1: never
    int64_t pack_gran = 0, pack_blocks = 0, pack_cap_blocks = 0, pack_cap_docs = 0;
    bool pack_use = false;         // the copy exists for pack_docs docs and pays (or is forced)
    int pack_mode = 0;             // option maxsim_pack8 at the time of that decision
    // search scratch
    float* qtok = nullptr;         // [kMsCols, dpad] per launch
    float* dist = nullptr;         // [max queries per launch (4), cap_docs]
    uint64_t* pk[2] = {nullptr, nullptr};
    int32_t* pr[2] = {nullptr, nullptr};
    int64_t part_cap = 0;
    float* out_d = nullptr;        // [kKMax]
    int64_t* out_r = nullptr;
    int64_t dist_cap_docs = 0;
};

bool multivec_view(const mi355dr_index* idx, MultiVecView* out) {
    const MultiVecStore* m = idx->mv;
    if (!m || m->n_docs == 0) return false;
    out->tok = m->tok;
    out->blk_off = m->blk_off;
    out->blk_off_host = m->blk_off_host.data();
    out->dpad = m->dpad;
    out->n_docs = m->n_docs;
    return true;
}

void multivec_destroy(mi355dr_index* idx) {
    MultiVecStore* m = idx->mv;
    if (!m) return;
    void* ptrs[] = {m->tok, m->blk_off, m->qtok, m->dist, m->pk[0], m->pk[1], m->pr[0], m->pr[1], m->out_d, m->out_r,
                    m->tok16, m->qfrag, m->dist16, m->cand_list, m->cand_dist, m->cand_ctl, m->sel[0], m->sel[1],
                    m->two_e_dev, m->cand_sd, m->tok16p, m->goff};
    if (m->cand_ctl_host) (void)hipHostFree(m->cand_ctl_host);
    if (m->stage_host) (void)hipHostFree(m->stage_host);
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete m;
    idx->mv = nullptr;
}

struct MsArgs {
    const float* tok;
    const int64_t* blk_off;
    const float* qtok;      // [kMsCols, dpad] zero-padded
    float* dist;            // [nq_launch, n_items]
    const int32_t* doc_list;  // optional [n_items]: the docs to score (nullptr: item i = doc i); < 0 or >= n_docs: NaN
    int64_t n_items;        // work items (= n_docs without a list)
    const int* n_items_dev; // optional: the real number of items (<= n_items, which then only sizes the grid / dist rows)
    int64_t list_stride;    // > 0: grid.y = query of the launch, each with its own doc_list / dist row (stride) and
                            //      n_items_dev pair (stride 2); the workgroup scores that query only
    int64_t n_docs;
    int dpad;
    int nq_launch;          // queries in this launch (<= 4 scored together; list mode with list_stride: <= kMsPassQueries, one per grid.y)
    int q_col0[kMsPassQueries];  // first column of each query in the staged image (any column)
    int q_len[kMsPassQueries];   // real token count of each query
    int clamp0;             // 1: every query token contributes max(0, max_j <q_i, d_j>)  (ColBERT reranker, rerankers/colbert.py:79)
    // a query with more vectors than one launch stages (ms_cols_for(dpad) <= 128 columns) is scored in TILES of its vectors:
    // the launch of tile t starts every item's sum from the value tile t-1 left (same layout as `dist`; may be `dist` itself),
    // so the fp32 sum still runs over the query's vectors in order -- bit for bit the one-launch chain
    const float* dist_in;
    // list mode, long documents (ColPali pages: 33 blocks): the four waves of a workgroup share ONE item -- wave w multiplies
    // blocks w, w + 4, ... -- and their per-column maxima meet in LDS (byte offset red_off of the dynamic segment, 2 KiB) before
    // wave 0 adds them up in token order.  A maximum does not depend on the order: the same bits as one wave per item.  With a
    // whole page per wave a 570-candidate list kept 570 of the chip's 1024 SIMDs busy for 33 blocks each, two deep where two
    // workgroups shared a CU: 0.5 ms per launch against 0.13 ms of fp32 MFMA work.
    int coop;
    int red_off;
};

// query-token columns one launch of k_maxsim stages: whole 32-column blocks, at most kMsCols, inside the 160 KiB of LDS
// (d = 128: 128 columns; d = 768, the hidden size the ColBERT reranker scores with: 32)
inline int ms_cols_for(int dpad) {
    const int c = (int)(((size_t)160 * 1024 - kMsRedBytes) / ((size_t)(dpad + 4) * sizeof(float))) / 32 * 32;
    return c < kMsCols ? c : kMsCols;
}

// Column order inside every group of 8 dims, for BOTH stored token rows and the staged query rows:
// position j holds original column kPerm[j] = {0,4,2,6,1,5,3,7}[j].  A lane of the lower half (k-slot 0 of the
// 32x32x2 MFMA) reads positions 0..3 = columns (0,4,2,6), a lane of the upper half positions 4..7 = (1,5,3,7),
// so issuing the MFMAs on components x, z, y, w walks k = (0|1), (2|3), (4|5), (6|7): ascending, no lane swaps.
__host__ __device__ inline int ms_perm(int j) {
    constexpr int P[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    return (j & ~7) | P[j & 7];
}

int multivec_col_perm(int j) { return ms_perm(j); }

__device__ __forceinline__ void ms_load_piece(float4 (&a)[16], const float* row, int chunk, int dpad) {
    // this lane's 4 floats of every 8-dim group of dims [128*chunk, +128); groups past dpad read as zero
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k0 = chunk * 128 + i * 8;
        a[i] = k0 < dpad ? load_gmem_f4(row + k0) : make_float4(0.f, 0.f, 0.f, 0.f);  // (global, not FLAT: dev_common.h)
    }
}

__device__ __forceinline__ void ms_compute_piece(f32x16 (&acc)[4], const float4 (&a)[16], const float* qs, int ld, int ncb,
                                                 int col, int half, int chunk, int dpad) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        if (cb >= ncb) break;
        const float* brow = qs + (cb * 32 + col) * ld + 4 * half + chunk * 128;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (chunk * 128 + i * 8 >= dpad) break;
            const float4 bv = *(const float4*)(brow + i * 8);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, bv.x, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, bv.z, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, bv.y, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, bv.w, acc[cb], 0, 0, 0);
        }
    }
}

// B fragments: LDS image [col][dpad + 4] floats (the +4 pad makes the 16-lane groups of ds_read_b128 hit 16
// distinct 16-B slots).  The doc-token piece (32 rows x 128 dims = 16 float4 per lane) is register-resident
// and reused for every query column block; the next piece is prefetched while the current one is consumed.
__global__ __launch_bounds__(kMsThreads, 2) void k_maxsim(const MsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qs = (float*)smem;
    const int ld = a.dpad + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The argument block is read-only and its two small arrays are only ever indexed at compile time (unrolled selects):
    // a kernel that writes its by-value arguments, or indexes them with a run-time value, gets the whole block copied to the
    // stack first (65 scratch instructions in the round-4 build; pinned by tests/test_build_pipeline.py).
    const int32_t* doc_list = a.doc_list;
    float* dist = a.dist;
    const float* dist_in = a.dist_in;
    const int* n_items_dev = a.n_items_dev;
    const float* qtok = a.qtok;
    int nql = a.nq_launch;           // queries scored together by this workgroup: <= 4 (list mode: the one of blockIdx.y)
    int qc0[4], qln[4];              // their first column / token count (registers: every index below is a constant)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        qc0[i] = a.q_col0[i];
        qln[i] = a.q_len[i];
    }
    if (a.list_stride > 0) {  // per-query candidate lists
        const int y = blockIdx.y;
        doc_list += (int64_t)y * a.list_stride;
        dist += (int64_t)y * a.list_stride;
        if (dist_in) dist_in += (int64_t)y * a.list_stride;
        n_items_dev += 2 * y;
        int c0 = a.q_col0[0], ln = a.q_len[0];
#pragma unroll
        for (int i = 1; i < kMsPassQueries; ++i)
            if (i == y) {
                c0 = a.q_col0[i];
                ln = a.q_len[i];
            }
        qtok += (int64_t)c0 * a.dpad;  // this query's columns become columns 0.. of the staged image (any c0)
        qc0[0] = 0;
        qln[0] = ln;
        nql = 1;
    }
    const int64_t n_items = n_items_dev ? min((int64_t)*n_items_dev, a.n_items) : a.n_items;
    const bool coop = a.coop != 0;
    if ((int64_t)blockIdx.x * (coop ? 1 : 4) >= n_items) return;  // nothing for this workgroup: skip staging the query block
    float* red = (float*)(smem + a.red_off);
    int ncb = 0;   // column blocks in use
#pragma unroll
    for (int qi = 0; qi < 4; ++qi)
        if (qi < nql) ncb = max(ncb, (qc0[qi] + qln[qi] + 31) / 32);
    for (int i = tid; i < ncb * 32 * (a.dpad / 4); i += kMsThreads) {
        const int c = i / (a.dpad / 4), k4 = i - c * (a.dpad / 4);
        *(float4*)(qs + c * ld + k4 * 4) = *(const float4*)(qtok + (int64_t)c * a.dpad + k4 * 4);
    }
    __syncthreads();
    const int half = lane >> 5, col = lane & 31;
    const int nchunk = (a.dpad + 127) / 128;
    // docs are dealt round-robin to the waves of the grid so long and short docs mix
    // (with a doc list the grid is small and fixed -- the real list length is only known on the device -- and the
    // waves stride over the list until it ends)
    for (int dw = 0; doc_list || dw < kMsDocsPerWave; ++dw) {
    const int64_t item = coop ? (int64_t)dw * gridDim.x + blockIdx.x : ((int64_t)dw * gridDim.x + blockIdx.x) * 4 + wave;
    if (item >= n_items) break;  // (cooperative: the same item in all four waves -- every branch on it is workgroup-uniform)
    const int64_t doc = doc_list ? (int64_t)doc_list[item] : item;
    if (doc < 0 || doc >= a.n_docs) {  // (subset scoring) not a stored doc
        if (lane == 0 && (!coop || wave == 0))
            for (int qi = 0; qi < nql; ++qi) dist[(int64_t)qi * a.n_items + item] = __uint_as_float(0x7FC00000u);
        continue;
    }
    const int64_t b0 = a.blk_off[doc], b1 = a.blk_off[doc + 1];
    const int bfirst = coop ? wave : 0, bstep = coop ? 4 : 1;  // this wave's blocks of the document: b0 + bfirst + bstep * i
    const int64_t nb_w = b1 - b0 > bfirst ? (b1 - b0 - bfirst + bstep - 1) / bstep : 0;
    float run[4];  // running max per column block (this lane's column)
#pragma unroll
    for (int c = 0; c < 4; ++c) run[c] = -__builtin_inff();

    const int64_t npieces = nb_w * nchunk;
    f32x16 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
    float4 pa[16], pb[16];
    auto row_of = [&](int64_t p) { return a.tok + ((b0 + bfirst + bstep * (p / nchunk)) * kMsBlkRows + col) * (int64_t)a.dpad + 4 * half; };
    auto finish_block = [&]() {  // block max per column: 16 rows in this lane, the other 16 in lane^32
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            if (cb >= ncb) break;
            float m = acc[cb][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[cb][r]);
            m = fmaxf(m, __shfl_xor(m, 32, kWave));
            run[cb] = fmaxf(run[cb], m);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
        }
    };
    if (npieces > 0) ms_load_piece(pa, row_of(0), 0, a.dpad);
    for (int64_t p = 0; p < npieces; p += 2) {
        if (p + 1 < npieces) ms_load_piece(pb, row_of(p + 1), (int)((p + 1) % nchunk), a.dpad);
        ms_compute_piece(acc, pa, qs, ld, ncb, col, half, (int)(p % nchunk), a.dpad);
        if ((p + 1) % nchunk == 0) finish_block();
        if (p + 1 < npieces) {
            if (p + 2 < npieces) ms_load_piece(pa, row_of(p + 2), (int)((p + 2) % nchunk), a.dpad);
            ms_compute_piece(acc, pb, qs, ld, ncb, col, half, (int)((p + 1) % nchunk), a.dpad);
            if ((p + 2) % nchunk == 0) finish_block();
        }
    }
    if (coop) {  // the four waves' column maxima -> wave 0
        if (lane < 32) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
                if (cb < ncb) red[(wave * 4 + cb) * 32 + lane] = run[cb];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                if (cb >= ncb) break;
                float m = red[cb * 32 + col];
#pragma unroll
                for (int w = 1; w < 4; ++w) m = fmaxf(m, red[(w * 4 + cb) * 32 + col]);
                run[cb] = m;
            }
        }
        __syncthreads();  // (red is rewritten by the next item)
        if (wave != 0) continue;
    }
    // per query: distance = sum over its tokens (in order) of -(max dot); empty docs are skipped by the select
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
        if (qi >= nql) break;
        float accd = dist_in ? dist_in[(int64_t)qi * a.n_items + item] : 0.0f;  // (wave-uniform address)
        for (int j = 0; j < qln[qi]; ++j) {
            const int c = qc0[qi] + j;
            float v = 0.0f;
#pragma unroll
            for (int cbi = 0; cbi < 4; ++cbi)
                if (cbi == (c >> 5)) v = run[cbi];
            v = __shfl(v, c & 31, kWave);
            if (a.clamp0) v = fmaxf(v, 0.0f);
            accd = accd + (-v);
        }
        if (lane == 0) dist[(int64_t)qi * a.n_items + item] = b1 > b0 ? accd : __uint_as_float(0x7FC00000u);
    }
    }  // docs of this wave
}


// fp32 -> sortable key (distance asc, NaN last)
__device__ __forceinline__ uint64_t f32_to_key(float f) {
    if (f != f) return kKeyNaN;
    uint32_t b = __float_as_uint(f);
    b = (b >> 31) ? ~b : (b | 0x80000000u);
    return (uint64_t)b;
}
__device__ __forceinline__ float key_to_f32(uint64_t k) {
    if (k == kKeyNaN) return __uint_as_float(0x7FC00000u);
    uint32_t b = (uint32_t)k;
    b = (b >> 31) ? (b & 0x7FFFFFFFu) : ~b;
    return __uint_as_float(b);
}

// one workgroup per segment of kSegSort entries: sort by (key,row), write the first k.
// first stage reads distances (and skips empty docs), later stages read (key,row) partials.
// row_map / n_in_dev (first stage only): entry g is doc row_map[g], and only the first *n_in_dev entries exist.
__global__ __launch_bounds__(256) void k_topk_segments(const float* dist, const int64_t* blk_off,
                                                        const uint64_t* key_in, const int32_t* row_in, int64_t n_in,
                                                        int k, int seg, uint64_t* key_out, int32_t* row_out,
                                                        const int32_t* row_map, const int* n_in_dev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* SK = (uint64_t*)smem;
    int32_t* SR = (int32_t*)(smem + (size_t)kSegSort * 8);
    const int64_t base = (int64_t)blockIdx.x * seg;
    if (dist && n_in_dev) n_in = min(n_in, (int64_t)*n_in_dev);
    for (int i = threadIdx.x; i < seg; i += blockDim.x) {
        const int64_t g = base + i;
        uint64_t key = kKeyNaN;
        int32_t row = 0x7FFFFFFF;
        if (g < n_in) {
            if (dist) {
                const int64_t doc = row_map ? (int64_t)row_map[g] : g;
                if (blk_off[doc + 1] > blk_off[doc]) {  // docs without vectors are not rows of the result
                    key = f32_to_key(dist[g]);
                    row = (int32_t)doc;
                }
            } else {
                key = key_in[g];
                row = row_in[g];
            }
        }
        SK[i] = key;
        SR[i] = row;
    }
    __syncthreads();
    bitonic_asc_key_row(SK, SR, seg);
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        key_out[(int64_t)blockIdx.x * k + i] = SK[i];
        row_out[(int64_t)blockIdx.x * k + i] = SR[i];
    }
}

// candidates of one query: every doc whose screen distance is within 2E of the k-th best screen distance
// (kth_key = last entry of the screen's top-k; NaN key = fewer than k docs with vectors -> every doc is a candidate)
__global__ void k_ms_candidates(const float* dist16, const int64_t* blk_off, int64_t n_docs, const uint64_t* topk_keys, int k,
                                float two_e, int32_t* list, int cap, int* ctl) {
    const int64_t doc = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (doc >= n_docs || blk_off[doc + 1] <= blk_off[doc]) return;
    const uint64_t kth = topk_keys[k - 1];
    float thr = __builtin_inff();
    if (kth != kKeyNaN) {
        thr = key_to_f32(kth) + two_e;
        thr += fabsf(thr) * 1.2e-7f + 1e-30f;  // round the sum up
    }
    if (!(dist16[doc] > thr)) {  // (a NaN screen value stays a candidate)
        const int slot = atomicAdd(&ctl[0], 1);
        if (slot < cap) list[slot] = (int32_t)doc;
        else ctl[1] = 1;
    }
}

// ---- fast selection path of the screened search (k <= kMsFastK): all queries of a launch at once (grid.y) ----
constexpr int kMsFastK = 64;
constexpr int kMsSelSeg = kWave * kSelPerLane;  // 1024 entries per wave

// One wave per segment of 1024 entries: the k smallest distances of the segment, as order keys (unsorted, padded
// with 0xFFFFFFFF).  First stage reads screen distances (docs without vectors / NaN rank last), later stages keys.
// Only VALUES travel: the stages exist to find the k-th best screen distance.
__global__ __launch_bounds__(kWave) void k_ms_select(const float* dist, const int64_t* blk_off, const uint32_t* key_in,
                                                     int64_t n_in, int64_t in_stride, int k, uint32_t* key_out,
                                                     int64_t out_stride) {
    const int lane = threadIdx.x, y = blockIdx.y;
    const int64_t base = (int64_t)blockIdx.x * kMsSelSeg;
    uint32_t inv[kSelPerLane];  // inverted key: the smallest distance has the largest inv; 0 = absent
#pragma unroll
    for (int j = 0; j < kSelPerLane; ++j) {
        const int64_t g = base + j * kWave + lane;
        uint32_t key = 0xFFFFFFFFu;
        if (g < n_in) {
            if (dist) {
                const float v = dist[(int64_t)y * in_stride + g];
                if (blk_off[g + 1] > blk_off[g] && v == v) key = f32_order_key(v);
            } else {
                key = key_in[(int64_t)y * in_stride + g];
            }
        }
        inv[j] = ~key;
    }
    uint32_t* out = key_out + (int64_t)y * out_stride + (int64_t)blockIdx.x * k;
    for (int i = lane; i < k; i += kWave) out[i] = 0xFFFFFFFFu;
    const int n_valid = wave_count_ge(inv, 1u);
    const int kk = min(k, n_valid);
    if (kk == 0) return;
    const uint32_t x = wave_nth_largest(inv, kk);
    int n = 0;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {  // strictly better than the k-th first, then ties up to k
#pragma unroll
        for (int j = 0; j < kSelPerLane; ++j) {
            const bool want = pass == 0 ? inv[j] > x : inv[j] == x;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(want);
            const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
            if (want && pos < kk) out[pos] = ~inv[j];
            n += __builtin_popcountll(bal);
        }
    }
}

// candidates of every query of the launch (grid.y): docs whose screen distance is within 2E of the k-th best one -> the WIDE
// list (+ each entry's screen distance, when sd_out is given); the docs AT OR ABOVE the k-th best screen distance -> the STARTER
// list (when list_a is given): k_ms_tighten narrows the wide list with the starter's exact distances
constexpr int kMsCandPerThread = 4;  // a workgroup of 256 threads looks at 1024 docs
__global__ __launch_bounds__(256) void k_ms_candidates_y(const float* dist16, int64_t dist_stride, const int64_t* blk_off,
                                                          int64_t n_docs, const uint32_t* topk_keys, int64_t key_stride, int k,
                                                          const float* two_e, int32_t* list, int cap, int* ctl, float* sd_out,
                                                          int32_t* list_a, int* ctl_a) {
    __shared__ uint32_t kth_s;
    const int y = blockIdx.y;
    if (threadIdx.x < kWave) {  // k <= kMsFastK = 64: one key per lane of the first wave
        uint32_t key = threadIdx.x < k ? topk_keys[(int64_t)y * key_stride + threadIdx.x] : 0u;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) key = max(key, (uint32_t)__shfl_xor((int)key, o, kWave));
        if (threadIdx.x == 0) kth_s = key;
    }
    __syncthreads();
    const uint32_t kth = kth_s;
    float thr = __builtin_inff();
    if (kth != 0xFFFFFFFFu) {
        const uint32_t ub = (kth & 0x80000000u) ? (kth & 0x7FFFFFFFu) : ~kth;
        thr = __uint_as_float(ub) + two_e[y];
        thr += fabsf(thr) * 1.2e-7f + 1e-30f;
    }
#pragma unroll
    for (int u = 0; u < kMsCandPerThread; ++u) {
        const int64_t doc = ((int64_t)blockIdx.x * kMsCandPerThread + u) * blockDim.x + threadIdx.x;
        if (doc >= n_docs || blk_off[doc + 1] <= blk_off[doc]) continue;
        const float v = dist16[(int64_t)y * dist_stride + doc];
        if (!(v > thr)) {
            const int slot = atomicAdd(&ctl[2 * y], 1);
            if (slot < cap) {
                list[(int64_t)y * cap + slot] = (int32_t)doc;
                if (sd_out) sd_out[(int64_t)y * cap + slot] = v;
            } else {
                ctl[2 * y + 1] = 1;
            }
            if (list_a && v == v && f32_order_key(v) <= kth) {  // (the select ranks exactly these keys)
                const int sa = atomicAdd(&ctl_a[2 * y], 1);
                if (sa < cap) list_a[(int64_t)y * cap + sa] = (int32_t)doc;
                else ctl_a[2 * y + 1] = 1;
            }
        }
    }
}

// The wide list -> the final list (grid.y = query, one workgroup).  The starter docs (>= k of them: the screen's top-k and its
// ties) carry their EXACT distances: their k-th smallest, D, is an upper bound of the true k-th best exact distance, and a doc
// of the exact top-k (ties included) has exact <= D, hence screen <= exact + E <= D + E.  The wide list's threshold is
// x_k + 2E with x_k the k-th best SCREEN distance; D <= x_k + E always (every starter doc has exact <= screen + E), and
// D ~ x_k in practice: the band halves and the docs to re-score drop by ~6 x (the band sits in the tail of the score
// distribution).  The starter is part of the final list (screen <= x_k <= D + E).  Without a usable starter (fewer than k docs
// with vectors, a starter list beyond kMsTightenMax entries or overflown) the final list is the wide list.
constexpr int kMsTightenMax = 1024;
__global__ __launch_bounds__(256) void k_ms_tighten(const float* dist_a, const int* ctl_a, const int32_t* list_c, const float* sd_c,
                                                     const int* ctl_c, int cap, int k, const float* two_e, int32_t* list_b, int* ctl_b) {
    __shared__ uint32_t key_a[kMsTightenMax];
    __shared__ float thr_s;
    __shared__ int n_b;
    const int y = blockIdx.y, tid = threadIdx.x;
    const int n_c = min(ctl_c[2 * y], cap);
    if (ctl_c[2 * y + 1] != 0) {  // the wide list overflowed: the caller's exact full scan
        if (tid == 0) {
            ctl_b[2 * y] = 0;
            ctl_b[2 * y + 1] = 1;
        }
        return;
    }
    const int n_a = ctl_a[2 * y];
    const bool usable = ctl_a[2 * y + 1] == 0 && n_a >= k && n_a <= kMsTightenMax;  // workgroup-uniform
    if (tid == 0) {
        thr_s = __builtin_inff();
        n_b = 0;
    }
    if (usable) {
        for (int i = tid; i < n_a; i += blockDim.x) key_a[i] = f32_order_key(dist_a[(int64_t)y * cap + i]);
        __syncthreads();
        for (int i = tid; i < n_a; i += blockDim.x) {  // rank under the strict order (key, position): exactly one entry has rank k - 1
            const uint32_t ki = key_a[i];
            int rank = 0;
            for (int j = 0; j < n_a; ++j) rank += (key_a[j] < ki || (key_a[j] == ki && j < i)) ? 1 : 0;
            if (rank == k - 1) {
                const uint32_t ub = (ki & 0x80000000u) ? (ki & 0x7FFFFFFFu) : ~ki;
                // D + E rounded UP (two_e holds 2E rounded up); a NaN here keeps every entry (the comparison below)
                thr_s = __double2float_ru((double)__uint_as_float(ub) + 0.5 * (double)two_e[y]);
            }
        }
    }
    __syncthreads();
    const float thr = thr_s;
    for (int i = tid; i < n_c; i += blockDim.x) {
        if (!(sd_c[(int64_t)y * cap + i] > thr)) list_b[(int64_t)y * cap + atomicAdd(&n_b, 1)] = list_c[(int64_t)y * cap + i];
    }
    __syncthreads();
    if (tid == 0) {
        ctl_b[2 * y] = n_b;
        ctl_b[2 * y + 1] = 0;
    }
}

// exact top-k of one query's re-scored candidates (grid.y = query): sort by (distance, doc) in LDS, write the result
__global__ __launch_bounds__(256) void k_ms_final(const float* cand_dist, const int32_t* cand_list, const int* ctl, int cap,
                                                   int k, int64_t row_offset, float* out_d, int64_t* out_r) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int y = blockIdx.y;
    const int n = min(ctl[2 * y], cap);
    const int np = next_pow2(max(n, 1));
    uint64_t* SK = (uint64_t*)smem;
    int32_t* SR = (int32_t*)(smem + (size_t)np * 8);
    for (int i = threadIdx.x; i < np; i += blockDim.x) {
        uint64_t key = kKeyNaN;
        int32_t row = 0x7FFFFFFF;
        if (i < n) {
            key = f32_to_key(cand_dist[(int64_t)y * cap + i]);
            row = cand_list[(int64_t)y * cap + i];
        }
        SK[i] = key;
        SR[i] = row;
    }
    __syncthreads();
    bitonic_asc_key_row(SK, SR, np);
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        const bool ok = i < n && SR[i] != 0x7FFFFFFF;
        out_d[(int64_t)y * k + i] = ok ? key_to_f32(SK[i]) : __uint_as_float(0x7FC00000u);
        out_r[(int64_t)y * k + i] = ok ? (int64_t)SR[i] + row_offset : -1;
    }
}

__global__ void k_ms_write_out(const uint64_t* key, const int32_t* row, int k, int64_t row_offset, float* out_d,
                               int64_t* out_r) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const bool ok = row[i] != 0x7FFFFFFF;
    out_d[i] = ok ? key_to_f32(key[i]) : __uint_as_float(0x7FC00000u);
    out_r[i] = ok ? (int64_t)row[i] + row_offset : -1;
}

}  // namespace mi355

namespace {

inline float host_bf16_to_f32(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

inline uint16_t host_bf16_rn(float f) {  // round-to-nearest-even; NaN/Inf keep their class (the screen is off for them)
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)((u >> 16) | ((u & 0xFFFFu) ? 0x40u : 0u));
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

int ms_reserve(mi355dr_index* idx, MultiVecStore* m, int64_t want_blocks, int64_t want_docs) {
    if (want_blocks > m->cap_blocks) {
        int64_t nb = std::max<int64_t>(want_blocks, m->cap_blocks + m->cap_blocks / 2);
        float* t = nullptr;
        HIPCHECK(idx, hipMalloc(&t, (size_t)nb * kMsBlkRows * m->dpad * sizeof(float)));
        if (m->n_blocks > 0)
            HIPCHECK(idx, hipMemcpy(t, m->tok, (size_t)m->n_blocks * kMsBlkRows * m->dpad * sizeof(float),
                                    hipMemcpyDeviceToDevice));
        if (m->tok) (void)hipFree(m->tok);
        m->tok = t;
        uint4* t16 = nullptr;
        HIPCHECK(idx, hipMalloc(&t16, (size_t)nb * m->nkk * 64 * sizeof(uint4)));
        if (m->n_blocks > 0)
            HIPCHECK(idx, hipMemcpy(t16, m->tok16, (size_t)m->n_blocks * m->nkk * 64 * sizeof(uint4), hipMemcpyDeviceToDevice));
        if (m->tok16) (void)hipFree(m->tok16);
        m->tok16 = t16;
        m->cap_blocks = nb;
    }
    if (want_docs > m->cap_docs) {
        int64_t nd = std::max<int64_t>(want_docs, m->cap_docs + m->cap_docs / 2);
        int64_t* b = nullptr;
        HIPCHECK(idx, hipMalloc(&b, (size_t)(nd + 1) * sizeof(int64_t)));
        if (m->blk_off) (void)hipFree(m->blk_off);
        m->blk_off = b;
        m->cap_docs = nd;
    }
    return MI355DR_OK;
}

}  // namespace

extern "C" {

int mi355dr_add_multivec(mi355dr_index* idx, const float* vecs, const int64_t* offsets, int64_t n_docs) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (n_docs < 0 || !offsets || (n_docs > 0 && offsets[n_docs] > 0 && !vecs))
        return fail(idx, MI355DR_E_INVALID, "bad multi-vector arguments");
    if (n_docs == 0) return MI355DR_OK;
    for (int64_t i = 0; i < n_docs; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(idx, MI355DR_E_INVALID, "offsets must be non-decreasing");
    HIPCHECK(idx, hipSetDevice(idx->device));
    if (!idx->mv) {
        idx->mv = new MultiVecStore();
        idx->mv->dpad = (int)round_up(idx->dim, 8);
        idx->mv->nkk = (int)round_up(idx->dim, 16) / 16;
        idx->mv->blk_off_host.push_back(0);
    }
    MultiVecStore* m = idx->mv;
    if (m->n_docs + n_docs >= ((int64_t)1 << 31)) return fail(idx, MI355DR_E_UNSUPPORTED, "too many docs");
    // the block-offset table grows while the images are built; a failure below (reservation, copies) takes it back
    struct Rollback {
        std::vector<int64_t>& v;
        size_t n;
        bool keep = false;
        ~Rollback() {
            if (!keep) v.resize(n);
        }
    } rollback{m->blk_off_host, m->blk_off_host.size()};
    // padded host image of the new docs: whole 32-row blocks, tail = copies of the last token, dim zero-padded
    int64_t new_blocks = 0;
    for (int64_t i = 0; i < n_docs; ++i) new_blocks += (offsets[i + 1] - offsets[i] + kMsBlkRows - 1) / kMsBlkRows;
    std::vector<float> img((size_t)new_blocks * kMsBlkRows * m->dpad, 0.0f);
    int64_t blk = 0;
    const int d = idx->dim, dp = m->dpad;
    for (int64_t i = 0; i < n_docs; ++i) {
        const int64_t T = offsets[i + 1] - offsets[i];
        const int64_t nb = (T + kMsBlkRows - 1) / kMsBlkRows;
        for (int64_t r = 0; r < nb * kMsBlkRows; ++r) {
            const int64_t src = offsets[i] + std::min<int64_t>(r, T - 1);
            float* dst = &img[(size_t)(blk * kMsBlkRows + r) * dp];
            const float* sv = vecs + src * d;
            for (int j = 0; j < dp; ++j) {  // stored position j holds column ms_perm(j) (zero beyond dim)
                const int c = ms_perm(j);
                dst[j] = c < d ? sv[c] : 0.0f;
            }
        }
        blk += nb;
        m->blk_off_host.push_back(m->n_blocks + blk);
    }
    // bf16 fragment image of the same padded blocks + the store-wide quantities of the screen bound
    std::vector<uint16_t> img16((size_t)new_blocks * m->nkk * 64 * 8, 0);
    blk = 0;
    for (int64_t i = 0; i < n_docs; ++i) {
        const int64_t T = offsets[i + 1] - offsets[i];
        const int64_t nb = (T + kMsBlkRows - 1) / kMsBlkRows;
        for (int64_t t = 0; t < T; ++t) {
            const float* sv = vecs + (offsets[i] + t) * d;
            double n2 = 0.0, n16 = 0.0, r2 = 0.0;
            for (int c = 0; c < d; ++c) {
                if (!std::isfinite(sv[c])) m->finite = false;
                const double x = sv[c], x16 = host_bf16_to_f32(host_bf16_rn(sv[c]));
                n2 += x * x;
                n16 += x16 * x16;
                r2 += (x - x16) * (x - x16);
            }
            if (std::isfinite(n2)) {
                m->tok_norm_max = std::max(m->tok_norm_max, std::sqrt(n2));
                m->tok16_norm_max = std::max(m->tok16_norm_max, std::sqrt(n16));
                m->tok_res_max = std::max(m->tok_res_max, std::sqrt(r2));
            }
        }
        for (int64_t b = 0; b < nb; ++b)
            for (int kk = 0; kk < m->nkk; ++kk)
                for (int lane = 0; lane < 64; ++lane) {
                    const int64_t r = b * kMsBlkRows + (lane & 31);
                    const float* sv = vecs + (offsets[i] + std::min<int64_t>(r, T - 1)) * d;
                    uint16_t* dst = &img16[(((size_t)(blk + b) * m->nkk + kk) * 64 + lane) * 8];
                    for (int j = 0; j < 8; ++j) {
                        const int c = kk * 16 + (lane >> 5) * 8 + j;
                        dst[j] = c < d ? host_bf16_rn(sv[c]) : (uint16_t)0;
                    }
                }
        blk += nb;
    }
    CHECK(ms_reserve(idx, m, m->n_blocks + new_blocks, m->n_docs + n_docs));
    if (new_blocks > 0)
        HIPCHECK(idx, hipMemcpy(m->tok16 + (size_t)m->n_blocks * m->nkk * 64, img16.data(), img16.size() * sizeof(uint16_t),
                                hipMemcpyHostToDevice));
    if (new_blocks > 0)
        HIPCHECK(idx, hipMemcpy(m->tok + (size_t)m->n_blocks * kMsBlkRows * dp, img.data(), img.size() * sizeof(float),
                                hipMemcpyHostToDevice));
    HIPCHECK(idx, hipMemcpy(m->blk_off, m->blk_off_host.data(), m->blk_off_host.size() * sizeof(int64_t),
                            hipMemcpyHostToDevice));
    rollback.keep = true;
    for (int64_t i = 0; i < n_docs; ++i) m->tok_cnt_host.push_back((int32_t)(offsets[i + 1] - offsets[i]));
    m->n_blocks += new_blocks;
    m->n_docs += n_docs;
    return MI355DR_OK;
}

namespace {

__device__ __forceinline__ uint16_t dev_bf16_rn(float f) {  // same rounding as host_bf16_rn
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)((u >> 16) | ((u & 0xFFFFu) ? 0x40u : 0u));
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// one workgroup (64 lanes) per NEW 32-row block: the padded fp32 image (columns permuted like the host path), the bf16
// fragment image, and the store-wide maxima of the screen bound (non-negative doubles order like their bit patterns)
__global__ __launch_bounds__(64) void k_ms_build(const float* __restrict__ vecs, const int64_t* __restrict__ doc_tok0,
                                                  const int64_t* __restrict__ doc_T, const int32_t* __restrict__ blk_doc,
                                                  const int64_t* __restrict__ doc_blk0, int d, int dp, int nkk, int64_t blk_base,
                                                  float* tok, uint16_t* tok16, unsigned long long* stats, int* not_finite) {
    const int64_t b = blockIdx.x;
    const int lane = threadIdx.x;
    const int i = blk_doc[b];
    const int64_t T = doc_T[i], bi = b - doc_blk0[i];
    for (int r = 0; r < kMsBlkRows; ++r) {
        const float* sv = vecs + (doc_tok0[i] + min(bi * kMsBlkRows + r, T - 1)) * (int64_t)d;
        float* dst = tok + ((blk_base + b) * kMsBlkRows + r) * (int64_t)dp;
        for (int j = lane; j < dp; j += 64) {
            const int c = ms_perm(j);
            dst[j] = c < d ? sv[c] : 0.0f;
        }
    }
    {
        const int r = lane & 31, hf = lane >> 5;
        const float* sv = vecs + (doc_tok0[i] + min(bi * kMsBlkRows + r, T - 1)) * (int64_t)d;
        for (int kk = 0; kk < nkk; ++kk) {
            uint16_t* dst = tok16 + ((((blk_base + b) * nkk + kk) * 64 + lane) * (int64_t)8);
            for (int j = 0; j < 8; ++j) {
                const int c = kk * 16 + hf * 8 + j;
                dst[j] = c < d ? dev_bf16_rn(sv[c]) : (uint16_t)0;
            }
        }
    }
    if (lane < kMsBlkRows && bi * kMsBlkRows + lane < T) {  // the real tokens of this block: norms, residual, finiteness
        const float* sv = vecs + (doc_tok0[i] + bi * kMsBlkRows + lane) * (int64_t)d;
        double n2 = 0.0, n16 = 0.0, r2 = 0.0;
        bool fin = true;
        for (int c = 0; c < d; ++c) {
            const float f = sv[c];
            fin = fin && (fabsf(f) <= 3.402823466e38f);
            const double x = f, x16 = __uint_as_float((uint32_t)dev_bf16_rn(f) << 16);
            n2 += x * x;
            n16 += x16 * x16;
            r2 += (x - x16) * (x - x16);
        }
        if (!fin) atomicExch(not_finite, 1);
        if (n2 == n2 && n2 <= 1.7976931348623157e308) {
            atomicMax(&stats[0], (unsigned long long)__double_as_longlong(sqrt(n2)));
            atomicMax(&stats[1], (unsigned long long)__double_as_longlong(sqrt(n16)));
            atomicMax(&stats[2], (unsigned long long)__double_as_longlong(sqrt(r2)));
        }
    }
}

}  // namespace

int mi355dr_add_multivec_device(mi355dr_index* idx, const float* vecs_dev, const int64_t* offsets, int64_t n_docs) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (n_docs < 0 || !offsets || (n_docs > 0 && offsets[n_docs] > 0 && !vecs_dev))
        return fail(idx, MI355DR_E_INVALID, "bad multi-vector arguments");
    if (n_docs == 0) return MI355DR_OK;
    for (int64_t i = 0; i < n_docs; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(idx, MI355DR_E_INVALID, "offsets must be non-decreasing");
    HIPCHECK(idx, hipSetDevice(idx->device));
    if (!idx->mv) {
        idx->mv = new MultiVecStore();
        idx->mv->dpad = (int)round_up(idx->dim, 8);
        idx->mv->nkk = (int)round_up(idx->dim, 16) / 16;
        idx->mv->blk_off_host.push_back(0);
    }
    MultiVecStore* m = idx->mv;
    if (m->n_docs + n_docs >= ((int64_t)1 << 31)) return fail(idx, MI355DR_E_UNSUPPORTED, "too many docs");
    std::vector<int64_t> tok0(n_docs), T(n_docs), blk0(n_docs);
    std::vector<int32_t> blk_doc;
    // the new docs' block offsets are built on the side and appended to the store's table only after everything below
    // succeeded: a failed reservation / copy / kernel leaves the table as it was (n_docs and n_blocks are untouched too)
    std::vector<int64_t> new_off;
    new_off.reserve((size_t)n_docs);
    int64_t new_blocks = 0;
    for (int64_t i = 0; i < n_docs; ++i) {
        tok0[i] = offsets[i];
        T[i] = offsets[i + 1] - offsets[i];
        blk0[i] = new_blocks;
        const int64_t nb = (T[i] + kMsBlkRows - 1) / kMsBlkRows;
        for (int64_t b = 0; b < nb; ++b) blk_doc.push_back((int32_t)i);
        new_blocks += nb;
        new_off.push_back(m->n_blocks + new_blocks);
    }
    CHECK(ms_reserve(idx, m, m->n_blocks + new_blocks, m->n_docs + n_docs));
    double v[3] = {0.0, 0.0, 0.0};
    int nf = 0;
    if (new_blocks > 0) {
        struct Scratch {  // released on every exit
            int64_t *tok0 = nullptr, *T = nullptr, *blk0 = nullptr;
            int32_t* blk_doc = nullptr;
            unsigned long long* stats = nullptr;
            int* nf = nullptr;
            ~Scratch() {
                for (void* p : {(void*)tok0, (void*)T, (void*)blk0, (void*)blk_doc, (void*)stats, (void*)nf})
                    if (p) (void)hipFree(p);
            }
        } sc;
        HIPCHECK(idx, hipMalloc(&sc.tok0, n_docs * sizeof(int64_t)));
        HIPCHECK(idx, hipMalloc(&sc.T, n_docs * sizeof(int64_t)));
        HIPCHECK(idx, hipMalloc(&sc.blk0, n_docs * sizeof(int64_t)));
        HIPCHECK(idx, hipMalloc(&sc.blk_doc, blk_doc.size() * sizeof(int32_t)));
        HIPCHECK(idx, hipMalloc(&sc.stats, 3 * sizeof(unsigned long long)));
        HIPCHECK(idx, hipMalloc(&sc.nf, sizeof(int)));
        hipStream_t s = idx->stream;
        HIPCHECK(idx, hipMemcpyAsync(sc.tok0, tok0.data(), n_docs * sizeof(int64_t), hipMemcpyHostToDevice, s));
        HIPCHECK(idx, hipMemcpyAsync(sc.T, T.data(), n_docs * sizeof(int64_t), hipMemcpyHostToDevice, s));
        HIPCHECK(idx, hipMemcpyAsync(sc.blk0, blk0.data(), n_docs * sizeof(int64_t), hipMemcpyHostToDevice, s));
        HIPCHECK(idx, hipMemcpyAsync(sc.blk_doc, blk_doc.data(), blk_doc.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
        HIPCHECK(idx, hipMemsetAsync(sc.stats, 0, 3 * sizeof(unsigned long long), s));
        HIPCHECK(idx, hipMemsetAsync(sc.nf, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_ms_build, dim3((unsigned)new_blocks), dim3(64), 0, s, vecs_dev, sc.tok0, sc.T, sc.blk_doc, sc.blk0,
                           idx->dim, m->dpad, m->nkk, m->n_blocks, m->tok, (uint16_t*)m->tok16, sc.stats, sc.nf);
        HIPCHECK(idx, hipGetLastError());
        unsigned long long st[3] = {0, 0, 0};
        HIPCHECK(idx, hipMemcpyAsync(st, sc.stats, sizeof(st), hipMemcpyDeviceToHost, s));
        HIPCHECK(idx, hipMemcpyAsync(&nf, sc.nf, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHECK(idx, hipStreamSynchronize(s));
        memcpy(v, st, sizeof(v));
    }
    // the device copy of the table first (the host vector is its source of truth): append, upload, roll back on failure
    const size_t old_size = m->blk_off_host.size();
    m->blk_off_host.insert(m->blk_off_host.end(), new_off.begin(), new_off.end());
    const hipError_t ce = hipMemcpy(m->blk_off, m->blk_off_host.data(), m->blk_off_host.size() * sizeof(int64_t),
                                    hipMemcpyHostToDevice);
    if (ce != hipSuccess) {
        m->blk_off_host.resize(old_size);
        HIPCHECK(idx, ce);
    }
    m->tok_norm_max = std::max(m->tok_norm_max, v[0]);
    m->tok16_norm_max = std::max(m->tok16_norm_max, v[1]);
    m->tok_res_max = std::max(m->tok_res_max, v[2]);
    if (nf) m->finite = false;
    for (int64_t i = 0; i < n_docs; ++i) m->tok_cnt_host.push_back((int32_t)T[i]);
    m->n_blocks += new_blocks;
    m->n_docs += n_docs;
    return MI355DR_OK;
}

namespace {

// ---- the granule-packed bf16 copy (k_maxsim_wg8.h) ----
// one wave per PACKED block: lane = (row = lane & 31, half = lane >> 5) copies its 16-byte fragment of every k-group from the padded
// copy -- same bf16 values, so the two copies screen to bit-identical distances.  Row r of packed block p is token
// min(8 (granule - goff[doc]) + r % 8, T - 1) of the doc that owns granule 4 p + r / 8; rows past the last granule repeat the
// stream's last token (no workgroup ever folds them).
__global__ __launch_bounds__(64) void k_ms_pack8(const uint4* __restrict__ tok16, const int64_t* __restrict__ blk_off,
                                                 const int32_t* __restrict__ tok_cnt, const int64_t* __restrict__ goff,
                                                 int64_t n_docs, int64_t n_gran, int nkk, uint4* __restrict__ out, int64_t p0) {
    const int64_t p = p0 + blockIdx.x;
    const int lane = threadIdx.x, r = lane & 31, hf = lane >> 5;
    int64_t gi = p * 4 + (r >> 3);
    int rr = r & 7;
    if (gi >= n_gran) {
        gi = n_gran - 1;
        rr = 7;
    }
    int64_t lo = 0, hi = n_docs - 1;  // the doc with goff[doc] <= gi < goff[doc + 1]
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (goff[mid + 1] > gi) hi = mid;
        else lo = mid + 1;
    }
    const int64_t t = min((gi - goff[lo]) * 8 + rr, (int64_t)tok_cnt[lo] - 1);
    const uint4* src = tok16 + ((blk_off[lo] + (t >> 5)) * nkk) * 64 + (int)(t & 31) + 32 * hf;
    uint4* dst = out + (p * nkk) * 64 + lane;
    for (int kk = 0; kk < nkk; ++kk) dst[(int64_t)kk * 64] = src[(int64_t)kk * 64];
}

// Makes the packed copy current for the store's docs, or decides it is not to be used (m->pack_use).  Called with the index lock
// held, before a screen launch on stream `s` that could take it.  A failed allocation is not an error: the padded copy serves.
int ms_pack8_ensure(mi355dr_index* idx, MultiVecStore* m, hipStream_t s) {
    if (m->pack_docs == m->n_docs && (m->pack_use || m->pack_mode == idx->maxsim_pack8)) return MI355DR_OK;
    // a store that grew since the copy was built is packed from the block its new granules start in (an ingest loop that searches
    // between its adds pays for the new documents only); a copy that was never built, was judged not to pay, or must move is
    // packed whole
    int64_t first_block = m->pack_use && m->pack_docs >= 0 && m->pack_docs < m->n_docs ? m->pack_gran / 4 : 0;
    m->pack_docs = m->n_docs;
    m->pack_mode = idx->maxsim_pack8;
    m->pack_use = false;
    if (m->nkk != 8 || (int64_t)m->tok_cnt_host.size() != m->n_docs || m->n_blocks == 0) return MI355DR_OK;
    std::vector<int64_t> goff((size_t)m->n_docs + 1);
    goff[0] = 0;
    for (int64_t i = 0; i < m->n_docs; ++i) goff[i + 1] = goff[i] + (m->tok_cnt_host[i] + 7) / 8;
    const int64_t n_gran = goff[m->n_docs], n_pb = (n_gran + 3) / 4;
    if (n_gran == 0) return MI355DR_OK;
    if (idx->maxsim_pack8 < 0 && (double)n_pb > 0.95 * (double)m->n_blocks) return MI355DR_OK;  // (long documents: nothing to gain)
    if (n_pb > m->pack_cap_blocks) {
        first_block = 0;
        if (m->tok16p) (void)hipFree(m->tok16p);
        m->tok16p = nullptr;
        m->pack_cap_blocks = 0;
        const int64_t want = std::max<int64_t>(n_pb, std::min<int64_t>(m->cap_blocks, n_pb + n_pb / 2));
        if (hipMalloc(&m->tok16p, (size_t)want * m->nkk * 64 * sizeof(uint4)) != hipSuccess) {
            (void)hipGetLastError();
            m->tok16p = nullptr;
            return MI355DR_OK;
        }
        m->pack_cap_blocks = want;
    }
    if (m->n_docs > m->pack_cap_docs) {
        if (m->goff) (void)hipFree(m->goff);
        m->goff = nullptr;
        m->pack_cap_docs = 0;
        if (hipMalloc(&m->goff, (size_t)(m->cap_docs + 1) * sizeof(int64_t)) != hipSuccess) {
            (void)hipGetLastError();
            m->goff = nullptr;
            return MI355DR_OK;
        }
        m->pack_cap_docs = m->cap_docs;
    }
    int32_t* cnt = nullptr;
    if (hipMalloc(&cnt, (size_t)m->n_docs * sizeof(int32_t)) != hipSuccess) {
        (void)hipGetLastError();
        return MI355DR_OK;
    }
    struct Free {
        void* p;
        ~Free() { (void)hipFree(p); }
    } free_cnt{cnt};
    HIPCHECK(idx, hipMemcpyAsync(cnt, m->tok_cnt_host.data(), (size_t)m->n_docs * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHECK(idx, hipMemcpyAsync(m->goff, goff.data(), goff.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
    if (n_pb > first_block) {  // (new documents without vectors add no granule)
        hipLaunchKernelGGL(k_ms_pack8, dim3((unsigned)(n_pb - first_block)), dim3(64), 0, s, m->tok16, m->blk_off, cnt, m->goff, m->n_docs,
                           n_gran, m->nkk, m->tok16p, first_block);
        HIPCHECK(idx, hipGetLastError());
        idx->s_ms_packed_built += n_pb - first_block;
    }
    HIPCHECK(idx, hipStreamSynchronize(s));  // (the host vectors above are the copies' sources)
    m->pack_gran = n_gran;
    m->pack_blocks = n_pb;
    m->pack_use = true;
    idx->s_ms_packed_blocks = n_pb;
    return MI355DR_OK;
}

}  // namespace

int64_t mi355dr_size_multivec(const mi355dr_index* idx) { return idx && idx->mv ? idx->mv->n_docs : 0; }

namespace {

// segment-wise top-k of n_in distances (first stage) until one segment is left; returns the buffer index holding it
int ms_topk(mi355dr_index* idx, MultiVecStore* m, hipStream_t s, const float* dist, int64_t n_in, int k, int seg,
            const int32_t* row_map, const int* n_in_dev, int* cur_out) {
    int cur = 0;
    bool first_stage = true;
    while (true) {
        const int64_t nseg = (n_in + seg - 1) / seg;
        hipLaunchKernelGGL(k_topk_segments, dim3((unsigned)nseg), dim3(256), (size_t)kSegSort * 12, s,
                           first_stage ? dist : nullptr, m->blk_off, first_stage ? nullptr : m->pk[cur ^ 1],
                           first_stage ? nullptr : m->pr[cur ^ 1], n_in, k, seg, m->pk[cur], m->pr[cur],
                           first_stage ? row_map : nullptr, first_stage ? n_in_dev : nullptr);
        HIPCHECK(idx, hipGetLastError());
        first_stage = false;
        if (nseg == 1) break;
        n_in = nseg * k;
        cur ^= 1;
    }
    *cur_out = cur;
    return MI355DR_OK;
}

int ms_emit_result(mi355dr_index* idx, MultiVecStore* m, hipStream_t s, int cur, int k, float* out_dist, int64_t* out_rows,
                   bool out_dev) {
    if (out_dev) {  // straight into the caller's device buffers
        hipLaunchKernelGGL(k_ms_write_out, dim3((k + 255) / 256), dim3(256), 0, s, m->pk[cur], m->pr[cur], k, idx->row_offset,
                           out_dist, out_rows);
        HIPCHECK(idx, hipGetLastError());
        return MI355DR_OK;
    }
    hipLaunchKernelGGL(k_ms_write_out, dim3((k + 255) / 256), dim3(256), 0, s, m->pk[cur], m->pr[cur], k, idx->row_offset,
                       m->out_d, m->out_r);
    HIPCHECK(idx, hipGetLastError());
    HIPCHECK(idx, hipMemcpyAsync(out_dist, m->out_d, k * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHECK(idx, hipMemcpyAsync(out_rows, m->out_r, k * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    return MI355DR_OK;
}

__global__ void k_ms_fill_empty(float* d, int64_t* r, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        d[i] = __uint_as_float(0x7FC00000u);
        r[i] = -1;
    }
}

}  // namespace

// qtok: HOST [sum_nq, dim]; outputs on the host (out_dev = false) or in device memory of the index's GPU (out_dev = true:
// written by kernels / device copies on the index's stream, complete on return)
//
// Round 4: a PASS = up to kMsPassGroups groups of <= 4 queries (dims <= 128, k <= kMsFastK): one screen launch for all of them,
// then ONE selection / candidate / exact re-score / final sequence for all of them (grid.y = query of the pass) and one host
// synchronisation -- rounds 2-3 ran that sequence (ten launches, three copies, one synchronisation) once per group of four.
static int search_maxsim_impl(mi355dr_index* idx, const float* qtok, const int32_t* q_offsets, int B, int k, float* out_dist,
                              int64_t* out_rows, bool out_dev) {
    if (k > kKMax) return fail(idx, MI355DR_E_UNSUPPORTED, "k exceeds 1024");
    if (out_dev) {
        if (B > 0) {
            HIPCHECK(idx, hipSetDevice(idx->device));
            const int64_t n = (int64_t)B * k;
            hipLaunchKernelGGL(k_ms_fill_empty, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, idx->stream, out_dist, out_rows, n);
            HIPCHECK(idx, hipGetLastError());
            HIPCHECK(idx, hipStreamSynchronize(idx->stream));
        }
    } else {
        for (int64_t i = 0; i < (int64_t)B * k; ++i) {
            out_dist[i] = NAN;
            out_rows[i] = -1;
        }
    }
    MultiVecStore* m = idx->mv;
    if (B == 0 || !m || m->n_docs == 0) return MI355DR_OK;
    for (int b = 0; b < B; ++b) {
        const int nq = q_offsets[b + 1] - q_offsets[b];
        if (nq < 0) return fail(idx, MI355DR_E_INVALID, "q_offsets must be non-decreasing");
    }
    HIPCHECK(idx, hipSetDevice(idx->device));
    hipStream_t s = idx->stream;
    const int dp = m->dpad, d = idx->dim, nkk = m->nkk;
    const int cols = ms_cols_for(dp);  // query vectors one launch of the EXACT kernel stages; longer queries are scored in tiles
    if (cols < 32) return fail(idx, MI355DR_E_UNSUPPORTED, "dim too large for the MaxSim kernel's LDS budget (dim <= 1272)");
    const size_t lds = (size_t)cols * (dp + 4) * sizeof(float);
    const size_t lds16 = (size_t)4 * nkk * 64 * sizeof(uint4);
    // columns of the fp32 query image of a pass: every group's columns behind the previous group's (+ 32 columns of slack: the
    // list form of k_maxsim stages whole 32-column blocks from a query's FIRST column on; those columns' results are never read)
    constexpr int kImgCols = kMsPassBlocks * 32 + 32;
    constexpr int kPQ = kMsPassQueries;
    // scratch
    if (!m->qtok) {
        HIPCHECK(idx, hipMalloc(&m->qtok, (size_t)kImgCols * dp * sizeof(float)));
        HIPCHECK(idx, hipMemsetAsync(m->qtok, 0, (size_t)kImgCols * dp * sizeof(float), s));
        HIPCHECK(idx, hipMalloc(&m->qfrag, kMsPassGroups * lds16));
        HIPCHECK(idx, hipMalloc(&m->out_d, (size_t)kPQ * kKMax * sizeof(float)));
        HIPCHECK(idx, hipMalloc(&m->out_r, (size_t)kPQ * kKMax * sizeof(int64_t)));
        HIPCHECK(idx, hipMalloc(&m->cand_list, (size_t)3 * kPQ * kMsCandCap * sizeof(int32_t)));
        HIPCHECK(idx, hipMalloc(&m->cand_dist, (size_t)kPQ * kMsCandCap * sizeof(float)));
        HIPCHECK(idx, hipMalloc(&m->cand_sd, (size_t)kPQ * kMsCandCap * sizeof(float)));
        HIPCHECK(idx, hipMalloc(&m->cand_ctl, 3 * 2 * kPQ * sizeof(int)));
        HIPCHECK(idx, hipMalloc(&m->two_e_dev, kPQ * sizeof(float)));
        HIPCHECK(idx, hipHostMalloc(&m->cand_ctl_host, 2 * kPQ * sizeof(int)));
        HIPCHECK(idx, hipFuncSetAttribute((const void*)k_ms_final, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          kMsCandCap * 12));
        HIPCHECK(idx, hipFuncSetAttribute((const void*)k_maxsim, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + kMsRedBytes)));
        CHECK(ms16_prepare(idx, lds16));
        HIPCHECK(idx, hipFuncSetAttribute((const void*)k_topk_segments, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          kSegSort * 12));
    }
    if (m->dist_cap_docs < m->n_docs) {
        if (m->dist) (void)hipFree(m->dist);
        if (m->dist16) (void)hipFree(m->dist16);
        HIPCHECK(idx, hipMalloc(&m->dist, (size_t)4 * m->cap_docs * sizeof(float)));
        HIPCHECK(idx, hipMalloc(&m->dist16, (size_t)kPQ * m->cap_docs * sizeof(float)));
        for (int i = 0; i < 2; ++i) {
            if (m->sel[i]) (void)hipFree(m->sel[i]);
            HIPCHECK(idx, hipMalloc(&m->sel[i], (size_t)kPQ * ((m->cap_docs + kMsSelSeg - 1) / kMsSelSeg) * kMsFastK * sizeof(uint32_t)));
        }
        m->dist_cap_docs = m->cap_docs;
    }
    // segment size: small segments = many workgroups; it must hold k and shrink the list by >= 4x per stage
    int seg = 512;
    while (seg < 4 * k) seg <<= 1;
    if (seg > kSegSort) seg = kSegSort;
    const int64_t nseg0 = (m->n_docs + seg - 1) / seg;
    if (m->part_cap < nseg0 * kKMax) {
        for (int i = 0; i < 2; ++i) {
            if (m->pk[i]) (void)hipFree(m->pk[i]);
            if (m->pr[i]) (void)hipFree(m->pr[i]);
            HIPCHECK(idx, hipMalloc(&m->pk[i], (size_t)nseg0 * kKMax * sizeof(uint64_t)));
            HIPCHECK(idx, hipMalloc(&m->pr[i], (size_t)nseg0 * kKMax * sizeof(int32_t)));
        }
        m->part_cap = nseg0 * kKMax;
    }
    const unsigned grid_all = (unsigned)((m->n_docs + 4 * kMsDocsPerWave - 1) / (4 * kMsDocsPerWave));
    const int64_t n_cand_max = std::min<int64_t>(kMsCandCap, m->n_docs);
    // bf16 round-to-nearest: unit roundoff 2^-8 per operand -> 2^-7 + 2^-16 per product
    const double eps = std::ldexp(1.0, -7) + std::ldexp(1.0, -15) + 3.0 * d * std::ldexp(1.0, -24);
    // pinned staging (pageable copies are synchronous and cost ~20 us each)
    const size_t qimg_n = (size_t)kImgCols * dp, qf16_n = (size_t)kMsPassBlocks * nkk * 64 * 8;
    const size_t need_stage = qimg_n * 4 + qf16_n * 2 + (size_t)kPQ * kKMax * 12 + 256;
    if (m->stage_bytes < need_stage) {
        if (m->stage_host) (void)hipHostFree(m->stage_host);
        HIPCHECK(idx, hipHostMalloc(&m->stage_host, need_stage));
        m->stage_bytes = need_stage;
    }
    float* const qimg = (float*)m->stage_host;
    uint16_t* const qf16 = (uint16_t*)(m->stage_host + qimg_n * 4);
    float* const hd = (float*)(m->stage_host + qimg_n * 4 + qf16_n * 2);
    int64_t* const hr = (int64_t*)(hd + (size_t)kPQ * kKMax + 16);
    // one group = up to 4 queries whose token counts fit `cols` columns (packed tightly: a query may start anywhere in a
    // column block).  pack() writes the group's bf16 fragments at column base `fcol0` of qf16 (fcol0 < 0: none) and its fp32
    // image for the exact kernel at column base `icol0` of qimg (icol0 < 0: none).
    struct Group {
        int nql = 0, col = 0, b_end = 0;
        int q_col0[4] = {0, 0, 0, 0}, q_len[4] = {0, 0, 0, 0};
        double two_e[4] = {0, 0, 0, 0};
        bool finite = true;
    };
    auto pack = [&](int b0, int fcol0, int icol0) {
        Group g;
        int bb = b0;
        while (bb < B && g.nql < 4) {
            const int nq = q_offsets[bb + 1] - q_offsets[bb];
            const int need = std::max(nq, 1);  // (columns are packed tightly: no padding of a query to whole 32-column blocks)
            if (g.col + need > cols) break;  // (a query longer than `cols` never fits: the caller scores it in tiles)
            g.q_col0[g.nql] = g.col;
            g.q_len[g.nql] = nq;
            double norm_sum = 0.0, res_sum = 0.0;
            for (int j = 0; j < nq; ++j) {
                const float* sv = qtok + (int64_t)(q_offsets[bb] + j) * d;
                if (icol0 >= 0) {
                    float* dst = &qimg[(size_t)(icol0 + g.col + j) * dp];
                    for (int c = 0; c < dp; ++c) {
                        const int oc = ms_perm(c);
                        dst[c] = oc < d ? sv[oc] : 0.0f;
                    }
                }
                // bf16 fragment of the same column: block cb = column / 32, lane = (column & 31) + 32 * half
                double n2 = 0.0, r2 = 0.0;
                const int cc = fcol0 + g.col + j;
                for (int c = 0; c < d; ++c) {
                    if (!std::isfinite(sv[c])) g.finite = false;
                    const uint16_t h = host_bf16_rn(sv[c]);
                    const double x = sv[c], x16 = host_bf16_to_f32(h);
                    n2 += x * x;
                    r2 += (x - x16) * (x - x16);
                    if (fcol0 >= 0) {
                        const int kk = c / 16, half = (c % 16) / 8, jj = c % 8;
                        qf16[((((size_t)(cc >> 5) * nkk + kk) * 64) + (cc & 31) + 32 * half) * 8 + jj] = h;
                    }
                }
                norm_sum += std::sqrt(n2);
                res_sum += std::sqrt(r2);
            }
            // per token pair: |q16.d16 - q.d| <= |r_q||d16| + |q||r_d| with the residuals MEASURED (round-to-nearest leaves
            // about half of the a-priori 2^-8 |x|), + fp32 accumulation of both dot products and of the per-doc sums
            const double e_pair = res_sum * m->tok16_norm_max + norm_sum * m->tok_res_max;
            const double e_acc = (3.0 * d + 2.0 * nq) * std::ldexp(1.0, -24) * m->tok_norm_max * norm_sum;
            g.two_e[g.nql] = 2.0 * std::min(e_pair + e_acc, (eps + 2.0 * nq * std::ldexp(1.0, -24)) * m->tok_norm_max * norm_sum) *
                             (1.0 + 1e-6);
            g.col += need;
            ++g.nql;
            ++bb;
        }
        g.b_end = bb;
        return g;
    };
    MsArgs a0{};
    a0.tok = m->tok;
    a0.blk_off = m->blk_off;
    a0.qtok = m->qtok;
    a0.dist = m->dist;
    a0.n_docs = m->n_docs;
    a0.n_items = m->n_docs;
    a0.doc_list = nullptr;
    a0.n_items_dev = nullptr;
    a0.dpad = dp;
    // exact kernel over EVERY doc for one query whose image sits at column c0 of m->qtok -> m->dist row 0 -> top-k -> outputs
    auto full_scan_query = [&](int c0, int len, float* od, int64_t* orow) -> int {
        MsArgs f = a0;
        f.qtok = m->qtok + (int64_t)c0 * dp;
        f.nq_launch = 1;
        f.q_col0[0] = 0;
        f.q_len[0] = len;
        hipLaunchKernelGGL(k_maxsim, dim3(grid_all), dim3(kMsThreads), lds, s, f);
        HIPCHECK(idx, hipGetLastError());
        int cur = 0;
        CHECK(ms_topk(idx, m, s, m->dist, m->n_docs, k, seg, nullptr, nullptr, &cur));
        CHECK(ms_emit_result(idx, m, s, cur, k, od, orow, out_dev));
        HIPCHECK(idx, hipStreamSynchronize(s));
        return MI355DR_OK;
    };
    int b = 0;
    while (b < B) {
        HIPCHECK(idx, hipStreamSynchronize(s));  // the staging buffers are free again
        if (q_offsets[b + 1] - q_offsets[b] > cols) {
            // ---- a query with more vectors than one launch stages (VectorChord's `@#` has no such limit: base.py:518-524):
            // the exact kernel over every doc, one launch per tile of <= cols query vectors, each continuing the per-doc sums
            const int nq = q_offsets[b + 1] - q_offsets[b];
            for (int t0 = 0; t0 < nq; t0 += cols) {
                const int tl = std::min(cols, nq - t0);
                if (t0 > 0) HIPCHECK(idx, hipStreamSynchronize(s));  // (the previous tile's launch has read the staging image)
                std::fill(qimg, qimg + (size_t)cols * dp, 0.0f);
                for (int j = 0; j < tl; ++j) {
                    const float* sv = qtok + (int64_t)(q_offsets[b] + t0 + j) * d;
                    float* dst = &qimg[(size_t)j * dp];
                    for (int c = 0; c < dp; ++c) {
                        const int oc = ms_perm(c);
                        dst[c] = oc < d ? sv[oc] : 0.0f;
                    }
                }
                HIPCHECK(idx, hipMemcpyAsync(m->qtok, qimg, (size_t)cols * dp * sizeof(float), hipMemcpyHostToDevice, s));
                MsArgs f = a0;
                f.nq_launch = 1;
                f.q_col0[0] = 0;
                f.q_len[0] = tl;
                f.dist_in = t0 > 0 ? m->dist : nullptr;
                hipLaunchKernelGGL(k_maxsim, dim3(grid_all), dim3(kMsThreads), lds, s, f);
                HIPCHECK(idx, hipGetLastError());
            }
            idx->s_ms_fallbacks++;
            int cur = 0;
            CHECK(ms_topk(idx, m, s, m->dist, m->n_docs, k, seg, nullptr, nullptr, &cur));
            CHECK(ms_emit_result(idx, m, s, cur, k, out_dist + (int64_t)b * k, out_rows + (int64_t)b * k, out_dev));
            HIPCHECK(idx, hipStreamSynchronize(s));
            ++b;
            continue;
        }
        // ---- the groups of this pass
        const auto t_pack0 = std::chrono::steady_clock::now();
        std::fill(qimg, qimg + qimg_n, 0.0f);
        std::fill(qf16, qf16 + qf16_n, (uint16_t)0);
        const int first = b;
        Group gs[kMsPassGroups];
        int gbase[kMsPassGroups] = {0, 0, 0, 0}, gfirst[kMsPassGroups] = {b, 0, 0, 0};
        gs[0] = pack(b, 0, 0);
        int n_acc = 1, total_col = gs[0].col, bn = gs[0].b_end;
        const bool screen = idx->maxsim_screen && m->finite && gs[0].finite && lds16 <= 160 * 1024;
        if (screen && nkk == 8 && k <= kMsFastK) {
            // The NEXT groups ride the same pass over the token stream (dims <= 128, the single-launch selection path): their
            // columns packed behind the previous group's
            while (n_acc < kMsPassGroups && n_acc < idx->maxsim_pass_groups && bn < B && q_offsets[bn + 1] - q_offsets[bn] <= cols) {
                const Group H = pack(bn, total_col, total_col);
                if (H.nql == 0 || !H.finite) {
                    // (what pack() may have written for H is not used: rebuild the accepted groups alone)
                    std::fill(qimg, qimg + qimg_n, 0.0f);
                    std::fill(qf16, qf16 + qf16_n, (uint16_t)0);
                    for (int g2 = 0; g2 < n_acc; ++g2) (void)pack(gfirst[g2], gbase[g2], gbase[g2]);
                    break;
                }
                gs[n_acc] = H;
                gbase[n_acc] = total_col;
                gfirst[n_acc] = bn;
                total_col += H.col;
                bn = H.b_end;
                ++n_acc;
            }
        }
        b = bn;
        // the LIVE queries of the pass (a query without vectors: reference `if not query_vectors: return []`): row r of the
        // screen distances, of the candidate lists, of the results
        int pq_n = 0, pq_b[kPQ], pq_col0[kPQ], pq_len[kPQ];
        double pq_two_e[kPQ];
        for (int g = 0; g < n_acc; ++g)
            for (int qi = 0; qi < gs[g].nql; ++qi) {
                if (gs[g].q_len[qi] == 0) continue;
                pq_b[pq_n] = gfirst[g] + qi;
                pq_col0[pq_n] = gbase[g] + gs[g].q_col0[qi];
                pq_len[pq_n] = gs[g].q_len[qi];
                pq_two_e[pq_n] = gs[g].two_e[qi];
                ++pq_n;
            }
        if (pq_n == 0) continue;
        if (idx->profile)
            idx->s_ms_pack_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_pack0).count();
        const size_t img_cols = std::min<size_t>(kImgCols, (size_t)((total_col + 31) / 32 * 32 + 32));
        HIPCHECK(idx, hipMemcpyAsync(m->qtok, qimg, img_cols * dp * sizeof(float), hipMemcpyHostToDevice, s));
        if (!screen) {
            // ---- the exact kernel over every doc for the whole group (one launch, <= 4 queries), then a top-k per query
            MsArgs a = a0;
            a.nq_launch = gs[0].nql;
            for (int qi = 0; qi < 4; ++qi) {
                a.q_col0[qi] = gs[0].q_col0[qi];
                a.q_len[qi] = gs[0].q_len[qi];
            }
            hipLaunchKernelGGL(k_maxsim, dim3(grid_all), dim3(kMsThreads), lds, s, a);
            HIPCHECK(idx, hipGetLastError());
            for (int qi = 0; qi < gs[0].nql; ++qi) {
                if (gs[0].q_len[qi] == 0) continue;
                int cur = 0;
                CHECK(ms_topk(idx, m, s, m->dist + (int64_t)qi * m->n_docs, m->n_docs, k, seg, nullptr, nullptr, &cur));
                CHECK(ms_emit_result(idx, m, s, cur, k, out_dist + (int64_t)(first + qi) * k, out_rows + (int64_t)(first + qi) * k, out_dev));
                HIPCHECK(idx, hipStreamSynchronize(s));
            }
            continue;
        }
        // ---- the screen: one launch for every live query of the pass
        Ms16Args sa{};
        sa.tok16 = m->tok16;
        sa.blk_off = m->blk_off;
        sa.qfrag = m->qfrag;
        sa.dist = m->dist16;
        sa.n_docs = m->n_docs;
        sa.nkk = nkk;
        sa.nq_launch = pq_n;
        sa.aligned = idx->maxsim_aligned ? 1 : 0;
        for (int r = 0; r < pq_n; ++r) {
            sa.q_col0[r] = pq_col0[r];
            sa.q_len[r] = pq_len[r];
            if (pq_col0[r] != 32 * r || pq_len[r] > 32) sa.aligned = 0;
        }
        const int ncb_launch = (total_col + 31) / 32;
        HIPCHECK(idx, hipMemcpyAsync(m->qfrag, qf16, (size_t)std::max(ncb_launch, 4) * nkk * 64 * 8 * sizeof(uint16_t),
                                     hipMemcpyHostToDevice, s));
        if (idx->profile) {
            for (auto& e : idx->ms_ev)
                if (!e) HIPCHECK(idx, hipEventCreate(&e));
            HIPCHECK(idx, hipEventRecord(idx->ms_ev[0], s));
        }
        if (nkk == 8) {  // dims <= 128: the compile-time-unrolled forms, only as many column blocks as the pass has
            Ms16Pack pk{};
            bool packed = false;
            // the granule-packed copy serves the workgroup form's aligned passes (k_maxsim_wg8.h) and the passes of up to four column
            // blocks (k_maxsim16_d128<.., PK>: one to four queries per call are bound by the token stream's bytes: 10 % fewer)
            const bool wg_form = ms16_takes_wg(idx, ncb_launch, m->n_docs, m->n_blocks);
            if (idx->maxsim_pack8 != 0 && (wg_form ? sa.aligned != 0 : ncb_launch <= 4)) {
                CHECK(ms_pack8_ensure(idx, m, s));
                if (m->pack_use) {
                    pk.tok16p = m->tok16p;
                    pk.goff = m->goff;
                    pk.n_gran = m->pack_gran;
                    pk.n_pblocks = m->pack_blocks;
                    packed = true;
                }
            }
            idx->s_ms_packed_launches += packed ? 1 : 0;
            CHECK(ms16_d128_launch(idx, s, ncb_launch, m->n_docs, m->n_blocks, idx->maxsim_persistent != 0, sa, packed ? &pk : nullptr));
        } else {
            CHECK(ms16_generic_launch(idx, s, grid_all, lds16, sa));
        }
        idx->s_ms_screen_cols += 32 * (int64_t)ncb_launch;
        if (idx->profile) HIPCHECK(idx, hipEventRecord(idx->ms_ev[1], s));
        bool handled[kPQ];
        for (int r = 0; r < kPQ; ++r) handled[r] = false;
        if (k <= kMsFastK) {
            // ---- fast path: every step handles all queries of the pass at once (grid.y), one host sync per pass
            const int64_t sel_stride = ((m->cap_docs + kMsSelSeg - 1) / kMsSelSeg) * kMsFastK;
            int64_t n_in = m->n_docs;
            int cur = 0;
            bool first_stage = true;
            while (true) {  // k best screen distances per 1024-entry segment, until one segment is left
                const int64_t nseg = (n_in + kMsSelSeg - 1) / kMsSelSeg;
                hipLaunchKernelGGL(k_ms_select, dim3((unsigned)nseg, pq_n), dim3(kWave), 0, s,
                                   first_stage ? m->dist16 : nullptr, m->blk_off, first_stage ? nullptr : m->sel[cur ^ 1], n_in,
                                   first_stage ? m->n_docs : sel_stride, k, m->sel[cur], sel_stride);
                HIPCHECK(idx, hipGetLastError());
                first_stage = false;
                if (nseg == 1) break;
                n_in = nseg * k;
                cur ^= 1;
            }
            float* const te = hd;  // (staging: pinned; the results overwrite it after the synchronisation below)
            for (int r = 0; r < pq_n; ++r) {
                te[r] = (float)pq_two_e[r];
                if ((double)te[r] < pq_two_e[r]) te[r] = std::nextafter(te[r], INFINITY);
            }
            HIPCHECK(idx, hipMemcpyAsync(m->two_e_dev, te, pq_n * sizeof(float), hipMemcpyHostToDevice, s));
            // wide list (+ starter) -> [starter re-scored exactly -> final list] -> final list re-scored exactly -> exact top-k
            const bool tighten = idx->maxsim_tighten != 0;
            int32_t* const list_c = m->cand_list;
            int32_t* const list_a = m->cand_list + (size_t)kPQ * kMsCandCap;
            int32_t* const list_b = m->cand_list + (size_t)2 * kPQ * kMsCandCap;
            int* const ctl_c = m->cand_ctl;
            int* const ctl_a = m->cand_ctl + 2 * kPQ;
            int* const ctl_b = m->cand_ctl + 4 * kPQ;
            HIPCHECK(idx, hipMemsetAsync(m->cand_ctl, 0, 3 * 2 * kPQ * sizeof(int), s));
            hipLaunchKernelGGL(k_ms_candidates_y, dim3((unsigned)((m->n_docs + 256 * kMsCandPerThread - 1) / (256 * kMsCandPerThread)), pq_n),
                               dim3(256), 0, s, m->dist16,
                               m->n_docs, m->blk_off, m->n_docs, m->sel[cur], sel_stride, k, m->two_e_dev, list_c, kMsCandCap, ctl_c,
                               tighten ? m->cand_sd : nullptr, tighten ? list_a : nullptr, tighten ? ctl_a : nullptr);
            HIPCHECK(idx, hipGetLastError());
            MsArgs c = a0;
            c.dist = m->cand_dist;
            c.n_items = n_cand_max;
            c.list_stride = kMsCandCap;
            c.nq_launch = pq_n;
            for (int r = 0; r < pq_n; ++r) {
                c.q_col0[r] = pq_col0[r];
                c.q_len[r] = pq_len[r];
            }
            // long documents (>= 8 blocks on average: pages): one workgroup per candidate, its four waves share the blocks
            const bool coop = idx->maxsim_coop < 0 ? m->n_blocks >= 8 * m->n_docs : idx->maxsim_coop != 0;
            c.coop = coop ? 1 : 0;
            // LDS for the column blocks the longest query of the pass has (a workgroup stages ONE query's columns), and grids sized
            // for the lists that are usual (the workgroups stride over a list until it ends): 256 x 16 workgroups of 68 KiB each,
            // nearly all of which find nothing to do, cost more than the re-scoring itself
            int len_max = 1;
            for (int r = 0; r < pq_n; ++r) len_max = std::max(len_max, pq_len[r]);
            const size_t lds_list = (size_t)((len_max + 31) / 32) * 32 * (dp + 4) * sizeof(float);
            c.red_off = (int)lds_list;
            const int64_t want_a = k + 8, want_f = tighten ? 256 : n_cand_max;  // documents a launch should cover in ONE round
            const dim3 grid_a((unsigned)std::min<int64_t>({coop ? want_a : (want_a + 3) / 4, n_cand_max, (int64_t)kMsListGrid}), pq_n);
            const dim3 list_grid((unsigned)std::min<int64_t>({coop ? want_f : (want_f + 3) / 4, n_cand_max, (int64_t)kMsListGrid}), pq_n);
            if (idx->profile) HIPCHECK(idx, hipEventRecord(idx->ms_ev[2], s));
            const int32_t* list_f = list_c;
            const int* ctl_f = ctl_c;
            if (tighten) {
                c.doc_list = list_a;
                c.n_items_dev = ctl_a;
                hipLaunchKernelGGL(k_maxsim, grid_a, dim3(kMsThreads), lds_list + kMsRedBytes, s, c);
                HIPCHECK(idx, hipGetLastError());
                hipLaunchKernelGGL(k_ms_tighten, dim3(1, pq_n), dim3(256), 0, s, m->cand_dist, ctl_a, list_c, m->cand_sd, ctl_c,
                                   kMsCandCap, k, m->two_e_dev, list_b, ctl_b);
                HIPCHECK(idx, hipGetLastError());
                list_f = list_b;
                ctl_f = ctl_b;
            }
            c.doc_list = list_f;
            c.n_items_dev = ctl_f;
            hipLaunchKernelGGL(k_maxsim, list_grid, dim3(kMsThreads), lds_list + kMsRedBytes, s, c);
            HIPCHECK(idx, hipGetLastError());
            if (idx->profile) HIPCHECK(idx, hipEventRecord(idx->ms_ev[3], s));
            hipLaunchKernelGGL(k_ms_final, dim3(1, pq_n), dim3(256), (size_t)kMsCandCap * 12, s, m->cand_dist, list_f, ctl_f,
                               kMsCandCap, k, idx->row_offset, m->out_d, m->out_r);
            HIPCHECK(idx, hipGetLastError());
            HIPCHECK(idx, hipMemcpyAsync(m->cand_ctl_host, ctl_f, 2 * kPQ * sizeof(int), hipMemcpyDeviceToHost, s));
            if (!out_dev) {  // (hd also staged `te`: its H2D copy precedes these copies in stream order)
                HIPCHECK(idx, hipMemcpyAsync(hd, m->out_d, (size_t)pq_n * k * sizeof(float), hipMemcpyDeviceToHost, s));
                HIPCHECK(idx, hipMemcpyAsync(hr, m->out_r, (size_t)pq_n * k * sizeof(int64_t), hipMemcpyDeviceToHost, s));
            }
            HIPCHECK(idx, hipStreamSynchronize(s));
            if (idx->profile) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, idx->ms_ev[0], idx->ms_ev[1]) == hipSuccess) {
                    idx->s_ms_screen_ns += (int64_t)(ms * 1e6);
                    idx->s_ms_screen_launches++;
                }
                if (hipEventElapsedTime(&ms, idx->ms_ev[2], idx->ms_ev[3]) == hipSuccess) {
                    idx->s_ms_exact_ns += (int64_t)(ms * 1e6);
                    idx->s_ms_exact_launches++;
                }
            }
            for (int r = 0; r < pq_n; ++r) {
                if (m->cand_ctl_host[2 * r + 1] != 0) continue;  // list overflow: exact full scan below
                handled[r] = true;
                idx->s_ms_screened++;
                idx->s_ms_candidates += m->cand_ctl_host[2 * r];
                if (out_dev) {
                    HIPCHECK(idx, hipMemcpyAsync(out_dist + (int64_t)pq_b[r] * k, m->out_d + (size_t)r * k, k * sizeof(float),
                                                 hipMemcpyDeviceToDevice, s));
                    HIPCHECK(idx, hipMemcpyAsync(out_rows + (int64_t)pq_b[r] * k, m->out_r + (size_t)r * k,
                                                 k * sizeof(int64_t), hipMemcpyDeviceToDevice, s));
                } else {
                    memcpy(out_dist + (int64_t)pq_b[r] * k, &hd[(size_t)r * k], k * sizeof(float));
                    memcpy(out_rows + (int64_t)pq_b[r] * k, &hr[(size_t)r * k], k * sizeof(int64_t));
                }
            }
        } else if (idx->profile) {  // (the screen launch of a slow-path pass is timed too)
            HIPCHECK(idx, hipStreamSynchronize(s));
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, idx->ms_ev[0], idx->ms_ev[1]) == hipSuccess) {
                idx->s_ms_screen_ns += (int64_t)(ms * 1e6);
                idx->s_ms_screen_launches++;
            }
        }
        for (int r = 0; r < pq_n; ++r) {
            if (handled[r]) continue;
            float* od = out_dist + (int64_t)pq_b[r] * k;
            int64_t* orow = out_rows + (int64_t)pq_b[r] * k;
            if (k <= kMsFastK) {  // the fast path gave this query up (candidate list overflow): exact full scan
                idx->s_ms_fallbacks++;
                CHECK(full_scan_query(pq_col0[r], pq_len[r], od, orow));
                continue;
            }
            // k above the fast path's: screen top-k -> candidates -> exact kernel on the candidates -> exact top-k
            int cur = 0;
            const float* dist16 = m->dist16 + (int64_t)r * m->n_docs;
            CHECK(ms_topk(idx, m, s, dist16, m->n_docs, k, seg, nullptr, nullptr, &cur));
            HIPCHECK(idx, hipMemsetAsync(m->cand_ctl, 0, 2 * sizeof(int), s));
            float te = (float)pq_two_e[r];
            if ((double)te < pq_two_e[r]) te = std::nextafter(te, INFINITY);
            hipLaunchKernelGGL(k_ms_candidates, dim3((unsigned)((m->n_docs + 255) / 256)), dim3(256), 0, s, dist16, m->blk_off,
                               m->n_docs, m->pk[cur], k, te, m->cand_list, kMsCandCap, m->cand_ctl);
            HIPCHECK(idx, hipGetLastError());
            MsArgs c = a0;
            c.qtok = m->qtok + (int64_t)pq_col0[r] * dp;
            c.dist = m->cand_dist;
            c.doc_list = m->cand_list;
            c.n_items = n_cand_max;
            c.n_items_dev = m->cand_ctl;
            c.nq_launch = 1;
            c.q_col0[0] = 0;
            c.q_len[0] = pq_len[r];
            hipLaunchKernelGGL(k_maxsim, dim3((unsigned)std::min<int64_t>((n_cand_max + 3) / 4, kMsListGrid)),
                               dim3(kMsThreads), lds, s, c);
            HIPCHECK(idx, hipGetLastError());
            CHECK(ms_topk(idx, m, s, m->cand_dist, n_cand_max, k, seg, m->cand_list, m->cand_ctl, &cur));
            CHECK(ms_emit_result(idx, m, s, cur, k, od, orow, out_dev));
            HIPCHECK(idx, hipMemcpyAsync(m->cand_ctl_host, m->cand_ctl, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
            HIPCHECK(idx, hipStreamSynchronize(s));
            if (m->cand_ctl_host[1] == 0) {
                idx->s_ms_screened++;
                idx->s_ms_candidates += m->cand_ctl_host[0];
            } else {
                idx->s_ms_fallbacks++;  // more candidates than the list holds: this query takes the exact full scan
                CHECK(full_scan_query(pq_col0[r], pq_len[r], od, orow));
            }
        }
    }
    HIPCHECK(idx, hipStreamSynchronize(s));  // (device outputs: the last copies)
    return MI355DR_OK;
}

int mi355dr_search_maxsim(mi355dr_index* idx, const float* qtok, const int32_t* q_offsets, int B, int k,
                          float* out_dist, int64_t* out_rows) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (B < 0 || k <= 0 || !q_offsets || (B > 0 && (!out_dist || !out_rows)))
        return fail(idx, MI355DR_E_INVALID, "bad maxsim arguments");
    return search_maxsim_impl(idx, qtok, q_offsets, B, k, out_dist, out_rows, false);
}

int mi355dr_search_maxsim_device(mi355dr_index* idx, const float* qtok_dev, const int32_t* q_offsets, int B, int k,
                                 float* out_dist_dev, int64_t* out_rows_dev, void* stream) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (B < 0 || k <= 0 || !q_offsets || (B > 0 && (!out_dist_dev || !out_rows_dev)))
        return fail(idx, MI355DR_E_INVALID, "bad maxsim arguments");
    if (B == 0) return MI355DR_OK;
    for (int b = 0; b < B; ++b)
        if (q_offsets[b + 1] < q_offsets[b]) return fail(idx, MI355DR_E_INVALID, "q_offsets must be non-decreasing");
    const int64_t n_tok = q_offsets[B] - q_offsets[0];
    if (n_tok > 0 && !qtok_dev) return fail(idx, MI355DR_E_INVALID, "null query vectors");
    HIPCHECK(idx, hipSetDevice(idx->device));
    // The query side of a pass is tiny (8 queries x 32 vectors x 128 dims = 128 KiB) and its bound is evaluated in double on
    // the host: the vectors come down once (ordered behind the caller's stream), the k results of every query never leave HBM.
    if (stream) HIPCHECK(idx, hipStreamSynchronize((hipStream_t)stream));
    std::vector<float> qh((size_t)std::max<int64_t>(n_tok, 1) * idx->dim);
    if (n_tok > 0)
        HIPCHECK(idx, hipMemcpy(qh.data(), qtok_dev + (int64_t)q_offsets[0] * idx->dim, (size_t)n_tok * idx->dim * sizeof(float),
                                hipMemcpyDeviceToHost));
    std::vector<int32_t> off(B + 1);
    for (int b = 0; b <= B; ++b) off[b] = q_offsets[b] - q_offsets[0];
    return search_maxsim_impl(idx, qh.data(), off.data(), B, k, out_dist_dev, out_rows_dev, true);
}

static int maxsim_subset_impl(mi355dr_index* idx, const float* qtok, const int32_t* q_offsets, int B, const int64_t* doc_ids,
                              int m_ids, int clamp0, float* out_dist) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (B < 0 || m_ids < 0 || !q_offsets || (B > 0 && m_ids > 0 && (!doc_ids || !out_dist)))
        return fail(idx, MI355DR_E_INVALID, "bad maxsim_subset arguments");
    for (int64_t i = 0; i < (int64_t)B * m_ids; ++i) out_dist[i] = NAN;
    MultiVecStore* m = idx->mv;
    if (B == 0 || m_ids == 0 || !m || m->n_docs == 0) return MI355DR_OK;
    for (int b = 0; b < B; ++b)
        if (q_offsets[b + 1] - q_offsets[b] < 0) return fail(idx, MI355DR_E_INVALID, "q_offsets must be non-decreasing");
    HIPCHECK(idx, hipSetDevice(idx->device));
    hipStream_t s = idx->stream;
    const int dp = m->dpad, d = idx->dim;
    const int cols = ms_cols_for(dp);  // query vectors per launch; a longer query is scored in tiles (MsArgs::dist_in)
    if (cols < 32) return fail(idx, MI355DR_E_UNSUPPORTED, "dim too large for the MaxSim kernel's LDS budget (dim <= 1272)");
    const size_t lds = (size_t)cols * (dp + 4) * sizeof(float);
    HIPCHECK(idx, hipFuncSetAttribute((const void*)k_maxsim, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + kMsRedBytes)));
    // per call scratch (candidate lists are small: a few hundred docs per query), released on every exit
    struct Scratch {
        int32_t* list = nullptr;
        float* dist = nullptr;
        float* q = nullptr;
        ~Scratch() {
            if (list) (void)hipFree(list);
            if (dist) (void)hipFree(dist);
            if (q) (void)hipFree(q);
        }
    } sc;
    std::vector<int32_t> list((size_t)B * m_ids);
    for (int64_t i = 0; i < (int64_t)B * m_ids; ++i) {
        const int64_t v = doc_ids[i] - idx->row_offset;  // ids are global rows, like the search results
        list[i] = (v >= 0 && v < m->n_docs) ? (int32_t)v : -1;
    }
    HIPCHECK(idx, hipMalloc(&sc.list, list.size() * sizeof(int32_t)));
    HIPCHECK(idx, hipMalloc(&sc.dist, list.size() * sizeof(float)));
    HIPCHECK(idx, hipMalloc(&sc.q, (size_t)cols * dp * sizeof(float)));
    HIPCHECK(idx, hipMemcpyAsync(sc.list, list.data(), list.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    std::vector<float> qimg((size_t)cols * dp);
    for (int b = 0; b < B; ++b) {
        const int nq = q_offsets[b + 1] - q_offsets[b];
        if (nq == 0) continue;  // reference heaven.py:251-252: no query vectors -> every score 0 (host side)
        for (int t0 = 0; t0 < nq; t0 += cols) {  // tiles of the query's vectors: each launch continues the per-doc sums
            const int tl = std::min(cols, nq - t0);
            std::fill(qimg.begin(), qimg.end(), 0.0f);
            for (int j = 0; j < tl; ++j) {
                float* dst = &qimg[(size_t)j * dp];
                const float* sv = qtok + (int64_t)(q_offsets[b] + t0 + j) * d;
                for (int c = 0; c < dp; ++c) {
                    const int oc = ms_perm(c);
                    dst[c] = oc < d ? sv[oc] : 0.0f;
                }
            }
            // the staging buffer is reused: the previous launch must have consumed it (stream order + pageable copy)
            HIPCHECK(idx, hipMemcpyAsync(sc.q, qimg.data(), qimg.size() * sizeof(float), hipMemcpyHostToDevice, s));
            HIPCHECK(idx, hipStreamSynchronize(s));
            MsArgs a{};
            a.tok = m->tok;
            a.blk_off = m->blk_off;
            a.qtok = sc.q;
            a.dist = sc.dist + (int64_t)b * m_ids;
            a.dist_in = t0 > 0 ? a.dist : nullptr;
            a.doc_list = sc.list + (int64_t)b * m_ids;
            a.n_items = m_ids;
            a.n_docs = m->n_docs;
            a.dpad = dp;
            a.nq_launch = 1;
            a.q_col0[0] = 0;
            a.q_len[0] = tl;
            a.clamp0 = clamp0;
            hipLaunchKernelGGL(k_maxsim, dim3((unsigned)std::min<int64_t>((m_ids + 3) / 4, kMsListGrid)), dim3(kMsThreads),
                               lds, s, a);
            HIPCHECK(idx, hipGetLastError());
        }
        HIPCHECK(idx, hipMemcpyAsync(out_dist + (int64_t)b * m_ids, sc.dist + (int64_t)b * m_ids, m_ids * sizeof(float),
                                     hipMemcpyDeviceToHost, s));
    }
    HIPCHECK(idx, hipStreamSynchronize(s));
    return MI355DR_OK;
}

int mi355dr_maxsim_subset(mi355dr_index* idx, const float* qtok, const int32_t* q_offsets, int B, const int64_t* doc_ids,
                          int m_ids, float* out_dist) {
    return maxsim_subset_impl(idx, qtok, q_offsets, B, doc_ids, m_ids, 0, out_dist);
}

int mi355dr_maxsim_subset_ex(mi355dr_index* idx, const float* qtok, const int32_t* q_offsets, int B, const int64_t* doc_ids,
                             int m_ids, int flags, float* out_dist) {
    if (flags & ~MI355DR_MAXSIM_CLAMP0) return fail(idx, MI355DR_E_INVALID, "unknown maxsim flag");
    return maxsim_subset_impl(idx, qtok, q_offsets, B, doc_ids, m_ids, (flags & MI355DR_MAXSIM_CLAMP0) ? 1 : 0, out_dist);
}

}  // extern "C"
