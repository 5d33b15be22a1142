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
#include "index.h"

using namespace mi355;

namespace mi355 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMsCols = 128;        // query-token columns per launch (4 column blocks of 32)
constexpr int kMsBlkRows = 32;
constexpr int kMsThreads = 256;     // 4 waves
constexpr int kMsDocsPerWave = 4;   // docs a wave walks per workgroup (amortises staging the query block in LDS)
constexpr int kSegSort = kSortMax;  // select: largest segment (entries sorted per workgroup)

struct MultiVecStore {
    int64_t n_docs = 0;
    int64_t n_blocks = 0, cap_blocks = 0;  // 32-row blocks stored / allocated
    int64_t cap_docs = 0;
    int dpad = 0;                  // dim rounded up to 8
    float* tok = nullptr;          // [cap_blocks*32, dpad]
    int64_t* blk_off = nullptr;    // [cap_docs+1] first block of each doc (device)
    std::vector<int64_t> blk_off_host;
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

void multivec_destroy(mi355dr_index* idx) {
    MultiVecStore* m = idx->mv;
    if (!m) return;
    void* ptrs[] = {m->tok, m->blk_off, m->qtok, m->dist, m->pk[0], m->pk[1], m->pr[0], m->pr[1], m->out_d, m->out_r};
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
    int64_t n_docs;
    int dpad;
    int nq_launch;          // queries in this launch (<= 4)
    int q_col0[4];          // first column of each query (multiple of 32)
    int q_len[4];           // real token count of each query
};

// Column order inside every group of 8 dims, for BOTH stored token rows and the staged query rows:
// position j holds original column kPerm[j] = {0,4,2,6,1,5,3,7}[j].  A lane of the lower half (k-slot 0 of the
// 32x32x2 MFMA) reads positions 0..3 = columns (0,4,2,6), a lane of the upper half positions 4..7 = (1,5,3,7),
// so issuing the MFMAs on components x, z, y, w walks k = (0|1), (2|3), (4|5), (6|7): ascending, no lane swaps.
__host__ __device__ inline int ms_perm(int j) {
    constexpr int P[8] = {0, 4, 2, 6, 1, 5, 3, 7};
    return (j & ~7) | P[j & 7];
}

__device__ __forceinline__ void ms_load_piece(float4 (&a)[16], const float* row, int chunk, int dpad) {
    // this lane's 4 floats of every 8-dim group of dims [128*chunk, +128); groups past dpad read as zero
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k0 = chunk * 128 + i * 8;
        a[i] = k0 < dpad ? *(const float4*)(row + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
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
__global__ __launch_bounds__(kMsThreads, 2) void k_maxsim(MsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* qs = (float*)smem;
    const int ld = a.dpad + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kMsCols * (a.dpad / 4); i += kMsThreads) {
        const int c = i / (a.dpad / 4), k4 = i - c * (a.dpad / 4);
        *(float4*)(qs + c * ld + k4 * 4) = *(const float4*)(a.qtok + (int64_t)c * a.dpad + k4 * 4);
    }
    __syncthreads();
    const int half = lane >> 5, col = lane & 31;
    int ncb = 0;   // column blocks in use
    for (int qi = 0; qi < a.nq_launch; ++qi) ncb = max(ncb, (a.q_col0[qi] + a.q_len[qi] + 31) / 32);
    const int nchunk = (a.dpad + 127) / 128;
    // docs are dealt round-robin to the waves of the grid so long and short docs mix
    for (int dw = 0; dw < kMsDocsPerWave; ++dw) {
    const int64_t item = ((int64_t)dw * gridDim.x + blockIdx.x) * 4 + wave;
    if (item >= a.n_items) break;
    const int64_t doc = a.doc_list ? (int64_t)a.doc_list[item] : item;
    if (doc < 0 || doc >= a.n_docs) {  // (subset scoring) not a stored doc
        if (lane == 0)
            for (int qi = 0; qi < a.nq_launch; ++qi) a.dist[(int64_t)qi * a.n_items + item] = __uint_as_float(0x7FC00000u);
        continue;
    }
    const int64_t b0 = a.blk_off[doc], b1 = a.blk_off[doc + 1];
    float run[4];  // running max per column block (this lane's column)
#pragma unroll
    for (int c = 0; c < 4; ++c) run[c] = -__builtin_inff();

    const int64_t npieces = (b1 - b0) * nchunk;
    f32x16 acc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
    float4 pa[16], pb[16];
    auto row_of = [&](int64_t p) { return a.tok + ((b0 + p / nchunk) * kMsBlkRows + col) * (int64_t)a.dpad + 4 * half; };
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
    // per query: distance = sum over its tokens (in order) of -(max dot); empty docs are skipped by the select
    for (int qi = 0; qi < a.nq_launch; ++qi) {
        float accd = 0.0f;
        for (int j = 0; j < a.q_len[qi]; ++j) {
            const int c = a.q_col0[qi] + j;
            float v = 0.0f;
#pragma unroll
            for (int cbi = 0; cbi < 4; ++cbi)
                if (cbi == (c >> 5)) v = run[cbi];
            v = __shfl(v, c & 31, kWave);
            accd = accd + (-v);
        }
        if (lane == 0) a.dist[(int64_t)qi * a.n_items + item] = b1 > b0 ? accd : __uint_as_float(0x7FC00000u);
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
__global__ __launch_bounds__(256) void k_topk_segments(const float* dist, const int64_t* blk_off,
                                                        const uint64_t* key_in, const int32_t* row_in, int64_t n_in,
                                                        int k, int seg, uint64_t* key_out, int32_t* row_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* SK = (uint64_t*)smem;
    int32_t* SR = (int32_t*)(smem + (size_t)kSegSort * 8);
    const int64_t base = (int64_t)blockIdx.x * seg;
    for (int i = threadIdx.x; i < seg; i += blockDim.x) {
        const int64_t g = base + i;
        uint64_t key = kKeyNaN;
        int32_t row = 0x7FFFFFFF;
        if (g < n_in) {
            if (dist) {
                if (blk_off[g + 1] > blk_off[g]) {  // docs without vectors are not rows of the result
                    key = f32_to_key(dist[g]);
                    row = (int32_t)g;
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
        idx->mv->blk_off_host.push_back(0);
    }
    MultiVecStore* m = idx->mv;
    if (m->n_docs + n_docs >= ((int64_t)1 << 31)) return fail(idx, MI355DR_E_UNSUPPORTED, "too many docs");
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
    CHECK(ms_reserve(idx, m, m->n_blocks + new_blocks, m->n_docs + n_docs));
    if (new_blocks > 0)
        HIPCHECK(idx, hipMemcpy(m->tok + (size_t)m->n_blocks * kMsBlkRows * dp, img.data(), img.size() * sizeof(float),
                                hipMemcpyHostToDevice));
    HIPCHECK(idx, hipMemcpy(m->blk_off, m->blk_off_host.data(), m->blk_off_host.size() * sizeof(int64_t),
                            hipMemcpyHostToDevice));
    m->n_blocks += new_blocks;
    m->n_docs += n_docs;
    return MI355DR_OK;
}

int64_t mi355dr_size_multivec(const mi355dr_index* idx) { return idx && idx->mv ? idx->mv->n_docs : 0; }

int mi355dr_search_maxsim(mi355dr_index* idx, const float* qtok, const int32_t* q_offsets, int B, int k,
                          float* out_dist, int64_t* out_rows) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (B < 0 || k <= 0 || !q_offsets || (B > 0 && (!out_dist || !out_rows)))
        return fail(idx, MI355DR_E_INVALID, "bad maxsim arguments");
    if (k > kKMax) return fail(idx, MI355DR_E_UNSUPPORTED, "k exceeds 1024");
    for (int64_t i = 0; i < (int64_t)B * k; ++i) {
        out_dist[i] = NAN;
        out_rows[i] = -1;
    }
    MultiVecStore* m = idx->mv;
    if (B == 0 || !m || m->n_docs == 0) return MI355DR_OK;
    for (int b = 0; b < B; ++b) {
        const int nq = q_offsets[b + 1] - q_offsets[b];
        if (nq < 0) return fail(idx, MI355DR_E_INVALID, "q_offsets must be non-decreasing");
        if (nq > kMsCols) return fail(idx, MI355DR_E_UNSUPPORTED, "more than 128 query vectors per query");
    }
    HIPCHECK(idx, hipSetDevice(idx->device));
    hipStream_t s = idx->stream;
    const int dp = m->dpad, d = idx->dim;
    const size_t lds = (size_t)kMsCols * (dp + 4) * sizeof(float);
    if (lds > 160 * 1024) return fail(idx, MI355DR_E_UNSUPPORTED, "dim too large for the MaxSim kernel's LDS budget");
    // scratch
    if (!m->qtok) {
        HIPCHECK(idx, hipMalloc(&m->qtok, (size_t)kMsCols * dp * sizeof(float)));
        HIPCHECK(idx, hipMalloc(&m->out_d, kKMax * sizeof(float)));
        HIPCHECK(idx, hipMalloc(&m->out_r, kKMax * sizeof(int64_t)));
        HIPCHECK(idx, hipFuncSetAttribute((const void*)k_maxsim, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHECK(idx, hipFuncSetAttribute((const void*)k_topk_segments, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          kSegSort * 12));
    }
    if (m->dist_cap_docs < m->n_docs) {
        if (m->dist) (void)hipFree(m->dist);
        HIPCHECK(idx, hipMalloc(&m->dist, (size_t)4 * m->cap_docs * sizeof(float)));
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
    std::vector<float> qimg((size_t)kMsCols * dp);
    int b = 0;
    while (b < B) {
        // pack queries into one launch while their 32-padded token counts fit 128 columns (max 4 queries)
        MsArgs a{};
        a.tok = m->tok;
        a.blk_off = m->blk_off;
        a.qtok = m->qtok;
        a.dist = m->dist;
        a.n_docs = m->n_docs;
        a.n_items = m->n_docs;
        a.doc_list = nullptr;
        a.dpad = dp;
        std::fill(qimg.begin(), qimg.end(), 0.0f);
        int col = 0, nql = 0, first = b;
        while (b < B && nql < 4) {
            const int nq = q_offsets[b + 1] - q_offsets[b];
            const int need = (int)round_up(std::max(nq, 1), 32);
            if (col + need > kMsCols) break;
            a.q_col0[nql] = col;
            a.q_len[nql] = nq;
            for (int j = 0; j < nq; ++j) {
                float* dst = &qimg[(size_t)(col + j) * dp];
                const float* sv = qtok + (int64_t)(q_offsets[b] + j) * d;
                for (int c = 0; c < dp; ++c) {
                    const int oc = ms_perm(c);
                    dst[c] = oc < d ? sv[oc] : 0.0f;
                }
            }
            col += need;
            ++nql;
            ++b;
        }
        a.nq_launch = nql;
        HIPCHECK(idx, hipMemcpyAsync(m->qtok, qimg.data(), qimg.size() * sizeof(float), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_maxsim, dim3((unsigned)((m->n_docs + 4 * kMsDocsPerWave - 1) / (4 * kMsDocsPerWave))),
                           dim3(kMsThreads), lds, s, a);
        HIPCHECK(idx, hipGetLastError());
        for (int qi = 0; qi < nql; ++qi) {
            if (a.q_len[qi] == 0) continue;  // reference: `if not query_vectors: return []`
            // segment-wise top-k until one segment is left
            int64_t n_in = m->n_docs;
            int cur = 0;
            bool first_stage = true;
            while (true) {
                const int64_t nseg = (n_in + seg - 1) / seg;
                hipLaunchKernelGGL(k_topk_segments, dim3((unsigned)nseg), dim3(256), (size_t)kSegSort * 12, s,
                                   first_stage ? m->dist + (int64_t)qi * m->n_docs : nullptr, m->blk_off,
                                   first_stage ? nullptr : m->pk[cur ^ 1], first_stage ? nullptr : m->pr[cur ^ 1], n_in,
                                   k, seg, m->pk[cur], m->pr[cur]);
                HIPCHECK(idx, hipGetLastError());
                first_stage = false;
                if (nseg == 1) break;
                n_in = nseg * k;
                cur ^= 1;
            }
            hipLaunchKernelGGL(k_ms_write_out, dim3((k + 255) / 256), dim3(256), 0, s, m->pk[cur], m->pr[cur], k, idx->row_offset,
                               m->out_d, m->out_r);
            HIPCHECK(idx, hipGetLastError());
            HIPCHECK(idx, hipMemcpyAsync(out_dist + (int64_t)(first + qi) * k, m->out_d, k * sizeof(float),
                                         hipMemcpyDeviceToHost, s));
            HIPCHECK(idx, hipMemcpyAsync(out_rows + (int64_t)(first + qi) * k, m->out_r, k * sizeof(int64_t),
                                         hipMemcpyDeviceToHost, s));
            HIPCHECK(idx, hipStreamSynchronize(s));
        }
    }
    return MI355DR_OK;
}

int mi355dr_maxsim_subset(mi355dr_index* idx, const float* qtok, const int32_t* q_offsets, int B, const int64_t* doc_ids,
                          int m_ids, float* out_dist) {
    if (!idx) return fail(nullptr, MI355DR_E_INVALID, "null index");
    std::lock_guard<std::mutex> g(idx->mu);
    if (B < 0 || m_ids < 0 || !q_offsets || (B > 0 && m_ids > 0 && (!doc_ids || !out_dist)))
        return fail(idx, MI355DR_E_INVALID, "bad maxsim_subset arguments");
    for (int64_t i = 0; i < (int64_t)B * m_ids; ++i) out_dist[i] = NAN;
    MultiVecStore* m = idx->mv;
    if (B == 0 || m_ids == 0 || !m || m->n_docs == 0) return MI355DR_OK;
    for (int b = 0; b < B; ++b) {
        const int nq = q_offsets[b + 1] - q_offsets[b];
        if (nq < 0) return fail(idx, MI355DR_E_INVALID, "q_offsets must be non-decreasing");
        if (nq > kMsCols) return fail(idx, MI355DR_E_UNSUPPORTED, "more than 128 query vectors per query");
    }
    HIPCHECK(idx, hipSetDevice(idx->device));
    hipStream_t s = idx->stream;
    const int dp = m->dpad, d = idx->dim;
    const size_t lds = (size_t)kMsCols * (dp + 4) * sizeof(float);
    if (lds > 160 * 1024) return fail(idx, MI355DR_E_UNSUPPORTED, "dim too large for the MaxSim kernel's LDS budget");
    HIPCHECK(idx, hipFuncSetAttribute((const void*)k_maxsim, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // per call scratch (candidate lists are small: a few hundred docs per query)
    int32_t* list_dev = nullptr;
    float* dist_dev = nullptr;
    float* q_dev = nullptr;
    std::vector<int32_t> list((size_t)B * m_ids);
    for (int64_t i = 0; i < (int64_t)B * m_ids; ++i) {
        const int64_t v = doc_ids[i] - idx->row_offset;  // ids are global rows, like the search results
        list[i] = (v >= 0 && v < m->n_docs) ? (int32_t)v : -1;
    }
    HIPCHECK(idx, hipMalloc(&list_dev, list.size() * sizeof(int32_t)));
    HIPCHECK(idx, hipMalloc(&dist_dev, list.size() * sizeof(float)));
    HIPCHECK(idx, hipMalloc(&q_dev, (size_t)kMsCols * dp * sizeof(float)));
    HIPCHECK(idx, hipMemcpyAsync(list_dev, list.data(), list.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    std::vector<float> qimg((size_t)kMsCols * dp);
    for (int b = 0; b < B; ++b) {
        const int nq = q_offsets[b + 1] - q_offsets[b];
        if (nq == 0) continue;  // reference heaven.py:251-252: no query vectors -> every score 0 (host side)
        std::fill(qimg.begin(), qimg.end(), 0.0f);
        for (int j = 0; j < nq; ++j) {
            float* dst = &qimg[(size_t)j * dp];
            const float* sv = qtok + (int64_t)(q_offsets[b] + j) * d;
            for (int c = 0; c < dp; ++c) {
                const int oc = ms_perm(c);
                dst[c] = oc < d ? sv[oc] : 0.0f;
            }
        }
        // the staging buffer is reused: the previous launch must have consumed it (stream order + pageable copy)
        HIPCHECK(idx, hipMemcpyAsync(q_dev, qimg.data(), qimg.size() * sizeof(float), hipMemcpyHostToDevice, s));
        HIPCHECK(idx, hipStreamSynchronize(s));
        MsArgs a{};
        a.tok = m->tok;
        a.blk_off = m->blk_off;
        a.qtok = q_dev;
        a.dist = dist_dev + (int64_t)b * m_ids;
        a.doc_list = list_dev + (int64_t)b * m_ids;
        a.n_items = m_ids;
        a.n_docs = m->n_docs;
        a.dpad = dp;
        a.nq_launch = 1;
        a.q_col0[0] = 0;
        a.q_len[0] = nq;
        hipLaunchKernelGGL(k_maxsim, dim3((unsigned)((m_ids + 4 * kMsDocsPerWave - 1) / (4 * kMsDocsPerWave))),
                           dim3(kMsThreads), lds, s, a);
        HIPCHECK(idx, hipGetLastError());
        HIPCHECK(idx, hipMemcpyAsync(out_dist + (int64_t)b * m_ids, dist_dev + (int64_t)b * m_ids, m_ids * sizeof(float),
                                     hipMemcpyDeviceToHost, s));
    }
    HIPCHECK(idx, hipStreamSynchronize(s));
    (void)hipFree(list_dev);
    (void)hipFree(dist_dev);
    (void)hipFree(q_dev);
    return MI355DR_OK;
}

}  // extern "C"
