// maxsim_common.h -- what the translation units of the multi-vector (MaxSim) path share (internal): pass geometry, the screen
// kernels' argument block, and the launch entry points of mi355dr_maxsim_screen.hip.
//   mi355dr_maxsim.hip         store (add_multivec[_device]), exact kernel k_maxsim, select / candidates / tighten / final, the C ABI
//   mi355dr_maxsim_screen.hip  the bf16 MFMA screens k_maxsim16 / k_maxsim16_d128 / k_maxsim16_wg (k_maxsim_wg.h) and their tables
// (Round 6: one 110 KB unit until then; the screen's ~70 template instantiations are what its 68 s of compile time were.)
#pragma once
#include <array>
#include <utility>

#include "index.h"

namespace mi355 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMsCols = 128;        // query-token columns per launch (4 column blocks of 32)
constexpr int kMsBlkRows = 32;
constexpr int kMsThreads = 256;     // 4 waves
constexpr int kMsDocsPerWave = 4;   // docs a wave walks per workgroup (amortises staging the query block in LDS)
constexpr int kSegSort = kSortMax;  // select: largest segment (entries sorted per workgroup)
constexpr int kMsListGrid = 256;    // workgroups of a doc-list launch of k_maxsim (4 waves each stride over the list)
constexpr int kMsRedBytes = 4 * 4 * 32 * 4;  // k_maxsim, cooperative list mode: [wave][column block][column] maxima
constexpr int kMsCandCap = 8192;    // docs the screen may hand to the exact kernel per query (more: exact full scan)
// One pass of the bf16 screen over the token store serves up to kMsPassGroups groups of <= 4 queries (dims <= 128): the pass is
// bound by the HBM stream of the fragment copy up to ~8 column blocks and by the matrix pipe beyond, so every further query
// that rides a pass costs MFMA time only -- 16 queries x 32 vectors = 16 column blocks = 128 KiB of query fragments in LDS.
constexpr int kMsPassGroups = 4;
constexpr int kMsPassQueries = 4 * kMsPassGroups;
constexpr int kMsPassBlocks = 4 * kMsPassGroups;  // column blocks of 32 query vectors

// ---- bf16 screen: same walk as k_maxsim, operands are 16-byte MFMA fragments (1 KiB per wave instruction, fully
// coalesced), v_mfma_f32_32x32x16_bf16, 8 fragments of the doc block in flight while the previous 8 are consumed ----
typedef __bf16 ms_bf16x8 __attribute__((ext_vector_type(8)));

struct Ms16Args {
    const uint4* tok16;
    const int64_t* blk_off;
    const uint4* qfrag;     // [column blocks][nkk][64]
    float* dist;            // [nq_launch, n_docs]
    int64_t n_docs;
    int nkk;
    int nq_launch;
    int q_col0[kMsPassQueries];  // (k_maxsim16_d128 serves up to FOUR groups of <= 4 queries per launch: rows 4 g .. 4 g + 3)
    int q_len[kMsPassQueries];
    int aligned;             // k_maxsim16_wg: query r of the launch is exactly column block r (q_col0[r] = 32 r, q_len[r] <= 32)
};

// the granule-packed bf16 copy (k_maxsim_wg8.h): documents rounded up to whole 8-token granules, the stream cut into 32-token blocks
struct Ms16Pack {
    const uint4* tok16p;   // [n_pblocks][nkk = 8][64] fragments, the padded copy's order inside a block
    const int64_t* goff;   // [n_docs + 1] first granule of each doc (device)
    int64_t n_gran;        // goff[n_docs]
    int64_t n_pblocks;     // ceil(n_gran / 4)
};

// ---- mi355dr_maxsim_screen.hip ----
// dynamic-LDS attributes of every screen kernel (lds16 = the generic form's query-fragment bytes; > 160 KiB: that form is not used)
int ms16_prepare(mi355dr_index* idx, size_t lds16);
// one screen launch over every doc, dims <= 128 (compile-time column-block count): one wave per document, or the
// workgroup-cooperative form from 8 / 9 column blocks up (options maxsim_wg*, maxsim_persistent)
// `pk` (may be null): the packed copy; taken when the pass is aligned and the workgroup form serves it (ms16_takes_wg)
int ms16_d128_launch(mi355dr_index* idx, hipStream_t s, int ncb, int64_t n_docs, int64_t n_blocks, bool persistent, const Ms16Args& sa,
                     const Ms16Pack* pk = nullptr);
bool ms16_takes_wg(const mi355dr_index* idx, int ncb, int64_t n_docs, int64_t n_blocks);
// ... dims > 128: the generic form (k_maxsim16)
int ms16_generic_launch(mi355dr_index* idx, hipStream_t s, unsigned grid, size_t lds16, const Ms16Args& sa);

}  // namespace mi355
