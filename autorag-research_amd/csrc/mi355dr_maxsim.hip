// mi355dr_maxsim.hip -- multi-vector (late interaction, VectorChord `@#`) store and search.
#include "index.h"

using namespace mi355;

namespace mi355 {
struct MultiVecStore {
    int64_t n_docs = 0;
};
void multivec_destroy(mi355dr_index* idx) {
    delete idx->mv;
    idx->mv = nullptr;
}
}  // namespace mi355

extern "C" {

int mi355dr_add_multivec(mi355dr_index* idx, const float*, const int64_t*, int64_t) {
    return fail(idx, MI355DR_E_UNSUPPORTED, "multi-vector store not built yet");
}
int64_t mi355dr_size_multivec(const mi355dr_index* idx) { return idx && idx->mv ? idx->mv->n_docs : 0; }
int mi355dr_search_maxsim(mi355dr_index* idx, const float*, const int32_t*, int, int, float*, int64_t*) {
    return fail(idx, MI355DR_E_UNSUPPORTED, "multi-vector search not built yet");
}

}  // extern "C"
