// mi355dr_comm.hip -- row-sharded search inside the C ABI: RCCL communicator per index, one packed all-gather of the
// per-shard [B,k] lists over xGMI and the world*k -> k merge, all on the caller's stream.
//
// RCCL is bound at run time (dlopen + dlsym), not at link time:
//   * a host that never shards does not need librccl at all (the library keeps loading without it);
//   * a host that already carries an RCCL (PyTorch-ROCm bundles its own librccl.so with the same soname) must not get a
//     second copy: RTLD_NOLOAD finds the one already mapped, only then the default search path is tried.
// The ncclUniqueId travels through the HOST's own channel (MPI, a torch store, a file): rank 0 calls
// mi355dr_comm_unique_id, every rank calls mi355dr_comm_init with the same 128 bytes.
#include <dlfcn.h>

#include "index.h"

namespace {

// the few RCCL entry points used, with the types of rccl.h (ncclResult_t = int, ncclDataType_t ncclInt64 = 4)
struct NcclUniqueId {
    char internal[128];
};
typedef int (*fn_get_unique_id)(NcclUniqueId*);
typedef int (*fn_comm_init_rank)(void** comm, int nranks, NcclUniqueId id, int rank);
typedef int (*fn_comm_destroy)(void* comm);
typedef int (*fn_all_gather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t s);
typedef const char* (*fn_get_error_string)(int);
typedef int (*fn_comm_count)(const void* comm, int* count);
constexpr int kNcclInt64 = 4;

struct RcclApi {
    void* handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_get_error_string error_string = nullptr;
    fn_comm_count comm_count = nullptr;
    std::string why;
};

RcclApi& rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names)
            if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);  // a copy the process already has
        for (const char* n : names)
            if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!api.handle) api.handle = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!api.handle) {
            const char* e = dlerror();
            api.why = std::string("librccl not found: ") + (e ? e : "?");
            return;
        }
        api.get_unique_id = (fn_get_unique_id)dlsym(api.handle, "ncclGetUniqueId");
        api.comm_init_rank = (fn_comm_init_rank)dlsym(api.handle, "ncclCommInitRank");
        api.comm_destroy = (fn_comm_destroy)dlsym(api.handle, "ncclCommDestroy");
        api.all_gather = (fn_all_gather)dlsym(api.handle, "ncclAllGather");
        api.error_string = (fn_get_error_string)dlsym(api.handle, "ncclGetErrorString");
        api.comm_count = (fn_comm_count)dlsym(api.handle, "ncclCommCount");
        if (!api.get_unique_id || !api.comm_init_rank || !api.comm_destroy || !api.all_gather) api.why = "librccl lacks an entry point";
    });
    return api;
}

int nccl_fail(mi355dr_index* idx, const char* what, int rc) {
    RcclApi& r = rccl();
    return mi355::fail(idx, MI355DR_E_HIP, std::string(what) + " failed: " + (r.error_string ? r.error_string(rc) : "rccl error"));
}

}  // namespace

namespace mi355 {
void comm_destroy(mi355dr_index* idx) {
    if (idx->comm) {
        RcclApi& r = rccl();
        if (r.comm_destroy) (void)r.comm_destroy(idx->comm);
        idx->comm = nullptr;
    }
    idx->comm_custom = nullptr;
    idx->comm_custom_user = nullptr;
    for (int i = 0; i < 2; ++i) {
        if (idx->comm_packed[i]) (void)hipFree(idx->comm_packed[i]);
        if (idx->comm_packed_all[i]) (void)hipFree(idx->comm_packed_all[i]);
        idx->comm_packed[i] = idx->comm_packed_all[i] = nullptr;
        if (idx->comm_done[i]) (void)hipEventDestroy(idx->comm_done[i]);
        idx->comm_done[i] = nullptr;
        idx->comm_done_armed[i] = false;
    }
    if (idx->comm_stream) (void)hipStreamDestroy(idx->comm_stream);
    idx->comm_stream = nullptr;
    idx->comm_cap = 0;
}
}  // namespace mi355

extern "C" {

int mi355dr_comm_unique_id(void* out, size_t len) {
    if (!out || len < sizeof(NcclUniqueId)) return mi355::fail(nullptr, MI355DR_E_INVALID, "unique id buffer must hold 128 bytes");
    RcclApi& r = rccl();
    if (!r.why.empty()) return mi355::fail(nullptr, MI355DR_E_UNSUPPORTED, r.why);
    NcclUniqueId id;
    const int rc = r.get_unique_id(&id);
    if (rc != 0) return nccl_fail(nullptr, "ncclGetUniqueId", rc);
    memcpy(out, &id, sizeof(id));
    return MI355DR_OK;
}

int mi355dr_comm_init(mi355dr_index* idx, int rank, int world, const void* nccl_unique_id, size_t id_len) {
    if (!idx) return mi355::fail(nullptr, MI355DR_E_INVALID, "null index");
    if (world < 1 || rank < 0 || rank >= world) return mi355::fail(idx, MI355DR_E_INVALID, "need 0 <= rank < world");
    if (!nccl_unique_id || id_len < sizeof(NcclUniqueId)) return mi355::fail(idx, MI355DR_E_INVALID, "unique id must be 128 bytes");
    RcclApi& r = rccl();
    if (!r.why.empty()) return mi355::fail(idx, MI355DR_E_UNSUPPORTED, r.why);
    std::lock_guard<std::mutex> g(idx->mu);
    HIPCHECK(idx, hipSetDevice(idx->device));
    mi355::comm_destroy(idx);
    NcclUniqueId id;
    memcpy(&id, nccl_unique_id, sizeof(id));
    void* comm = nullptr;
    const int rc = r.comm_init_rank(&comm, world, id, rank);
    if (rc != 0) return nccl_fail(idx, "ncclCommInitRank", rc);
    idx->comm = comm;
    idx->comm_rank = rank;
    idx->comm_world = world;
    return MI355DR_OK;
}

int mi355dr_comm_init_custom(mi355dr_index* idx, int rank, int world, mi355dr_allgather_fn fn, void* user) {
    if (!idx) return mi355::fail(nullptr, MI355DR_E_INVALID, "null index");
    if (world < 1 || rank < 0 || rank >= world) return mi355::fail(idx, MI355DR_E_INVALID, "need 0 <= rank < world");
    if (!fn) return mi355::fail(idx, MI355DR_E_INVALID, "all-gather function is null");
    std::lock_guard<std::mutex> g(idx->mu);
    HIPCHECK(idx, hipSetDevice(idx->device));
    mi355::comm_destroy(idx);
    idx->comm_custom = fn;
    idx->comm_custom_user = user;
    idx->comm_rank = rank;
    idx->comm_world = world;
    return MI355DR_OK;
}

int mi355dr_comm_world(const mi355dr_index* idx) { return idx && (idx->comm || idx->comm_custom) ? idx->comm_world : 0; }

int mi355dr_comm_count(mi355dr_index* idx, int* out) {
    if (!idx || !out) return mi355::fail(idx, MI355DR_E_INVALID, "null argument");
    *out = 0;
    if (idx->comm_custom) {
        *out = idx->comm_world;
        return MI355DR_OK;
    }
    if (!idx->comm) return MI355DR_OK;
    RcclApi& r = rccl();
    if (!r.comm_count) return mi355::fail(idx, MI355DR_E_UNSUPPORTED, "librccl lacks ncclCommCount");
    int n = 0;
    const int rc = r.comm_count(idx->comm, &n);
    if (rc != 0) return nccl_fail(idx, "ncclCommCount", rc);
    *out = n;
    return MI355DR_OK;
}

// Two packed blocks, two gathered blocks, a second stream: block i + 1 is put on the caller's stream BEFORE the host waits for
// block i (mi355dr_search_device_async / _wait), and block i's all-gather + merge run on the index's communication stream under
// block i + 1's search.  A block's packed buffer is reused two blocks later: the search that writes it waits (on the stream) for
// the gather that read it.  On return the caller's stream has been made to wait for the last merges, so the outputs are ordered
// behind `stream` exactly as with the serial form.
int mi355dr_search_sharded_device(mi355dr_index* idx, const float* queries_dev, int B, int k, double* out_dist_dev,
                                  int64_t* out_rows_dev, void* stream) {
    if (!idx) return mi355::fail(nullptr, MI355DR_E_INVALID, "null index");
    if (!idx->comm && !idx->comm_custom)
        return mi355::fail(idx, MI355DR_E_INVALID, "mi355dr_comm_init has not been called on this index");
    if (B < 0 || k <= 0 || k > mi355::kKMax) return mi355::fail(idx, MI355DR_E_INVALID, "bad B / k");
    if ((int64_t)idx->comm_world * k > mi355::kSortMax) return mi355::fail(idx, MI355DR_E_UNSUPPORTED, "world*k exceeds 4096");
    if (B == 0) return MI355DR_OK;
    if (!queries_dev || !out_dist_dev || !out_rows_dev) return mi355::fail(idx, MI355DR_E_INVALID, "null buffer");
    RcclApi& r = rccl();
    HIPCHECK(idx, hipSetDevice(idx->device));
    hipStream_t s = stream ? (hipStream_t)stream : idx->stream;
    const int world = idx->comm_world;
    {   // staging: per buffer one packed [2, nb, k] int64 block per rank (plane 0 = float8 distance bits, plane 1 = global rows)
        std::lock_guard<std::mutex> g(idx->mu);
        const size_t need = (size_t)2 * mi355::kQBlockMax * k;
        if (idx->comm_cap < need) {
            HIPCHECK(idx, hipDeviceSynchronize());  // (a previous call's gathers may still read the old buffers)
            for (int i = 0; i < 2; ++i) {
                if (idx->comm_packed[i]) (void)hipFree(idx->comm_packed[i]);
                if (idx->comm_packed_all[i]) (void)hipFree(idx->comm_packed_all[i]);
                idx->comm_packed[i] = idx->comm_packed_all[i] = nullptr;
                idx->comm_done_armed[i] = false;
            }
            idx->comm_cap = 0;
            for (int i = 0; i < 2; ++i) {
                HIPCHECK(idx, hipMalloc(&idx->comm_packed[i], need * sizeof(int64_t)));
                HIPCHECK(idx, hipMalloc(&idx->comm_packed_all[i], need * sizeof(int64_t) * world));
            }
            idx->comm_cap = need;
        }
        if (!idx->comm_stream) HIPCHECK(idx, hipStreamCreateWithFlags(&idx->comm_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i)
            if (!idx->comm_done[i]) HIPCHECK(idx, hipEventCreateWithFlags(&idx->comm_done[i], hipEventDisableTiming));
    }
    struct InFlight {
        int64_t ticket = -1;
        int buf = 0, b0 = 0, nb = 0;
    } pend;
    auto finish = [&](const InFlight& p) -> int {
        // complete on the host (the library re-does the rare flagged queries here): the packed block is final and visible
        CHECK(mi355dr_search_wait(idx, p.ticket));
        const size_t plane = (size_t)p.nb * k;
        if (idx->comm_custom) {   // the host's transport (ordered on the communication stream, or complete on return)
            const int rc = idx->comm_custom(idx->comm_packed[p.buf], idx->comm_packed_all[p.buf], 2 * plane * sizeof(int64_t),
                                            (void*)idx->comm_stream, idx->comm_custom_user);
            if (rc != 0) return mi355::fail(idx, MI355DR_E_HIP, "the host's all-gather failed with code " + std::to_string(rc));
        } else {
            const int rc = r.all_gather(idx->comm_packed[p.buf], idx->comm_packed_all[p.buf], 2 * plane, kNcclInt64, idx->comm,
                                        idx->comm_stream);
            if (rc != 0) return nccl_fail(idx, "ncclAllGather", rc);
        }
        CHECK(mi355dr_merge_topk_packed_device(idx, idx->comm_packed_all[p.buf], world, p.nb, k, out_dist_dev + (int64_t)p.b0 * k,
                                               out_rows_dev + (int64_t)p.b0 * k, idx->comm_stream));
        HIPCHECK(idx, hipEventRecord(idx->comm_done[p.buf], idx->comm_stream));
        idx->comm_done_armed[p.buf] = true;
        return MI355DR_OK;
    };
    // An error anywhere below must not leave work behind the caller's back: every ticket still open is waited for, and `s` is
    // ordered behind whatever the communication stream was already given (merges of earlier blocks write the caller's
    // outputs) before the error code is returned -- the outputs are then garbage, but nothing writes them after the return.
    auto drain = [&](int64_t t_a, int64_t t_b) {
        if (t_a >= 0) (void)mi355dr_search_wait(idx, t_a);
        if (t_b >= 0) (void)mi355dr_search_wait(idx, t_b);
        if (idx->comm_stream) (void)hipStreamSynchronize(idx->comm_stream);
        for (int b = 0; b < 2; ++b) idx->comm_done_armed[b] = false;   // (everything they stood for has completed)
    };
#define SHARDED_TRY(expr, TA, TB)        \
    do {                                 \
        const int rc__ = (expr);         \
        if (rc__ != MI355DR_OK) {        \
            drain((TA), (TB));           \
            return rc__;                 \
        }                                \
    } while (0)
    int i = 0;
    for (int b0 = 0; b0 < B; b0 += mi355::kQBlockMax, ++i) {
        const int nb = std::min(mi355::kQBlockMax, B - b0), buf = i & 1;
        const size_t plane = (size_t)nb * k;
        // the gather that read this packed block two blocks (or one call) ago
        if (idx->comm_done_armed[buf] && hipStreamWaitEvent(s, idx->comm_done[buf], 0) != hipSuccess) {
            drain(pend.ticket, -1);
            return mi355::fail(idx, MI355DR_E_HIP, "hipStreamWaitEvent failed in the sharded search");
        }
        InFlight cur;
        cur.buf = buf;
        cur.b0 = b0;
        cur.nb = nb;
        // the shard's list goes straight into the packed block (the float8 plane is written as doubles)
        SHARDED_TRY(mi355dr_search_device_async(idx, queries_dev + (int64_t)b0 * idx->dim, nb, k, (double*)idx->comm_packed[buf],
                                                idx->comm_packed[buf] + plane, s, &cur.ticket), pend.ticket, -1);
        if (pend.ticket >= 0) {
            const int64_t t = pend.ticket;
            pend.ticket = -1;   // finish() waits for it first thing: whatever happens inside, it is no longer open
            (void)t;
            InFlight done = pend;
            done.ticket = t;
            SHARDED_TRY(finish(done), cur.ticket, -1);
        }
        pend = cur;
    }
    if (pend.ticket >= 0) SHARDED_TRY(finish(pend), -1, -1);
#undef SHARDED_TRY
    // outputs ordered behind the caller's stream
    for (int b = 0; b < 2; ++b)
        if (idx->comm_done_armed[b] && hipStreamWaitEvent(s, idx->comm_done[b], 0) != hipSuccess) {
            drain(-1, -1);
            return mi355::fail(idx, MI355DR_E_HIP, "hipStreamWaitEvent failed in the sharded search");
        }
    return MI355DR_OK;  // asynchronous on `s`: the last merges run on the communication stream, `s` waits for them
}

}  // extern "C"
