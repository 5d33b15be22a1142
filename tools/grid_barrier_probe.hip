// grid_barrier_probe.hip -- what ONE persistent launch per pass would pay for its rendezvous.
//
// Fusing the screen launches and the prunes of a pass into one persistent kernel replaces every dependent-launch boundary by
// a grid-wide barrier with agent-scope release / acquire (the prune's candidate lists and the thresholds it publishes cross
// XCDs: per-XCD L2s are not coherent).  This probe measures that barrier in the screen kernel's own geometry -- 256 workgroups
// of 512 threads, one per CU (159 KiB of LDS each) -- in two forms: one device-scope counter, and the XCD-hierarchical form
// (per-XCD counter, the XCD's last arriver goes to the top counter).  Every workgroup dirties a little global memory before the
// barrier (a prune publishes 4 thresholds and a few hundred bytes of kept lists per workgroup) and reads another workgroup's
// words after it, checking them: a stale read fails the run.  Next to it: the same number of dependent LAUNCHES of a trivial
// kernel of the same geometry, i.e. what the barrier replaces.
//
// build: hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_probe.hip -o /tmp/grid_barrier_probe ; run: /tmp/grid_barrier_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e__ = (x);                                                                   \
        if (e__ != hipSuccess) {                                                                \
            fprintf(stderr, "%s failed: %s (%s:%d)\n", #x, hipGetErrorString(e__), __FILE__, __LINE__); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

constexpr int kWG = 256, kThreads = 512, kLds = 159 * 1024;
constexpr unsigned kSpinLimit = 1u << 22;  // every spin is bounded: a lost arrival ends the kernel with the timeout word set

struct Bar {
    unsigned* top;      // [1]   monotonic arrivals (flat form) / XCD arrivals (hierarchical form)
    unsigned* xcd;      // [8]   per-XCD arrivals
    unsigned* gen;      // [8]   per-XCD generation, written by the XCD's leader
    unsigned* timeout;  // [1]
};

__device__ __forceinline__ unsigned ld_relaxed(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one counter: every workgroup's lane 0 releases, arrives, polls, acquires
__device__ __forceinline__ bool barrier_flat(const Bar& b, unsigned epoch) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (ld_relaxed(b.top) < epoch * gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpinLimit) {
                atomicExch(b.timeout, 1u);
                ok = false;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

// XCD-hierarchical: arrive on the XCD's counter; its last arriver releases, arrives on the top counter, waits for the 8 XCDs,
// acquires and bumps the XCD's generation; the others poll their XCD's generation and acquire
__device__ __forceinline__ bool barrier_xcd(const Bar& b, unsigned epoch, int xcd, int per_xcd) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(&b.xcd[xcd], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        if (prev + 1 == epoch * (unsigned)per_xcd) {  // this XCD's last arriver: its L2 holds everybody's dirty lines
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_relaxed(b.top) < epoch * 8u) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) {
                    atomicExch(b.timeout, 2u);
                    ok = false;
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(&b.gen[xcd], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld_relaxed(&b.gen[xcd]) < epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > kSpinLimit) {
                    atomicExch(b.timeout, 3u);
                    ok = false;
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
    return ok;
}

// FORM 0: flat, 1: xcd.  Every round: write 128 words of "published state", barrier, read the words of the workgroup
// (g + 37) % G (another XCD) and check them, barrier (the readers are done before the next round overwrites).
template <int FORM>
__global__ __launch_bounds__(kThreads, 2) void k_rounds(Bar b, unsigned* state, int rounds, unsigned* bad, int work_iters) {
    extern __shared__ char smem[];
    (void)smem;
    const int g = blockIdx.x, G = gridDim.x;
    const int xcd = g & 7, per_xcd = G / 8;
    unsigned epoch = 0;
    unsigned sink = 0;
    for (int r = 1; r <= rounds; ++r) {
        for (int i = 0; i < work_iters; ++i) sink = sink * 1664525u + 1013904223u + threadIdx.x;  // (a little uneven work)
        if (threadIdx.x < 128) state[g * 128 + threadIdx.x] = (unsigned)r * 1000003u + g * 131u + threadIdx.x + (sink & 0u);
        if (!(FORM == 0 ? barrier_flat(b, ++epoch) : barrier_xcd(b, ++epoch, xcd, per_xcd))) return;
        const int o = (g + 37) % G;
        if (threadIdx.x < 128) {
            const unsigned v = state[o * 128 + threadIdx.x];
            if (v != (unsigned)r * 1000003u + o * 131u + threadIdx.x) atomicAdd(bad, 1u);
        }
        if (!(FORM == 0 ? barrier_flat(b, ++epoch) : barrier_xcd(b, ++epoch, xcd, per_xcd))) return;
    }
    if (sink == 0xFFFFFFFFu) bad[1] = sink;
}

__global__ __launch_bounds__(kThreads, 2) void k_trivial(unsigned* state, int r) {
    extern __shared__ char smem[];
    (void)smem;
    if (threadIdx.x < 128) state[blockIdx.x * 128 + threadIdx.x] += (unsigned)r;
}

int main() {
    unsigned *top, *xcd, *gen, *tmo, *state, *bad;
    CK(hipMalloc(&top, 4));
    CK(hipMalloc(&xcd, 32));
    CK(hipMalloc(&gen, 32));
    CK(hipMalloc(&tmo, 4));
    CK(hipMalloc(&state, kWG * 128 * 4));
    CK(hipMalloc(&bad, 8));
    CK(hipFuncSetAttribute((const void*)k_rounds<0>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    CK(hipFuncSetAttribute((const void*)k_rounds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    CK(hipFuncSetAttribute((const void*)k_trivial, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int rounds = 200;
    for (int form = 0; form < 2; ++form)
        for (int work : {0, 2000}) {
            float best = 1e9f;
            unsigned hbad = 0, htmo = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemset(top, 0, 4));
                CK(hipMemset(xcd, 0, 32));
                CK(hipMemset(gen, 0, 32));
                CK(hipMemset(tmo, 0, 4));
                CK(hipMemset(bad, 0, 8));
                Bar b{top, xcd, gen, tmo};
                CK(hipEventRecord(e0));
                if (form == 0) hipLaunchKernelGGL(k_rounds<0>, dim3(kWG), dim3(kThreads), kLds, 0, b, state, rounds, bad, work);
                else hipLaunchKernelGGL(k_rounds<1>, dim3(kWG), dim3(kThreads), kLds, 0, b, state, rounds, bad, work);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(&htmo, tmo, 4, hipMemcpyDeviceToHost));
            }
            printf("%-28s work %4d iters: %7.2f us per barrier (%d rounds x 2 barriers, 512 B published + checked per workgroup and round; "
                   "stale reads %u, timeouts %u)\n",
                   form == 0 ? "one device-scope counter" : "XCD-hierarchical", work, best * 1e3f / (2 * rounds), rounds, hbad, htmo);
        }
    // what a barrier replaces: a dependent launch of the same geometry (same stream)
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < 2 * rounds; ++r) hipLaunchKernelGGL(k_trivial, dim3(kWG), dim3(kThreads), kLds, 0, state, r);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("dependent launches of a trivial kernel, same geometry (256 x 512 threads, 159 KiB LDS): %7.2f us per launch + boundary\n",
               ms * 1e3f / (2 * rounds));
    }
    return 0;
}
