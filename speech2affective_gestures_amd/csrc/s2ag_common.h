// Shared device helpers for the gfx950 kernels of libs2ag_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "s2ag_hip.h"

#define S2AG_LAUNCH_CHECK()                  \
    do {                                     \
        hipError_t e__ = hipGetLastError();  \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

// Debug flavour (python -m speech2affective_gestures_amd.build --debug, -DS2AG_DEBUG=1): device-side asserts on the indices the
// loaders compute (LDS image offsets, staged-segment indices, row / column ranges).  A failing assert aborts the kernel and
// the next HIP call reports it; the release build compiles them away.
#if defined(S2AG_DEBUG) && S2AG_DEBUG
#include <assert.h>
#define S2AG_DBG_ASSERT(cond) assert(cond)
#else
#define S2AG_DBG_ASSERT(cond) ((void)0)
#endif

namespace s2ag {

constexpr int WAVE = 64;

// Run-time options of the library: set through s2ag_set_option (misc.hip) by the ONE registry of switches,
// speech2affective_gestures_amd/config.py.  The library itself never reads the environment.
// (OPT_GRU_SPLIT: a product mode.  The others: opt-in kernel VARIANTS that live in files of their own, default 0 -- each is
//  A/B-timed against the default by tools/ab_variants.sh and held bit-identical / to the same tolerance by tests/test_gpu_zy_variants.py)
enum Option { OPT_GRU_SPLIT = 0, OPT_WGRAD32_PIPE, OPT_TCN32_PAIR, OPT_EMB_BWD_ROWS, OPT_COUNT };
int option(Option o);

__host__ __device__ inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- counter-based randomness -----------------------------------------------------------------
// 32-bit avalanche (two multiply/xorshift rounds).  The stream of a site is keyed by
// (seed, step counter, site); element i of the site hashes (key, i) twice.  Stateless, so the
// backward pass regenerates a forward keep-mask instead of storing it.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

struct SiteKey {
    uint32_t k0, k1;
};

__device__ __forceinline__ SiteKey site_key(const unsigned long long* rng, unsigned site) {
    const unsigned long long seed = rng[0], ctr = rng[1];
    SiteKey k;
    k.k0 = mix32((uint32_t)seed ^ mix32((uint32_t)(ctr & 0xffffffffULL) + 0x9e3779b9U * (site + 1)));
    k.k1 = mix32((uint32_t)(seed >> 32) ^ mix32((uint32_t)(ctr >> 32) + 0x85ebca6bU * (site + 0x632be5abU)));
    return k;
}

__device__ __forceinline__ uint32_t rand_u32(SiteKey k, unsigned long long idx) {
    uint32_t x = mix32((uint32_t)idx * 0x9e3779b1U + k.k0);
    x = mix32(x ^ k.k1 ^ (uint32_t)(idx >> 32) * 0xc2b2ae35U);
    return x;
}

// keep mask already scaled by 1/(1-p)
__device__ __forceinline__ float keep_scale(SiteKey k, unsigned long long idx, float p, float inv_keep) {
    const float u = (float)(rand_u32(k, idx) >> 8) * (1.0f / 16777216.0f);
    return u >= p ? inv_keep : 0.0f;
}

__device__ __forceinline__ float normal_dev(SiteKey k, unsigned long long idx) {
    const uint32_t a = rand_u32(k, 2 * idx), b = rand_u32(k, 2 * idx + 1);
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0,1]
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);            // [0,1)
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

__device__ __forceinline__ float leaky(float x, float slope) { return x > 0.f ? x : x * slope; }

__device__ __forceinline__ float apply_act(float x, int act, float slope) {
    if (act == S2AG_ACT_LEAKY) return leaky(x, slope);
    if (act == S2AG_ACT_SIGMOID) return 1.0f / (1.0f + expf(-x));
    return x;
}

// Accumulators are cleared by a KERNEL, never by hipMemsetAsync: inside a replayed hipGraph (ROCm 7.2) a memset node
// was observed not to be ordered against the kernel that follows it (see gru_coop.hip), which would silently drop
// atomically accumulated sums.
static __global__ void zero_words_k(unsigned* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

static inline hipError_t zero_async(void* p, size_t bytes, hipStream_t stream) {
    const size_t n = (bytes + 3) / 4;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(zero_words_k, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<unsigned*>(p), n);
    return hipGetLastError();
}

// ---- deterministic mode (build flavour `det`: -DS2AG_DET=1, libs2ag_hip_det.so; s2ag_set_deterministic) ------------------
// fp32 sums formed by atomics depend on the order in which workgroups (and, inside a workgroup, wavefronts) arrive: two runs
// of the same step differ in the last bits, which is why tests compare runs with a tolerance and why a divergence between
// data-parallel replicas cannot be bisected.  In the det flavour with the mode on (g_det_turn = a zero device word; the host
// also serialises the passes of a step onto one stream) every workgroup passes its accumulation phase IN THE ORDER OF ITS
// LINEAR INDEX: det_enter() waits until the turn word equals the index, det_leave() drains this workgroup's atomics and hands
// the turn on (the last workgroup re-arms the word).  Progress: a workgroup only ever waits for workgroups with a smaller
// index; that those were started earlier is a property of the in-order workgroup dispatcher of each gfx950 XCD, not of HIP
// -- so the poll is bounded and a time-out raises the sticky error word (g_det_err, the trainer's error flag: the step fails
// loudly instead of hanging).  In-workgroup accumulation into LDS goes wavefront by wavefront (S2AG_DET_WAVES_BEGIN..END).
// In the RELEASE library (no S2AG_DET) the three helpers are empty at compile time: the default kernels carry no trace of the
// mode (r04 had it as a run-time word in every accumulating kernel, which changed 39 default binaries).
#if defined(S2AG_DET) && S2AG_DET
static __device__ int* g_det_turn = nullptr;             // one copy per translation unit, installed by s2ag_det_hook_<file>
static __device__ unsigned* g_det_err = nullptr;         // sticky error word the trainer reads every step (bit 3: a turn never came)
constexpr unsigned DET_ERR_BIT = 8u;
// polls of >= 64 cycles each.  The longest legitimate wait is the LAST workgroup's: the serialised accumulation phases of a
// whole launch, milliseconds at most; 2^22 polls (~0.2 s) is far beyond that and keeps a broken ordering assumption from
// costing more than a fraction of a second per launch -- and once the error bit is up nobody waits at all any more.
#ifndef S2AG_DET_SPIN_LIMIT
#define S2AG_DET_SPIN_LIMIT (1 << 22)
#endif
constexpr int DET_SPIN_LIMIT = S2AG_DET_SPIN_LIMIT;

__device__ __forceinline__ void det_enter() {            // workgroup-uniform call
    int* w = g_det_turn;
    if (!w) return;
    if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
        const int me = (int)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        int spins = 0;
        const bool failed = g_det_err && (__hip_atomic_load(g_det_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & DET_ERR_BIT);
        while (!failed && __hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != me) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > DET_SPIN_LIMIT) {              // give up: the sums of this launch are no longer ordered
                if (g_det_err) atomicOr(g_det_err, DET_ERR_BIT);
                break;
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void det_leave() {            // by every thread of the workgroup that is still alive
    int* w = g_det_turn;
    if (!w) return;
    __threadfence();                                     // this thread's atomics have been performed
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
        const int n = (int)(gridDim.x * gridDim.y * gridDim.z);
        const int me = (int)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        __hip_atomic_store(w, me + 1 == n ? 0 : me + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// S2AG_DET_WAVES_BEGIN ... S2AG_DET_WAVES_END bracket a block that accumulates into LDS with atomics from several wavefronts
// (reached by the whole workgroup): deterministic mode runs it wavefront by wavefront.  Macros, not a lambda taking helper: in
// the release flavour they must leave the enclosed statements EXACTLY as they were written before the mode existed.
#define S2AG_DET_WAVES_BEGIN                                                                                              \
    {                                                                                                                     \
        const bool det_on__ = s2ag::g_det_turn != nullptr;                                                                \
        const int det_nw__ = det_on__ ? ((int)(blockDim.x * blockDim.y * blockDim.z) + 63) / 64 : 1;                      \
        const int det_me__ = (int)((threadIdx.z * blockDim.y + threadIdx.y) * blockDim.x + threadIdx.x) >> 6;             \
        for (int det_w__ = 0; det_w__ < det_nw__; ++det_w__) {                                                            \
            if (!det_on__ || det_w__ == det_me__) {
#define S2AG_DET_WAVES_END                                                                                                \
            }                                                                                                             \
            if (det_on__) __syncthreads();                                                                                \
        }                                                                                                                 \
    }

static inline int det_install_here(int* word, unsigned* err) {
    int rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_det_turn), &word, sizeof(word));
    if (!rc) rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(g_det_err), &err, sizeof(err));
    return rc;
}
#else
__device__ __forceinline__ void det_enter() {}
__device__ __forceinline__ void det_leave() {}
#define S2AG_DET_WAVES_BEGIN {
#define S2AG_DET_WAVES_END }
static inline int det_install_here(int* word, unsigned*) { return word ? -2 /* S2AG_E_UNSUPPORTED: not the det flavour */ : 0; }
#endif
// every translation unit with accumulating kernels exports a hook that installs the turn word in ITS copy of g_det_turn
#define S2AG_DET_HOOK(file) \
    extern "C" int s2ag_det_hook_##file(int* word, unsigned* err) { return s2ag::det_install_here(word, err); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace s2ag
