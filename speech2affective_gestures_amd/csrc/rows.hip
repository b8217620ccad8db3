// Touched-row exchange of the word-embedding gradient between data-parallel replicas.
//
// nn.Embedding(n_words, 300) of the text encoder (net/multimodal_context_net_v2.py:70-73 of the reference) receives a
// gradient in at most B*T of its n_words rows per step -- in practice a few hundred (a 34-frame clip holds a handful of
// words; every other frame is the PAD token).  A dense all-reduce would move all 24 MB (n_words = 20 000) per step over
// xGMI.  Instead every replica (1) lists the rows its batch touches (sorted, unique, fixed capacity), (2) packs
// [row id | 300 floats] records out of its dense gradient, (3) all-gathers the records (RCCL, one collective) and
// (4) rebuilds the touched rows of the dense gradient as the sum over replicas IN RANK ORDER -- bit-identical on every
// replica, like an all-reduce; the fused Adam then runs on the dense arena exactly as before (the reference's dense
// torch.optim.Adam moves every row every step, so a row-sparse optimizer would not be its drop-in).
#include "s2ag_common.h"

using namespace s2ag;

namespace {

__global__ __launch_bounds__(256) void rows_mark_k(const long long* __restrict__ ids, int n_tokens, int n_entries,
                                                   int* __restrict__ mark) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tokens) return;
    const long long id = ids[i];
    if (id >= 0 && id < n_entries) mark[id] = 1;
}

// One block: ordered compaction of the marked ids (block-wide exclusive scan per 1024-id chunk), sentinel padding,
// and the mark array is left zero again for the next step.
__global__ __launch_bounds__(1024) void rows_compact_k(int* __restrict__ mark, int n_entries, int cap,
                                                       int* __restrict__ uids, int* __restrict__ count,
                                                       int* __restrict__ overflow) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n_entries; c0 += 1024) {
        const int id = c0 + tid;
        const int m = id < n_entries ? mark[id] : 0;
        if (id < n_entries && m) mark[id] = 0;
        const unsigned long long bal = __ballot(m != 0);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wv; ++w) off += wsum[w];
        if (m) {
            const int s = off + before;
            if (s < cap) uids[s] = id;
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += wsum[w];
            base += t;
        }
        __syncthreads();
    }
    const int n = base;
    for (int s = n + tid; s < cap; s += 1024) uids[s] = n_entries;      // sentinel: sorts behind every real id
    if (tid == 0) {
        *count = n;
        if (n > cap && overflow) __hip_atomic_fetch_or(overflow, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// record s = [bit pattern of uids[s] | dense[uids[s], 0:dim]]; sentinel slots carry zeros
__global__ __launch_bounds__(128) void rows_pack_k(const float* __restrict__ dense, const int* __restrict__ uids, int dim,
                                                   int n_entries, float* __restrict__ out) {
    const int s = blockIdx.x;
    const int id = uids[s];
    float* o = out + (size_t)s * (dim + 1);
    if (threadIdx.x == 0) o[0] = __int_as_float(id);
    const bool live = id >= 0 && id < n_entries;
    const float* src = dense + (size_t)(live ? id : 0) * dim;
    for (int j = threadIdx.x; j < dim; j += blockDim.x) o[1 + j] = live ? src[j] : 0.f;
}

__device__ __forceinline__ int find_slot(const float* __restrict__ recs, int cap, int stride, int id) {
    int lo = 0, hi = cap;      // first slot whose id >= id
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (__float_as_int(recs[(size_t)mid * stride]) < id) lo = mid + 1;
        else hi = mid;
    }
    return (lo < cap && __float_as_int(recs[(size_t)lo * stride]) == id) ? lo : -1;
}

// block (s, r): record s of replica r.  The lowest replica that lists a row owns it and writes the sum over all
// replicas that list it, added in rank order.
__global__ __launch_bounds__(128) void rows_merge_k(const float* __restrict__ gathered, int world, int cap, int dim,
                                                    int n_entries, float* __restrict__ dense) {
    const int s = blockIdx.x, r = blockIdx.y, stride = dim + 1;
    const float* mine = gathered + ((size_t)r * cap + s) * stride;
    const int id = __float_as_int(mine[0]);
    if (id < 0 || id >= n_entries) return;
    __shared__ int slot[64];
    if (threadIdx.x < world) {
        const int q = threadIdx.x;
        slot[q] = q == r ? s : find_slot(gathered + (size_t)q * cap * stride, cap, stride, id);
    }
    __syncthreads();
    for (int q = 0; q < r; ++q)
        if (slot[q] >= 0) return;                  // a lower replica owns this row
    for (int j = threadIdx.x; j < dim; j += blockDim.x) {
        float acc = 0.f;
        bool first = true;
        for (int q = 0; q < world; ++q) {
            if (slot[q] < 0) continue;
            const float v = gathered[((size_t)q * cap + slot[q]) * stride + 1 + j];
            acc = first ? v : acc + v;
            first = false;
        }
        dense[(size_t)id * dim + j] = acc;
    }
}
}  // namespace

extern "C" int s2ag_rows_unique(const long long* ids, int n_tokens, int n_entries, int cap, int* mark, int* uids,
                                int* count, int* overflow_flag, void* stream) {
    if (!ids || !mark || !uids || !count || n_tokens <= 0 || n_entries <= 0 || cap <= 0) return S2AG_E_BADARG;
    hipLaunchKernelGGL(rows_mark_k, dim3(cdiv(n_tokens, 256)), dim3(256), 0, (hipStream_t)stream, ids, n_tokens,
                       n_entries, mark);
    hipLaunchKernelGGL(rows_compact_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, mark, n_entries, cap, uids, count,
                       overflow_flag);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_rows_pack(const float* dense, const int* uids, int cap, int dim, int n_entries, float* records,
                              void* stream) {
    if (!dense || !uids || !records || cap <= 0 || dim <= 0) return S2AG_E_BADARG;
    hipLaunchKernelGGL(rows_pack_k, dim3(cap), dim3(128), 0, (hipStream_t)stream, dense, uids, dim, n_entries, records);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_rows_merge(const float* gathered, int world, int cap, int dim, int n_entries, float* dense,
                               void* stream) {
    if (!gathered || !dense || world <= 0 || world > 64 || cap <= 0 || dim <= 0) return S2AG_E_BADARG;
    hipLaunchKernelGGL(rows_merge_k, dim3(cap, world), dim3(128), 0, (hipStream_t)stream, gathered, world, cap, dim,
                       n_entries, dense);
    S2AG_LAUNCH_CHECK();
    return 0;
}
