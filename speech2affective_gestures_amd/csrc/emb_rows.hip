// OPT-IN VARIANT (option EMB_BWD_ROWS, default OFF) of embedding_bwd_k (csrc/misc.hip): the gradient of nn.Embedding
// (TextEncoderTCN.embedding, net/multimodal_context_net_v2.py:69-72,84; the dense dW of loss.backward(), processor_v2.py:937).
// This is the kernel r03 wrote after the last GPU-run build and micro-timed ONCE on an MI355X (fp32, B = 256: 42 -> 17 us;
// DESIGN section 5 of r03, profiles/HISTORY.md) but never took through the suite on hardware; r05 removed it from the default
// path for that reason.  Back as a variant in a file of its own: parity against the default and against index_add_ in
// tests/test_gpu_zy_variants.py, timing in tools/ab_variants.py.
//
// Rows that repeat an id would hammer the same `dim` addresses with atomics, and one id does: PAD (id 0: vocab.py's PAD_token)
// fills ~85 % of every transcript.  The default merges runs of equal ids over 32 consecutive rows (one atomic per run and
// column); what is left are 272 workgroups x ~4 PAD runs, i.e. ~1 000 dependent read-modify-writes on each of the PAD row's 19
// cache lines -- the L2 serialises atomics to one line at ~30 ns apiece, and that IS the default kernel's time.  Here a
// workgroup owns 256 rows x 64 columns, its four waves take 64 rows each (16 loads in flight per lane), sum the PAD rows in a
// register, meet in LDS and leave ONE atomic per column: 34 per PAD line.  Word rows (a few per clip, rarely the same twice)
// go out as direct atomics.  Exact for any id pattern.
#include "s2ag_common.h"

namespace {
using namespace s2ag;
constexpr int EMB_RB = 256, EMB_CB = 64;
__global__ __launch_bounds__(256) void embedding_bwd_rows_k(const long long* ids, const float* __restrict__ g, int ldg,
                                                            int rows, int dim, int n_entries, float* dtable, float drop_p,
                                                            float inv_keep, const unsigned long long* rng, unsigned site) {
    __shared__ int sid[EMB_RB];
    __shared__ float pad_s[4][EMB_CB];
    SiteKey key{0, 0};
    const bool drop = drop_p > 0.f;
    if (drop) key = site_key(rng, site);
    const int r0 = blockIdx.x * EMB_RB;
    const int nr = min(EMB_RB, rows - r0);
    {
        long long id = (int)threadIdx.x < nr ? ids[r0 + threadIdx.x] : 0;
        sid[threadIdx.x] = (int)(id < 0 ? 0 : (id >= n_entries ? n_entries - 1 : id));
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.y * EMB_CB + lane;
    float pad_acc = 0.f;
    // deterministic flavour: workgroups in index order, and inside a workgroup wave by wave (two waves may hold the same word)
    s2ag::det_enter();
    S2AG_DET_WAVES_BEGIN
    if (c < dim) {
        const int i0 = wave * (EMB_RB / 4), i1 = min(nr, i0 + EMB_RB / 4);
        // 16 rows are LOADED before any of them is added: a load behind an atomic to memory it may alias waits for it, and
        // the loop was a chain of 64 memory round trips (33 us)
        for (int ib = i0; ib < i1; ib += 16) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = ib + j < i1 ? g[(long long)(r0 + ib + j) * ldg + c] : 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (ib + j >= i1) break;
                const long long r = r0 + ib + j;
                float x = v[j];
                if (drop) x *= keep_scale(key, (unsigned long long)r * dim + c, drop_p, inv_keep);
                const int id = sid[ib + j];
                if (id == 0) pad_acc += x;
                else atomicAdd(dtable + (long long)id * dim + c, x);
            }
        }
    }
    S2AG_DET_WAVES_END
    pad_s[wave][lane] = pad_acc;
    __syncthreads();
    if (wave == 0 && c < dim) {
        const float t = (pad_s[0][lane] + pad_s[1][lane]) + (pad_s[2][lane] + pad_s[3][lane]);
        if (t != 0.f) atomicAdd(dtable + c, t);
    }
    s2ag::det_leave();
}
}  // namespace

namespace s2ag {
// the accumulate / clear decision stays with s2ag_embedding_bwd (csrc/misc.hip), which calls this in place of its own launch
int embedding_bwd_rows_launch(const long long* ids, const float* g, int ldg, int rows, int dim, int n_entries, float* dtable,
                              float drop_p, const unsigned long long* rng, unsigned site, hipStream_t st) {
    hipLaunchKernelGGL(embedding_bwd_rows_k, dim3(cdiv(rows, EMB_RB), cdiv(dim, EMB_CB)), dim3(256), 0, st, ids, g, ldg, rows,
                       dim, n_entries, dtable, drop_p, drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, rng, site);
    return (int)hipGetLastError();
}
}  // namespace s2ag
S2AG_DET_HOOK(emb_rows)
