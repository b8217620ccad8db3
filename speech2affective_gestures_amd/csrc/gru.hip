// Persistent recurrent kernels of one bidirectional GRU layer (PyTorch gate order r, z, n).
//
// The input projections W_ih x + b_ih of all T frames are hoisted into one MFMA GEMM per direction
// (s2ag_conv1d_nlc_fwd), so only the strictly sequential part h_t = f(gi_t, W_hh h_{t-1}) runs here.
// One workgroup owns a slice of BS clips of one direction for all T steps (no inter-workgroup
// exchange, no grid barrier): per step it multiplies its BS x H state (k-major in LDS, broadcast
// reads) with W_hh^T streamed from L2 in 16-byte coalesced loads, K split over thread slices and
// reduced through LDS in a fixed order (deterministic), then applies the gate math.
// With BS < 16 rows the f32 MFMA pipe (same rate as VALU) would idle half its tile, so the matvec
// is plain f32 FMA.  Bound: L2 -> CU streaming of W_hh (3H*H*4 bytes per step per workgroup).
#include "s2ag_common.h"

namespace {
using namespace s2ag;

constexpr int NT = 1024;   // 16 waves: enough loads in flight to cover the L2 latency of the W_hh stream

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int BS>
__global__ __launch_bounds__(NT) void gru_seq_fwd_k(const float* __restrict__ gi, const float* __restrict__ whhT,
                                                    const float* __restrict__ bhh, float* __restrict__ y,
                                                    float* __restrict__ ydrop, float* __restrict__ gates, int B,
                                                    int T, int H, float drop_p, float inv_keep,
                                                    const unsigned long long* rng, unsigned site) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H3 = 3 * H;
    const int NC4 = H3 / 4;                    // column groups of 4 gate columns
    const int KS = max(1, NT / NC4);           // K slices
    const int kchunk = (H + KS - 1) / KS;
    float* hs = smem;                          // [H][BS]   k-major state
    float* red = smem + H * BS;                // [KS][BS][3H]
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * BS;
    const int nb = min(BS, B - b0);
    const int tid = threadIdx.x;
    const float* W = whhT + (long long)dir * H * H3;
    const float* bh = bhh + dir * H3;

    for (int i = tid; i < H * BS; i += NT) hs[i] = 0.f;
    SiteKey key{0, 0};
    const bool drop = ydrop != nullptr && drop_p > 0.f;
    if (drop) key = site_key(rng, site);
    __syncthreads();

    const int ngroups = NC4 * KS;   // active (cg, ks) work items when NC4 <= NT
    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        // ---- phase 1: partial products  red[ks][b][col] = sum_{k in slice} h[b][k] * W^T[k][col]
        for (int item = tid; item < ngroups; item += NT) {
            const int cg = item % NC4, ks = item / NC4;
            const int kbeg = ks * kchunk, kend = min(H, kbeg + kchunk);
            float acc[BS][4];
#pragma unroll
            for (int b = 0; b < BS; ++b) acc[b][0] = acc[b][1] = acc[b][2] = acc[b][3] = 0.f;
            const float* wp = W + (long long)kbeg * H3 + cg * 4;
#pragma unroll 8
            for (int k = kbeg; k < kend; ++k) {
                const float4 w = *reinterpret_cast<const float4*>(wp);
                wp += H3;
                const float* hk = hs + k * BS;
#pragma unroll
                for (int b = 0; b < BS; ++b) {
                    const float hv = hk[b];
                    acc[b][0] = fmaf(hv, w.x, acc[b][0]);
                    acc[b][1] = fmaf(hv, w.y, acc[b][1]);
                    acc[b][2] = fmaf(hv, w.z, acc[b][2]);
                    acc[b][3] = fmaf(hv, w.w, acc[b][3]);
                }
            }
#pragma unroll
            for (int b = 0; b < BS; ++b)
                *reinterpret_cast<float4*>(red + ((long long)ks * BS + b) * H3 + cg * 4) =
                    make_float4(acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
        }
        __syncthreads();
        // ---- phase 2: gates, one thread per (clip, hidden unit)
        for (int e = tid; e < nb * H; e += NT) {
            const int b = e / H, i = e - b * H;
            float ghr = bh[i], ghz = bh[H + i], ghn = bh[2 * H + i];
            for (int ks = 0; ks < KS; ++ks) {
                const float* rp = red + ((long long)ks * BS + b) * H3;
                ghr += rp[i];
                ghz += rp[H + i];
                ghn += rp[2 * H + i];
            }
            const long long row = (long long)(b0 + b) * T + t;
            const float* gp = gi + row * (2 * H3) + dir * H3;
            const float r = sigmoidf_(gp[i] + ghr);
            const float z = sigmoidf_(gp[H + i] + ghz);
            const float n = tanhf(gp[2 * H + i] + r * ghn);
            const float hp = hs[i * BS + b];
            const float hn = (1.f - z) * n + z * hp;
            hs[i * BS + b] = hn;
            const long long yi = row * (2 * H) + dir * H + i;
            y[yi] = hn;
            if (ydrop) ydrop[yi] = drop ? hn * keep_scale(key, (unsigned long long)yi, drop_p, inv_keep) : hn;
            if (gates) {
                float* gs = gates + ((long long)dir * B * T + row) * (4 * H);
                *reinterpret_cast<float4*>(gs + 4 * i) = make_float4(r, z, n, ghn);
            }
        }
        __syncthreads();
    }
}

template <int BS>
__global__ __launch_bounds__(NT) void gru_seq_bwd_k(const float* __restrict__ dy, int lddy, int dy_dir_stride,
                                                    const float* __restrict__ whh, const float* __restrict__ y,
                                                    const float* __restrict__ gates, float* __restrict__ dgi,
                                                    float* __restrict__ dgh, int B, int T, int H, float drop_p,
                                                    float inv_keep, const unsigned long long* rng, unsigned site) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H3 = 3 * H;
    const int NC4 = H / 4;                     // column groups of dh
    const int KS = max(1, NT / NC4);
    const int kchunk = (H3 + KS - 1) / KS;
    float* dh = smem;                          // [BS][H]   running grad w.r.t. h_t
    float* gs = smem + BS * H;                 // [3H][BS]  dgh of this step, k-major
    float* red = gs + H3 * BS;                 // [KS][BS][H]
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * BS;
    const int nb = min(BS, B - b0);
    const int tid = threadIdx.x;
    const float* W = whh + (long long)dir * H3 * H;   // (3H, H) row-major

    for (int i = tid; i < BS * H; i += NT) dh[i] = 0.f;
    for (int i = tid; i < H3 * BS; i += NT) gs[i] = 0.f;
    SiteKey key{0, 0};
    const bool drop = drop_p > 0.f;
    if (drop) key = site_key(rng, site);
    __syncthreads();

    const int ngroups = NC4 * KS;
    for (int step = 0; step < T; ++step) {
        // reverse of the forward visiting order
        const int t = dir ? step : (T - 1 - step);
        const int tprev = dir ? t + 1 : t - 1;   // time index of h_{prev} in the forward recurrence
        // ---- phase A: gate gradients
        for (int e = tid; e < nb * H; e += NT) {
            const int b = e / H, i = e - b * H;
            const long long row = (long long)(b0 + b) * T + t;
            float g = dy[row * lddy + dir * dy_dir_stride + i];
            if (drop) g *= keep_scale(key, (unsigned long long)(row * (2 * H) + dir * H + i), drop_p, inv_keep);
            const float dht = dh[b * H + i] + g;
            const float* gp = gates + ((long long)dir * B * T + row) * (4 * H);
            const float4 sv = *reinterpret_cast<const float4*>(gp + 4 * i);
            const float r = sv.x, z = sv.y, n = sv.z, hn = sv.w;
            float hp = 0.f;
            if (tprev >= 0 && tprev < T) hp = y[((long long)(b0 + b) * T + tprev) * (2 * H) + dir * H + i];
            const float dn = dht * (1.f - z) * (1.f - n * n);
            const float dz = dht * (hp - n) * z * (1.f - z);
            const float dr = dn * hn * r * (1.f - r);
            float* gi_o = dgi + row * (2 * H3) + dir * H3;
            gi_o[i] = dr;
            gi_o[H + i] = dz;
            gi_o[2 * H + i] = dn;
            float* gh_o = dgh + ((long long)dir * B * T + row) * H3;
            const float dnr = dn * r;
            gh_o[i] = dr;
            gh_o[H + i] = dz;
            gh_o[2 * H + i] = dnr;
            gs[i * BS + b] = dr;
            gs[(H + i) * BS + b] = dz;
            gs[(2 * H + i) * BS + b] = dnr;
            dh[b * H + i] = dht * z;
        }
        __syncthreads();
        // ---- phase B: dh += dgh @ W_hh   (K = 3H split over slices)
        for (int item = tid; item < ngroups; item += NT) {
            const int cg = item % NC4, ks = item / NC4;
            const int kbeg = ks * kchunk, kend = min(H3, kbeg + kchunk);
            float acc[BS][4];
#pragma unroll
            for (int b = 0; b < BS; ++b) acc[b][0] = acc[b][1] = acc[b][2] = acc[b][3] = 0.f;
            const float* wp = W + (long long)kbeg * H + cg * 4;
#pragma unroll 8
            for (int k = kbeg; k < kend; ++k) {
                const float4 w = *reinterpret_cast<const float4*>(wp);
                wp += H;
                const float* gk = gs + k * BS;
#pragma unroll
                for (int b = 0; b < BS; ++b) {
                    const float gv = gk[b];
                    acc[b][0] = fmaf(gv, w.x, acc[b][0]);
                    acc[b][1] = fmaf(gv, w.y, acc[b][1]);
                    acc[b][2] = fmaf(gv, w.z, acc[b][2]);
                    acc[b][3] = fmaf(gv, w.w, acc[b][3]);
                }
            }
#pragma unroll
            for (int b = 0; b < BS; ++b)
                *reinterpret_cast<float4*>(red + ((long long)ks * BS + b) * H + cg * 4) =
                    make_float4(acc[b][0], acc[b][1], acc[b][2], acc[b][3]);
        }
        __syncthreads();
        for (int e = tid; e < nb * H; e += NT) {
            const int b = e / H, i = e - b * H;
            float s = dh[b * H + i];
            for (int ks = 0; ks < KS; ++ks) s += red[((long long)ks * BS + b) * H + i];
            dh[b * H + i] = s;
        }
        __syncthreads();
    }
}

inline size_t fwd_smem(int BS, int H) {
    const int H3 = 3 * H, NC4 = H3 / 4, KS = NT / NC4 > 1 ? NT / NC4 : 1;
    return sizeof(float) * ((size_t)H * BS + (size_t)KS * BS * H3);
}
inline size_t bwd_smem(int BS, int H) {
    const int H3 = 3 * H, NC4 = H / 4, KS = NT / NC4 > 1 ? NT / NC4 : 1;
    return sizeof(float) * ((size_t)BS * H + (size_t)H3 * BS + (size_t)KS * BS * H);
}

template <typename K>
int allow_smem(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return S2AG_E_UNSUPPORTED;
    static size_t granted = 64 * 1024;   // per template instantiation; set once, outside any capture replay
    if (bytes > granted) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return (int)e;
        granted = bytes;
    }
    return 0;
}
}  // namespace

// small-H specialisation (gru_small.hip): W_hh in registers, one thread per (clip, unit)
int s2ag_gru_small_supported(int H);
int s2ag_gru_small_fwd(const float* gi, const float* whh, const float* bhh, float* y, float* ydrop, float* gates, int B,
                       int T, int H, float p, const unsigned long long* rng, unsigned site, hipStream_t stream);
int s2ag_gru_small_bwd(const float* dy, int lddy, int dy_dir_stride, const float* whh, const float* y,
                       const float* gates, float* dgi, float* dgh, int B, int T, int H, float p,
                       const unsigned long long* rng, unsigned site, hipStream_t stream);

extern "C" int s2ag_gru_seq_needs_transposed(int H) { return s2ag_gru_small_supported(H) ? 0 : 1; }

extern "C" int s2ag_gru_seq_fwd(const float* gi, const float* whh, const float* whhT, const float* bhh, float* y,
                                float* ydrop, float* gates, int B, int T, int H, const s2ag_epilogue* e, void* stream) {
    if (!gi || !whh || !bhh || !y || B <= 0 || T <= 0 || H <= 0) return S2AG_E_BADARG;
    if (H % 4 != 0 || (3 * H) / 4 > NT) return S2AG_E_UNSUPPORTED;
    const float p = (e && ydrop) ? e->drop_p : 0.f;
    if (p > 0.f && !e->rng) return S2AG_E_BADARG;
    if (s2ag_gru_small_supported(H))
        return s2ag_gru_small_fwd(gi, whh, bhh, y, ydrop, gates, B, T, H, p, e ? e->rng : nullptr, e ? e->site : 0u,
                                  (hipStream_t)stream);
    if (!whhT) return S2AG_E_BADARG;     // the L2-streaming kernel wants W_hh^T for coalesced 16-byte loads
    constexpr int BS = 8;
    const size_t sm = fwd_smem(BS, H);
    int rc = allow_smem(gru_seq_fwd_k<BS>, sm);
    if (rc) return rc;
    hipLaunchKernelGGL(gru_seq_fwd_k<BS>, dim3(cdiv(B, BS), 2), dim3(NT), sm, (hipStream_t)stream, gi, whhT, bhh, y,
                       ydrop, gates, B, T, H, p, p > 0.f ? 1.f / (1.f - p) : 1.f, e ? e->rng : nullptr,
                       e ? e->site : 0u);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_gru_seq_bwd(const float* dy, int lddy, int dy_dir_stride, const float* whh, const float* y,
                                const float* gates, float* dgi, float* dgh, int B, int T, int H,
                                const s2ag_epilogue* e, void* stream) {
    if (!dy || !whh || !y || !gates || !dgi || !dgh || B <= 0 || T <= 0 || H <= 0) return S2AG_E_BADARG;
    if (H % 4 != 0 || H / 4 > NT) return S2AG_E_UNSUPPORTED;
    const float p = e ? e->drop_p : 0.f;
    if (p > 0.f && !e->rng) return S2AG_E_BADARG;
    if (s2ag_gru_small_supported(H))
        return s2ag_gru_small_bwd(dy, lddy, dy_dir_stride, whh, y, gates, dgi, dgh, B, T, H, p, e ? e->rng : nullptr,
                                  e ? e->site : 0u, (hipStream_t)stream);
    constexpr int BS = 8;
    const size_t sm = bwd_smem(BS, H);
    int rc = allow_smem(gru_seq_bwd_k<BS>, sm);
    if (rc) return rc;
    hipLaunchKernelGGL(gru_seq_bwd_k<BS>, dim3(cdiv(B, BS), 2), dim3(NT), sm, (hipStream_t)stream, dy, lddy,
                       dy_dir_stride, whh, y, gates, dgi, dgh, B, T, H, p, p > 0.f ? 1.f / (1.f - p) : 1.f,
                       e ? e->rng : nullptr, e ? e->site : 0u);
    S2AG_LAUNCH_CHECK();
    return 0;
}
