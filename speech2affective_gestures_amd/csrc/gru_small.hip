// GRU recurrence for small hidden sizes (H = 64: the discriminators; H = 32: reduced-width test configs).
//
// At H <= 64 the three W_hh rows that produce one hidden unit's gates are 3H <= 192 floats: they fit in ONE thread's
// registers.  So: one thread per (clip, hidden unit), W_hh resident in registers for all T steps, the state h
// broadcast from LDS (a wave = one clip at H = 64, so every h read is a broadcast), gate math in the same thread --
// no partial sums, no L2 traffic in the loop, one barrier per step (double-buffered state).  A workgroup is 256
// threads = 256/H clips of one direction (one wave per SIMD, so the 192 weight registers never spill).  The backward kernel mirrors it with W_hh's COLUMN j (3H floats) per thread and d(gh)
// broadcast from LDS (two barriers per step).
// The generic L2-streaming kernels (gru.hip) took 3.7 / 5.3 us per time step here; nothing in them was busy -- they
// were a chain of latencies (L2 weight loads, 21-way partial sums, two phases, global gi loads).
#include "s2ag_common.h"

namespace {
using namespace s2ag;

constexpr int SNT = 256;  // threads per workgroup = 4 waves = one per SIMD, so each thread may use the whole 512-entry
                          // register file (192 weight registers at H = 64 spilled under a 2-waves-per-SIMD budget)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int H>
__global__ __launch_bounds__(SNT) void gru_small_fwd_k(const float* __restrict__ gi, const float* __restrict__ whh,
                                                          const float* __restrict__ bhh, float* __restrict__ y,
                                                          float* __restrict__ ydrop, float* __restrict__ gates, int B,
                                                          int T, float drop_p, float inv_keep,
                                                          const unsigned long long* rng, unsigned site) {
    constexpr int H3 = 3 * H;
    constexpr int SBS = SNT / H;                        // clips per workgroup (4 at H = 64, 8 at H = 32)
    __shared__ __attribute__((aligned(16))) float hs[2][SBS][H];
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * SBS;
    const int tid = threadIdx.x;
    const int b = tid / H, i = tid - b * H;
    const bool valid = (b0 + b) < B;
    const float* W = whh + (size_t)dir * H3 * H;       // (3H, H) reference layout
    float wr[H], wz[H], wn[H];                          // this unit's three gate rows, resident for the launch
#pragma unroll
    for (int k = 0; k < H; k += 4) {                    // 16-byte loads along the contiguous k axis
        const float4 a = *reinterpret_cast<const float4*>(W + (size_t)i * H + k);
        const float4 bq = *reinterpret_cast<const float4*>(W + (size_t)(H + i) * H + k);
        const float4 c = *reinterpret_cast<const float4*>(W + (size_t)(2 * H + i) * H + k);
        wr[k] = a.x; wr[k + 1] = a.y; wr[k + 2] = a.z; wr[k + 3] = a.w;
        wz[k] = bq.x; wz[k + 1] = bq.y; wz[k + 2] = bq.z; wz[k + 3] = bq.w;
        wn[k] = c.x; wn[k + 1] = c.y; wn[k + 2] = c.z; wn[k + 3] = c.w;
    }
    const float bhr = bhh[dir * H3 + i], bhz = bhh[dir * H3 + H + i], bhn = bhh[dir * H3 + 2 * H + i];
    hs[0][b][i] = 0.f;
    SiteKey key{0, 0};
    const bool drop = ydrop != nullptr && drop_p > 0.f;
    if (drop) key = site_key(rng, site);
    __syncthreads();
    float hp = 0.f;
    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        const int cur = step & 1;
        const long long row = (long long)(b0 + b) * T + t;
        float gir = 0.f, giz = 0.f, gin = 0.f;
        if (valid) {                                     // the only global loads of the loop: in flight under the FMAs
            const float* gp = gi + row * (2 * H3) + dir * H3;
            gir = gp[i];
            giz = gp[H + i];
            gin = gp[2 * H + i];
        }
        float ar = bhr, az = bhz, an = bhn;
#pragma unroll
        for (int k = 0; k < H; k += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(&hs[cur][b][k]);    // broadcast within the clip
            ar = fmaf(hv.x, wr[k], ar); az = fmaf(hv.x, wz[k], az); an = fmaf(hv.x, wn[k], an);
            ar = fmaf(hv.y, wr[k + 1], ar); az = fmaf(hv.y, wz[k + 1], az); an = fmaf(hv.y, wn[k + 1], an);
            ar = fmaf(hv.z, wr[k + 2], ar); az = fmaf(hv.z, wz[k + 2], az); an = fmaf(hv.z, wn[k + 2], an);
            ar = fmaf(hv.w, wr[k + 3], ar); az = fmaf(hv.w, wz[k + 3], az); an = fmaf(hv.w, wn[k + 3], an);
        }
        const float r = sigmoidf_(gir + ar);
        const float z = sigmoidf_(giz + az);
        const float n = tanhf(gin + r * an);
        const float hn = (1.f - z) * n + z * hp;
        hp = hn;
        hs[cur ^ 1][b][i] = hn;
        if (valid) {
            const long long yi = row * (2 * H) + dir * H + i;
            y[yi] = hn;
            if (ydrop) ydrop[yi] = drop ? hn * keep_scale(key, (unsigned long long)yi, drop_p, inv_keep) : hn;
            if (gates) {
                float* gs = gates + ((long long)dir * B * T + row) * (4 * H);
                gs[i] = r;
                gs[H + i] = z;
                gs[2 * H + i] = n;
                gs[3 * H + i] = an;
            }
        }
        __syncthreads();
    }
}

template <int H>
__global__ __launch_bounds__(SNT) void gru_small_bwd_k(const float* __restrict__ dy, int lddy, int dy_dir_stride,
                                                          const float* __restrict__ whh, const float* __restrict__ y,
                                                          const float* __restrict__ gates, float* __restrict__ dgi,
                                                          float* __restrict__ dgh, int B, int T, float drop_p,
                                                          float inv_keep, const unsigned long long* rng, unsigned site) {
    constexpr int H3 = 3 * H;
    constexpr int SBS = SNT / H;
    __shared__ __attribute__((aligned(16))) float gs[SBS][H3];      // d(gh) of this step
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * SBS;
    const int tid = threadIdx.x;
    const int b = tid / H, i = tid - b * H;
    const bool valid = (b0 + b) < B;
    const float* W = whh + (size_t)dir * H3 * H;        // (3H, H) row-major: this thread keeps column i
    float wc[H3];
#pragma unroll
    for (int k = 0; k < H3; ++k) wc[k] = W[k * H + i];
    SiteKey key{0, 0};
    const bool drop = drop_p > 0.f;
    if (drop) key = site_key(rng, site);
    float dh = 0.f;
    for (int step = 0; step < T; ++step) {
        const int t = dir ? step : (T - 1 - step);
        const int tprev = dir ? t + 1 : t - 1;
        const long long row = (long long)(b0 + b) * T + t;
        float dr = 0.f, dz = 0.f, dnr = 0.f, carry = 0.f;
        if (valid) {
            float g = dy[row * lddy + dir * dy_dir_stride + i];
            if (drop) g *= keep_scale(key, (unsigned long long)(row * (2 * H) + dir * H + i), drop_p, inv_keep);
            const float dht = dh + g;
            const float* gp = gates + ((long long)dir * B * T + row) * (4 * H);
            const float r = gp[i], z = gp[H + i], n = gp[2 * H + i], hn = gp[3 * H + i];
            float hp = 0.f;
            if (tprev >= 0 && tprev < T) hp = y[((long long)(b0 + b) * T + tprev) * (2 * H) + dir * H + i];
            const float dn = dht * (1.f - z) * (1.f - n * n);
            dz = dht * (hp - n) * z * (1.f - z);
            dr = dn * hn * r * (1.f - r);
            dnr = dn * r;
            float* gi_o = dgi + row * (2 * H3) + dir * H3;
            gi_o[i] = dr;
            gi_o[H + i] = dz;
            gi_o[2 * H + i] = dn;
            float* gh_o = dgh + ((long long)dir * B * T + row) * H3;
            gh_o[i] = dr;
            gh_o[H + i] = dz;
            gh_o[2 * H + i] = dnr;
            carry = dht * z;
        }
        gs[b][i] = dr;
        gs[b][H + i] = dz;
        gs[b][2 * H + i] = dnr;
        __syncthreads();
        float acc = carry;
#pragma unroll
        for (int k = 0; k < H3; k += 4) {
            const float4 gv = *reinterpret_cast<const float4*>(&gs[b][k]);        // broadcast within the clip
            acc = fmaf(gv.x, wc[k], acc);
            acc = fmaf(gv.y, wc[k + 1], acc);
            acc = fmaf(gv.z, wc[k + 2], acc);
            acc = fmaf(gv.w, wc[k + 3], acc);
        }
        dh = acc;
        __syncthreads();
    }
}
}  // namespace

// internal entry points used by s2ag_gru_seq_fwd / s2ag_gru_seq_bwd (gru.hip)
int s2ag_gru_small_supported(int H) { return (H == 64 || H == 32) ? 1 : 0; }

int s2ag_gru_small_fwd(const float* gi, const float* whh, const float* bhh, float* y, float* ydrop, float* gates, int B,
                       int T, int H, float p, const unsigned long long* rng, unsigned site, hipStream_t stream) {
    const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
    dim3 grid(s2ag::cdiv(B, SNT / H), 2);
    if (H == 64)
        hipLaunchKernelGGL(gru_small_fwd_k<64>, grid, dim3(SNT), 0, stream, gi, whh, bhh, y, ydrop, gates, B, T, p, ik, rng,
                           site);
    else
        hipLaunchKernelGGL(gru_small_fwd_k<32>, grid, dim3(SNT), 0, stream, gi, whh, bhh, y, ydrop, gates, B, T, p, ik, rng,
                           site);
    S2AG_LAUNCH_CHECK();
    return 0;
}

int s2ag_gru_small_bwd(const float* dy, int lddy, int dy_dir_stride, const float* whh, const float* y,
                       const float* gates, float* dgi, float* dgh, int B, int T, int H, float p,
                       const unsigned long long* rng, unsigned site, hipStream_t stream) {
    const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
    dim3 grid(s2ag::cdiv(B, SNT / H), 2);
    if (H == 64)
        hipLaunchKernelGGL(gru_small_bwd_k<64>, grid, dim3(SNT), 0, stream, dy, lddy, dy_dir_stride, whh, y, gates, dgi, dgh,
                           B, T, p, ik, rng, site);
    else
        hipLaunchKernelGGL(gru_small_bwd_k<32>, grid, dim3(SNT), 0, stream, dy, lddy, dy_dir_stride, whh, y, gates, dgi, dgh,
                           B, T, p, ik, rng, site);
    S2AG_LAUNCH_CHECK();
    return 0;
}
