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

#ifdef S2AG_COOP_TRACE
__device__ unsigned long long g_small_trace[64 * 8];
#define SM_TR(slot)                                                                   \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && step < 64) g_small_trace[step * 8 + (slot)] = wall_clock64()
#else
#define SM_TR(slot)
#endif

constexpr int SNT = 512;  // threads per workgroup = 8 waves.  TWO threads share one (clip, unit): each holds half of the
constexpr int KSPLIT = 2; // unit's W_hh rows (96 registers at H = 64) and half of every dot product; the halves meet in
                          // one DPP exchange.  One thread per unit (192 serial FMAs per step, one wave per SIMD) left
                          // nothing to overlap the LDS reads and transcendentals with: 1.3 us per step.

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter, i.e. every
// step would wait for the acknowledgement of its own y / gate stores (~1 us) although the threads only talk through LDS.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int H>
__global__ __launch_bounds__(SNT) void gru_small_fwd_k(const float* __restrict__ gi, const float* __restrict__ whh,
                                                          const float* __restrict__ bhh, float* __restrict__ y,
                                                          float* __restrict__ ydrop, float* __restrict__ gates, int B,
                                                          int T, float drop_p, float inv_keep,
                                                          const unsigned long long* rng, unsigned site) {
    constexpr int H3 = 3 * H;
    constexpr int HK = H / KSPLIT;                      // k range of one thread
    constexpr int SBS = SNT / (KSPLIT * H);             // clips per workgroup (4 at H = 64, 8 at H = 32)
    __shared__ __attribute__((aligned(16))) float hs[2][SBS][H];
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * SBS;
    const int tid = threadIdx.x;
    const int half = tid & 1, pr = tid >> 1;            // the two lanes of a pair are neighbours (one DPP swap apart)
    const int b = pr / H, i = pr - b * H;
    const int k0 = half * HK;
    const bool valid = (b0 + b) < B;
    const float* W = whh + (size_t)dir * H3 * H;       // (3H, H) reference layout
    float wr[HK], wz[HK], wn[HK];                       // this thread's half of the unit's three gate rows
#pragma unroll
    for (int k = 0; k < HK; k += 4) {                   // 16-byte loads along the contiguous k axis
        const float4 a = *reinterpret_cast<const float4*>(W + (size_t)i * H + k0 + k);
        const float4 bq = *reinterpret_cast<const float4*>(W + (size_t)(H + i) * H + k0 + k);
        const float4 c = *reinterpret_cast<const float4*>(W + (size_t)(2 * H + i) * H + k0 + k);
        wr[k] = a.x; wr[k + 1] = a.y; wr[k + 2] = a.z; wr[k + 3] = a.w;
        wz[k] = bq.x; wz[k + 1] = bq.y; wz[k + 2] = bq.z; wz[k + 3] = bq.w;
        wn[k] = c.x; wn[k + 1] = c.y; wn[k + 2] = c.z; wn[k + 3] = c.w;
    }
    const float bhr = bhh[dir * H3 + i], bhz = bhh[dir * H3 + H + i], bhn = bhh[dir * H3 + 2 * H + i];
    if (half == 0) hs[0][b][i] = 0.f;
    SiteKey key{0, 0};
    const bool drop = ydrop != nullptr && drop_p > 0.f;
    if (drop) key = site_key(rng, site);
    __syncthreads();
    float hp = 0.f;
    // input projections are fetched ONE STEP AHEAD: with the dot products split over two threads a step's own FMAs are
    // too short to cover an L2/HBM round trip
    auto load_gi = [&](int step, float& a, float& bq, float& c) {
        a = bq = c = 0.f;
        if (valid && step < T) {
            const int t = dir ? (T - 1 - step) : step;
            const float* gp = gi + ((long long)(b0 + b) * T + t) * (2 * H3) + dir * H3;
            a = gp[i];
            bq = gp[H + i];
            c = gp[2 * H + i];
        }
    };
    float gir, giz, gin, gir_n, giz_n, gin_n;
    load_gi(0, gir, giz, gin);
    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        const int cur = step & 1;
        const long long row = (long long)(b0 + b) * T + t;
        SM_TR(0);
        load_gi(step + 1, gir_n, giz_n, gin_n);
        SM_TR(1);
        float ar = 0.f, az = 0.f, an = 0.f;
#pragma unroll
        for (int k = 0; k < HK; k += 4) {
            const float4 hv = *reinterpret_cast<const float4*>(&hs[cur][b][k0 + k]);
            ar = fmaf(hv.x, wr[k], ar); az = fmaf(hv.x, wz[k], az); an = fmaf(hv.x, wn[k], an);
            ar = fmaf(hv.y, wr[k + 1], ar); az = fmaf(hv.y, wz[k + 1], az); an = fmaf(hv.y, wn[k + 1], an);
            ar = fmaf(hv.z, wr[k + 2], ar); az = fmaf(hv.z, wz[k + 2], az); an = fmaf(hv.z, wn[k + 2], an);
            ar = fmaf(hv.w, wr[k + 3], ar); az = fmaf(hv.w, wz[k + 3], az); an = fmaf(hv.w, wn[k + 3], an);
        }
        SM_TR(2);
        ar += __shfl_xor(ar, 1, 64);                     // the pair's two halves (both lanes end up with the full sums)
        az += __shfl_xor(az, 1, 64);
        an += __shfl_xor(an, 1, 64);
        ar += bhr; az += bhz; an += bhn;
        const float r = sigmoidf_(gir + ar);
        const float z = sigmoidf_(giz + az);
        const float n = tanhf(gin + r * an);
        const float hn = (1.f - z) * n + z * hp;
        hp = hn;
        SM_TR(3);
        if (half == 0) hs[cur ^ 1][b][i] = hn;
        if (valid) {
            const long long yi = row * (2 * H) + dir * H + i;
            if (half == 0) {                             // the pair splits the stores
                y[yi] = hn;
                if (ydrop) ydrop[yi] = drop ? hn * keep_scale(key, (unsigned long long)yi, drop_p, inv_keep) : hn;
            } else if (gates) {
                float* gs = gates + ((long long)dir * B * T + row) * (4 * H);
                *reinterpret_cast<float4*>(gs + 4 * i) = make_float4(r, z, n, an);
            }
        }
        SM_TR(4);
        gir = gir_n;
        giz = giz_n;
        gin = gin_n;
        lds_barrier();
        SM_TR(5);
    }
}

template <int H>
__global__ __launch_bounds__(SNT) void gru_small_bwd_k(const float* __restrict__ dy, int lddy, int dy_dir_stride,
                                                          const float* __restrict__ whh, const float* __restrict__ y,
                                                          const float* __restrict__ gates, float* __restrict__ dgi,
                                                          float* __restrict__ dgh, int B, int T, float drop_p,
                                                          float inv_keep, const unsigned long long* rng, unsigned site) {
    constexpr int H3 = 3 * H;
    constexpr int HK3 = H3 / KSPLIT;                    // rows of this thread's part of W_hh's column i
    constexpr int SBS = SNT / (KSPLIT * H);
    __shared__ __attribute__((aligned(16))) float gs[SBS][H3];      // d(gh) of this step
    const int dir = blockIdx.y;
    const int b0 = blockIdx.x * SBS;
    const int tid = threadIdx.x;
    const int half = tid & 1, pr = tid >> 1;
    const int b = pr / H, i = pr - b * H;
    const int k0 = half * HK3;
    const bool valid = (b0 + b) < B;
    const float* W = whh + (size_t)dir * H3 * H;        // (3H, H) row-major: this pair keeps column i
    float wc[HK3];
#pragma unroll
    for (int k = 0; k < HK3; ++k) wc[k] = W[(size_t)(k0 + k) * H + i];
    SiteKey key{0, 0};
    const bool drop = drop_p > 0.f;
    if (drop) key = site_key(rng, site);
    float dh = 0.f;
    // saved gates, h_{t-1} and the incoming gradient do not depend on the recurrence: fetched one step ahead
    struct Pre {
        float g, r, z, n, hn, hp;
    };
    auto fetch = [&](int step) {
        Pre p{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (valid && step < T) {
            const int t = dir ? step : (T - 1 - step);
            const int tprev = dir ? t + 1 : t - 1;
            const long long row = (long long)(b0 + b) * T + t;
            const float* gp = gates + ((long long)dir * B * T + row) * (4 * H);
            p.g = dy[row * lddy + dir * dy_dir_stride + i];
            const float4 sv = *reinterpret_cast<const float4*>(gp + 4 * i);
            p.r = sv.x;
            p.z = sv.y;
            p.n = sv.z;
            p.hn = sv.w;
            if (tprev >= 0 && tprev < T) p.hp = y[((long long)(b0 + b) * T + tprev) * (2 * H) + dir * H + i];
        }
        return p;
    };
    Pre cu = fetch(0);
    for (int step = 0; step < T; ++step) {
        const int t = dir ? step : (T - 1 - step);
        const long long row = (long long)(b0 + b) * T + t;
        const Pre nx = fetch(step + 1);
        float dr = 0.f, dz = 0.f, dnr = 0.f, carry = 0.f;
        if (valid) {                                     // both lanes of the pair compute; they split the stores
            float g = cu.g;
            if (drop) g *= keep_scale(key, (unsigned long long)(row * (2 * H) + dir * H + i), drop_p, inv_keep);
            const float dht = dh + g;
            const float r = cu.r, z = cu.z, n = cu.n, hn = cu.hn, hp = cu.hp;
            const float dn = dht * (1.f - z) * (1.f - n * n);
            dz = dht * (hp - n) * z * (1.f - z);
            dr = dn * hn * r * (1.f - r);
            dnr = dn * r;
            if (half == 0) {
                float* gi_o = dgi + row * (2 * H3) + dir * H3;
                gi_o[i] = dr;
                gi_o[H + i] = dz;
                gi_o[2 * H + i] = dn;
            } else {
                float* gh_o = dgh + ((long long)dir * B * T + row) * H3;
                gh_o[i] = dr;
                gh_o[H + i] = dz;
                gh_o[2 * H + i] = dnr;
            }
            carry = dht * z;
        }
        if (half == 0) {
            gs[b][i] = dr;
            gs[b][H + i] = dz;
            gs[b][2 * H + i] = dnr;
        }
        lds_barrier();
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < HK3; k += 4) {
            const float4 gv = *reinterpret_cast<const float4*>(&gs[b][k0 + k]);   // broadcast within the clip
            acc = fmaf(gv.x, wc[k], acc);
            acc = fmaf(gv.y, wc[k + 1], acc);
            acc = fmaf(gv.z, wc[k + 2], acc);
            acc = fmaf(gv.w, wc[k + 3], acc);
        }
        acc += __shfl_xor(acc, 1, 64);
        dh = carry + acc;
        cu = nx;
        lds_barrier();
    }
}
}  // namespace

// internal entry points used by s2ag_gru_seq_fwd / s2ag_gru_seq_bwd (gru.hip)
int s2ag_gru_small_supported(int H) { return (H == 64 || H == 32) ? 1 : 0; }

int s2ag_gru_small_fwd(const float* gi, const float* whh, const float* bhh, float* y, float* ydrop, float* gates, int B,
                       int T, int H, float p, const unsigned long long* rng, unsigned site, hipStream_t stream) {
    const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
    dim3 grid(s2ag::cdiv(B, SNT / (KSPLIT * H)), 2);
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
    dim3 grid(s2ag::cdiv(B, SNT / (KSPLIT * H)), 2);
    if (H == 64)
        hipLaunchKernelGGL(gru_small_bwd_k<64>, grid, dim3(SNT), 0, stream, dy, lddy, dy_dir_stride, whh, y, gates, dgi, dgh,
                           B, T, p, ik, rng, site);
    else
        hipLaunchKernelGGL(gru_small_bwd_k<32>, grid, dim3(SNT), 0, stream, dy, lddy, dy_dir_stride, whh, y, gates, dgi, dgh,
                           B, T, p, ik, rng, site);
    S2AG_LAUNCH_CHECK();
    return 0;
}

#ifdef S2AG_COOP_TRACE
extern "C" int s2ag_gru_small_trace_read(unsigned long long* host64x8) {
    return (int)hipMemcpyFromSymbol(host64x8, HIP_SYMBOL(g_small_trace), sizeof(unsigned long long) * 64 * 8);
}
#endif
