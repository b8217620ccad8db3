// Cooperative GRU recurrence for large hidden sizes (H = 300 on this path): W_hh never leaves the chip.
//
// The streaming kernels in gru.hip re-read W_hh (3H*H*4 B = 1.08 MB at H = 300) from L2 on every time step --
// that stream (per-CU L2 bandwidth) bounds them at ~18 us/step.  Here a GROUP of S workgroups shares one
// (direction, 16-clip batch slice); workgroup s owns HW = 32 hidden units, i.e. 96 gate columns of W_hh, and
// keeps them for all T steps in REGISTERS as MFMA B-operands (12 waves x 38 k-steps x 1 VGPR).  Per step it
//   1. waits for the group's h_{t-1} (one relaxed agent-scope poll of a per-step arrival counter),
//   2. loads h_{t-1} (16 x H) from the exchange buffer into LDS, k-major (sc1 loads: bypass the stale L1),
//   3. multiplies 16 x H by H x 96 on the f32 MFMA pipe (v_mfma_f32_16x16x4_f32; K split over two wave halves),
//   4. applies the gate math for its 32 units, writes y / ydrop / saved gates with plain stores and its slice
//      of h_t with write-through (sc1) 8-byte stores,
//   5. drains its stores (s_waitcnt vmcnt(0) per wave), barrier, one lane bumps the step's counter.
// This is the publish/consume recipe of the CDNA4 guide (write-through payload + drained flag, sc1 consumer
// loads, no fences); results do not depend on dispatch order or XCD placement.  Every spin is bounded: on
// time-out the workgroup sets an error word and stops waiting, so a lost workgroup can never hang the GPU.
// Residency: S * ceil(B/16) * 2 workgroups of 768 threads, one per CU (160 at B = 128, H = 300 <= 256 CUs).
//
// The backward kernel has the same structure with W_hh[:, slice] (3H x 32) in registers and the group
// exchanging d(gh) (16 x 3H) per step.
#include "s2ag_common.h"

namespace {
using namespace s2ag;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int CBS = 16;          // clips per group
constexpr int HW = 32;           // hidden units per workgroup
constexpr int NW = 3 * HW;       // gate columns per workgroup
constexpr int CNT = 768;         // 12 waves
constexpr unsigned SPIN_LIMIT = 1u << 22;

typedef unsigned long long u64;

__device__ __forceinline__ void st_sc1(float* p, float a, float b) {
    const u64 bits = (u64)__float_as_uint(a) | ((u64)__float_as_uint(b) << 32);
    __hip_atomic_store(reinterpret_cast<u64*>(p), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld_sc1(const float* p) {
    const u64 bits = __hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)bits), __uint_as_float((unsigned)(bits >> 32)));
}

// one lane waits until *cnt == target (relaxed agent-scope polls); returns false on time-out
__device__ __forceinline__ bool wait_count(const int* cnt, int target, int* err) {
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) {
            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
template <int H, int HW_, int NSL>
__global__ __launch_bounds__(CNT) void gru_coop_fwd_k(const float* __restrict__ gi, const float* __restrict__ whh,
                                                      const float* __restrict__ bhh, float* __restrict__ y,
                                                      float* __restrict__ ydrop, float* __restrict__ gates,
                                                      float* xbuf, int* cnt, int* err, int B, int T, float drop_p,
                                                      float inv_keep, const unsigned long long* rng, unsigned site) {
    // NSL independent 16-clip slices per workgroup, visited round-robin inside every time step: while slice A's h_t is
    // travelling to its peers (store drain -> counter -> peers' polls -> their sc1 loads), the workgroup multiplies
    // slice B -- the exchange latency of one slice hides behind the MFMA + gate phase of the other.  The W_hh
    // registers are shared by the slices; only the LDS state is per slice.
    constexpr int H3 = 3 * H;
    constexpr int NW_ = 3 * HW_;                   // gate columns owned by this workgroup
    constexpr int NTILES = NW_ / 16;               // MFMA column tiles
    constexpr int KSPLIT = 12 / NTILES;            // wave groups splitting K (12 waves)
    constexpr int KSTEPS = (H + 3) / 4;            // MFMA k-steps over the whole K = H
    constexpr int KPW = (KSTEPS + KSPLIT - 1) / KSPLIT;
    constexpr int S = (H + HW_ - 1) / HW_;         // workgroups per group (1: no exchange at all)
    constexpr int GT = CBS * HW_ / 2;              // gate-phase threads (2 units x 1 clip each)
    static_assert(NTILES * KSPLIT == 12 && GT <= CNT, "12 waves must tile (column tiles x K groups)");
    __shared__ float hT[NSL][H * CBS];             // state per slice, k-major [k][clip]
    __shared__ float red[KSPLIT][CBS][NW_];        // partial products of the K groups (one slice at a time)
    __shared__ int ok_flag;

    const int s = blockIdx.x, dir = blockIdx.z;
    const int nbs = (B + CBS - 1) / CBS;           // 16-clip slices in the batch
    const int u0 = s * HW_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave % NTILES, kh = wave / NTILES;
    const float* W = whh + (size_t)dir * H3 * H;   // (3H, H) reference layout: W[gate row][k]
    const float* bh = bhh + dir * H3;

    // B operands of this wave, resident for the whole launch: B[k][j] = W_hh[gate col(nt*16 + j)][k]
    // (read once per launch straight from the state_dict layout -- no transposed copy of W_hh is ever made)
    const int kbeg = kh * KPW;
    float breg[KPW];
    {
        const int cl = nt * 16 + (lane & 15);      // local gate column
        const int g = cl / HW_, ul = cl - g * HW_;
        const int u = u0 + ul;
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int k = (kbeg + i) * 4 + (lane >> 4);
            breg[i] = (kbeg + i < KSTEPS && k < H && u < H) ? W[(size_t)(g * H + u) * H + k] : 0.f;
        }
    }
    for (int i = tid; i < NSL * H * CBS; i += CNT) (&hT[0][0])[i] = 0.f;
    SiteKey key{0, 0};
    const bool drop = ydrop != nullptr && drop_p > 0.f;
    if (drop) key = site_key(rng, site);
    if (tid == 0) ok_flag = 1;
    __syncthreads();

    // gate-phase mapping: clip fastest (coalesced exchange stores)
    const int gb = tid & 15, gup = tid >> 4;
    const int gu = u0 + 2 * gup;                   // first of the two units of this thread
    const bool gate_lane = tid < GT && gu < H;
    float bias_r[2], bias_z[2], bias_n[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bias_r[j] = gate_lane ? bh[gu + j] : 0.f;
        bias_z[j] = gate_lane ? bh[H + gu + j] : 0.f;
        bias_n[j] = gate_lane ? bh[2 * H + gu + j] : 0.f;
    }

    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
            const int bsl = blockIdx.y * NSL + sl;
            if (bsl >= nbs) continue;              // uniform per workgroup
            const int b0 = bsl * CBS;
            const int nb = min(CBS, B - b0);
            const bool gate_thread = gate_lane && gb < nb;
            const int group = dir * nbs + bsl;
            float* X = xbuf + (size_t)group * 2 * H * CBS; // [parity][H/2][CBS][2]
            int* C = cnt + (size_t)group * T;
            float* hs = hT[sl];
            const long long row = (long long)(b0 + gb) * T + t;
            // prefetch this step's input projections (latency hides behind the wait + MFMA)
            float2 gir = make_float2(0.f, 0.f), giz = gir, gin = gir;
            if (gate_thread) {
                const float* gp = gi + row * (2 * H3) + dir * H3 + gu;
                gir = *reinterpret_cast<const float2*>(gp);
                giz = *reinterpret_cast<const float2*>(gp + H);
                gin = *reinterpret_cast<const float2*>(gp + 2 * H);
            }
            if (S > 1 && step > 0) {
                if (tid == 0 && ok_flag) {
                    if (!wait_count(C + (step - 1), S, err)) ok_flag = 0;
                }
                __syncthreads();
                const float* Xp = X + (size_t)((step - 1) & 1) * H * CBS;
                for (int i = tid; i < (H / 2) * CBS; i += CNT) {
                    const float2 v = ld_sc1(Xp + 2 * i);          // i = kp*CBS + clip
                    const int kp = i / CBS, c = i - kp * CBS;
                    hs[(2 * kp) * CBS + c] = v.x;
                    hs[(2 * kp + 1) * CBS + c] = v.y;
                }
                __syncthreads();
            }
            // ---- 16 x H times H x 16 per wave on the f32 MFMA pipe
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < KPW; ++i) {
                const int k = (kbeg + i) * 4 + (lane >> 4);
                const float a = (kbeg + i < KSTEPS && k < H) ? hs[k * CBS + (lane & 15)] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, breg[i], acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) red[kh][(lane >> 4) * 4 + q][nt * 16 + (lane & 15)] = acc[q];
            __syncthreads();
            // ---- gates for 2 units x 1 clip per thread
            if (gate_thread) {
                const int ul = 2 * gup;
                float hn2[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int u = gu + j, c = ul + j;
                    float ghr = bias_r[j], ghz = bias_z[j], ghn = bias_n[j];
#pragma unroll
                    for (int q = 0; q < KSPLIT; ++q) {
                        ghr += red[q][gb][c];
                        ghz += red[q][gb][HW_ + c];
                        ghn += red[q][gb][2 * HW_ + c];
                    }
                    const float r = sigmoidf_((j ? gir.y : gir.x) + ghr);
                    const float z = sigmoidf_((j ? giz.y : giz.x) + ghz);
                    const float n = tanhf((j ? gin.y : gin.x) + r * ghn);
                    const float hp = hs[u * CBS + gb];
                    const float hnew = (1.f - z) * n + z * hp;
                    hn2[j] = hnew;
                    const long long yi = row * (2 * H) + dir * H + u;
                    y[yi] = hnew;
                    if (ydrop)
                        ydrop[yi] = drop ? hnew * keep_scale(key, (unsigned long long)yi, drop_p, inv_keep) : hnew;
                    if (gates) {
                        float* gs = gates + ((long long)dir * B * T + row) * (4 * H);
                        gs[u] = r;
                        gs[H + u] = z;
                        gs[2 * H + u] = n;
                        gs[3 * H + u] = ghn;
                    }
                    if (S == 1) hs[u * CBS + gb] = hnew;      // single workgroup per group: the state never leaves LDS
                }
                if (S > 1 && step + 1 < T)
                    st_sc1(X + (size_t)(step & 1) * H * CBS + ((size_t)(gu >> 1) * CBS + gb) * 2, hn2[0], hn2[1]);
            } else if (S > 1 && gate_lane && step + 1 < T) {
                // clips beyond B: publish zeros so the group's state stays defined
                st_sc1(X + (size_t)(step & 1) * H * CBS + ((size_t)(gu >> 1) * CBS + gb) * 2, 0.f, 0.f);
            }
            if (S > 1) {
                if (step + 1 < T) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
                    __syncthreads();
                    if (tid == 0) __hip_atomic_fetch_add(C + step, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    __syncthreads();                                  // red[] is reused by the next slice
                }
            } else {
                __syncthreads();
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward through time
// ---------------------------------------------------------------------------------------------------------
template <int H, int HW_>
__global__ __launch_bounds__(CNT) void gru_coop_bwd_k(const float* __restrict__ dy, int lddy, int dy_dir_stride,
                                                      const float* __restrict__ whh, const float* __restrict__ y,
                                                      const float* __restrict__ gates, float* __restrict__ dgi,
                                                      float* __restrict__ dgh, float* xbuf, int* cnt, int* err, int B,
                                                      int T, float drop_p, float inv_keep,
                                                      const unsigned long long* rng, unsigned site) {
    constexpr int H3 = 3 * H;
    constexpr int NT_B = HW_ / 16;                 // dh column tiles
    constexpr int NKS = 12 / NT_B;                 // K slices (12 waves)
    constexpr int KSTEPS = (H3 + 3) / 4;           // k-steps over K = 3H
    constexpr int KPW = (KSTEPS + NKS - 1) / NKS;  // k-steps per wave
    constexpr int S = (H + HW_ - 1) / HW_;
    constexpr int GT = CBS * HW_ / 2;
    extern __shared__ __attribute__((aligned(16))) float smem_bwd[];   // up to 72 KB: above the static limit
    float* gT = smem_bwd;                                      // [3H][CBS] d(gh) of this step, whole group, k-major
    float (*red)[CBS][HW_] = reinterpret_cast<float (*)[CBS][HW_]>(gT + H3 * CBS);          // [NKS][CBS][HW_]
    float (*dh)[HW_] = reinterpret_cast<float (*)[HW_]>(gT + H3 * CBS + NKS * CBS * HW_);   // [CBS][HW_] running dL/dh
    __shared__ int ok_flag;

    const int s = blockIdx.x, bsl = blockIdx.y, dir = blockIdx.z;
    const int nbs = gridDim.y;
    const int b0 = bsl * CBS;
    const int nb = min(CBS, B - b0);
    const int u0 = s * HW_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave % NT_B, ksl = wave / NT_B;
    const int group = dir * nbs + bsl;
    float* X = xbuf + (size_t)group * 2 * H3 * CBS;   // [parity][3H/2][CBS][2]
    int* C = cnt + (size_t)group * T;
    const float* W = whh + (size_t)dir * H3 * H;      // (3H, H) row-major

    const int kbeg = ksl * KPW;
    float breg[KPW];
    {
        const int j = u0 + nt * 16 + (lane & 15);     // dh column owned by this lane
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int k = (kbeg + i) * 4 + (lane >> 4);
            breg[i] = (kbeg + i < KSTEPS && k < H3 && j < H) ? W[(size_t)k * H + j] : 0.f;
        }
    }
    for (int i = tid; i < CBS * HW_; i += CNT) (&dh[0][0])[i] = 0.f;   // dh is contiguous [CBS][HW_]
    for (int i = tid; i < H3 * CBS; i += CNT) gT[i] = 0.f;
    SiteKey key{0, 0};
    const bool drop = drop_p > 0.f;
    if (drop) key = site_key(rng, site);
    if (tid == 0) ok_flag = 1;
    __syncthreads();

    const int gb = tid & 15, gup = tid >> 4;
    const int gu = u0 + 2 * gup;
    const bool gate_lane = tid < GT && gu < H;
    const bool gate_thread = gate_lane && gb < nb;

    for (int step = 0; step < T; ++step) {
        const int t = dir ? step : (T - 1 - step);
        const int tprev = dir ? t + 1 : t - 1;
        // ---- phase A: gate gradients of this workgroup's units, published to the group
        if (gate_lane) {
            float o[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            if (gate_thread) {
                const long long row = (long long)(b0 + gb) * T + t;
                const float* gp = gates + ((long long)dir * B * T + row) * (4 * H);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int u = gu + j, c = 2 * gup + j;
                    float g = dy[row * lddy + dir * dy_dir_stride + u];
                    if (drop) g *= keep_scale(key, (unsigned long long)(row * (2 * H) + dir * H + u), drop_p, inv_keep);
                    const float dht = dh[gb][c] + g;
                    const float r = gp[u], z = gp[H + u], n = gp[2 * H + u], hn = gp[3 * H + u];
                    float hp = 0.f;
                    if (tprev >= 0 && tprev < T) hp = y[((long long)(b0 + gb) * T + tprev) * (2 * H) + dir * H + u];
                    const float dn = dht * (1.f - z) * (1.f - n * n);
                    const float dz = dht * (hp - n) * z * (1.f - z);
                    const float dr = dn * hn * r * (1.f - r);
                    float* gi_o = dgi + row * (2 * H3) + dir * H3;
                    gi_o[u] = dr;
                    gi_o[H + u] = dz;
                    gi_o[2 * H + u] = dn;
                    float* gh_o = dgh + ((long long)dir * B * T + row) * H3;
                    gh_o[u] = dr;
                    gh_o[H + u] = dz;
                    gh_o[2 * H + u] = dn * r;
                    o[0][j] = dr;
                    o[1][j] = dz;
                    o[2][j] = dn * r;
                    dh[gb][c] = dht * z;
                }
            }
            if (step + 1 < T) {
                if (S > 1) {
                    float* Xp = X + (size_t)(step & 1) * H3 * CBS;
#pragma unroll
                    for (int gI = 0; gI < 3; ++gI)
                        st_sc1(Xp + ((size_t)((gI * H + gu) >> 1) * CBS + gb) * 2, o[gI][0], o[gI][1]);
                } else {
#pragma unroll
                    for (int gI = 0; gI < 3; ++gI) {
                        gT[(gI * H + gu) * CBS + gb] = o[gI][0];
                        gT[(gI * H + gu + 1) * CBS + gb] = o[gI][1];
                    }
                }
            }
        }
        if (step + 1 == T) break;                   // the last step's dh is never consumed
        if (S > 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(C + step, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ok_flag && !wait_count(C + step, S, err)) ok_flag = 0;
            }
            __syncthreads();
            const float* Xp = X + (size_t)(step & 1) * H3 * CBS;
            for (int i = tid; i < (H3 / 2) * CBS; i += CNT) {
                const float2 v = ld_sc1(Xp + 2 * i);
                const int kp = i / CBS, c = i - kp * CBS;
                gT[(2 * kp) * CBS + c] = v.x;
                gT[(2 * kp + 1) * CBS + c] = v.y;
            }
        }
        __syncthreads();
        // ---- phase B: dh[16 x HW] += d(gh)[16 x 3H] . W_hh[3H x HW]
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int k = (kbeg + i) * 4 + (lane >> 4);
            const float a = (kbeg + i < KSTEPS && k < H3) ? gT[k * CBS + (lane & 15)] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, breg[i], acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) red[ksl][(lane >> 4) * 4 + q][nt * 16 + (lane & 15)] = acc[q];
        __syncthreads();
        for (int e = tid; e < CBS * HW_; e += CNT) {
            const int b = e / HW_, c = e - b * HW_;
            float v = dh[b][c];
#pragma unroll
            for (int q = 0; q < NKS; ++q) v += red[q][b][c];
            dh[b][c] = v;
        }
        __syncthreads();
    }
}

// Counters are re-armed by a KERNEL (s2ag::zero_async), not by hipMemsetAsync: inside a replayed hipGraph (ROCm 7.2)
// a memset node was observed not to be ordered against the polling kernel that follows it -- peers lost arrivals
// from the second replay on (tools/diag_coop4.py reproduces it).
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
}  // namespace

// ---------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------
// H = 300: groups of 10 workgroups (32 units each) exchanging h per step;  H = 64: one workgroup per group
// (64 units, no exchange at all -- the small discriminator GRU simply lives in registers + LDS)
extern "C" int s2ag_gru_coop_supported(int H) { return (H == 300 || H == 64) ? 1 : 0; }

extern "C" long long s2ag_gru_coop_workspace_bytes(int B, int T, int H, int backward) {
    if (B <= 0 || T <= 0 || !s2ag_gru_coop_supported(H)) return 0;
    const size_t groups = (size_t)2 * cdiv(B, CBS);
    const size_t payload = (backward ? 3 : 1) * (size_t)H * CBS * 2 * sizeof(float);
    return (long long)(align_up(groups * payload, 256) + align_up(groups * T * sizeof(int), 256) + 256);
}

namespace {
struct Ws {
    float* x;
    int* cnt;
    int* err;
    size_t zero_bytes;
};
Ws carve(void* ws, int B, int T, int H, int backward) {
    const size_t groups = (size_t)2 * cdiv(B, CBS);
    const size_t payload = (backward ? 3 : 1) * (size_t)H * CBS * 2 * sizeof(float);
    char* p = static_cast<char*>(ws);
    Ws w;
    w.x = reinterpret_cast<float*>(p);
    p += align_up(groups * payload, 256);
    w.cnt = reinterpret_cast<int*>(p);
    w.zero_bytes = align_up(groups * T * sizeof(int), 256) + 256;
    p += align_up(groups * T * sizeof(int), 256);
    w.err = reinterpret_cast<int*>(p);
    return w;
}
}  // namespace

extern "C" int s2ag_gru_coop_fwd(const float* gi, const float* whh, const float* bhh, float* y, float* ydrop,
                                 float* gates, int B, int T, int H, const s2ag_epilogue* e, void* workspace,
                                 void* stream) {
    if (!gi || !whh || !bhh || !y || !workspace || B <= 0 || T <= 0) return S2AG_E_BADARG;
    if (!s2ag_gru_coop_supported(H)) return S2AG_E_UNSUPPORTED;
    const float p = (e && ydrop) ? e->drop_p : 0.f;
    if (p > 0.f && !e->rng) return S2AG_E_BADARG;
    Ws w = carve(workspace, B, T, H, 0);
    if (H == 300) {   // H = 64 runs one workgroup per group: no counters to re-arm
        hipError_t ze = zero_async(w.cnt, w.zero_bytes, (hipStream_t)stream);
        if (ze != hipSuccess) return (int)ze;
    }
    const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const unsigned long long* rg = e ? e->rng : nullptr;
    const unsigned site = e ? e->site : 0u;
    if (H == 300)
        // NSL = 1: interleaving two slices per workgroup (NSL = 2) measured 0.46 vs 0.25 ms -- the step is bound by the
        // workgroup's own dependent chain (sc1 loads -> MFMA -> gates -> store drain), not by waiting for peers
        hipLaunchKernelGGL((gru_coop_fwd_k<300, 32, 1>), dim3(10, cdiv(B, CBS), 2), dim3(CNT), 0,
                           (hipStream_t)stream, gi, whh, bhh, y, ydrop, gates, w.x, w.cnt, w.err, B, T, p, ik, rg, site);
    else
        hipLaunchKernelGGL((gru_coop_fwd_k<64, 64, 1>), dim3(1, cdiv(B, CBS), 2), dim3(CNT), 0, (hipStream_t)stream, gi,
                           whh, bhh, y, ydrop, gates, w.x, w.cnt, w.err, B, T, p, ik, rg, site);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_gru_coop_bwd(const float* dy, int lddy, int dy_dir_stride, const float* whh, const float* y,
                                 const float* gates, float* dgi, float* dgh, int B, int T, int H,
                                 const s2ag_epilogue* e, void* workspace, void* stream) {
    if (!dy || !whh || !y || !gates || !dgi || !dgh || !workspace || B <= 0 || T <= 0) return S2AG_E_BADARG;
    if (!s2ag_gru_coop_supported(H)) return S2AG_E_UNSUPPORTED;
    const float p = e ? e->drop_p : 0.f;
    if (p > 0.f && !e->rng) return S2AG_E_BADARG;
    Ws w = carve(workspace, B, T, H, 1);
    if (H == 300) {   // H = 64 runs one workgroup per group: no counters to re-arm
        hipError_t ze = zero_async(w.cnt, w.zero_bytes, (hipStream_t)stream);
        if (ze != hipSuccess) return (int)ze;
    }
    const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const unsigned long long* rg = e ? e->rng : nullptr;
    const unsigned site = e ? e->site : 0u;
    if (H == 300) {
        constexpr size_t smem = sizeof(float) * (3 * 300 * CBS + 6 * CBS * 32 + CBS * 32);
        static bool granted = false;
        if (!granted) {
            hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(gru_coop_bwd_k<300, 32>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (ae != hipSuccess) return (int)ae;
            granted = true;
        }
        hipLaunchKernelGGL((gru_coop_bwd_k<300, 32>), dim3(10, cdiv(B, CBS), 2), dim3(CNT), smem, (hipStream_t)stream,
                           dy, lddy, dy_dir_stride, whh, y, gates, dgi, dgh, w.x, w.cnt, w.err, B, T, p, ik, rg, site);
    } else {
        constexpr size_t smem = sizeof(float) * (3 * 64 * CBS + 3 * CBS * 64 + CBS * 64);
        hipLaunchKernelGGL((gru_coop_bwd_k<64, 64>), dim3(1, cdiv(B, CBS), 2), dim3(CNT), smem, (hipStream_t)stream, dy,
                           lddy, dy_dir_stride, whh, y, gates, dgi, dgh, w.x, w.cnt, w.err, B, T, p, ik, rg, site);
    }
    S2AG_LAUNCH_CHECK();
    return 0;
}

/* 1 if the last cooperative launch that used `workspace` timed out waiting for a peer (results invalid) */
extern "C" int s2ag_gru_coop_error_word_offset(int B, int T, int H, int backward, long long* offset) {
    if (!offset || !s2ag_gru_coop_supported(H)) return S2AG_E_BADARG;
    const size_t groups = (size_t)2 * cdiv(B, CBS);
    const size_t payload = (backward ? 3 : 1) * (size_t)H * CBS * 2 * sizeof(float);
    *offset = (long long)(align_up(groups * payload, 256) + align_up(groups * T * sizeof(int), 256));
    return 0;
}
