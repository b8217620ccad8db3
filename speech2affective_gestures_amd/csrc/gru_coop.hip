// Cooperative GRU recurrence for large hidden sizes (H = 300 on this path): W_hh never leaves the chip.
//
// The streaming kernels in gru.hip re-read W_hh (3H*H*4 B = 1.08 MB at H = 300) from L2 on every time step --
// that stream (per-CU L2 bandwidth) bounds them at ~18 us/step.  Here a GROUP of S = 10 workgroups shares one
// (direction, 16-clip batch slice); workgroup s owns HW = 32 hidden units, i.e. 96 gate columns of W_hh, and
// keeps them for all T steps in REGISTERS as MFMA B-operands (12 waves x 38 k-steps x 1 VGPR).  Per step it
//   1. polls the group's h_{t-1} out of the exchange buffer: every value travels as an 8-byte (value, step tag) cell,
//      so the payload is its own flag -- a consumer re-reads (L1-bypassing agent-scope loads) until every cell it
//      needs carries this step's tag, then puts the values into LDS ([clip][k], pitch = 4 mod 64 words: both the
//      lane-contiguous fill and the MFMA A-operand reads are bank-conflict free),
//   2. multiplies 16 x H by H x 96 on the f32 MFMA pipe (v_mfma_f32_16x16x4_f32; K split over two wave groups),
//   3. applies the gate math, one (clip, unit) per thread with the unit index running along the lanes, publishes its
//      slice of h_t FIRST (write-through 8-byte cell stores, fire and forget) and then writes y / ydrop / saved gates
//      with plain stores -- all of them 128-byte contiguous per clip row.  h_{t-1} of the thread's own unit stays in a
//      register.
// There is no counter, no store drain and no flag round trip on the step's critical path (the first version published
// with "stores -> vmcnt(0) -> barrier -> counter atomic -> peers poll the counter -> peers load"; see DESIGN.md for
// the per-phase timeline, tools/diag_coop_trace.py measures it).  8-byte stores are single-copy atomic, so a cell is
// either the old (value, tag) or the new one; the buffer is double-buffered by step parity, and a producer can only
// reach step s+2 after every peer has consumed step s (it needs their s+1 cells, which they publish after reading s),
// so a cell is never overwritten while someone still needs it.  Tags are step+1 and the buffer is cleared by a
// kernel before the launch, so stale cells of an earlier launch can never match.  Results do not depend on dispatch
// order or XCD placement.  Every spin is bounded: on time-out the thread sets an error word and stops waiting, so a
// lost workgroup can never hang the GPU.
// Residency: S * ceil(B/16) * 2 workgroups of 768 threads, one per CU (160 at B = 128, H = 300 <= 256 CUs).
//
// The backward kernel has the same structure with W_hh[:, slice] (3H x 32) in registers and the group
// exchanging d(gh) (16 x 3H) cells per step; the running dL/dh of a (clip, unit) lives in its thread's register.
//
// Default build: the products run on the bf16 matrix pipe from bf16-piece splits of the fp32 operands (two pieces = three
// products, 16 mantissa bits, by default; three pieces = six products, fp32-equivalent, with S2AG_GRU_SPLIT=3; see "fp32
// products on the bf16 matrix pipe" below); the f32-MFMA kernels described above remain selectable (S2AG_GRU_SPLIT=0) and
// are what the split kernels are validated against.
#include <stdlib.h>

#include "s2ag_common.h"

namespace {
using namespace s2ag;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int CBS = 16;          // clips per group
constexpr int CNT = 768;         // 12 waves
constexpr unsigned SPIN_LIMIT = 1u << 22;

typedef unsigned long long u64;

// Phase timestamps of workgroup (0,0,0) for tools/diag_coop_trace.py (compiled only with -DS2AG_COOP_TRACE)
#ifdef S2AG_COOP_TRACE
__device__ u64 g_coop_trace[64 * 8];
#define COOP_TR(slot)                                                                                     \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && step < 64)          \
    g_coop_trace[step * 8 + (slot)] = wall_clock64()
#define COOP_TRV(slot, val)                                                                               \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && step < 64)          \
    g_coop_trace[step * 8 + (slot)] = (u64)(val)
#else
#define COOP_TR(slot)
#define COOP_TRV(slot, val)
#endif

// exchange cell = (float value, u32 tag) in one 8-byte word
__device__ __forceinline__ void st_cell(u64* p, float v, unsigned tag) {
    __hip_atomic_store(p, (u64)__float_as_uint(v) | ((u64)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 ld_cell(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every thread fetches its NL cells of the group's (CBS x ROW)-cell matrix until all carry `tag`, then puts the values
// into LDS at dst[clip * PITCH + k].  All NL loads of a round are in flight together.  Returns false on time-out.
// (Measured and dropped: placing a group's workgroups on one XCD and polling with L1-only-bypassing loads that hit
// the XCD's L2 -- correct thanks to the tags, but 15 % slower than the memory-side loads, and peers strided across
// the grid dead-lock two concurrent launches.  See DESIGN.md.)
template <int NL, int ROW, int PITCH>
__device__ __forceinline__ bool gather_cells(const u64* X, unsigned tag, float* dst, int* err, unsigned* rounds) {
    constexpr int n = CBS * ROW;
    const int tid = threadIdx.x;
    u64 v[NL];
    unsigned spins = 0;
    for (;;) {
        bool all = true;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int i = tid + j * CNT;
            v[j] = ld_cell(X + (i < n ? i : n - 1));
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) all = all && ((unsigned)(v[j] >> 32) == tag);
        if (all) break;
        if (++spins > SPIN_LIMIT) {
            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int i = tid + j * CNT;
        if (i < n) {
            const int c = i / ROW, k = i - c * ROW;
            dst[c * PITCH + k] = __uint_as_float((unsigned)v[j]);
        }
    }
    *rounds = spins;
    return true;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

constexpr int lds_pitch(int n) {          // smallest pitch >= n with pitch % 64 == 4: rows land 4 banks apart
    int p = n;
    while (p % 64 != 4) ++p;
    return p;
}

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
template <int H, int HW_>
__global__ __launch_bounds__(CNT) void gru_coop_fwd_k(const float* __restrict__ gi, const float* __restrict__ whh,
                                                      const float* __restrict__ bhh, float* __restrict__ y,
                                                      float* __restrict__ ydrop, float* __restrict__ gates,
                                                      u64* xbuf, int* err, int B, int T, float drop_p,
                                                      float inv_keep, const unsigned long long* rng, unsigned site) {
    constexpr int H3 = 3 * H;
    constexpr int NW_ = 3 * HW_;                   // gate columns owned by this workgroup
    constexpr int NTILES = NW_ / 16;               // MFMA column tiles
    constexpr int KSPLIT = 12 / NTILES;            // wave groups splitting K (12 waves)
    constexpr int KSTEPS = (H + 3) / 4;            // MFMA k-steps over the whole K = H
    constexpr int KPW = (KSTEPS + KSPLIT - 1) / KSPLIT;
    constexpr int GT = CBS * HW_;                  // gate-phase threads: one (clip, unit) each
    constexpr int HP = lds_pitch(KSPLIT * KPW * 4);// LDS row pitch of the state (covers the padded K range: no
                                                   // bounds test -- and no branch -- in the MFMA loop)
    constexpr int RP = NW_ + 4;                    // pitch of the partial products (rows 0,4,8,12 -> 16 banks apart)
    constexpr int NLF = (H * CBS + CNT - 1) / CNT; // exchange cells per thread
    static_assert(NTILES * KSPLIT == 12 && GT <= CNT && HW_ == 32, "12 waves must tile (column tiles x K groups)");
    __shared__ float hs[CBS * HP];                 // h_{t-1} of the whole group, [clip][k]
    __shared__ float red[KSPLIT][CBS][RP];         // partial products of the K groups

    // peers of a group are consecutive in dispatch order (blockIdx.x fastest): a launch that is only partly resident
    // (another cooperative launch holds the other CUs) still has complete groups that run to the end and free CUs
    const int s = blockIdx.x, bsl = blockIdx.y, dir = blockIdx.z;
    const int nbs = gridDim.y;                     // 16-clip slices in the batch
    const int group = dir * nbs + bsl;
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
        const int g = cl / HW_, cu = cl - g * HW_;
        const int wu = u0 + cu;
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int k = (kbeg + i) * 4 + (lane >> 4);
            breg[i] = (kbeg + i < KSTEPS && k < H && wu < H) ? W[(size_t)(g * H + wu) * H + k] : 0.f;
        }
    }
    for (int i = tid; i < CBS * HP; i += CNT) hs[i] = 0.f;
    SiteKey key{0, 0};
    const bool drop = ydrop != nullptr && drop_p > 0.f;
    if (drop) key = site_key(rng, site);
    bool ok = true;                                // this thread has not timed out

    // gate-phase mapping: unit along the lanes (every global access of a wave = two 128-byte row segments)
    const int gc = tid >> 5, ul = tid & 31;
    const int u = u0 + ul;
    const int b0 = bsl * CBS;
    const int nb = min(CBS, B - b0);
    const bool gate_lane = tid < GT && u < H;
    const bool gate_thread = gate_lane && gc < nb;
    const float bias_r = gate_lane ? bh[u] : 0.f, bias_z = gate_lane ? bh[H + u] : 0.f,
                bias_n = gate_lane ? bh[2 * H + u] : 0.f;
    u64* X = xbuf + (size_t)group * 2 * H * CBS;   // [parity][clip][H] cells
    float hp = 0.f;                                // h_{t-1} of this thread's (clip, unit)
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        const long long row = (long long)(b0 + gc) * T + t;
        // prefetch this step's input projections (latency hides behind the gather + MFMA)
        float gir = 0.f, giz = 0.f, gin = 0.f;
        if (gate_thread) {
            const float* gp = gi + row * (2 * H3) + dir * H3 + u;
            gir = gp[0];
            giz = gp[H];
            gin = gp[2 * H];
        }
        COOP_TR(0);
        if (step > 0) {
            // h_{t-1} of the whole group: cells tagged `step` (published at step-1 with tag (step-1)+1)
            unsigned rounds = 0;
            if (ok)
                ok = gather_cells<NLF, H, HP>(X + (size_t)((step - 1) & 1) * H * CBS, (unsigned)step, hs, err, &rounds);
            COOP_TR(1);
            COOP_TRV(7, rounds);
            __syncthreads();
        }
        COOP_TR(2);
#ifdef S2AG_COOP_TRACE
        const u64 cyc0 = clock64();
#endif
        // ---- 16 x H times H x 16 per wave on the f32 MFMA pipe
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};      // two independent accumulation chains
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int k = (kbeg + i) * 4 + (lane >> 4);
            const float a = hs[(lane & 15) * HP + k];     // k >= H: zero pad columns (never written), breg = 0 too
            if (i & 1)
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, breg[i], acc1, 0, 0, 0);
            else
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, breg[i], acc, 0, 0, 0);
        }
        acc += acc1;
#pragma unroll
        for (int q = 0; q < 4; ++q) red[kh][(lane >> 4) * 4 + q][nt * 16 + (lane & 15)] = acc[q];
#ifdef S2AG_COOP_TRACE
        const u64 cyc1 = clock64();
#endif
        __syncthreads();
        COOP_TR(3);
        COOP_TRV(6, cyc1 - cyc0);
        // ---- gates: one (clip, unit) per thread.  No barrier closes the step: red[] is next written after the
        // post-gather barrier of step+1, which every thread reaches only after this phase; hs[] is not read here.
        if (gate_lane) {
            float hnew = 0.f, r = 0.f, z = 0.f, n = 0.f, ghn = bias_n;
            if (gate_thread) {
                float ghr = bias_r, ghz = bias_z;
#pragma unroll
                for (int q = 0; q < KSPLIT; ++q) {
                    ghr += red[q][gc][ul];
                    ghz += red[q][gc][HW_ + ul];
                    ghn += red[q][gc][2 * HW_ + ul];
                }
                r = sigmoidf_(gir + ghr);
                z = sigmoidf_(giz + ghz);
                n = tanhf(gin + r * ghn);
                hnew = (1.f - z) * n + z * hp;
                hp = hnew;
            }
            COOP_TR(4);
            // publish first: the peers are waiting for exactly these cells (clips beyond B publish zeros)
            if (step + 1 < T)
                st_cell(X + (size_t)(step & 1) * H * CBS + (size_t)gc * H + u, hnew, (unsigned)(step + 1));
            if (gate_thread) {
                const long long yi = row * (2 * H) + dir * H + u;
                y[yi] = hnew;
                if (ydrop) ydrop[yi] = drop ? hnew * keep_scale(key, (unsigned long long)yi, drop_p, inv_keep) : hnew;
                if (gates) {
                    float* gs = gates + ((long long)dir * B * T + row) * (4 * H);
                    *reinterpret_cast<float4*>(gs + 4 * u) = make_float4(r, z, n, ghn);
                }
            }
        }
        COOP_TR(5);
    }
}


// ---------------------------------------------------------------------------------------------------------
// fp32 products on the bf16 matrix pipe
// ---------------------------------------------------------------------------------------------------------
// The step's matrix product runs on the f32 MFMA (v_mfma_f32_16x16x4_f32: 32 cycles per SIMD for 1024 MACs) in the
// kernels above -- 3 waves x 38 of them per SIMD are 1.5 us of the ~4.8 us step.  The bf16 pipe does 8192 MACs in ~17
// cycles.  An fp32 value splits EXACTLY into bf16 pieces v = p0 + p1 (+ p2) + residual, |residual| <= 2^-17 |v| with two
// pieces, 2^-25 |v| with three (each piece is the bf16 rounding of what the previous ones left; the differences are exact
// in fp32).  The product a*b is then the sum of the piece products, accumulated in fp32 inside the MFMA:
//   NP = 2:  a0b0 + a0b1 + a1b0                          (3 MFMAs, dropped terms <= 2^-16 |ab|: 16-bit mantissa)
//   NP = 3:  ... + a1b1 + a0b2 + a2b0                    (6 MFMAs, dropped terms <= 2^-24 |ab|: fp32-equivalent)
// W_hh is split once per launch into registers; a new state value is split once by the thread that produces it and
// travels through the exchange already split: a cell is (p0, p1, p2, 16-bit step tag) in one 8-byte word.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ unsigned bf16_rn(float v) {            // bf16 bits of v, round to nearest even (finite v)
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
template <int NP>
__device__ __forceinline__ void split_bf16(float v, unsigned (&pc)[3]) {
    pc[0] = bf16_rn(v);
    pc[1] = pc[2] = 0u;
    if (NP == 1) return;                            // bf16 step mode: one piece, one product
    float r = v - __uint_as_float(pc[0] << 16);
    pc[1] = bf16_rn(r);
    if (NP == 3) {
        r -= __uint_as_float(pc[1] << 16);
        pc[2] = bf16_rn(r);
    }
}
__device__ __forceinline__ void st_cell_sp(u64* p, const unsigned (&pc)[3], unsigned tag) {
    const u64 w = (u64)(pc[0] | (pc[1] << 16)) | ((u64)(pc[2] | (tag << 16)) << 32);
    __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// gather_cells for split cells: piece p of (clip c, index k) goes to dst[p * PLANE + c * PITCH + k] (bf16 elements)
template <int NL, int ROW, int PITCH, int PLANE, int NP>
__device__ __forceinline__ bool gather_cells_sp(const u64* X, unsigned tag, unsigned short* dst, int* err,
                                                unsigned* rounds) {
    constexpr int n = CBS * ROW;
    const int tid = threadIdx.x;
    u64 v[NL];
    unsigned spins = 0;
    for (;;) {
        bool all = true;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const int i = tid + j * CNT;
            v[j] = ld_cell(X + (i < n ? i : n - 1));
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) all = all && ((unsigned)(v[j] >> 48) == tag);
        if (all) break;
        if (++spins > SPIN_LIMIT) {
            __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int i = tid + j * CNT;
        if (i < n) {
            const int c = i / ROW, k = i - c * ROW;
            unsigned short* d = dst + c * PITCH + k;
            d[0] = (unsigned short)v[j];
            if (NP >= 2) d[PLANE] = (unsigned short)(v[j] >> 16);
            if (NP == 3) d[2 * PLANE] = (unsigned short)(v[j] >> 32);
        }
    }
    *rounds = spins;
    return true;
}

// forward recurrence, products on the bf16 pipe (structure, mappings and exchange protocol of gru_coop_fwd_k)
template <int H, int HW_, int NP>
__global__ __launch_bounds__(CNT) void gru_coop_fwd_sp_k(const float* __restrict__ gi, const float* __restrict__ whh,
                                                         const float* __restrict__ bhh, float* __restrict__ y,
                                                         float* __restrict__ ydrop, float* __restrict__ gates,
                                                         u64* xbuf, int* err, int B, int T, float drop_p,
                                                         float inv_keep, const unsigned long long* rng, unsigned site) {
    constexpr int H3 = 3 * H;
    constexpr int NW_ = 3 * HW_;
    constexpr int NTILES = NW_ / 16;
    constexpr int KSPLIT = 12 / NTILES;
    constexpr int KSTEPS = (H + 31) / 32;          // MFMA k-steps (K = 32 each) over the whole K = H
    constexpr int KPW = (KSTEPS + KSPLIT - 1) / KSPLIT;
    constexpr int GT = CBS * HW_;
    constexpr int HPB = KSPLIT * KPW * 32 + 8;     // LDS row pitch in bf16 elements: 656 B = 164 words, rows 4 banks
                                                   // apart (a lane's 16-byte operand chunk covers 4 banks)
    constexpr int PLANE = CBS * HPB;               // one piece of the whole state
    constexpr int RP = NW_ + 4;
    constexpr int NLF = (H * CBS + CNT - 1) / CNT;
    static_assert(NTILES * KSPLIT == 12 && GT <= CNT && HW_ == 32, "12 waves must tile (column tiles x K groups)");
    static_assert((HPB * 2) % 16 == 0 && (HPB / 2) % 32 == 4, "operand chunks 16-byte aligned, rows 4 banks apart");
    __shared__ __attribute__((aligned(16))) unsigned short hsb[NP * PLANE];   // h_{t-1} pieces, [piece][clip][k]
    __shared__ float red[KSPLIT][CBS][RP];

    const int s = blockIdx.x, bsl = blockIdx.y, dir = blockIdx.z;
    const int nbs = gridDim.y;
    const int group = dir * nbs + bsl;
    const int u0 = s * HW_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave % NTILES, kh = wave / NTILES;
    const float* W = whh + (size_t)dir * H3 * H;
    const float* bh = bhh + dir * H3;

    // B operands: lane = (gate column nt*16 + (lane & 15), k chunk (lane >> 4) * 8 ... + 7 of every 32-wide k-step)
    const int kbeg = kh * KPW;
    u32x4 breg[KPW][NP];
    {
        const int cl = nt * 16 + (lane & 15);
        const int g = cl / HW_, cu = cl - g * HW_;
        const int wu = u0 + cu;
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int k0 = (kbeg + i) * 32 + (lane >> 4) * 8;
            unsigned pk[8][3];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                const float w = (k < H && wu < H) ? W[(size_t)(g * H + wu) * H + k] : 0.f;
                split_bf16<NP>(w, pk[j]);
            }
#pragma unroll
            for (int pc = 0; pc < NP; ++pc)
#pragma unroll
                for (int d = 0; d < 4; ++d) breg[i][pc][d] = pk[2 * d][pc] | (pk[2 * d + 1][pc] << 16);
        }
    }
    for (int i = tid; i < NP * PLANE / 2; i += CNT) reinterpret_cast<unsigned*>(hsb)[i] = 0u;
    SiteKey key{0, 0};
    const bool drop = ydrop != nullptr && drop_p > 0.f;
    if (drop) key = site_key(rng, site);
    bool ok = true;

    const int gc = tid >> 5, ul = tid & 31;
    const int u = u0 + ul;
    const int b0 = bsl * CBS;
    const int nb = min(CBS, B - b0);
    const bool gate_lane = tid < GT && u < H;
    const bool gate_thread = gate_lane && gc < nb;
    const float bias_r = gate_lane ? bh[u] : 0.f, bias_z = gate_lane ? bh[H + u] : 0.f,
                bias_n = gate_lane ? bh[2 * H + u] : 0.f;
    u64* X = xbuf + (size_t)group * 2 * H * CBS;
    float hp = 0.f;
    const unsigned short* a_base = hsb + (lane & 15) * HPB + kbeg * 32 + (lane >> 4) * 8;
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
        const long long row = (long long)(b0 + gc) * T + t;
        float gir = 0.f, giz = 0.f, gin = 0.f;
        if (gate_thread) {
            const float* gp = gi + row * (2 * H3) + dir * H3 + u;
            gir = gp[0];
            giz = gp[H];
            gin = gp[2 * H];
        }
        COOP_TR(0);
        if (step > 0) {
            unsigned rounds = 0;
            if (ok)
                ok = gather_cells_sp<NLF, H, HPB, PLANE, NP>(X + (size_t)((step - 1) & 1) * H * CBS, (unsigned)step, hsb,
                                                             err, &rounds);
            COOP_TR(1);
            COOP_TRV(7, rounds);
            __syncthreads();
        }
        COOP_TR(2);
#ifdef S2AG_COOP_TRACE
        const u64 cyc0 = clock64();
#endif
        // ---- 16 x H times H x 16 per wave: KPW k-steps x (3 or 6) piece products, independent accumulation chains
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            bf16x8 a[NP], b[NP];
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
                a[pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(a_base + pc * PLANE + i * 32));
                b[pc] = __builtin_bit_cast(bf16x8, breg[i][pc]);
            }
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc0, 0, 0, 0);
            if constexpr (NP >= 2) {
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[NP - 1 ? 1 : 0], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[NP - 1 ? 1 : 0], b[0], acc2, 0, 0, 0);
            }
            if constexpr (NP == 3) {
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc2, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc1, 0, 0, 0);
            }
        }
        const f32x4 acc = (acc1 + acc2) + acc0;     // small terms first
#pragma unroll
        for (int q = 0; q < 4; ++q) red[kh][(lane >> 4) * 4 + q][nt * 16 + (lane & 15)] = acc[q];
#ifdef S2AG_COOP_TRACE
        const u64 cyc1 = clock64();
#endif
        __syncthreads();
        COOP_TR(3);
        COOP_TRV(6, cyc1 - cyc0);
        if (gate_lane) {
            float hnew = 0.f, r = 0.f, z = 0.f, n = 0.f, ghn = bias_n;
            if (gate_thread) {
                float ghr = bias_r, ghz = bias_z;
#pragma unroll
                for (int q = 0; q < KSPLIT; ++q) {
                    ghr += red[q][gc][ul];
                    ghz += red[q][gc][HW_ + ul];
                    ghn += red[q][gc][2 * HW_ + ul];
                }
                r = sigmoidf_(gir + ghr);
                z = sigmoidf_(giz + ghz);
                n = tanhf(gin + r * ghn);
                hnew = (1.f - z) * n + z * hp;
                hp = hnew;
            }
            COOP_TR(4);
            if (step + 1 < T) {
                unsigned pc[3];
                split_bf16<NP>(hnew, pc);
                st_cell_sp(X + (size_t)(step & 1) * H * CBS + (size_t)gc * H + u, pc, (unsigned)(step + 1));
            }
            if (gate_thread) {
                const long long yi = row * (2 * H) + dir * H + u;
                y[yi] = hnew;
                if (ydrop) ydrop[yi] = drop ? hnew * keep_scale(key, (unsigned long long)yi, drop_p, inv_keep) : hnew;
                if (gates) {
                    float* gs = gates + ((long long)dir * B * T + row) * (4 * H);
                    *reinterpret_cast<float4*>(gs + 4 * u) = make_float4(r, z, n, ghn);
                }
            }
        }
        COOP_TR(5);
    }
}


// Two 16-clip slices per workgroup (B > 16): with the product on the bf16 pipe a step is ~1.3 us of waiting for the
// peers' cells against ~1.8 us of work, so a workgroup alternates between two independent slices -- while slice A's new
// state travels, slice B (whose cells arrived during A's work) is multiplied, gated and published.  Same W_hh registers
// serve both.  A launch then needs 10 x ceil(B/32) x 2 = 80 workgroups at B = 128 instead of 160: two generator passes
// (or a pass and the weight-gradient GEMMs of the previous layer) fit on the chip side by side.
// Roles: waves 0..3 do the gate math, two (clip, unit) pairs per thread, and all the stores; waves 4..11 gather -- a
// wave's loads cannot be consumed before its earlier stores are acknowledged (one in-order counter), and the gate waves
// have just published the other slice.  Ten cells per gathering thread, all in flight at once.
// Several PASSES per launch: the trainer runs the generator three times per step on the same weights (for D, for the
// loss, with shuffled speakers; processor_v2.py:798, :823, :909); run layer by layer in lockstep, the recurrences of a
// layer are independent slices of one launch -- blockIdx.y = pass * pairs_per_pass + slice pair -- with their own
// input projections, outputs, exchange cells and noise snapshot.  A launch is bound by the latency of its T exchanges,
// not by its width: three passes cost one pass's time.
constexpr int COOP_MAX_PASSES = 4;
struct CoopFwdPasses {
    const float* gi[COOP_MAX_PASSES];
    float* y[COOP_MAX_PASSES];
    float* ydrop[COOP_MAX_PASSES];
    float* gates[COOP_MAX_PASSES];
    const unsigned long long* rng[COOP_MAX_PASSES];
    int pairs_per_pass;
    long long cells_per_pass;
};

template <int H, int HW_, int NP, int NS>
__global__ __launch_bounds__(CNT) void gru_coop_fwd_sp2_k(const CoopFwdPasses P, const float* __restrict__ whh,
                                                          const float* __restrict__ bhh, u64* xbuf, int* err, int B,
                                                          int T, float drop_p, float inv_keep, unsigned site) {
    const int pass = blockIdx.y / P.pairs_per_pass;
    const float* __restrict__ gi = P.gi[pass];
    float* __restrict__ y = P.y[pass];
    float* __restrict__ ydrop = P.ydrop[pass];
    float* __restrict__ gates = P.gates[pass];
    const unsigned long long* rng = P.rng[pass];
    xbuf += (size_t)pass * P.cells_per_pass;
    constexpr int H3 = 3 * H;
    constexpr int NW_ = 3 * HW_;
    constexpr int NTILES = NW_ / 16;
    constexpr int KSPLIT = 12 / NTILES;
    constexpr int KSTEPS = (H + 31) / 32;
    constexpr int KPW = (KSTEPS + KSPLIT - 1) / KSPLIT;
    constexpr int CPT = 2;                         // (clip, unit) pairs per gate thread
    constexpr int GT = CBS * HW_ / CPT;            // gate threads (waves 0..3)
    constexpr int NGA = CNT - GT;                  // gathering threads (waves 4..11)
    constexpr int HPB = KSPLIT * KPW * 32 + 8;
    constexpr int PLANE = CBS * HPB;
    constexpr int RP = NW_ + 4;
    constexpr int NLG = (H * CBS + NGA - 1) / NGA; // exchange cells per gathering thread (10)
    static_assert(NTILES * KSPLIT == 12 && GT < CNT && HW_ == 32, "12 waves must tile (column tiles x K groups)");
    static_assert((HPB * 2) % 16 == 0 && (HPB / 2) % 32 == 4, "operand chunks 16-byte aligned, rows 4 banks apart");
    extern __shared__ __attribute__((aligned(16))) unsigned short hsb2[];          // [slice][piece][clip][k]
    __shared__ float red[KSPLIT][CBS][RP];

    const int s = blockIdx.x, bpair = blockIdx.y - pass * P.pairs_per_pass, dir = blockIdx.z;
    const int nbs = (B + CBS - 1) / CBS;           // 16-clip slices in the batch
    const int u0 = s * HW_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave % NTILES, kh = wave / NTILES;
    const float* W = whh + (size_t)dir * H3 * H;
    const float* bh = bhh + dir * H3;

    const int kbeg = kh * KPW;
    u32x4 breg[KPW][NP];
    {
        const int cl = nt * 16 + (lane & 15);
        const int g = cl / HW_, cu = cl - g * HW_;
        const int wu = u0 + cu;
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int k0 = (kbeg + i) * 32 + (lane >> 4) * 8;
            unsigned pk[8][3];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                const float w = (k < H && wu < H) ? W[(size_t)(g * H + wu) * H + k] : 0.f;
                split_bf16<NP>(w, pk[j]);
            }
#pragma unroll
            for (int pc = 0; pc < NP; ++pc)
#pragma unroll
                for (int d = 0; d < 4; ++d) breg[i][pc][d] = pk[2 * d][pc] | (pk[2 * d + 1][pc] << 16);
        }
    }
    for (int i = tid; i < NS * NP * PLANE / 2; i += CNT) reinterpret_cast<unsigned*>(hsb2)[i] = 0u;
    SiteKey key{0, 0};
    const bool drop = ydrop != nullptr && drop_p > 0.f;
    if (drop) key = site_key(rng, site);
    bool ok = true;

    const int gc = tid >> 5, ul = tid & 31;        // gate threads: clips gc and gc + 8 of a slice
    const int u = u0 + ul;
    const bool gate_lane = tid < GT && u < H;
    const float bias_r = gate_lane ? bh[u] : 0.f, bias_z = gate_lane ? bh[H + u] : 0.f,
                bias_n = gate_lane ? bh[2 * H + u] : 0.f;
    int b0[NS];
    bool live[NS];
    u64* X[NS];
    float hp[NS][CPT];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
        const int bsl = bpair * NS + sl;
        live[sl] = bsl < nbs;                      // block-uniform: a trailing half-empty pair skips its second slice
        b0[sl] = bsl * CBS;
        X[sl] = xbuf + (size_t)(dir * nbs + (live[sl] ? bsl : 0)) * 2 * H * CBS;
#pragma unroll
        for (int c2 = 0; c2 < CPT; ++c2) hp[sl][c2] = 0.f;
    }
    const int a_off = (lane & 15) * HPB + kbeg * 32 + (lane >> 4) * 8;
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir ? (T - 1 - step) : step;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            if (!live[sl]) continue;
            unsigned short* hs = hsb2 + sl * (NP * PLANE);
            const int nb = min(CBS, B - b0[sl]);
            float gir[CPT], giz[CPT], gin[CPT];
#pragma unroll
            for (int c2 = 0; c2 < CPT; ++c2) {
                gir[c2] = giz[c2] = gin[c2] = 0.f;
                const int c = gc + c2 * (CBS / CPT);
                if (gate_lane && c < nb) {
                    const float* gp = gi + ((long long)(b0[sl] + c) * T + t) * (2 * H3) + dir * H3 + u;
                    gir[c2] = gp[0];
                    giz[c2] = gp[H];
                    gin[c2] = gp[2 * H];
                }
            }
            if (step > 0 && tid >= GT) {
                // cells tagged `step` of this slice, fetched by the store-free waves
                const u64* Xs = X[sl] + (size_t)((step - 1) & 1) * H * CBS;
                constexpr int n = CBS * H;
                const int gt = tid - GT;
                u64 v[NLG];
                unsigned spins = 0;
                while (ok) {
                    bool all = true;
#pragma unroll
                    for (int j = 0; j < NLG; ++j) {
                        const int i = gt + j * NGA;
                        v[j] = ld_cell(Xs + (i < n ? i : n - 1));
                    }
#pragma unroll
                    for (int j = 0; j < NLG; ++j) all = all && ((unsigned)(v[j] >> 48) == (unsigned)step);
                    if (all) break;
                    if (++spins > SPIN_LIMIT) {
                        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = false;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
#pragma unroll
                for (int j = 0; j < NLG; ++j) {
                    const int i = gt + j * NGA;
                    if (i < n) {
                        const int c = i / H, k = i - c * H;
                        unsigned short* d = hs + c * HPB + k;
                        d[0] = (unsigned short)v[j];
                        if (NP >= 2) d[PLANE] = (unsigned short)(v[j] >> 16);
                        if (NP == 3) d[2 * PLANE] = (unsigned short)(v[j] >> 32);
                    }
                }
            }
            __syncthreads();      // also at step 0: red[] may still be read by the other slice's gate phase
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < KPW; ++i) {
                bf16x8 a[NP], b[NP];
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) {
                    a[pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(hs + a_off + pc * PLANE + i * 32));
                    b[pc] = __builtin_bit_cast(bf16x8, breg[i][pc]);
                }
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc0, 0, 0, 0);
                if constexpr (NP >= 2) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[NP - 1 ? 1 : 0], acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[NP - 1 ? 1 : 0], b[0], acc2, 0, 0, 0);
                }
                if constexpr (NP == 3) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc2, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc1, 0, 0, 0);
                }
            }
            const f32x4 acc = (acc1 + acc2) + acc0;
#pragma unroll
            for (int q = 0; q < 4; ++q) red[kh][(lane >> 4) * 4 + q][nt * 16 + (lane & 15)] = acc[q];
            __syncthreads();
            // gates; red[] is next written behind the other slice's (or the next step's) pre-MFMA barrier
            if (gate_lane) {
                float hnew[CPT], r[CPT], z[CPT], n[CPT], ghn[CPT];
#pragma unroll
                for (int c2 = 0; c2 < CPT; ++c2) {
                    const int c = gc + c2 * (CBS / CPT);
                    hnew[c2] = r[c2] = z[c2] = n[c2] = 0.f;
                    ghn[c2] = bias_n;
                    if (c < nb) {
                        float ghr = bias_r, ghz = bias_z;
#pragma unroll
                        for (int q = 0; q < KSPLIT; ++q) {
                            ghr += red[q][c][ul];
                            ghz += red[q][c][HW_ + ul];
                            ghn[c2] += red[q][c][2 * HW_ + ul];
                        }
                        r[c2] = sigmoidf_(gir[c2] + ghr);
                        z[c2] = sigmoidf_(giz[c2] + ghz);
                        n[c2] = tanhf(gin[c2] + r[c2] * ghn[c2]);
                        hnew[c2] = (1.f - z[c2]) * n[c2] + z[c2] * hp[sl][c2];
                        hp[sl][c2] = hnew[c2];
                    }
                }
                if (step + 1 < T) {
#pragma unroll
                    for (int c2 = 0; c2 < CPT; ++c2) {
                        unsigned pc[3];
                        split_bf16<NP>(hnew[c2], pc);
                        st_cell_sp(X[sl] + (size_t)(step & 1) * H * CBS + (size_t)(gc + c2 * (CBS / CPT)) * H + u, pc,
                                   (unsigned)(step + 1));
                    }
                }
#pragma unroll
                for (int c2 = 0; c2 < CPT; ++c2) {
                    const int c = gc + c2 * (CBS / CPT);
                    if (c < nb) {
                        const long long row = (long long)(b0[sl] + c) * T + t;
                        const long long yi = row * (2 * H) + dir * H + u;
                        y[yi] = hnew[c2];
                        if (ydrop)
                            ydrop[yi] = drop ? hnew[c2] * keep_scale(key, (unsigned long long)yi, drop_p, inv_keep)
                                             : hnew[c2];
                        if (gates) {
                            float* gs = gates + ((long long)dir * B * T + row) * (4 * H);
                            *reinterpret_cast<float4*>(gs + 4 * u) = make_float4(r[c2], z[c2], n[c2], ghn[c2]);
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// backward through time
// ---------------------------------------------------------------------------------------------------------
// dL/dh_{t-1}[16 x H] = carry + d(gh)_t[16 x 3H] . W_hh[3H x H].  The contraction is partitioned over K, not over the
// output: a workgroup multiplies ITS OWN 96 rows of d(gh)_t (the gate gradients of its 32 units, which it has just
// computed -- no exchange needed for them) by W_hh[own rows, all H columns] and the group reduce-scatters the 16 x H
// partial sums: every workgroup publishes 16 x 300 tagged cells and reads the 10 x 16 x 32 cells of its own columns
// (41 KB per workgroup and step).  The output-partitioned first version had every workgroup read the whole group's
// d(gh) -- 14 400 cells, 115 KB, 18 MB per step over all groups: the gather ran at the memory-side fabric's ~4 TB/s and
// was 4 of the step's 8.8 us.
template <int H, int HW_, int NP>
__global__ __launch_bounds__(CNT) void gru_coop_bwd_k(const float* __restrict__ dy, int lddy, int dy_dir_stride,
                                                      const float* __restrict__ whh, const float* __restrict__ y,
                                                      const float* __restrict__ gates, float* __restrict__ dgi,
                                                      float* __restrict__ dgh, u64* xbuf, int* err, int B,
                                                      int T, float drop_p, float inv_keep,
                                                      const unsigned long long* rng, unsigned site) {
    constexpr int H3 = 3 * H;
    constexpr int S = (H + HW_ - 1) / HW_;         // workgroups (= producers) per group
    constexpr int NTL = (H + 15) / 16;             // 16-column tiles of the partial product (19)
    constexpr int NITEM = NTL * 3;                 // work items: (column tile, gate) = 8 MFMAs each
    constexpr int IPW = (NITEM + 11) / 12;         // items per wave (5)
    constexpr int KS = HW_ / 4;                    // k-steps per gate (8)
    constexpr int GT = CBS * HW_;
    constexpr int GOP = lds_pitch(3 * HW_);        // LDS pitch of the own d(gh) rows (132)
    constexpr int RP = NTL * 16 + 4;               // pitch of the per-gate partial products (308)
    // NP > 0: products on the bf16 pipe from NP pieces per fp32 value (see gru_coop_fwd_sp_k); own d(gh) then sits in
    // LDS as bf16 pieces [piece][clip][k'], pitch 104 elements = 52 words: 8 rows' 16-byte operand chunks tile the banks
    constexpr int GOPB = 3 * HW_ + 8;
    constexpr int GO_FLOATS = NP ? (NP * CBS * GOPB) / 2 : CBS * GOP;
    static_assert(HW_ == 32 && GT <= CNT, "one (clip, unit) per gate thread");
    static_assert((GOPB * 2) % 16 == 0 && GO_FLOATS % 4 == 0, "16-byte operand chunks");
    extern __shared__ __attribute__((aligned(16))) float smem_bwd[];
    float* gO = smem_bwd;                                       // [CBS][GOP]: own d(gh), k' = gate*32 + unit
    unsigned short* gOb = reinterpret_cast<unsigned short*>(smem_bwd);
    float (*red)[CBS][RP] = reinterpret_cast<float (*)[CBS][RP]>(smem_bwd + GO_FLOATS);   // [3][CBS][RP]

    const int s = blockIdx.x, bsl = blockIdx.y, dir = blockIdx.z;
    const int nbs = gridDim.y;
    const int b0 = bsl * CBS;
    const int nb = min(CBS, B - b0);
    const int u0 = s * HW_;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = dir * nbs + bsl;
    constexpr size_t XPAR = (size_t)S * CBS * H;      // cells per parity: [producer][clip][column]
    u64* X = xbuf + (size_t)group * 2 * XPAR;
    const float* W = whh + (size_t)dir * H3 * H;      // (3H, H) row-major

    // B operands: item it = wave + 12*i -> (column tile nt = it / 3, gate g = it % 3); B[k][j] = W[g*H + u0 + k][nt*16 + j]
    float breg[NP ? 1 : IPW][NP ? 1 : KS];
    u32x4 bsp[NP ? IPW : 1][NP ? NP : 1];          // split: one K = 32 step per item, lane = (column, units (lane>>4)*8..+7)
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int it = wave + 12 * i;
        const int nt = it / 3, g = it - nt * 3;
        const int col = nt * 16 + (lane & 15);
        if constexpr (NP == 0) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int ul = ks * 4 + (lane >> 4);
                const int u = u0 + ul;
                breg[i][ks] = (it < NITEM && u < H && col < H) ? W[(size_t)(g * H + u) * H + col] : 0.f;
            }
        } else {
            unsigned pk[8][3];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int u = u0 + (lane >> 4) * 8 + j;
                const float w = (it < NITEM && u < H && col < H) ? W[(size_t)(g * H + u) * H + col] : 0.f;
                split_bf16<(NP ? NP : 2)>(w, pk[j]);
            }
#pragma unroll
            for (int pc = 0; pc < NP; ++pc)
#pragma unroll
                for (int d = 0; d < 4; ++d) bsp[i][pc][d] = pk[2 * d][pc] | (pk[2 * d + 1][pc] << 16);
        }
    }
    for (int i = tid; i < GO_FLOATS; i += CNT) gO[i] = 0.f;    // pad columns / clips beyond B stay zero
    SiteKey key{0, 0};
    const bool drop = drop_p > 0.f;
    if (drop) key = site_key(rng, site);
    bool ok = true;

    const int gc = tid >> 5, ul = tid & 31;
    const int u = u0 + ul;
    const bool gate_lane = tid < GT && u < H;
    const bool gate_thread = gate_lane && gc < nb;
    float dh = 0.f;                                   // running dL/dh_t of this thread's (clip, unit)

    // Operands of the gate gradients (saved gates, h_{t-1}, incoming dy) do not depend on the recurrence: they are
    // fetched one step ahead, so their HBM/L2 latency sits under the previous step's MFMA phase.
    struct Pre {
        float g, r, z, n, hn, hp;
    };
    auto fetch = [&](int step) {
        Pre p{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (gate_thread && step < T) {
            const int t = dir ? step : (T - 1 - step);
            const int tprev = dir ? t + 1 : t - 1;
            const long long row = (long long)(b0 + gc) * T + t;
            const float* gp = gates + ((long long)dir * B * T + row) * (4 * H);
            p.g = dy[row * lddy + dir * dy_dir_stride + u];
            const float4 sv = *reinterpret_cast<const float4*>(gp + 4 * u);
            p.r = sv.x;
            p.z = sv.y;
            p.n = sv.z;
            p.hn = sv.w;
            if (tprev >= 0 && tprev < T) p.hp = y[((long long)(b0 + gc) * T + tprev) * (2 * H) + dir * H + u];
        }
        return p;
    };
    Pre cur = fetch(0);
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir ? step : (T - 1 - step);
        // ---- phase A: gate gradients of this workgroup's units -> global (dgi, dgh) and LDS (own rows of d(gh))
        COOP_TR(0);
        float carry = 0.f;
        if (gate_lane) {
            float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f;
            const long long row = (long long)(b0 + gc) * T + t;
            if (gate_thread) {
                float g = cur.g;
                if (drop) g *= keep_scale(key, (unsigned long long)(row * (2 * H) + dir * H + u), drop_p, inv_keep);
                const float dht = dh + g;
                const float r = cur.r, z = cur.z, n = cur.n, hn = cur.hn, hp = cur.hp;
                dn = dht * (1.f - z) * (1.f - n * n);
                dz = dht * (hp - n) * z * (1.f - z);
                dr = dn * hn * r * (1.f - r);
                dnr = dn * r;
                carry = dht * z;
                float* gi_o = dgi + row * (2 * H3) + dir * H3;
                gi_o[u] = dr;
                gi_o[H + u] = dz;
                gi_o[2 * H + u] = dn;
                float* gh_o = dgh + ((long long)dir * B * T + row) * H3;
                gh_o[u] = dr;
                gh_o[H + u] = dz;
                gh_o[2 * H + u] = dnr;
            }
            if constexpr (NP == 0) {
                gO[gc * GOP + ul] = dr;
                gO[gc * GOP + HW_ + ul] = dz;
                gO[gc * GOP + 2 * HW_ + ul] = dnr;
            } else {
                unsigned pr[3], pz[3], pn[3];
                split_bf16<(NP ? NP : 2)>(dr, pr);
                split_bf16<(NP ? NP : 2)>(dz, pz);
                split_bf16<(NP ? NP : 2)>(dnr, pn);
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) {
                    unsigned short* d = gOb + pc * (CBS * GOPB) + gc * GOPB + ul;
                    d[0] = (unsigned short)pr[pc];
                    d[HW_] = (unsigned short)pz[pc];
                    d[2 * HW_] = (unsigned short)pn[pc];
                }
            }
        }
        if (step + 1 == T) break;                   // the last step's dh is never consumed
        COOP_TR(1);
        __syncthreads();
        COOP_TR(2);
        cur = fetch(step + 1);                      // latency hides under the MFMA phase
        // ---- phase B: partial[16 x H] of this workgroup = own d(gh)[16 x 96] . W_hh[own 96 rows, :]
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const int it = wave + 12 * i;
            if (it >= NITEM) break;                 // wave-uniform
            const int nt = it / 3, g = it - nt * 3;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if constexpr (NP == 0) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const float a = gO[(lane & 15) * GOP + g * HW_ + ks * 4 + (lane >> 4)];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, breg[i][ks], acc, 0, 0, 0);
                }
            } else {
                bf16x8 a[NP ? NP : 1], b[NP ? NP : 1];
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) {
                    a[pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                                                           gOb + pc * (CBS * GOPB) + (lane & 15) * GOPB + g * HW_ +
                                                           (lane >> 4) * 8));
                    b[pc] = __builtin_bit_cast(bf16x8, bsp[i][pc]);
                }
                f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
                if constexpr (NP >= 2) {
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[NP - 1 ? 1 : 0], acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[NP - 1 ? 1 : 0], b[0], acc1, 0, 0, 0);
                }
                if constexpr (NP == 3) {
                    f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc2, 0, 0, 0);
                    acc1 += acc2;
                }
                acc += acc1;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) red[g][(lane >> 4) * 4 + q][nt * 16 + (lane & 15)] = acc[q];
        }
        __syncthreads();
        COOP_TR(3);
        // ---- publish the 16 x H partial sums (three gate slabs added) as cells tagged step+1: [producer][clip][column]
        {
            u64* Xp = X + (size_t)(step & 1) * XPAR + (size_t)s * CBS * H;
            for (int i = tid; i < CBS * H; i += CNT) {
                const int c = i / H, j = i - c * H;
                st_cell(Xp + i, red[0][c][j] + red[1][c][j] + red[2][c][j], (unsigned)(step + 1));
            }
        }
        COOP_TR(4);
        // ---- reduce-scatter: this thread's (clip, unit) column from every producer of the group
        if (gate_lane) {
            const u64* Xc = X + (size_t)(step & 1) * XPAR + (size_t)gc * H + u;
            u64 v[S];
            unsigned spins = 0;
            while (ok) {
                bool all = true;
#pragma unroll
                for (int q = 0; q < S; ++q) v[q] = ld_cell(Xc + (size_t)q * CBS * H);
#pragma unroll
                for (int q = 0; q < S; ++q) all = all && ((unsigned)(v[q] >> 32) == (unsigned)(step + 1));
                if (all) break;
                if (++spins > SPIN_LIMIT) {
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = false;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            float acc = carry;
#pragma unroll
            for (int q = 0; q < S; ++q) acc += __uint_as_float((unsigned)v[q]);
            dh = acc;
            COOP_TRV(7, spins);
        }
        COOP_TR(5);
        // no barrier closes the step: gO[] is rewritten in phase A of step+1, after every wave finished phase B (barrier
        // above); red[] is rewritten in phase B of step+1, behind that step's first barrier, which a thread reaches only
        // after its part of the publish loop
    }
}

// The exchange buffer is cleared by a KERNEL (s2ag::zero_async), not by hipMemsetAsync: inside a replayed hipGraph
// (ROCm 7.2) a memset node was observed not to be ordered against the polling kernel that follows it
// (tools/diag_coop4.py reproduces it).
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline size_t coop_payload_bytes(int B, int H, int backward) {
    const size_t groups = (size_t)2 * cdiv(B, CBS);
    const size_t producers = backward ? (size_t)((H + 31) / 32) : 1;      // backward: a 16 x H slab per producer
    return align_up(groups * producers * 2 * (size_t)H * CBS * sizeof(u64), 256);   // [group][parity][...] cells
}

// option GRU_SPLIT = 0: f32 MFMA; 2 (default) / 3: fp32 products from 2 / 3 bf16 pieces on the bf16 pipe (see above); 1: the
// bf16 step mode's single piece.  s2ag_gru_coop_set_split_pieces overrides it for an extent (precision contexts, tests).
int g_split_override = -1;
inline int coop_split_pieces() {
    const int n = s2ag::option(s2ag::OPT_GRU_SPLIT);
    return g_split_override >= 0 ? g_split_override : ((n >= 1 && n <= 3) ? n : 0);
}

// two 16-clip slices per forward workgroup whenever B > 16 (see gru_coop_fwd_sp2_k; the one-slice launch of 160 workgroups
// lost its A/B inside the step, r01-j)
inline constexpr bool coop_two_slices() { return true; }

// sticky time-out flag of the process (s2ag_gru_coop_set_error_flag): when set, every launch reports a peer time-out
// THERE (never cleared by the library) instead of in its own workspace word, so a trainer reads one word per step
int* g_sticky_err = nullptr;

struct Ws {
    u64* x;
    int* err;
    size_t zero_bytes;
};
Ws carve(void* ws, int B, int H, int backward) {
    char* p = static_cast<char*>(ws);
    Ws w;
    w.x = reinterpret_cast<u64*>(p);
    w.err = g_sticky_err ? g_sticky_err : reinterpret_cast<int*>(p + coop_payload_bytes(B, H, backward));
    w.zero_bytes = coop_payload_bytes(B, H, backward) + 256;     // cells (tag 0 = never valid) and the error word
    return w;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------
// H = 300: groups of 10 workgroups (32 units each) exchanging h per step.  (Small hidden sizes -- the
// discriminators' H = 64 -- live entirely in registers: gru_small.hip.)
extern "C" int s2ag_gru_coop_supported(int H) { return H == 300 ? 1 : 0; }
extern "C" int s2ag_gru_coop_split_pieces(void) { return coop_split_pieces(); }
extern "C" int s2ag_gru_coop_fwd_slices(int B) { return (coop_split_pieces() != 0 && B > CBS && coop_two_slices()) ? 2 : 1; }
extern "C" int s2ag_gru_coop_split_override(void) { return g_split_override; }
extern "C" int s2ag_gru_coop_set_split_pieces(int pieces) {
    const int prev = g_split_override;          // the OVERRIDE (-1: none), so that restoring it un-pins option GRU_SPLIT again
    g_split_override = (pieces >= 1 && pieces <= 3) ? pieces : (pieces == 0 ? 0 : -1);
    return prev;
}

extern "C" int s2ag_gru_coop_set_error_flag(int* device_word) {
    g_sticky_err = device_word;
    return 0;
}

extern "C" long long s2ag_gru_coop_workspace_bytes(int B, int T, int H, int backward) {
    if (B <= 0 || T <= 0 || !s2ag_gru_coop_supported(H)) return 0;
    return (long long)(coop_payload_bytes(B, H, backward) + 256);
}

namespace {
// one launch of the two-slice kernel over ``n`` passes (see CoopFwdPasses); workspace = n exchange buffers + error word
int launch_fwd_sp2(int n, const float* const* gi, const float* whh, const float* bhh, float* const* y,
                   float* const* ydrop, float* const* gates, int B, int T, float p,
                   const unsigned long long* const* rng, unsigned site, void* workspace, hipStream_t stream) {
    const size_t per_pass = coop_payload_bytes(B, 300, 0);
    u64* x = reinterpret_cast<u64*>(workspace);
    int* err = g_sticky_err ? g_sticky_err : reinterpret_cast<int*>(static_cast<char*>(workspace) + n * per_pass);
    hipError_t ze = zero_async(x, n * per_pass + 256, stream);
    if (ze != hipSuccess) return (int)ze;
    const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const int np = coop_split_pieces();
    int smem2 = 2 * np * CBS * (5 * 2 * 32 + 8) * 2;                   // [slice][piece][clip][k] bf16
    const void* fn2 = np == 3   ? reinterpret_cast<const void*>(gru_coop_fwd_sp2_k<300, 32, 3, 2>)
                      : np == 2 ? reinterpret_cast<const void*>(gru_coop_fwd_sp2_k<300, 32, 2, 2>)
                                : reinterpret_cast<const void*>(gru_coop_fwd_sp2_k<300, 32, 1, 2>);
    static bool granted2[4] = {false, false, false, false};
    if (!granted2[np]) {
        hipError_t ae = hipFuncSetAttribute(fn2, hipFuncAttributeMaxDynamicSharedMemorySize, smem2);
        if (ae != hipSuccess) return (int)ae;
        granted2[np] = true;
    }
    CoopFwdPasses P{};
    for (int i = 0; i < n; ++i) {
        P.gi[i] = gi[i];
        P.y[i] = y[i];
        P.ydrop[i] = ydrop ? ydrop[i] : nullptr;
        P.gates[i] = gates ? gates[i] : nullptr;
        P.rng[i] = rng ? rng[i] : nullptr;
    }
    P.pairs_per_pass = cdiv(B, 2 * CBS);
    P.cells_per_pass = (long long)(per_pass / sizeof(u64));
    const dim3 grid2(10, n * P.pairs_per_pass, 2);
    if (np == 3)
        hipLaunchKernelGGL((gru_coop_fwd_sp2_k<300, 32, 3, 2>), grid2, dim3(CNT), smem2, stream, P, whh, bhh, x, err, B, T,
                           p, ik, site);
    else if (np == 2)
        hipLaunchKernelGGL((gru_coop_fwd_sp2_k<300, 32, 2, 2>), grid2, dim3(CNT), smem2, stream, P, whh, bhh, x, err, B, T,
                           p, ik, site);
    else
        hipLaunchKernelGGL((gru_coop_fwd_sp2_k<300, 32, 1, 2>), grid2, dim3(CNT), smem2, stream, P, whh, bhh, x, err, B, T,
                           p, ik, site);
    S2AG_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int s2ag_gru_coop_fwd(const float* gi, const float* whh, const float* bhh, float* y, float* ydrop,
                                 float* gates, int B, int T, int H, const s2ag_epilogue* e, void* workspace,
                                 void* stream) {
    if (!gi || !whh || !bhh || !y || !workspace || B <= 0 || T <= 0) return S2AG_E_BADARG;
    if (!s2ag_gru_coop_supported(H)) return S2AG_E_UNSUPPORTED;
    const float p = (e && ydrop) ? e->drop_p : 0.f;
    if (p > 0.f && !e->rng) return S2AG_E_BADARG;
    const unsigned long long* rg = e ? e->rng : nullptr;
    const unsigned site = e ? e->site : 0u;
    if (coop_split_pieces() != 0 && B > CBS && coop_two_slices())
        return launch_fwd_sp2(1, &gi, whh, bhh, &y, &ydrop, &gates, B, T, p, &rg, site, workspace, (hipStream_t)stream);
    Ws w = carve(workspace, B, H, 0);
    hipError_t ze = zero_async(w.x, w.zero_bytes, (hipStream_t)stream);
    if (ze != hipSuccess) return (int)ze;
    const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const dim3 grid(10, cdiv(B, CBS), 2);
    switch (coop_split_pieces()) {
        case 1:
            hipLaunchKernelGGL((gru_coop_fwd_sp_k<300, 32, 1>), grid, dim3(CNT), 0, (hipStream_t)stream, gi, whh, bhh, y,
                               ydrop, gates, w.x, w.err, B, T, p, ik, rg, site);
            break;
        case 2:
            hipLaunchKernelGGL((gru_coop_fwd_sp_k<300, 32, 2>), grid, dim3(CNT), 0, (hipStream_t)stream, gi, whh, bhh, y,
                               ydrop, gates, w.x, w.err, B, T, p, ik, rg, site);
            break;
        case 3:
            hipLaunchKernelGGL((gru_coop_fwd_sp_k<300, 32, 3>), grid, dim3(CNT), 0, (hipStream_t)stream, gi, whh, bhh, y,
                               ydrop, gates, w.x, w.err, B, T, p, ik, rg, site);
            break;
        default:
            hipLaunchKernelGGL((gru_coop_fwd_k<300, 32>), grid, dim3(CNT), 0, (hipStream_t)stream, gi, whh, bhh, y, ydrop,
                               gates, w.x, w.err, B, T, p, ik, rg, site);
    }
    S2AG_LAUNCH_CHECK();
    return 0;
}

// The same layer of up to four passes over the SAME weights in one launch (passes differ in input projections, outputs
// and noise snapshot; drop_p / site are the layer's).  1 = the launch exists for this configuration, else the caller
// issues s2ag_gru_coop_fwd per pass.
extern "C" int s2ag_gru_coop_fwd_multi_supported(int n, int B, int H) {
    return (s2ag_gru_coop_supported(H) && n >= 1 && n <= COOP_MAX_PASSES && coop_split_pieces() != 0 && B > CBS &&
            coop_two_slices() && 10 * n * cdiv(B, 2 * CBS) * 2 <= 256) ? 1 : 0;
}
extern "C" long long s2ag_gru_coop_fwd_multi_workspace_bytes(int n, int B, int T, int H) {
    if (n <= 0 || B <= 0 || T <= 0 || !s2ag_gru_coop_supported(H)) return 0;
    return (long long)(n * coop_payload_bytes(B, H, 0) + 256);
}
extern "C" int s2ag_gru_coop_fwd_multi(int n, const float* const* gi, const float* whh, const float* bhh,
                                       float* const* y, float* const* ydrop, float* const* gates, int B, int T, int H,
                                       float drop_p, const unsigned long long* const* rng, unsigned site,
                                       void* workspace, void* stream) {
    if (!gi || !whh || !bhh || !y || !workspace || B <= 0 || T <= 0) return S2AG_E_BADARG;
    if (!s2ag_gru_coop_fwd_multi_supported(n, B, H)) return S2AG_E_UNSUPPORTED;
    float p = 0.f;
    for (int i = 0; i < n; ++i) {
        if (!gi[i] || !y[i]) return S2AG_E_BADARG;
        if (ydrop && ydrop[i]) p = drop_p;
    }
    if (p > 0.f)
        for (int i = 0; i < n; ++i)
            if (!ydrop[i] || !rng || !rng[i]) return S2AG_E_BADARG;       // a dropping layer drops in every pass
    return launch_fwd_sp2(n, gi, whh, bhh, y, ydrop, gates, B, T, p, rng, site, workspace, (hipStream_t)stream);
}
extern "C" int s2ag_gru_coop_fwd_multi_error_word_offset(int n, int B, int T, int H, long long* offset) {
    (void)T;
    if (!offset || !s2ag_gru_coop_supported(H) || n <= 0) return S2AG_E_BADARG;
    *offset = (long long)(n * coop_payload_bytes(B, H, 0));
    return 0;
}

extern "C" int s2ag_gru_coop_bwd(const float* dy, int lddy, int dy_dir_stride, const float* whh, const float* y,
                                 const float* gates, float* dgi, float* dgh, int B, int T, int H,
                                 const s2ag_epilogue* e, void* workspace, void* stream) {
    if (!dy || !whh || !y || !gates || !dgi || !dgh || !workspace || B <= 0 || T <= 0) return S2AG_E_BADARG;
    if (!s2ag_gru_coop_supported(H)) return S2AG_E_UNSUPPORTED;
    const float p = e ? e->drop_p : 0.f;
    if (p > 0.f && !e->rng) return S2AG_E_BADARG;
    Ws w = carve(workspace, B, H, 1);
    hipError_t ze = zero_async(w.x, w.zero_bytes, (hipStream_t)stream);
    if (ze != hipSuccess) return (int)ze;
    const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const unsigned long long* rg = e ? e->rng : nullptr;
    const unsigned site = e ? e->site : 0u;
    const int np = coop_split_pieces();
    const size_t red_bytes = sizeof(float) * 3 * CBS * (19 * 16 + 4);
    size_t smem = red_bytes + (np ? (size_t)np * CBS * (96 + 8) * 2 : sizeof(float) * CBS * lds_pitch(96));   // gO + red
    // CU reservation: a workgroup that asks for (almost) all of a CU's LDS keeps every other LDS-using workgroup -- the
    // weight-gradient GEMMs that run beside the recurrence on a forked stream -- off its CU.  The recurrence is a chain of
    // latency-bound steps: sharing the CU's issue slots and LDS pipe with a GEMM made a launch 299 us inside the step
    // against 158 us alone, while 96 of the 256 CUs have no recurrence workgroup at all.
    const void* fn = np == 3   ? reinterpret_cast<const void*>(gru_coop_bwd_k<300, 32, 3>)
                     : np == 2 ? reinterpret_cast<const void*>(gru_coop_bwd_k<300, 32, 2>)
                     : np == 1 ? reinterpret_cast<const void*>(gru_coop_bwd_k<300, 32, 1>)
                               : reinterpret_cast<const void*>(gru_coop_bwd_k<300, 32, 0>);
    static bool granted[4] = {false, false, false, false};
    if (!granted[np]) {
        hipError_t ae = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (ae != hipSuccess) return (int)ae;
        granted[np] = true;
    }
    const dim3 grid(10, cdiv(B, CBS), 2);
    if (np == 3)
        hipLaunchKernelGGL((gru_coop_bwd_k<300, 32, 3>), grid, dim3(CNT), smem, (hipStream_t)stream, dy, lddy,
                           dy_dir_stride, whh, y, gates, dgi, dgh, w.x, w.err, B, T, p, ik, rg, site);
    else if (np == 2)
        hipLaunchKernelGGL((gru_coop_bwd_k<300, 32, 2>), grid, dim3(CNT), smem, (hipStream_t)stream, dy, lddy,
                           dy_dir_stride, whh, y, gates, dgi, dgh, w.x, w.err, B, T, p, ik, rg, site);
    else if (np == 1)
        hipLaunchKernelGGL((gru_coop_bwd_k<300, 32, 1>), grid, dim3(CNT), smem, (hipStream_t)stream, dy, lddy,
                           dy_dir_stride, whh, y, gates, dgi, dgh, w.x, w.err, B, T, p, ik, rg, site);
    else
        hipLaunchKernelGGL((gru_coop_bwd_k<300, 32, 0>), grid, dim3(CNT), smem, (hipStream_t)stream, dy, lddy,
                           dy_dir_stride, whh, y, gates, dgi, dgh, w.x, w.err, B, T, p, ik, rg, site);
    S2AG_LAUNCH_CHECK();
    return 0;
}

/* byte offset of the error word inside `workspace`: non-zero after a launch = a thread timed out waiting for a peer
 * (results invalid) */
extern "C" int s2ag_gru_coop_error_word_offset(int B, int T, int H, int backward, long long* offset) {
    (void)T;
    if (!offset || !s2ag_gru_coop_supported(H)) return S2AG_E_BADARG;
    *offset = (long long)coop_payload_bytes(B, H, backward);
    return 0;
}

#ifdef S2AG_COOP_TRACE
extern "C" int s2ag_gru_coop_trace_read(unsigned long long* host64x8) {
    return (int)hipMemcpyFromSymbol(host64x8, HIP_SYMBOL(g_coop_trace), sizeof(u64) * 64 * 8);
}
#endif
