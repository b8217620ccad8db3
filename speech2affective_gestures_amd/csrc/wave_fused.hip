// The wave encoder's strided convolutions with BatchNorm folded into their neighbours (bf16 mode).
//
// net/multimodal_context_net_v2.py:14-33 of the reference: Conv1d(1,16,15,s5,p1600) BN LeakyReLU(0.3) Conv1d(16,32,15,s6) BN
// LeakyReLU Conv1d(32,64,15,s6) BN LeakyReLU Conv1d(64,32,15,s6).  Layer by layer that is, per BatchNorm, an apply pass
// forward and a reduction + an apply pass backward over tensors the neighbouring conv has just written / is about to read:
// ~220 us of the 600 us the chain took at B = 256, and for conv1's (256, 7891, 16) output alone five extra HBM passes.  Here
// a training-mode BatchNorm never runs as a kernel of its own:
//
//   forward   every conv stores its RAW output y_i (bf16) and leaves fp64 column sums behind (one partial row per
//             workgroup; folded into scale / shift / mean / invstd by s2ag_bn_fold); the NEXT conv applies
//             a_i = leaky(scale_i * y_i + shift_i) in its loader, on the way from the global load to LDS;
//   backward  the data gradient of conv_{i+1} multiplies its result by leaky'(scale_i y_i + shift_i) in its epilogue
//             (dz_i, bf16) and leaves the column sums of dz_i and dz_i * xhat_i behind -- taken from the fp32 accumulators,
//             before any rounding; a one-block fold turns them into the gradients of gamma / beta and three coefficients
//             per channel, and BOTH consumers of dy_i (weight and data gradient of conv_i) form
//                 dy_i = A dz_i + C y_i + B        (A = gamma r, C = -gamma r^2 m2, B = gamma r (r mu m2 - m1);
//                                                   r = invstd, m1 = mean(dz), m2 = mean(dz xhat))
//             in their loaders; the weight gradient of conv_{i+1} recomputes a_i from y_i the same way the forward did.
//
// All three kernel families use the fact that the taps of a window are contiguous in a channels-last tensor ("flat
// window": output frame l reads the K = 15 Cin elements from element 6 Cin l on) and the poly-phase form of a stride-6
// conv (input frame p = 6 q + r only meets the taps t = r + 6 i, i = 0..2, of output frame q - i):
//
//   wv_fwd_k     y[l, co] = b[co] + sum_k a[6 Cin l + k] W[co, k]          A = W (registers), B = staged activations
//   wv_dgrad_k   da[6q + r, ci] = sum_i sum_co dy[q - i, co] W[co, ci, r + 6i]
//   wv_wgrad_k   dW[co, r + 6i, ci] = sum_q dy[q - i, co] a[6q + r, ci]   (contraction over q: LDS transpose reads)
//
// Every global access is a coalesced 16-byte load / store of a contiguous span; every operand element is read from HBM
// once per kernel; the weights (<= 120 KB) live in registers.  v_mfma_f32_16x16x32_bf16, fp32 accumulation.
#include "s2ag_common.h"
#include "bn_fold_inl.h"

namespace {
using namespace s2ag;
using namespace s2ag_fold;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef unsigned short bf16_t;

constexpr int WS = 6;            // stride of conv2..4
constexpr int WKS = 15;          // taps
constexpr int WNT = 3;           // taps per phase (max)

__device__ __forceinline__ unsigned bf_rn(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
// two fp32 -> packed bf16 pair, round to nearest even: ONE v_cvt_pk_bf16_f32 (the integer form is 5 instructions per element)
__device__ __forceinline__ unsigned bf_pack(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

// a = leaky(scale * y + shift) on a 16-byte chunk of 8 consecutive channels (the chunk's channels are the lane's own:
// every lane of these kernels sees the same 8 channels in all its chunks, see the loaders)
__device__ __forceinline__ u32x4 bn_act8(u32x4 v, const float (&sc)[8], const float (&sh)[8], float slope) {
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    unsigned o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        o[j] = bf_pack(leaky(fmaf(sc[2 * j], bf_lo(w[j]), sh[2 * j]), slope),
                       leaky(fmaf(sc[2 * j + 1], bf_hi(w[j]), sh[2 * j + 1]), slope));
    return u32x4{o[0], o[1], o[2], o[3]};
}

// dy = A dz + C y + B on a chunk (see the header)
__device__ __forceinline__ u32x4 bn_bwd8(u32x4 dz, u32x4 y, const float (&ca)[8], const float (&cb)[8],
                                         const float (&cc)[8]) {
    const unsigned d[4] = {dz.x, dz.y, dz.z, dz.w}, w[4] = {y.x, y.y, y.z, y.w};
    unsigned o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        o[j] = bf_pack(fmaf(ca[2 * j], bf_lo(d[j]), fmaf(cc[2 * j], bf_lo(w[j]), cb[2 * j])),
                       fmaf(ca[2 * j + 1], bf_hi(d[j]), fmaf(cc[2 * j + 1], bf_hi(w[j]), cb[2 * j + 1])));
    return u32x4{o[0], o[1], o[2], o[3]};
}

__device__ __forceinline__ u32x4 bn_bwd8(u32x4 dz, u32x4 y, const float* ca, const float* cb, const float* cc) {
    const unsigned d[4] = {dz.x, dz.y, dz.z, dz.w}, w[4] = {y.x, y.y, y.z, y.w};
    unsigned o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        o[j] = bf_pack(fmaf(ca[2 * j], bf_lo(d[j]), fmaf(cc[2 * j], bf_lo(w[j]), cb[2 * j])),
                       fmaf(ca[2 * j + 1], bf_hi(d[j]), fmaf(cc[2 * j + 1], bf_hi(w[j]), cb[2 * j + 1])));
    return u32x4{o[0], o[1], o[2], o[3]};
}

// =====================================================================================================================
// forward
// =====================================================================================================================
struct WvFwdP {
    const bf16_t* x;          // (N, Lin, CIN) raw output of the previous conv
    const float* sc;          // CIN: scale / shift of the previous BatchNorm
    const float* sh;
    float slope;
    const bf16_t* w;          // (COUT, KP) bf16, k = tap * CIN + ci, zero for k >= 15 CIN
    const float* bias;        // COUT, nullable
    void* y;                  // (N, Lout, COUT) bf16, or fp32 (OUT_F32)
    double* stats;            // (2, gridDim.x, COUT) or null
    int N, Lin, Lout, KP;
    int chunks, LC;           // output-frame chunks per clip, frames per chunk (multiple of 16 * teams)
    FwdFold fold;             // fold.ticket != null: the last workgroup turns the partials into BatchNorm coefficients
};

// TEAM = 1: every wave owns sub-tiles of 16 output frames (all COUT channels; its own LDS image, no block barrier).
// TEAM = 4: the four waves share a sub-tile: wave w multiplies channel tile (w % NCT) with K part (w / NCT) of KSPLIT.
template <int CIN, int COUT, int TEAM, bool OUT_F32>
__global__ __launch_bounds__(256) void wv_fwd_k(const WvFwdP p) {
    constexpr int K = WKS * CIN;
    constexpr int KP = (K + 31) / 32 * 32;
    constexpr int RS = WS * CIN;                          // elements between the windows of consecutive frames
    constexpr int PITCH = RS + 16;                        // LDS pitch of an RS-row (bank spread of the 16-byte fragment reads)
    constexpr int SPAN = 15 * RS + KP;                    // elements under a sub-tile of 16 frames
    constexpr int IROWS = (SPAN + RS - 1) / RS;
    constexpr int NCT = COUT / 16;                        // channel tiles
    constexpr int WCT = TEAM == 1 ? NCT : 1;              // channel tiles per wave
    constexpr int KSPLIT = TEAM == 1 ? 1 : 4 / NCT;       // K parts (TEAM = 4)
    constexpr int NKT = KP / 32 / KSPLIT;                 // K tiles per wave
    constexpr int NTH = TEAM == 1 ? 64 : 256;             // threads that stage one image
    constexpr int NLD = (SPAN / 8 + NTH - 1) / NTH;
    static_assert(RS % 32 == 0 && (KP / 32) % KSPLIT == 0 && (TEAM == 1 || NCT * KSPLIT == 4), "shape");
    static_assert((NTH * 8) % CIN == 0, "a lane's chunks must all start at the same channel");
    constexpr int NIMG = TEAM == 1 ? 4 : 2;               // wave-private images / double buffer of the shared one
    __shared__ __attribute__((aligned(16))) bf16_t img_s[NIMG][IROWS * PITCH];
    __shared__ double red[4][2][COUT];
    __shared__ __attribute__((aligned(16))) float kred[TEAM == 1 ? 1 : 4][16 * 16 + 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int st_id = TEAM == 1 ? lane : tid;             // index of this thread among the stagers of its image
    const int ct0 = TEAM == 1 ? 0 : wave % NCT, kpart = TEAM == 1 ? 0 : wave / NCT;

    // weights: a[ct][kt] = W[co = 16 (ct0 + ct) + (lane & 15)][k = 32 (kpart NKT + kt) + 8 (lane >> 4) .. + 8]
    bf16x8 wa[WCT][NKT];
#pragma unroll
    for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
            wa[ct][kt] = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const u32x4*>(p.w + (long long)(16 * (ct0 + ct) + (lane & 15)) * p.KP +
                                                        32 * (kpart * NKT + kt) + 8 * (lane >> 4)));
    float bias[WCT][4];
#pragma unroll
    for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
        for (int v = 0; v < 4; ++v) bias[ct][v] = p.bias ? p.bias[16 * (ct0 + ct) + 4 * (lane >> 4) + v] : 0.f;
    // BatchNorm coefficients of this thread's 8 channels
    float sc[8], sh[8];
    {
        const int c0 = (st_id * 8) % CIN;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j] = p.sc[c0 + j];
            sh[j] = p.sh[c0 + j];
        }
    }

    const int n = blockIdx.x / p.chunks;
    const int l_lo = (blockIdx.x - n * p.chunks) * p.LC;
    int l_hi = l_lo + p.LC;
    if (l_hi > p.Lout) l_hi = p.Lout;
    const bf16_t* xc = p.x + (long long)n * p.Lin * CIN;
    const long long xlen = (long long)p.Lin * CIN;

    // RING sub-tiles' raw rows in flight in registers: a sub-tile is ~0.3 us of work behind a ~2 us load, and with one set a
    // workgroup's few sub-tiles were a chain of memory round trips (conv3: 26 us for 29 MB).  The BatchNorm transform runs
    // when a set goes to LDS, not when it is requested (the request must not wait for the previous set's arithmetic).
    constexpr int RING = 3;
    u32x4 st[RING][NLD];
    auto fetch = [&](int l0, int set) {
        const long long base = (long long)l0 * RS;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = (u * NTH + st_id) * 8;
            st[set][u] = u32x4{0u, 0u, 0u, 0u};
            if (e < SPAN && base + e + 7 < xlen) st[set][u] = *reinterpret_cast<const u32x4*>(xc + base + e);
        }
    };
    auto stash = [&](bf16_t* img, int l0, int set) {
        const long long base = (long long)l0 * RS;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = (u * NTH + st_id) * 8;
            S2AG_DBG_ASSERT(e >= SPAN || e + 16 * (e / RS) + 8 <= IROWS * PITCH);
            if (e < SPAN)
                *reinterpret_cast<u32x4*>(img + e + 16 * (e / RS)) =
                    base + e + 7 < xlen ? bn_act8(st[set][u], sc, sh, p.slope) : u32x4{0u, 0u, 0u, 0u};
        }
    };

    double s1[WCT][4], s2[WCT][4];
#pragma unroll
    for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
        for (int v = 0; v < 4; ++v) s1[ct][v] = s2[ct][v] = 0.0;

    constexpr int STEP = TEAM == 1 ? 64 : 16;             // frames a workgroup covers per iteration
    const int l_first = l_lo + (TEAM == 1 ? wave * 16 : 0);
    int buf = 0;
#pragma unroll
    for (int r = 0; r < RING; ++r)
        if (l_first + r * STEP < l_hi) fetch(l_first + r * STEP, r);
    for (int lb = l_first; lb < l_hi; lb += RING * STEP)
#pragma unroll
    for (int rs = 0; rs < RING; ++rs) {
        const int l0 = lb + rs * STEP;
        if (l0 >= l_hi) break;                            // uniform over the workgroup (TEAM = 4) / the wave (TEAM = 1)
        bf16_t* img = TEAM == 1 ? img_s[wave] : img_s[buf];
        if (TEAM == 1) __builtin_amdgcn_wave_barrier();
        stash(img, l0, rs);
        if (TEAM == 1) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
        if (l0 + RING * STEP < l_hi) fetch(l0 + RING * STEP, rs);
        f32x4 acc[WCT];
#pragma unroll
        for (int ct = 0; ct < WCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bf16_t* frow = img + (lane & 15) * PITCH + 8 * (lane >> 4);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const int k0 = 32 * (kpart * NKT + kt);       // wave-uniform (compile time for TEAM = 1)
            const int j = k0 / RS;
            const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(frow + j * PITCH + (k0 - j * RS)));
#pragma unroll
            for (int ct = 0; ct < WCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ct][kt], b, acc[ct], 0, 0, 0);
        }
        if (KSPLIT > 1) {                                 // the K parts of a channel tile meet in LDS; part 0 finishes
            if (kpart > 0) *reinterpret_cast<f32x4*>(&kred[wave][lane * 4]) = acc[0];
            __syncthreads();
            if (kpart == 0) {
#pragma unroll
                for (int kp = 1; kp < KSPLIT; ++kp) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(&kred[kp * NCT + ct0][lane * 4]);
                    acc[0] += o;
                }
            }
        }
        // D[co][frame]: this lane holds channels 16 ct + 4 (lane >> 4) + v of frame l0 + (lane & 15)
        const int l = l0 + (lane & 15);
        if (kpart == 0 && l < l_hi) {
#pragma unroll
            for (int ct = 0; ct < WCT; ++ct) {
                const int co = 16 * (ct0 + ct) + 4 * (lane >> 4);
                float r[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) r[v] = acc[ct][v] + bias[ct][v];
                if constexpr (OUT_F32) {
                    *reinterpret_cast<f32x4*>(static_cast<float*>(p.y) + ((long long)n * p.Lout + l) * COUT + co) =
                        f32x4{r[0], r[1], r[2], r[3]};
                } else {
                    const unsigned h0 = bf_pack(r[0], r[1]), h1 = bf_pack(r[2], r[3]);
                    *reinterpret_cast<uint2*>(static_cast<bf16_t*>(p.y) + ((long long)n * p.Lout + l) * COUT + co) =
                        make_uint2(h0, h1);
                    r[0] = bf_lo(h0); r[1] = bf_hi(h0); r[2] = bf_lo(h1); r[3] = bf_hi(h1);      // what was stored
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    s1[ct][v] += (double)r[v];
                    s2[ct][v] += (double)r[v] * (double)r[v];
                }
            }
        }
        if (TEAM != 1 && KSPLIT > 1) __syncthreads();     // kred is reused by the next iteration
        buf ^= 1;
    }
    if (p.stats) {          // (2, gridDim.x, COUT): one partial row per workgroup
#pragma unroll
        for (int ct = 0; ct < WCT; ++ct)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                double a1 = s1[ct][v], a2 = s2[ct][v];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    a1 += __shfl_xor(a1, m, 64);
                    a2 += __shfl_xor(a2, m, 64);
                }
                if ((lane & 15) == 0) {
                    red[wave][0][16 * (ct0 + ct) + 4 * (lane >> 4) + v] = a1;
                    red[wave][1][16 * (ct0 + ct) + 4 * (lane >> 4) + v] = a2;
                }
            }
        __syncthreads();
        if (tid < 2 * COUT) {
            const int which = tid / COUT, c = tid - which * COUT;
            double v;
            if (TEAM == 1) v = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
            else v = red[(c / 16) % NCT][which][c];       // the K-part-0 wave of this channel tile
            st_agent(p.stats + ((size_t)which * gridDim.x + blockIdx.x) * COUT + c, v);
        }
        if (p.fold.ticket && two_level_done(p.stats, gridDim.x, COUT, p.fold.ticket)) {
            __shared__ double fred[2][256];
            bn_fwd_fold_body(p.stats + (size_t)2 * gridDim.x * COUT, fold_groups(gridDim.x), COUT, p.fold, fred);
        }
    }
}

template <int CIN, int COUT, int TEAM, bool OUT_F32>
int launch_fwd(WvFwdP p, hipStream_t stream) {
    // ~2 workgroups per CU; a chunk is a multiple of the frames a workgroup covers per iteration
    const int step = TEAM == 1 ? 64 : 16;
    int per_clip = cdiv(512, p.N);
    if (per_clip < 1) per_clip = 1;
    p.LC = cdiv(cdiv(p.Lout, per_clip), step) * step;
    p.chunks = cdiv(p.Lout, p.LC);
    hipLaunchKernelGGL((wv_fwd_k<CIN, COUT, TEAM, OUT_F32>), dim3(p.N * p.chunks), dim3(256), 0, stream, p);
    return p.N * p.chunks;
}

// =====================================================================================================================
// data gradient (poly-phase) with the BatchNorm backward on both sides
// =====================================================================================================================
struct WvDgP {
    const void* dz;           // (N, Lout, COUT): dz_i (bf16), or the fp32 output gradient of the last conv (G_F32)
    const bf16_t* y;          // (N, Lout, COUT) raw output of conv i (unused with G_F32)
    const float* ca;          // COUT: dy = ca dz + cc y + cb (unused with G_F32)
    const float* cb;
    const float* cc;
    const bf16_t* w;          // (6, CIN, 3, CPO) bf16: w[r][ci][i][co] = W[co][ci][r + 6 i], zero beyond the taps / COUT
    int CPO;
    const bf16_t* yp;         // (N, Lin, CIN) raw output of conv i-1
    const float* psc;         // CIN: scale / shift / mean / invstd of BatchNorm i-1
    const float* psh;
    const float* pmean;
    const float* pinv;
    float slope;
    bf16_t* dzp;              // (N, Lin, CIN) out: dz_{i-1} = da_{i-1} * leaky'(psc yp + psh)
    double* stats;            // (2, gridDim.x, CIN): column sums of dz_{i-1} and of dz_{i-1} * xhat_{i-1}
    int N, Lin, Lout, Q, chunks, QC;
    // fold of those sums by the workgroup that finishes last (ticket != null): gradients of gamma / beta of BatchNorm i-1
    // and the coefficients of dy_{i-1} = oca dz + occ y + ocb
    int* ticket;
    const float* pgamma;
    float* dgamma;
    float* dbeta;
    float* oca;
    float* ocb;
    float* occ;
    double inv_rows;
};

template <int COUT, int CIN, int TEAM, bool G_F32>
__global__ __launch_bounds__(256) void wv_dgrad_k(const WvDgP p) {
    constexpr int NCT = CIN / 16;                         // 16-channel tiles of the output
    constexpr int KC = COUT / 32;                         // MFMA K steps per tap
    constexpr int PSTEP = (TEAM == 4 && NCT == 2) ? 2 : 1;
    constexpr int NPH = WS / PSTEP;                       // phases per wave
    constexpr int DROWS = 16 + WNT - 1;                   // source frames under a sub-tile of 16 q
    constexpr int PD = COUT + 16;                         // pitch of the dy image
    constexpr int OP = CIN + 4;                           // pitch of the fp32 output image (96 frames x CIN)
    constexpr int NTH = TEAM == 1 ? 64 : 256;
    constexpr int NLD = (DROWS * COUT / 8 + NTH - 1) / NTH;
    constexpr int NOC = (16 * WS * CIN / 8 + NTH - 1) / NTH;  // output chunks per thread
    static_assert(TEAM == 1 ? NCT == 1 : (NCT == 2 || NCT == 4), "teams");
    static_assert((NTH * 8) % COUT == 0 && (NTH * 8) % CIN == 0, "a thread's chunks must all start at the same channel");
    constexpr int NIMG = TEAM == 1 ? 4 : 1;
    __shared__ __attribute__((aligned(16))) bf16_t dimg_s[NIMG][DROWS * PD];
    __shared__ __attribute__((aligned(16))) float oimg_s[NIMG][16 * WS * OP];
    __shared__ double red[4][2][CIN];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int st_id = TEAM == 1 ? lane : tid;
    const int ct = TEAM == 1 ? 0 : (NCT == 2 ? (wave & 1) : wave);
    const int pstart = (TEAM == 4 && NCT == 2) ? (wave >> 1) : 0;

    // weights of this wave: wa[j][i][kc] = w[r_j][ci = 16 ct + (lane & 15)][i][co = 32 kc + 8 (lane >> 4) .. + 8]
    bf16x8 wa[NPH][WNT][KC];
#pragma unroll
    for (int j = 0; j < NPH; ++j)
#pragma unroll
        for (int i = 0; i < WNT; ++i)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
                wa[j][i][kc] = __builtin_bit_cast(
                    bf16x8, *reinterpret_cast<const u32x4*>(
                                p.w + ((long long)((pstart + j * PSTEP) * CIN + 16 * ct + (lane & 15)) * WNT + i) * p.CPO +
                                32 * kc + 8 * (lane >> 4)));
    // coefficient sets of a thread's 8 source channels (loader) and 8 output channels (epilogue): in LDS, read per use -- as
    // 56 registers per lane they left no room for a ring of prefetched sub-tiles at two workgroups per CU
    __shared__ float cf_src[3][COUT], cf_out[4][CIN];
    for (int i = tid; i < COUT; i += 256) {
        cf_src[0][i] = G_F32 ? 0.f : p.ca[i];
        cf_src[1][i] = G_F32 ? 0.f : p.cb[i];
        cf_src[2][i] = G_F32 ? 0.f : p.cc[i];
    }
    for (int i = tid; i < CIN; i += 256) {
        cf_out[0][i] = p.psc[i];
        cf_out[1][i] = p.psh[i];
        cf_out[2][i] = p.pinv[i];
        cf_out[3][i] = -p.pmean[i] * p.pinv[i];           // xhat = y inv - mean inv
    }
    __syncthreads();
    const int c_src = (st_id * 8) % COUT, c_out = (st_id * 8) % CIN;
    // per lane a few dozen terms: fp32 here, fp64 from the cross-lane sums on
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;

    const int n = blockIdx.x / p.chunks;
    const int q_lo = (blockIdx.x - n * p.chunks) * p.QC;
    int q_hi = q_lo + p.QC;
    if (q_hi > p.Q) q_hi = p.Q;
    const long long src_clip = (long long)n * p.Lout * COUT;
    const bf16_t* ypc = p.yp + (long long)n * p.Lin * CIN;
    bf16_t* dzc = p.dzp + (long long)n * p.Lin * CIN;

    // RING sub-tiles' raw rows (the operands of dy and the rows of y_{i-1} the epilogue needs) in flight in registers: with
    // one set a workgroup's sub-tiles were a chain of memory round trips.  The dy transform runs when a set goes to LDS.
    constexpr int RING = TEAM == 1 ? 2 : 3;
    u32x4 sd[RING][NLD], sy[RING][NLD], yvr[RING][NOC];
    auto fetch = [&](int q0, int set) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = (u * NTH + st_id) * 8;
            const int row = e / COUT, col = e - row * COUT;
            const int l = q0 - (WNT - 1) + row;
            sd[set][u] = sy[set][u] = u32x4{0u, 0u, 0u, 0u};
            if (row < DROWS && (unsigned)l < (unsigned)p.Lout) {
                const long long off = src_clip + (long long)l * COUT + col;
                if constexpr (G_F32) {
                    const float* g = static_cast<const float*>(p.dz) + off;
                    sd[set][u] = *reinterpret_cast<const u32x4*>(g);
                    sy[set][u] = *reinterpret_cast<const u32x4*>(g + 4);
                } else {
                    sd[set][u] = *reinterpret_cast<const u32x4*>(static_cast<const bf16_t*>(p.dz) + off);
                    sy[set][u] = *reinterpret_cast<const u32x4*>(p.y + off);
                }
            }
        }
        // the rows of y_{i-1} under this sub-tile's 96 output frames, in the order the epilogue consumes them
        const int pos0 = WS * q0;
#pragma unroll
        for (int u = 0; u < NOC; ++u) {
            const int e = (u * NTH + st_id) * 8;
            const int row = e / CIN;
            yvr[set][u] = u32x4{0u, 0u, 0u, 0u};
            if (row < 16 * WS && pos0 + row < p.Lin) yvr[set][u] = *reinterpret_cast<const u32x4*>(ypc + (long long)pos0 * CIN + e);
        }
    };
    auto stash = [&](bf16_t* img, int q0, int set) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = (u * NTH + st_id) * 8;
            const int row = e / COUT, col = e - row * COUT;
            const int l = q0 - (WNT - 1) + row;
            if (row < DROWS) {
                S2AG_DBG_ASSERT(row * PD + col + 8 <= DROWS * PD);
                u32x4 v = u32x4{0u, 0u, 0u, 0u};
                if ((unsigned)l < (unsigned)p.Lout) {
                    if constexpr (G_F32) {
                        const f32x4 a = __builtin_bit_cast(f32x4, sd[set][u]), b = __builtin_bit_cast(f32x4, sy[set][u]);
                        v = u32x4{bf_pack(a[0], a[1]), bf_pack(a[2], a[3]), bf_pack(b[0], b[1]), bf_pack(b[2], b[3])};
                    } else {
                        v = bn_bwd8(sd[set][u], sy[set][u], cf_src[0] + c_src, cf_src[1] + c_src, cf_src[2] + c_src);
                    }
                }
                *reinterpret_cast<u32x4*>(img + row * PD + col) = v;
            }
        }
    };

    constexpr int STEP = TEAM == 1 ? 64 : 16;
    const int q_first = q_lo + (TEAM == 1 ? wave * 16 : 0);
#pragma unroll
    for (int r = 0; r < RING; ++r)
        if (q_first + r * STEP < q_hi) fetch(q_first + r * STEP, r);
    for (int qb = q_first; qb < q_hi; qb += RING * STEP)
#pragma unroll
    for (int rs = 0; rs < RING; ++rs) {
        const int q0 = qb + rs * STEP;
        if (q0 >= q_hi) break;                            // uniform over the workgroup (TEAM = 4) / the wave (TEAM = 1)
        bf16_t* dimg = TEAM == 1 ? dimg_s[wave] : dimg_s[0];
        float* oimg = TEAM == 1 ? oimg_s[wave] : oimg_s[0];
        if (TEAM == 1) __builtin_amdgcn_wave_barrier();
        stash(dimg, q0, rs);
        if (TEAM == 1) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
        u32x4 yv[NOC];
#pragma unroll
        for (int u = 0; u < NOC; ++u) yv[u] = yvr[rs][u];
        const int pos0 = WS * q0;
        if (q0 + RING * STEP < q_hi) fetch(q0 + RING * STEP, rs);
        // B fragments: b[i][kc] = dy[q0 + (lane & 15) - i][32 kc + 8 (lane >> 4) .. + 8]
        bf16x8 b[WNT][KC];
#pragma unroll
        for (int i = 0; i < WNT; ++i)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
                b[i][kc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                                                          dimg + ((lane & 15) + WNT - 1 - i) * PD + 32 * kc + 8 * (lane >> 4)));
        f32x4 acc[NPH];
#pragma unroll
        for (int j = 0; j < NPH; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < WNT; ++i)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int j = 0; j < NPH; ++j)
                    if (pstart + j * PSTEP + WS * i < WKS)       // wave-uniform: the last phases have one tap less
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j][i][kc], b[i][kc], acc[j], 0, 0, 0);
        // D[ci][q] -> fp32 image of the 96 output frames: row 6 (lane & 15) + r, channels 16 ct + 4 (lane >> 4) .. + 4
#pragma unroll
        for (int j = 0; j < NPH; ++j)
            *reinterpret_cast<f32x4*>(oimg + (WS * (lane & 15) + pstart + j * PSTEP) * OP + 16 * ct + 4 * (lane >> 4)) = acc[j];
        if (TEAM == 1) __builtin_amdgcn_wave_barrier();
        else __syncthreads();
        // epilogue: whole rows, 16 bytes per thread: dz = da * leaky'(z), statistics from the fp32 values
        int rows = (q_hi - q0) * WS;
        if (rows > 16 * WS) rows = 16 * WS;
        if (pos0 + rows > p.Lin) rows = p.Lin - pos0;
#pragma unroll
        for (int u = 0; u < NOC; ++u) {
            const int e = (u * NTH + st_id) * 8;
            const int row = e / CIN, col = e - row * CIN;
            if (row < rows) {
                S2AG_DBG_ASSERT(row * OP + col + 8 <= 16 * WS * OP && pos0 + row < p.Lin);
                const f32x4 d0 = *reinterpret_cast<const f32x4*>(oimg + row * OP + col);
                const f32x4 d1 = *reinterpret_cast<const f32x4*>(oimg + row * OP + col + 4);
                const float da[8] = {d0[0], d0[1], d0[2], d0[3], d1[0], d1[1], d1[2], d1[3]};
                const unsigned w[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
                float dzv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float yk = (k & 1) ? bf_hi(w[k >> 1]) : bf_lo(w[k >> 1]);
                    const float z = fmaf(cf_out[0][c_out + k], yk, cf_out[1][c_out + k]);
                    dzv[k] = z > 0.f ? da[k] : da[k] * p.slope;
                    s1[k] += dzv[k];
                    s2[k] = fmaf(dzv[k], fmaf(yk, cf_out[2][c_out + k], cf_out[3][c_out + k]), s2[k]);
                }
                *reinterpret_cast<u32x4*>(dzc + (long long)pos0 * CIN + e) =
                    u32x4{bf_pack(dzv[0], dzv[1]), bf_pack(dzv[2], dzv[3]), bf_pack(dzv[4], dzv[5]), bf_pack(dzv[6], dzv[7])};
            }
        }
    }
    // statistics: threads with the same (st_id * 8) % CIN hold the same 8 channels
    constexpr int G = CIN / 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double a1 = (double)s1[k], a2 = (double)s2[k];
#pragma unroll
        for (int m = G; m < 64; m <<= 1) {
            a1 += __shfl_xor(a1, m, 64);
            a2 += __shfl_xor(a2, m, 64);
        }
        if (lane < G) {
            red[wave][0][lane * 8 + k] = a1;
            red[wave][1][lane * 8 + k] = a2;
        }
    }
    __syncthreads();
    if (tid < 2 * CIN) {
        const int which = tid / CIN, c = tid - which * CIN;
        st_agent(p.stats + ((size_t)which * gridDim.x + blockIdx.x) * CIN + c,
                 red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c]);
    }
    if (p.ticket && two_level_done(p.stats, gridDim.x, CIN, p.ticket)) {
        __shared__ double fred[2][256];
        bn_bwd_fold_body<true>(p.stats + (size_t)2 * gridDim.x * CIN, fold_groups(gridDim.x), CIN, p.inv_rows, p.pgamma, p.pmean,
                               p.pinv, p.dgamma, p.dbeta, p.oca, p.ocb, p.occ, fred);
    }
}

template <int COUT, int CIN, int TEAM, bool G_F32>
int launch_dgrad(WvDgP p, hipStream_t stream) {
    const int step = TEAM == 1 ? 64 : 16;
    p.Q = cdiv(p.Lin, WS);
    int per_clip = cdiv(512, p.N);
    if (per_clip < 1) per_clip = 1;
    p.QC = cdiv(cdiv(p.Q, per_clip), step) * step;
    p.chunks = cdiv(p.Q, p.QC);
    hipLaunchKernelGGL((wv_dgrad_k<COUT, CIN, TEAM, G_F32>), dim3(p.N * p.chunks), dim3(256), 0, stream, p);
    return p.N * p.chunks;
}

// =====================================================================================================================
// weight gradient (poly-phase, contraction over output frames through the LDS transpose read)
// =====================================================================================================================
struct WvWgP {
    const void* dz;           // as in WvDgP: the operands of dy_i
    const bf16_t* y;
    const float* ca;
    const float* cb;
    const float* cc;
    const bf16_t* yp;         // (N, Lin, CIN) raw output of conv i-1; a = leaky(psc yp + psh)
    const float* psc;
    const float* psh;
    float slope;
    float* part;              // (gridDim.x, COUT, 15, CIN) fp32
    float* part_b;            // (gridDim.x, COUT)
    int N, Lin, Lout, QS, total_steps;
};

__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* img, int off, int pitch) {
    using lds_p = __attribute__((address_space(3))) s16x4*;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + off + 4 * pitch));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// One workgroup = 4 waves on shared, double-buffered images of a 32-q step; the raw loads of the next RING - 1 steps are in
// flight in registers (a step's ~10 KB arrive ~2 us after they are requested, its MFMAs take a fraction of that: with one
// step in flight the kernel was a chain of memory round trips).  The accumulator tiles (co tile, ci tile, phase r, tap
// index i) are split over the waves -- (32, 16): wave = (co tile, phase half);  (64, 32): wave = co tile;  (32, 64): wave =
// ci tile -- and, for the two small layers, the phases over blockIdx.y (PSPLIT = 3: a workgroup stages the frames of ITS two
// phases only, so the input is still read once; the 123 KB partial tile of a workgroup is shared by the three).
template <int COUT, int CIN, bool G_F32, int PSPLIT>
__global__ __launch_bounds__(256) void wv_wgrad_k(const WvWgP p) {
    constexpr int NCOT = COUT / 16, NCIT = CIN / 16;
    constexpr int WCOT = (NCOT == 2 && NCIT == 4) ? 2 : 1;    // co tiles per wave
    constexpr int WCIT = (NCOT == 4) ? 2 : 1;                  // ci tiles per wave
    constexpr int RSPLIT = (NCOT == 2 && NCIT == 1) ? 2 : 1;   // phase halves over the waves (32, 16)
    constexpr int NPW = WS / PSPLIT;                           // phases a workgroup stages
    constexpr int NR = NPW / RSPLIT;                           // phases per wave
    constexpr int QT = 32;                                     // q per step = one MFMA K
    constexpr int DROWS = QT + WNT - 1;
    constexpr int PG = COUT + 8, PA = CIN + 8;                 // image pitches
    constexpr int NDC = DROWS * COUT / 8, NAC = QT * NPW * CIN / 8;
    constexpr int NLD = (NDC + 255) / 256, NLA = (NAC + 255) / 256;
    constexpr int RING = 3;
    static_assert(2048 % COUT == 0 && 2048 % CIN == 0, "a thread's chunks must all start at the same channel");
    static_assert(WS % PSPLIT == 0 && NPW % RSPLIT == 0, "phase split");
    __shared__ __attribute__((aligned(16))) bf16_t dimg[2][DROWS * PG];
    __shared__ __attribute__((aligned(16))) bf16_t aimg[2][NPW * QT * PA];
    __shared__ float bred[4][COUT];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cot0 = NCOT == 4 ? wave : (NCIT == 1 ? (wave & 1) : 0);
    const int cit0 = (NCOT == 2 && NCIT == 4) ? wave : 0;
    const int ph0 = blockIdx.y * NPW;                          // first phase of this workgroup
    const int r0 = RSPLIT == 2 ? NR * (wave >> 1) : 0;         // first phase of this wave (relative to ph0)

    float ca[8], cb[8], cc[8], psc[8], psh[8];
    {
        const int c0 = (tid * 8) % COUT, d0 = (tid * 8) % CIN;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (!G_F32) {
                ca[j] = p.ca[c0 + j];
                cb[j] = p.cb[c0 + j];
                cc[j] = p.cc[c0 + j];
            }
            psc[j] = p.psc[d0 + j];
            psh[j] = p.psh[d0 + j];
        }
    }
    float bacc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bacc[j] = 0.f;
    f32x4 acc[WCOT][WCIT][NR][WNT];
#pragma unroll
    for (int a = 0; a < WCOT; ++a)
#pragma unroll
        for (int b = 0; b < WCIT; ++b)
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int i = 0; i < WNT; ++i) acc[a][b][r][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this workgroup's contiguous range of steps (a step = 32 q of one clip)
    const int per = p.total_steps / gridDim.x, extra = p.total_steps % gridDim.x;
    const int s_beg = blockIdx.x * per + min((int)blockIdx.x, extra);
    const int s_end = s_beg + per + ((int)blockIdx.x < extra ? 1 : 0);

    // raw loads of a step: rd = dz chunks (or the first four floats of g), ry = y chunks (or the second four), ra = y_prev
    u32x4 rd[RING][NLD], ry[RING][NLD], ra[RING][NLA];
    auto fetch = [&](int s, int set) {
        const bool live = s < s_end;
        const int n = live ? s / p.QS : 0, q0 = live ? (s - n * p.QS) * QT : 0;
        const long long gclip = (long long)n * p.Lout * COUT;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = (u * 256 + tid) * 8;
            const int row = e / COUT, col = e - row * COUT;
            const int l = q0 - (WNT - 1) + row;
            rd[set][u] = ry[set][u] = u32x4{0u, 0u, 0u, 0u};
            if (live && row < DROWS && (unsigned)l < (unsigned)p.Lout) {
                const long long off = gclip + (long long)l * COUT + col;
                if constexpr (G_F32) {
                    const float* g = static_cast<const float*>(p.dz) + off;
                    rd[set][u] = *reinterpret_cast<const u32x4*>(g);
                    ry[set][u] = *reinterpret_cast<const u32x4*>(g + 4);
                } else {
                    rd[set][u] = *reinterpret_cast<const u32x4*>(static_cast<const bf16_t*>(p.dz) + off);
                    ry[set][u] = *reinterpret_cast<const u32x4*>(p.y + off);
                }
            }
        }
        const bf16_t* ypc = p.yp + ((long long)n * p.Lin + (long long)WS * q0) * CIN;
        const int frames = live ? p.Lin - WS * q0 : 0;         // valid frames from the block start
#pragma unroll
        for (int u = 0; u < NLA; ++u) {
            const int e = (u * 256 + tid) * 8;                 // element among this workgroup's NPW phases x 32 q x CIN
            const int fp = e / CIN, col = e - fp * CIN;        // frame index among the staged ones: fp = ql * NPW + rl
            const int ql = fp / NPW, rl = fp - ql * NPW;
            const int fl = ql * WS + ph0 + rl;                 // frame inside the block of 192
            ra[set][u] = u32x4{0u, 0u, 0u, 0u};
            if (fp < QT * NPW && fl < frames) ra[set][u] = *reinterpret_cast<const u32x4*>(ypc + (long long)fl * CIN + col);
        }
    };
    // transform + LDS stores of a fetched step; a live flag travels with the data (zeros stay zeros: B != 0 otherwise)
    auto stash = [&](int s, int set, int buf) {
        const bool live = s < s_end;
        const int q0 = live ? (s - (s / p.QS) * p.QS) * QT : 0;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = (u * 256 + tid) * 8;
            const int row = e / COUT, col = e - row * COUT;
            const int l = q0 - (WNT - 1) + row;
            if (row < DROWS) {
                u32x4 v = u32x4{0u, 0u, 0u, 0u};
                if (live && (unsigned)l < (unsigned)p.Lout) {
                    if constexpr (G_F32) {
                        const f32x4 a = __builtin_bit_cast(f32x4, rd[set][u]), b = __builtin_bit_cast(f32x4, ry[set][u]);
                        v = u32x4{bf_pack(a[0], a[1]), bf_pack(a[2], a[3]), bf_pack(b[0], b[1]), bf_pack(b[2], b[3])};
                    } else {
                        v = bn_bwd8(rd[set][u], ry[set][u], ca, cb, cc);
                    }
                }
                *reinterpret_cast<u32x4*>(&dimg[buf][row * PG + col]) = v;
                if (row >= WNT - 1 && blockIdx.y == 0) {       // the halo rows are the previous step's core rows
                    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bacc[2 * j] += bf_lo(w[j]);
                        bacc[2 * j + 1] += bf_hi(w[j]);
                    }
                }
            }
        }
        const int frames = live ? p.Lin - WS * q0 : 0;
#pragma unroll
        for (int u = 0; u < NLA; ++u) {
            const int e = (u * 256 + tid) * 8;
            const int fp = e / CIN, col = e - fp * CIN;
            const int ql = fp / NPW, rl = fp - ql * NPW;
            if (fp < QT * NPW) {
                u32x4 v = u32x4{0u, 0u, 0u, 0u};
                if (ql * WS + ph0 + rl < frames) v = bn_act8(ra[set][u], psc, psh, p.slope);
                *reinterpret_cast<u32x4*>(&aimg[buf][(rl * QT + ql) * PA + col]) = v;
            }
        }
    };

    const int g = lane >> 4, t = lane & 15;
    const int tr_row = 8 * g + (t >> 2), tr_col = 4 * (t & 3);
    auto mma = [&](int buf) {
        // A = dy^T shifted by the tap index: af[i][a] holds dy[q0 + 8 g .. + 8 - i][co tile a]
        bf16x8 af[WNT][WCOT];
#pragma unroll
        for (int i = 0; i < WNT; ++i)
#pragma unroll
            for (int a = 0; a < WCOT; ++a)
                af[i][a] = tr_frag(dimg[buf], (tr_row + WNT - 1 - i) * PG + 16 * (cot0 + a) + tr_col, PG);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            bf16x8 bfr[WCIT];
#pragma unroll
            for (int b = 0; b < WCIT; ++b)
                bfr[b] = tr_frag(aimg[buf], ((r0 + r) * QT + tr_row) * PA + 16 * (cit0 + b) + tr_col, PA);
#pragma unroll
            for (int i = 0; i < WNT; ++i)
                if (ph0 + r0 + r + WS * i < WKS) {             // wave-uniform
#pragma unroll
                    for (int a = 0; a < WCOT; ++a)
#pragma unroll
                        for (int b = 0; b < WCIT; ++b)
                            acc[a][b][r][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][a], bfr[b], acc[a][b][r][i], 0, 0, 0);
                }
        }
    };

#pragma unroll
    for (int r = 0; r < RING; ++r) fetch(s_beg + r, r);
    int buf = 0;
    for (int s = s_beg; s < s_end; s += RING) {
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            stash(s + r, r, buf);
            fetch(s + r + RING, r);
            __syncthreads();
            mma(buf);
            buf ^= 1;
        }
    }
    // D[co][ci]: co = 16 (cot0 + a) + 4 (lane >> 4) + v, ci = 16 (cit0 + b) + (lane & 15); tap ph0 + r0 + r + 6 i
    float* dst = p.part + (size_t)blockIdx.x * COUT * WKS * CIN;
#pragma unroll
    for (int a = 0; a < WCOT; ++a)
#pragma unroll
        for (int b = 0; b < WCIT; ++b)
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int i = 0; i < WNT; ++i) {
                    const int tap = ph0 + r0 + r + WS * i;
                    if (tap < WKS) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int co = 16 * (cot0 + a) + 4 * (lane >> 4) + v, ci = 16 * (cit0 + b) + (lane & 15);
                            dst[((size_t)co * WKS + tap) * CIN + ci] = acc[a][b][r][i][v];
                        }
                    }
                }
    // bias gradient (the workgroups of phase group 0): threads with the same (tid * 8) % COUT hold the same 8 channels
    if (blockIdx.y != 0) return;
    constexpr int G = COUT / 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = bacc[j];
#pragma unroll
        for (int m = G; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
        if (lane < G) bred[wave][lane * 8 + j] = v;
    }
    __syncthreads();
    if (tid < COUT) p.part_b[(size_t)blockIdx.x * COUT + tid] = bred[0][tid] + bred[1][tid] + bred[2][tid] + bred[3][tid];
}

// dw (Cout, Cin, 15) += sum over the partials (b, Cout, 15, Cin); db += sum of the bias partials.  A block owns 32
// consecutive elements; its 8 groups of 32 threads sum every 8th partial with independent loads in flight and meet in LDS
// in a fixed order (one writer per element: bit-reproducible).
__global__ __launch_bounds__(256) void wv_wgrad_reduce_k(const float* __restrict__ part, const float* __restrict__ part_b,
                                                         int nparts, int Cout, int Cin, float* __restrict__ dw,
                                                         float* __restrict__ db) {
    __shared__ float red[8][32];
    const int total = Cout * WKS * Cin;
    const int i = blockIdx.x * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
    const bool is_w = i < total, is_b = !is_w && i < total + Cout;
    const float* src = is_w ? part + i : part_b + (i - total);
    const size_t stride = is_w ? (size_t)total : (size_t)Cout;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (is_w || is_b) {
        int b = grp;
        for (; b + 24 < nparts; b += 32) {
            s0 += src[(size_t)b * stride];
            s1 += src[(size_t)(b + 8) * stride];
            s2 += src[(size_t)(b + 16) * stride];
            s3 += src[(size_t)(b + 24) * stride];
        }
        for (; b < nparts; b += 8) s0 += src[(size_t)b * stride];
    }
    red[grp][threadIdx.x & 31] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp == 0 && (is_w || is_b)) {
        float s = red[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < 8; ++k) s += red[k][threadIdx.x];
        if (is_w) {
            const int ci = i % Cin, tap = (i / Cin) % WKS, co = i / (Cin * WKS);
            dw[((size_t)co * Cin + ci) * WKS + tap] += s;
        } else if (db) {
            db[i - total] += s;
        }
    }
}

// BatchNorm backward fold as a kernel of its own (the data-gradient kernels do it themselves when given a ticket word)
__global__ __launch_bounds__(256) void wv_bn_bwd_fold_k(const double* __restrict__ part, int R, int C, double inv_rows,
                                                        const float* __restrict__ gamma, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, float* __restrict__ ca,
                                                        float* __restrict__ cb, float* __restrict__ cc) {
    __shared__ double red[2][256];
    bn_bwd_fold_body<false>(part, R, C, inv_rows, gamma, mean, invstd, dgamma, dbeta, ca, cb, cc, red);
}
}  // namespace

// ---- C ABI ---------------------------------------------------------------------------------------------------------
extern "C" int s2ag_wave_fwd_rows(int N, int Lout, int Cin, int Cout) {
    if (N <= 0 || Lout <= 0) return S2AG_E_BADARG;
    const int step = (Cin == 16 && Cout == 32) ? 64 : 16;
    int per_clip = cdiv(512, N);
    if (per_clip < 1) per_clip = 1;
    const int LC = cdiv(cdiv(Lout, per_clip), step) * step;
    return N * cdiv(Lout, LC);
}

extern "C" int s2ag_wave_conv_fwd(const void* x, const float* in_scale, const float* in_shift, float slope, const void* w_packed,
                                  int KP, const float* bias, void* y, int out_f32, double* stats, const s2ag_bn_fold_args* fold,
                                  int N, int Lin, int Lout, int Cin, int Cout, void* stream) {
    if (!x || !in_scale || !in_shift || !w_packed || !y || N <= 0 || Lin <= 0 || Lout <= 0) return S2AG_E_BADARG;
    if (fold && (!stats || !fold->ticket || !fold->gamma || !fold->beta || !fold->running_mean || !fold->running_var ||
                 !fold->scale || !fold->shift || !fold->mean || !fold->invstd || fold->repeat < 1))
        return S2AG_E_BADARG;
    if ((long long)(Lout - 1) * WS + WKS > Lin) return S2AG_E_BADARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)w_packed & 15) || ((uintptr_t)y & 15)) return S2AG_E_BADARG;
    WvFwdP p{};
    p.x = static_cast<const bf16_t*>(x); p.sc = in_scale; p.sh = in_shift; p.slope = slope;
    p.w = static_cast<const bf16_t*>(w_packed); p.bias = bias; p.y = y; p.stats = stats;
    p.N = N; p.Lin = Lin; p.Lout = Lout; p.KP = KP;
    if (fold)
        p.fold = FwdFold{fold->ticket, fold->gamma, fold->beta, fold->running_mean, fold->running_var, fold->num_batches_tracked,
                         fold->eps, fold->momentum, fold->repeat, (long long)N * Lout, fold->scale, fold->shift, fold->mean,
                         fold->invstd};
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 16 && Cout == 32 && !out_f32 && KP >= 256) launch_fwd<16, 32, 1, false>(p, s);
    else if (Cin == 32 && Cout == 64 && !out_f32 && KP >= 480) launch_fwd<32, 64, 4, false>(p, s);
    else if (Cin == 64 && Cout == 32 && out_f32 && KP >= 960) launch_fwd<64, 32, 4, true>(p, s);
    else return S2AG_E_UNSUPPORTED;
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_wave_dgrad_rows(int N, int Lin, int Cin) {
    if (N <= 0 || Lin <= 0) return S2AG_E_BADARG;
    const int step = Cin == 16 ? 64 : 16;
    const int Q = cdiv(Lin, WS);
    int per_clip = cdiv(512, N);
    if (per_clip < 1) per_clip = 1;
    const int QC = cdiv(cdiv(Q, per_clip), step) * step;
    return N * cdiv(Q, QC);
}

extern "C" int s2ag_wave_conv_dgrad(const void* dz, const void* y, const float* ca, const float* cb, const float* cc, int g_f32,
                                    const void* w_phases, int CPO, const void* y_prev, const float* p_scale,
                                    const float* p_shift, const float* p_mean, const float* p_invstd, float slope, void* dz_prev,
                                    double* stats, int* ticket, const float* p_gamma, float* dgamma, float* dbeta, float* out_ca,
                                    float* out_cb, float* out_cc, int N, int Lin, int Lout, int Cin, int Cout, void* stream) {
    if (!dz || !w_phases || !y_prev || !p_scale || !p_shift || !p_mean || !p_invstd || !dz_prev || !stats || N <= 0 ||
        Lin <= 0 || Lout <= 0)
        return S2AG_E_BADARG;
    if (!g_f32 && (!y || !ca || !cb || !cc)) return S2AG_E_BADARG;
    if (ticket && (!p_gamma || !out_ca || !out_cb || !out_cc)) return S2AG_E_BADARG;
    if ((long long)(Lout - 1) * WS + WKS > Lin) return S2AG_E_BADARG;
    if (((uintptr_t)dz & 15) || ((uintptr_t)y & 15) || ((uintptr_t)w_phases & 15) || ((uintptr_t)y_prev & 15) ||
        ((uintptr_t)dz_prev & 15))
        return S2AG_E_BADARG;
    WvDgP p{};
    p.dz = dz; p.y = static_cast<const bf16_t*>(y); p.ca = ca; p.cb = cb; p.cc = cc;
    p.w = static_cast<const bf16_t*>(w_phases); p.CPO = CPO; p.yp = static_cast<const bf16_t*>(y_prev);
    p.psc = p_scale; p.psh = p_shift; p.pmean = p_mean; p.pinv = p_invstd; p.slope = slope;
    p.dzp = static_cast<bf16_t*>(dz_prev); p.stats = stats; p.N = N; p.Lin = Lin; p.Lout = Lout;
    p.ticket = ticket; p.pgamma = p_gamma; p.dgamma = dgamma; p.dbeta = dbeta; p.oca = out_ca; p.ocb = out_cb; p.occ = out_cc;
    p.inv_rows = 1.0 / ((double)N * Lin);
    hipStream_t s = (hipStream_t)stream;
    if (Cout == 32 && Cin == 16 && !g_f32 && CPO >= 32) launch_dgrad<32, 16, 1, false>(p, s);
    else if (Cout == 64 && Cin == 32 && !g_f32 && CPO >= 64) launch_dgrad<64, 32, 4, false>(p, s);
    else if (Cout == 32 && Cin == 64 && g_f32 && CPO >= 32) launch_dgrad<32, 64, 4, true>(p, s);
    else return S2AG_E_UNSUPPORTED;
    S2AG_LAUNCH_CHECK();
    return 0;
}

static int wgrad_blocks(int total_steps, int Cin, int Cout) {
    // the partials are (blocks, Cout, 15, Cin) fp32: keep them under ~16 MB (the two small layers then run three workgroups
    // -- one per pair of phases -- on every partial tile), and at least 64 workgroups streaming
    long long cap = (16ll << 20) / ((long long)Cout * WKS * Cin * 4);
    if (cap > 512) cap = 512;
    if (cap < 64) cap = 64;
    return (int)(total_steps < cap ? total_steps : cap);
}

extern "C" int s2ag_wave_wgrad_blocks(int N, int Lout, int Cin, int Cout) {
    if (N <= 0 || Lout <= 0 || Cin <= 0 || Cout <= 0) return S2AG_E_BADARG;
    return wgrad_blocks(N * cdiv(Lout + WNT - 1, 32), Cin, Cout);
}

extern "C" int s2ag_wave_conv_wgrad(const void* dz, const void* y, const float* ca, const float* cb, const float* cc, int g_f32,
                                    const void* y_prev, const float* p_scale, const float* p_shift, float slope, float* partials,
                                    float* partials_b, float* dw, float* db, int N, int Lin, int Lout, int Cin, int Cout,
                                    void* stream) {
    if (!dz || !y_prev || !p_scale || !p_shift || !partials || !partials_b || !dw || N <= 0 || Lin <= 0 || Lout <= 0)
        return S2AG_E_BADARG;
    if (!g_f32 && (!y || !ca || !cb || !cc)) return S2AG_E_BADARG;
    if ((long long)(Lout - 1) * WS + WKS > Lin) return S2AG_E_BADARG;
    if (((uintptr_t)dz & 15) || ((uintptr_t)y & 15) || ((uintptr_t)y_prev & 15)) return S2AG_E_BADARG;
    WvWgP p{};
    p.dz = dz; p.y = static_cast<const bf16_t*>(y); p.ca = ca; p.cb = cb; p.cc = cc;
    p.yp = static_cast<const bf16_t*>(y_prev); p.psc = p_scale; p.psh = p_shift; p.slope = slope;
    p.part = partials; p.part_b = partials_b; p.N = N; p.Lin = Lin; p.Lout = Lout;
    p.QS = cdiv(Lout + WNT - 1, 32);
    p.total_steps = N * p.QS;
    const int blocks = wgrad_blocks(p.total_steps, Cin, Cout);
    hipStream_t s = (hipStream_t)stream;
    if (Cout == 32 && Cin == 16 && !g_f32) hipLaunchKernelGGL((wv_wgrad_k<32, 16, false, 1>), dim3(blocks), dim3(256), 0, s, p);
    else if (Cout == 64 && Cin == 32 && !g_f32) hipLaunchKernelGGL((wv_wgrad_k<64, 32, false, 3>), dim3(blocks, 3), dim3(256), 0, s, p);
    else if (Cout == 32 && Cin == 64 && g_f32) hipLaunchKernelGGL((wv_wgrad_k<32, 64, true, 3>), dim3(blocks, 3), dim3(256), 0, s, p);
    else return S2AG_E_UNSUPPORTED;
    const int total = Cout * WKS * Cin + Cout;
    hipLaunchKernelGGL(wv_wgrad_reduce_k, dim3(cdiv(total, 32)), dim3(256), 0, s, partials, partials_b, blocks, Cout, Cin, dw, db);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_wave_bn_bwd_fold(const double* partials, int partial_rows, int C, long long rows, const float* gamma,
                                     const float* mean, const float* invstd, float* dgamma, float* dbeta, float* ca, float* cb,
                                     float* cc, void* stream) {
    if (!partials || partial_rows <= 0 || rows <= 0 || !gamma || !mean || !invstd || !ca || !cb || !cc) return S2AG_E_BADARG;
    if (C != 16 && C != 32 && C != 64) return S2AG_E_UNSUPPORTED;
    hipLaunchKernelGGL(wv_bn_bwd_fold_k, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, partial_rows, C, 1.0 / (double)rows,
                       gamma, mean, invstd, dgamma, dbeta, ca, cb, cc);
    S2AG_LAUNCH_CHECK();
    return 0;
}
