// The head of the wave encoder -- Conv1d(1,16,15,s5,p1600) BatchNorm LeakyReLU(0.3) Conv1d(16,32,15,s6)
// (net/multimodal_context_net_v2.py:18-21 of the reference) -- WITHOUT its (N, 7891, 16) tensor in HBM.
//
// Layer by layer that tensor (65 MB in bf16, 129 MB in fp32 at 256 clips) is written once and read four to six times per
// iteration (BatchNorm apply, conv2, the two gradients of conv2, BatchNorm backward, conv1's weight gradient), although
// every element is 15 multiply-adds away from the waveform.  Here it is recomputed where it is needed:
//
//   wv12_stats_k   waveform -> column sums of z1 and z1^2 (fp64 partial rows, folded into BatchNorm 1's coefficients by the
//                  workgroup that finishes last)                                                     reads x
//   wv12_fwd_k     waveform -> z1 -> a1 = leaky(scale1 z1 + shift1) in LDS -> z2 = conv2(a1) + the column sums of z2
//                                                                                                    reads x, writes z2
//   wv12_bwd_k     dy2 (+ waveform) -> da1 (poly-phase data gradient of conv2), z1 and a1 again, du1 = da1 leaky'(.), the
//                  column sums of du1 and du1 xhat1 (BatchNorm 1 backward), conv2's weight gradient dW2 += dy2^T a1, and the
//                  three sums conv1's weight gradient is made of (below)                               reads dy2, x
//   wv12_finish_k  partial tiles -> dW2, dW1
//
// conv1's weight gradient without dz1: BatchNorm backward is dz1 = A du1 + C z1 + B per channel (A = gamma r, C = -gamma r^2
// m2, B = gamma r (r mu m2 - m1), m1 = mean(du1), m2 = mean(du1 xhat1)), and A, B, C are only known when ALL of du1 has been
// seen -- which is why the layer-by-layer form stores du1 and reads it back.  But the weight gradient is linear in dz1:
//       dW1[c, t] = sum_f dz1[f, c] x[5 f + t] = A_c sum_f du1[f, c] x[5f + t] + C_c sum_f z1[f, c] x[5f + t] + B_c sum_f x[5f + t]
// and the three sums do not depend on A, B, C: wv12_bwd_k accumulates them (two 16 x 16 matrix products per 32 frames), the
// last kernel combines them.  Both biases feed a BatchNorm: their gradients are exactly zero (the column sums of a
// BatchNorm's input gradient vanish identically) and are not formed.
//
// conv1 itself is an MFMA here (16 channels x 16 frames x 16 taps; fp32 mode: four v_mfma_f32_16x16x4_f32 -- fp32 in, fp32
// out, the arithmetic of an fmaf chain --; bf16 mode, where z1 is rounded to bf16 anyway: three v_mfma_f32_16x16x16_bf16 on two
// bf16 pieces per operand): 4 LDS reads per lane and tile instead of 60 multiply-adds.  The kernels of a mode form z1 with the
// same instructions on the same operands, so they see bit-identical values.
//
// NP = 1: bf16 mode (z1, a1, z2 rounded to bf16 as the layer-by-layer kernels of wave_fused.hip store them; conv2 and the
// backward products on v_mfma_f32_16x16x32_bf16).  NP = 2: fp32 mode: the FORWARD pass (conv1, conv2) runs on the f32 MFMA --
// the reference's fp32 arithmetic, so activations and branch decisions are those of the layer-by-layer fp32 kernels --, the
// backward products take their operands as two bf16 pieces (hi = rn(v), lo = rn(v - hi); hi*lo + lo*hi + hi*hi in fp32: 16
// mantissa bits per product, the precision of the step's other large gradient products, bench.py `matrix_products`).
#include <type_traits>

#include "s2ag_common.h"
#include "bn_fold_inl.h"

namespace {
using namespace s2ag;
using namespace s2ag_fold;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
typedef unsigned short bf16_t;

constexpr int KS = 15;            // taps of both convs
constexpr int S1 = 5, S2 = 6;     // strides
constexpr int C1 = 16, C2 = 32;   // channels of z1 / z2
constexpr int NT = 3;             // taps per phase of conv2's poly-phase form
constexpr int K2P = 256;          // 15 * 16 padded to MFMA K tiles
// packed weights (bf16 elements, every section two piece planes: hi then lo)
constexpr int O_W2F = 0;                          // [2][32 co][256 k]        k = t * 16 + ci, zero from 240
constexpr int O_W2P = O_W2F + 2 * C2 * K2P;       // [2][6 r][16 ci][3 i][32 co] = W2[co][ci][r + 6 i], zero where r + 6 i >= 15
// ... and fp32 copies for the forward products on the f32 MFMA, k-major so that a wave-level load is one contiguous run
// (offsets in bf16 elements: an fp32 value takes two)
constexpr int O_W1F = O_W2P + 2 * S2 * C1 * NT * C2;   // fp32 [16 t][16 c]   t = 15: zero
constexpr int O_W2K = O_W1F + 2 * 16 * C1;             // fp32 [240 k][32 co]  k = t * 16 + ci
constexpr int O_W1 = O_W2K + 2 * KS * C1 * C2;         // bf16 [2][16 c][16 t]  t = 15: zero (conv1 of bf16 mode)
constexpr int W12_PACK = O_W1 + 2 * C1 * 16;

__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned bf_pack(float a, float b) {         // one v_cvt_pk_bf16_f32, round to nearest even
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
__device__ __forceinline__ float bf_round(float v) { return bf_lo(bf_pack(v, 0.f)); }
// what the hi piece leaves: exact in fp32
__device__ __forceinline__ unsigned bf_pack_rest(float a, float b, unsigned hi) { return bf_pack(a - bf_lo(hi), b - bf_hi(hi)); }

__device__ __forceinline__ f32x4 mfma32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ bf16x8 ld8(const bf16_t* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p)); }

// z1 of 16 frames x 16 channels on the f32 MFMA (v_mfma_f32_16x16x4_f32, bit-for-bit an fmaf chain: the forward pass of fp32
// mode keeps the reference's arithmetic, and no vector-ALU work goes into operand splitting): lane (n = lane & 15, g = lane >> 4)
// passes taps g, 4 + g, 8 + g, 12 + g of ITS frame's window (sp = &window[g] in LDS) and receives channels 4 g + v of that
// frame.  w[kb] = W1[c = n][4 kb + g] (tap 15: zero).  All three kernels form z1 this way: bit-identical values.
__device__ __forceinline__ f32x4 conv1_tile(const float* sp, const float (&w)[4], f32x4 bias) {
    f32x4 acc = bias;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[kb], sp[4 * kb], acc, 0, 0, 0);
    return acc;
}
// bf16 mode rounds z1 to bf16 anyway: there conv1 runs on v_mfma_f32_16x16x16_bf16 from two bf16 pieces per operand (hi =
// rn(v), lo = rn(v - hi); lo*hi + hi*lo + hi*hi: 16 mantissa bits), 24 MFMA cycles per tile instead of 128.  Lane (n, g) passes
// the samples under taps 4 g .. 4 g + 3 (sp = &window[4 g]) -- or, pre-split, words hi | lo << 16 (conv1_tile_sp).
__device__ __forceinline__ f32x4 mfma16(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 conv1_tile2(const float* sp, s16x4 wh, s16x4 wl, f32x4 bias) {
    const float x0 = sp[0], x1 = sp[1], x2 = sp[2], x3 = sp[3];
    const unsigned h0 = bf_pack(x0, x1), h1 = bf_pack(x2, x3);
    const unsigned l0 = bf_pack_rest(x0, x1, h0), l1 = bf_pack_rest(x2, x3, h1);
    const s16x4 xh = __builtin_bit_cast(s16x4, u32x2{h0, h1}), xl = __builtin_bit_cast(s16x4, u32x2{l0, l1});
    f32x4 acc = mfma16(wl, xh, bias);
    acc = mfma16(wh, xl, acc);
    return mfma16(wh, xh, acc);
}
__device__ __forceinline__ f32x4 conv1_tile_sp(const unsigned* sp, s16x4 wh, s16x4 wl, f32x4 bias) {
    const unsigned w0 = sp[0], w1 = sp[1], w2 = sp[2], w3 = sp[3];
    const s16x4 xh = __builtin_bit_cast(s16x4, u32x2{__builtin_amdgcn_perm(w1, w0, 0x05040100u), __builtin_amdgcn_perm(w3, w2, 0x05040100u)});
    const s16x4 xl = __builtin_bit_cast(s16x4, u32x2{__builtin_amdgcn_perm(w1, w0, 0x07060302u), __builtin_amdgcn_perm(w3, w2, 0x07060302u)});
    f32x4 acc = mfma16(wl, xh, bias);
    acc = mfma16(wh, xl, acc);
    return mfma16(wh, xh, acc);
}
// conv1's weights of a lane, for either form
struct W1Regs {
    float f[4];
    s16x4 h, l;
};
template <bool BF>
__device__ __forceinline__ W1Regs load_w1(const bf16_t* wp, int n, int g) {
    W1Regs w{};
    if (BF) {
        w.h = *reinterpret_cast<const s16x4*>(wp + O_W1 + n * 16 + 4 * g);
        w.l = *reinterpret_cast<const s16x4*>(wp + O_W1 + C1 * 16 + n * 16 + 4 * g);
    } else {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) w.f[kb] = reinterpret_cast<const float*>(wp + O_W1F)[(4 * kb + g) * C1 + n];
    }
    return w;
}
// win = the window of this lane's frame in an fp32 segment
template <bool BF>
__device__ __forceinline__ f32x4 conv1_any(const float* win, int g, const W1Regs& w, f32x4 bias) {
    if (BF) return conv1_tile2(win + 4 * g, w.h, w.l, bias);
    return conv1_tile(win + g, w.f, bias);
}

// ---- weights: fp32 masters -> the three operand layouts, two pieces each ---------------------------------------------------
__global__ __launch_bounds__(256) void wv12_pack_k(const float* __restrict__ w1, const float* __restrict__ w2,
                                                   bf16_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    constexpr int n2 = C2 * K2P, n3 = S2 * C1 * NT * C2, n4 = 16 * C1, n5 = KS * C1 * C2;
    float v = 0.f;
    int dst, plane;
    if (i < n2) {
        const int co = i / K2P, k = i % K2P, t = k / C1, ci = k % C1;
        if (t < KS) v = w2[(co * C1 + ci) * KS + t];
        dst = O_W2F + i; plane = n2;
    } else if (i < n2 + n3) {
        const int j = i - n2, co = j % C2, ii = (j / C2) % NT, ci = (j / (C2 * NT)) % C1, r = j / (C2 * NT * C1);
        if (r + S2 * ii < KS) v = w2[(co * C1 + ci) * KS + r + S2 * ii];
        dst = O_W2P + j; plane = n3;
    } else if (i < n2 + n3 + n4) {
        const int j = i - n2 - n3, t = j / C1, c = j % C1;
        reinterpret_cast<float*>(out + O_W1F)[j] = t < KS ? w1[c * KS + t] : 0.f;
        return;
    } else if (i < n2 + n3 + n4 + n5) {
        const int j = i - n2 - n3 - n4, k = j / C2, co = j % C2, t = k / C1, ci = k % C1;
        reinterpret_cast<float*>(out + O_W2K)[j] = w2[(co * C1 + ci) * KS + t];
        return;
    } else if (i < n2 + n3 + n4 + n5 + n4) {
        const int j = i - n2 - n3 - n4 - n5, c = j / 16, t = j % 16;
        if (t < KS) v = w1[c * KS + t];
        dst = O_W1 + j; plane = n4;
    } else {
        return;
    }
    const unsigned h = bf_pack(v, 0.f);
    out[dst] = (bf16_t)(h & 0xffffu);
    out[dst + plane] = (bf16_t)(bf_pack(v - bf_lo(h), 0.f) & 0xffffu);
}
constexpr int W12_PACK_THREADS = C2 * K2P + S2 * C1 * NT * C2 + 16 * C1 + KS * C1 * C2 + 16 * C1;

// =====================================================================================================================
// statistics of z1
// =====================================================================================================================
struct W12StatsP {
    const float* x;           // (N, Lin) waveform
    const bf16_t* wp;         // packed weights
    const float* b1;          // 16
    double* stats;            // (2, gridDim.x + groups, 16)
    FwdFold fold;
    int N, Lin, L1, pad, cpc, total;      // chunks per clip, chunks in all
};

// A wave owns chunks of 128 frames (its own LDS segment, the next chunk's samples in flight in registers, no block barrier).
template <bool RB>
__global__ __launch_bounds__(256) void wv12_stats_k(const W12StatsP p) {
    constexpr int CH = 128;
    constexpr int SEG = S1 * (CH - 1) + 16;
    constexpr int NLD = (SEG + 63) / 64;
    __shared__ float seg_s[4][NLD * 64];
    __shared__ double red[4][2][C1];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const W1Regs w1r = load_w1<RB>(p.wp, n, g);
    const f32x4 bias = f32x4{p.b1[4 * g], p.b1[4 * g + 1], p.b1[4 * g + 2], p.b1[4 * g + 3]};
    float* seg = seg_s[wave];
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    float sr[NLD];
    auto fetch = [&](int ch) {
        const int clip = ch / p.cpc, f0 = (ch - clip * p.cpc) * CH;
        const long long base = (long long)f0 * S1 - p.pad;
        const float* xc = p.x + (long long)clip * p.Lin;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = u * 64 + lane;
            const long long pos = base + i;
            sr[u] = (i < SEG && pos >= 0 && pos < p.Lin) ? xc[pos] : 0.f;
        }
    };
    if (wid < p.total) fetch(wid);
    for (int ch = wid; ch < p.total; ch += nw) {
        const int clip = ch / p.cpc, f0 = (ch - clip * p.cpc) * CH;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < NLD; ++u) seg[u * 64 + lane] = sr[u];
        __builtin_amdgcn_wave_barrier();
        if (ch + nw < p.total) fetch(ch + nw);
        float t1[4] = {0.f, 0.f, 0.f, 0.f}, t2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tile = 0; tile < CH / 16; ++tile) {
            S2AG_DBG_ASSERT(S1 * (16 * tile + n) + 15 < NLD * 64);
            const f32x4 z = conv1_any<RB>(seg + S1 * (16 * tile + n), g, w1r, bias);
            if (f0 + 16 * tile + n < p.L1) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float zr = RB ? bf_round(z[v]) : z[v];
                    t1[v] += zr;
                    t2[v] = fmaf(zr, zr, t2[v]);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            s1[v] += (double)t1[v];
            s2[v] += (double)t2[v];
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            s1[v] += __shfl_xor(s1[v], m, 64);
            s2[v] += __shfl_xor(s2[v], m, 64);
        }
        if (n == 0) {
            red[wave][0][4 * g + v] = s1[v];
            red[wave][1][4 * g + v] = s2[v];
        }
    }
    __syncthreads();
    if (tid < 2 * C1) {
        const int which = tid / C1, c = tid - which * C1;
        st_agent(p.stats + ((size_t)which * gridDim.x + blockIdx.x) * C1 + c,
                 (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]));
    }
    if (p.fold.ticket && two_level_done(p.stats, gridDim.x, C1, p.fold.ticket)) {
        __shared__ double fred[2][256];
        bn_fwd_fold_body(p.stats + (size_t)2 * gridDim.x * C1, fold_groups(gridDim.x), C1, p.fold, fred);
    }
}

// Branch decisions of the LeakyReLU behind BatchNorm 1 (parity tests: the oracle replays them, tests/s2ag_testing.py
// SignTap): sign[(clip, frame, channel)] = scale1 z1 + shift1 > 0, with z1 formed exactly as the three kernels form it.
struct W12SignP {
    const float* x;
    const bf16_t* wp;
    const float* b1;
    const float* sc1;
    const float* sh1;
    unsigned char* sign;      // (N, L1, 16)
    int N, Lin, L1, pad, cpc, total;
};

template <bool RB>
__global__ __launch_bounds__(256) void wv12_signs_k(const W12SignP p) {
    constexpr int CH = 128;
    constexpr int SEG = S1 * (CH - 1) + 16;
    constexpr int NLD = (SEG + 63) / 64;
    __shared__ float seg_s[4][NLD * 64];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const W1Regs w1r = load_w1<RB>(p.wp, n, g);
    const f32x4 bias = f32x4{p.b1[4 * g], p.b1[4 * g + 1], p.b1[4 * g + 2], p.b1[4 * g + 3]};
    float* seg = seg_s[wave];
    for (int ch = blockIdx.x * 4 + wave; ch < p.total; ch += gridDim.x * 4) {
        const int clip = ch / p.cpc, f0 = (ch - clip * p.cpc) * CH;
        const long long base = (long long)f0 * S1 - p.pad;
        const float* xc = p.x + (long long)clip * p.Lin;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = u * 64 + lane;
            const long long pos = base + i;
            seg[i] = (i < SEG && pos >= 0 && pos < p.Lin) ? xc[pos] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        for (int tile = 0; tile < CH / 16; ++tile) {
            const f32x4 z = conv1_any<RB>(seg + S1 * (16 * tile + n), g, w1r, bias);
            const int f = f0 + 16 * tile + n;
            if (f < p.L1) {
                unsigned w = 0u;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float zr = RB ? bf_round(z[v]) : z[v];
                    w |= (fmaf(p.sc1[4 * g + v], zr, p.sh1[4 * g + v]) > 0.f ? 1u : 0u) << (8 * v);
                }
                *reinterpret_cast<unsigned*>(p.sign + ((long long)clip * p.L1 + f) * C1 + 4 * g) = w;
            }
        }
    }
}

// =====================================================================================================================
// forward: waveform -> z2
// =====================================================================================================================
struct W12FwdP {
    const float* x;
    const bf16_t* wp;
    const float* b1;
    const float* sc1;         // 16: scale / shift of BatchNorm 1
    const float* sh1;
    float slope;
    const float* b2;          // 32, nullable
    void* z2;                 // (N, L2, 32) bf16 (NP = 1) / fp32 (NP = 2)
    double* stats;            // (2, gridDim.x (+ groups), 32) or null
    FwdFold fold;             // fold.ticket != null: the last workgroup folds the partials into BatchNorm 2's coefficients
    int N, Lin, L1, L2, pad, chunks, LC;
};

// Every wave owns sub-tiles of 16 output frames: the 571 samples under them go through its own LDS segment (next sub-tile's
// in flight in registers), seven conv1 tiles leave a1 (112 frames) in its own image in the flat-window layout of
// wave_fused.hip, then K = 240 against the weights in registers.
//   NP = 1  bf16 image, element e = 16 frame + channel at e + 16 (e / 96) (the 16 rows x 8 k of a fragment read spread over
//           the banks), 8 K tiles of v_mfma_f32_16x16x32_bf16, z2 stored as bf16;
//   NP = 2  fp32 image at e + 2 (e / 96) (row pitch 98 floats: the 16 frames x 2 taps of a half-wave read hit 32 different
//           banks), 60 K steps of v_mfma_f32_16x16x4_f32 -- the reference's fp32 arithmetic, no operand splitting --, z2 fp32.
template <int NP>
__global__ __launch_bounds__(256) void wv12_fwd_k(const W12FwdP p) {
    constexpr bool F32 = NP == 2;
    constexpr int RS = S2 * C1, NKT = K2P / 32, NKB = KS * C1 / 4;
    constexpr int RPAD = F32 ? 2 : 16, PITCH = RS + RPAD;         // elements between the windows of consecutive output frames
    constexpr int AFR = 112;                                      // a1 frames per sub-tile: 6 * 15 + 15 = 105 -> 7 tiles
    constexpr int IMG = AFR * C1 + RPAD * (AFR * C1 / RS + 1);
    constexpr int SEG = S1 * (AFR - 1) + 16;
    constexpr int NLD = (SEG + 63) / 64;
    __shared__ __attribute__((aligned(16))) unsigned char img_s[4][IMG * (F32 ? 4 : 2)];
    __shared__ float seg_s[4][NLD * 64];
    __shared__ double red[4][2][C2];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // conv2's weights: bf16 A[co = 16 ct + n][k = 32 kt + 8 g ..]  /  fp32 A[co = 16 ct + n][k = 4 kb + g]
    bf16x8 wa[F32 ? 1 : 2][F32 ? 1 : NKT];
    float wf[F32 ? 2 : 1][F32 ? NKB : 1];
    if constexpr (F32) {
        const float* w2k = reinterpret_cast<const float*>(p.wp + O_W2K);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) wf[ct][kb] = w2k[(4 * kb + g) * C2 + 16 * ct + n];
    } else {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) wa[ct][kt] = ld8(p.wp + O_W2F + (16 * ct + n) * K2P + 32 * kt + 8 * g);
    }
    const W1Regs w1r = load_w1<!F32>(p.wp, n, g);
    const f32x4 bias1 = f32x4{p.b1[4 * g], p.b1[4 * g + 1], p.b1[4 * g + 2], p.b1[4 * g + 3]};
    float sc[4], sh[4], bias2[2][4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        sc[v] = p.sc1[4 * g + v];
        sh[v] = p.sh1[4 * g + v];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) bias2[ct][v] = p.b2 ? p.b2[16 * ct + 4 * g + v] : 0.f;
    }
    const int clip = blockIdx.x / p.chunks;
    const int l_lo = (blockIdx.x - clip * p.chunks) * p.LC;
    int l_hi = l_lo + p.LC;
    if (l_hi > p.L2) l_hi = p.L2;
    const float* xc = p.x + (long long)clip * p.Lin;
    float* seg = seg_s[wave];
    bf16_t* img16 = reinterpret_cast<bf16_t*>(img_s[wave]);
    float* img32 = reinterpret_cast<float*>(img_s[wave]);

    float sr[NLD];
    auto fetch = [&](int l0) {
        const long long base = (long long)l0 * (S2 * S1) - p.pad;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = u * 64 + lane;
            const long long pos = base + i;
            sr[u] = (i < SEG && pos >= 0 && pos < p.Lin) ? xc[pos] : 0.f;
        }
    };
    double s1[2][4], s2[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int v = 0; v < 4; ++v) s1[ct][v] = s2[ct][v] = 0.0;

    int l0 = l_lo + wave * 16;
    if (l0 < l_hi) fetch(l0);
    for (; l0 < l_hi; l0 += 64) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < NLD; ++u) seg[u * 64 + lane] = sr[u];
        __builtin_amdgcn_wave_barrier();
        if (l0 + 64 < l_hi) fetch(l0 + 64);
        // a1 of the 112 frames from 6 l0 on
#pragma unroll
        for (int tile = 0; tile < AFR / 16; ++tile) {
            const f32x4 z = conv1_any<!F32>(seg + S1 * (16 * tile + n), g, w1r, bias1);
            float zr[4] = {z[0], z[1], z[2], z[3]}, a[4];
            if (!F32) {                                             // z1 as bf16 mode stores it
                const unsigned z0 = bf_pack(z[0], z[1]), z1 = bf_pack(z[2], z[3]);
                zr[0] = bf_lo(z0); zr[1] = bf_hi(z0); zr[2] = bf_lo(z1); zr[3] = bf_hi(z1);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float t = fmaf(sc[v], zr[v], sh[v]);
                a[v] = fmaxf(t, t * p.slope);                       // leaky, 0 <= slope <= 1
            }
            const int fl = 16 * tile + n;
            const int off = fl * C1 + 4 * g + RPAD * (fl / S2);
            S2AG_DBG_ASSERT(off >= 0 && off + 4 <= IMG && S1 * fl + 15 < NLD * 64);
            if constexpr (F32) {
                *reinterpret_cast<f32x2*>(img32 + off) = f32x2{a[0], a[1]};
                *reinterpret_cast<f32x2*>(img32 + off + 2) = f32x2{a[2], a[3]};
            } else {
                *reinterpret_cast<u32x2*>(img16 + off) = u32x2{bf_pack(a[0], a[1]), bf_pack(a[2], a[3])};
            }
        }
        __builtin_amdgcn_wave_barrier();
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if constexpr (F32) {
            // four accumulator chains (channel tile x K parity): a dependent f32 MFMA waits for its predecessor's result
            f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            const float* frow = img32 + n * PITCH + g;
#pragma unroll
            for (int kb = 0; kb < NKB; kb += 2) {
                const int k0 = 4 * kb, j0 = k0 / RS, k1 = k0 + 4, j1 = k1 / RS;      // compile time
                S2AG_DBG_ASSERT(n * PITCH + g + j1 * PITCH + (k1 - j1 * RS) < IMG);
                const float b0 = frow[j0 * PITCH + (k0 - j0 * RS)], b1 = frow[j1 * PITCH + (k1 - j1 * RS)];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ct][kb], b0, acc[ct], 0, 0, 0);
                    acc2[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[ct][kb + 1], b1, acc2[ct], 0, 0, 0);
                }
            }
            acc[0] += acc2[0];
            acc[1] += acc2[1];
        } else {
            const bf16_t* frow = img16 + n * PITCH + 8 * g;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const int k0 = 32 * kt, j = k0 / RS;
                S2AG_DBG_ASSERT(n * PITCH + 8 * g + j * PITCH + (k0 - j * RS) + 8 <= IMG);
                const bf16x8 b = ld8(frow + j * PITCH + (k0 - j * RS));
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[ct] = mfma32(wa[ct][kt], b, acc[ct]);
            }
        }
        // D[co][frame]: this lane holds channels 16 ct + 4 g + v of frame l0 + n
        const int l = l0 + n;
        if (l < l_hi) {
            S2AG_DBG_ASSERT(l >= 0 && l < p.L2 && clip < p.N);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int co = 16 * ct + 4 * g;
                float r[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) r[v] = acc[ct][v] + bias2[ct][v];
                if (F32) {
                    *reinterpret_cast<f32x4*>(static_cast<float*>(p.z2) + ((long long)clip * p.L2 + l) * C2 + co) =
                        f32x4{r[0], r[1], r[2], r[3]};
                } else {
                    const unsigned h0 = bf_pack(r[0], r[1]), h1 = bf_pack(r[2], r[3]);
                    *reinterpret_cast<u32x2*>(static_cast<bf16_t*>(p.z2) + ((long long)clip * p.L2 + l) * C2 + co) = u32x2{h0, h1};
                    r[0] = bf_lo(h0); r[1] = bf_hi(h0); r[2] = bf_lo(h1); r[3] = bf_hi(h1);      // what was stored
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    s1[ct][v] += (double)r[v];
                    s2[ct][v] += (double)r[v] * (double)r[v];
                }
            }
        }
    }
    if (p.stats) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                double a1 = s1[ct][v], a2 = s2[ct][v];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    a1 += __shfl_xor(a1, m, 64);
                    a2 += __shfl_xor(a2, m, 64);
                }
                if (n == 0) {
                    red[wave][0][16 * ct + 4 * g + v] = a1;
                    red[wave][1][16 * ct + 4 * g + v] = a2;
                }
            }
        __syncthreads();
        if (tid < 2 * C2) {
            const int which = tid / C2, c = tid - which * C2;
            st_agent(p.stats + ((size_t)which * gridDim.x + blockIdx.x) * C2 + c,
                     (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]));
        }
        if (p.fold.ticket && two_level_done(p.stats, gridDim.x, C2, p.fold.ticket)) {
            __shared__ double fred[2][256];
            bn_fwd_fold_body(p.stats + (size_t)2 * gridDim.x * C2, fold_groups(gridDim.x), C2, p.fold, fred);
        }
    }
}

// =====================================================================================================================
// backward: dy2 (+ waveform) -> dW2 tiles, the sums of BatchNorm 1's backward and of conv1's weight gradient
// =====================================================================================================================
constexpr int SROW = 2 * C1 * 16 + 16;     // a workgroup's sums for conv1: S_du (16 x 16), S_z (16 x 16), S_x (16)

struct W12BwdP {
    const float* x;
    const bf16_t* wp;
    const float* b1;
    const float* sc1;         // 16: scale / shift / mean / invstd of BatchNorm 1
    const float* sh1;
    const float* mean1;
    const float* inv1;
    float slope;
    // dy2: XF: ca du2 + cc z2 + cb from the bf16 rows of du2 and z2 (wave_fused.hip); else fp32 rows in `dz`
    const void* dz;
    const bf16_t* z2;
    const float* ca;
    const float* cb;
    const float* cc;
    float* part_w2;           // (gridDim.x, 32, 15, 16)
    float* part_s;            // (gridDim.x + groups, SROW): a row per workgroup, then a row per group of 16 workgroups
    double* stats;            // (2, gridDim.x + groups, 16): column sums of du1 and du1 * xhat1
    int* ticket;              // 1 + groups zero words (left zero): the last workgroup folds BatchNorm 1's backward
    const float* gamma1;
    float* dgamma1;           // nullable
    float* dbeta1;
    float* oca;               // out: dz1 = oca du1 + occ z1 + ocb
    float* ocb;
    float* occ;
    double inv_rows;
    int N, Lin, L1, L2, pad, QS, total_steps;
    unsigned long long* trace;  // diagnostics (s2ag_wave12_set_trace): s_memtime stamps of block 0, wave 0
};

__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* img, int off, int pitch) {
    using lds_p = __attribute__((address_space(3))) s16x4*;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + off + 4 * pitch));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ void st_agent_f(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent_f(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// A step = 32 q of one clip (q = frame of z2's grid; a1 frame 6 q + r, r = phase); a workgroup walks a contiguous range of
// steps, the raw loads of the next RING - 1 steps in flight in registers.  Per step:
//   stash   dy2 rows q0 - 2 .. q0 + 31 (bf16 pieces) and the 971 samples under the step (as they are, and as two bf16 pieces)
//           -> LDS (double buffered)
//   phase 2 wave = (q half h, phase triple pt): da1 of ITS 16 q x 3 phases (9 MFMAs against weights in registers) and z1 of
//           the same frames in the same lane layout (conv1 tiles whose 16 frames lie 6 apart), then in registers a1, du1 =
//           da1 leaky'(.), the BatchNorm sums; a1 / du1 / z1 -> LDS, phase major [r][q][16]
//   phase 3 wave = (co tile, phase half): dW2[co, r + 6 i, ci] += sum_q dy2[q - i, co] a1[6 q + r, ci] (transpose reads);
//           wave = (du1 | z1, phase half): S[c, t] += sum_q M[6 q + r, c] x[5 (6 q + r) + t]; the du1 waves also
//           S_x[t] += sum_q x[5 (6 q + r) + t] (a ones-matrix against the same windows)
template <int NP, bool XF>
__global__ __launch_bounds__(256, NP == 1 ? 2 : 1) void wv12_bwd_k(const W12BwdP p) {
    constexpr int QT = 32, DROWS = QT + NT - 1;
    constexpr int PG = C2 + 8, PA = C1 + 8;
    constexpr int SEGN = 1024;                                     // >= 30 * 31 + 5 * 5 + 15 + 1 = 971
    constexpr int RING = 3;                                       // steps whose raw loads are in flight in registers
    __shared__ __attribute__((aligned(16))) bf16_t dimg[2][NP][DROWS * PG];
    __shared__ __attribute__((aligned(16))) float seg_s[NP == 1 ? 1 : 2][NP == 1 ? 4 : SEGN];   // fp32 mode: the samples (conv1 on the f32 MFMA) ...
    __shared__ __attribute__((aligned(16))) unsigned spl_s[2][SEGN];   // ... and split: hi piece | lo piece << 16 (bf16-pipe products)
    __shared__ __attribute__((aligned(16))) bf16_t aimg[NP][S2 * QT * PA];
    __shared__ __attribute__((aligned(16))) bf16_t uimg[NP][S2 * QT * PA];
    __shared__ __attribute__((aligned(16))) bf16_t zimg[NP][S2 * QT * PA];
    __shared__ double red[4][2][C1];
    __shared__ float srow[2][SROW];

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave & 1, pt = wave >> 1;                        // phase 2 roles
    const int cot = wave & 1, ph = wave >> 1, mat = wave & 1;       // phase 3 roles

    // conv2's phase weights W2P[r][ci][i][co] (18 KB per piece) live in LDS: as registers (36 per piece and lane) they pushed
    // the kernel over the 256 registers two workgroups per CU can have; a fragment is re-read per use (9 reads per step)
    constexpr int W2PN = S2 * C1 * NT * C2;
    __shared__ __attribute__((aligned(16))) bf16_t wlds[NP][W2PN];
    // dy2 = ca du2 + cc z2 + cb (XF): read per use, not 24 registers; written HERE so that the barrier below orders it too
    // (r03 wrote it after that barrier: the first stash() of waves 2 / 3 could read it before waves 0 / 1 had written it)
    __shared__ float coef2[3][C2];
    if (XF && tid < 3 * C2) coef2[tid / C2][tid % C2] = (tid < C2 ? p.ca : tid < 2 * C2 ? p.cb : p.cc)[tid % C2];
#pragma unroll
    for (int np = 0; np < NP; ++np)
        for (int i = tid * 8; i < W2PN; i += 256 * 8)
            *reinterpret_cast<u32x4*>(&wlds[np][i]) = *reinterpret_cast<const u32x4*>(p.wp + O_W2P + np * W2PN + i);
    __syncthreads();
    const W1Regs w1r = load_w1<NP == 1>(p.wp, n, g);
    const f32x4 bias1 = f32x4{p.b1[4 * g], p.b1[4 * g + 1], p.b1[4 * g + 2], p.b1[4 * g + 3]};
    float sc[4], sh[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        sc[v] = p.sc1[4 * g + v];
        sh[v] = p.sh1[4 * g + v];
    }
    // loader roles: thread tid < 136 owns one chunk of 8 channels of a dy2 row
    const int d_row = (tid * 8) / C2, d_col = (tid * 8) % C2;
    const bool d_own = tid < DROWS * C2 / 8;
    f32x4 accw[3][NT], acc1 = f32x4{0.f, 0.f, 0.f, 0.f}, accx = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int i = 0; i < NT; ++i) accw[rr][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};   // per lane: a few hundred terms; fp64 from there on

    const int per = p.total_steps / gridDim.x, extra = p.total_steps % gridDim.x;
    const int s_beg = blockIdx.x * per + min((int)blockIdx.x, extra);
    const int s_end = s_beg + per + ((int)blockIdx.x < extra ? 1 : 0);
    // (clip, q0) of the step being fetched / being multiplied: running counters, no division per step
    int f_clip = s_beg / p.QS, f_q0 = (s_beg - f_clip * p.QS) * QT;
    int c_clip = f_clip, c_q0 = f_q0;
    const int q_wrap = p.QS * QT;

    u32x4 rd[RING], ry[RING];
    float rx[RING][4];
    auto fetch = [&](int s, int set) {
        const bool live = s < s_end;
        rd[set] = ry[set] = u32x4{0u, 0u, 0u, 0u};
        const int l = f_q0 - (NT - 1) + d_row;
        if (live && d_own && (unsigned)l < (unsigned)p.L2) {
            S2AG_DBG_ASSERT(f_clip < p.N);
            const long long off = ((long long)f_clip * p.L2 + l) * C2 + d_col;
            if constexpr (XF) {
                rd[set] = *reinterpret_cast<const u32x4*>(static_cast<const bf16_t*>(p.dz) + off);
                ry[set] = *reinterpret_cast<const u32x4*>(p.z2 + off);
            } else {
                const float* gp = static_cast<const float*>(p.dz) + off;
                rd[set] = *reinterpret_cast<const u32x4*>(gp);
                ry[set] = *reinterpret_cast<const u32x4*>(gp + 4);
            }
        }
        const long long base = (long long)f_q0 * (S2 * S1) - p.pad;
        const float* xc = p.x + (long long)f_clip * p.Lin;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long pos = base + u * 256 + tid;
            rx[set][u] = (live && pos >= 0 && pos < p.Lin) ? xc[pos] : 0.f;
        }
        f_q0 += QT;
        if (f_q0 >= q_wrap) {
            f_q0 = 0;
            ++f_clip;
        }
    };
    auto stash = [&](int set, int buf) {
        if (d_own) {
            const int l = c_q0 - (NT - 1) + d_row;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            if ((unsigned)l < (unsigned)p.L2) {
                if constexpr (XF) {
                    const unsigned d[4] = {rd[set].x, rd[set].y, rd[set].z, rd[set].w};
                    const unsigned w[4] = {ry[set].x, ry[set].y, ry[set].z, ry[set].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = d_col + 2 * j;
                        v[2 * j] = fmaf(coef2[0][c], bf_lo(d[j]), fmaf(coef2[2][c], bf_lo(w[j]), coef2[1][c]));
                        v[2 * j + 1] = fmaf(coef2[0][c + 1], bf_hi(d[j]), fmaf(coef2[2][c + 1], bf_hi(w[j]), coef2[1][c + 1]));
                    }
                } else {
                    const f32x4 a = __builtin_bit_cast(f32x4, rd[set]), b = __builtin_bit_cast(f32x4, ry[set]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = a[j];
                        v[4 + j] = b[j];
                    }
                }
            }
            unsigned hi[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) hi[j] = bf_pack(v[2 * j], v[2 * j + 1]);
            *reinterpret_cast<u32x4*>(&dimg[buf][0][d_row * PG + d_col]) = u32x4{hi[0], hi[1], hi[2], hi[3]};
            if (NP == 2)
                *reinterpret_cast<u32x4*>(&dimg[buf][NP - 1][d_row * PG + d_col]) =
                    u32x4{bf_pack_rest(v[0], v[1], hi[0]), bf_pack_rest(v[2], v[3], hi[1]), bf_pack_rest(v[4], v[5], hi[2]),
                          bf_pack_rest(v[6], v[7], hi[3])};
        }
        // samples, and their two bf16 pieces (split once per sample by the thread that stages it, not once per use)
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
            const float x0 = rx[set][u], x1 = rx[set][u + 1];
            const unsigned hh = bf_pack(x0, x1), ll = bf_pack_rest(x0, x1, hh);
            if (NP == 2) {
                seg_s[buf * (NP - 1)][u * 256 + tid] = x0;
                seg_s[buf * (NP - 1)][(u + 1) * 256 + tid] = x1;
            }
            spl_s[buf][u * 256 + tid] = __builtin_amdgcn_perm(ll, hh, 0x05040100u);
            spl_s[buf][(u + 1) * 256 + tid] = __builtin_amdgcn_perm(ll, hh, 0x07060302u);
        }
    };

    const int tr_row = 8 * g + (n >> 2), tr_col = 4 * (n & 3);
    int nst = 0;
    const bool tr_on = p.trace != nullptr && blockIdx.x == 0 && tid == 0;
#define W12_STAMP() do { if (tr_on && nst < 120) p.trace[nst++] = __builtin_amdgcn_s_memtime(); } while (0)
    // FULL: every frame of the step lies inside the clip (all but the last step of a clip): no masks
    auto step = [&](int buf, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        const float* seg = seg_s[buf * (NP - 1)];
        const unsigned* spl = spl_s[buf];
        W12_STAMP();
        // ---- phase 2 -------------------------------------------------------------------------------------------------
        {
            bf16x8 b[NP][NT];
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int i = 0; i < NT; ++i) b[np][i] = ld8(&dimg[buf][np][(16 * h + n + NT - 1 - i) * PG + 8 * g]);
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                const int r = 3 * pt + rr;                          // wave-uniform
                f32x4 da = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    if (r + S2 * i < KS) {
                        const int woff = ((r * C1 + n) * NT + i) * C2 + 8 * g;     // A[ci = n][co = 8 g ..]
                        const bf16x8 wf = ld8(&wlds[0][woff]);
                        if (NP == 2) {
                            da = mfma32(ld8(&wlds[NP - 1][woff]), b[0][i], da);
                            da = mfma32(wf, b[NP - 1][i], da);
                        }
                        da = mfma32(wf, b[0][i], da);
                    }
                const int woff = (S2 * S1) * (16 * h + n) + S1 * r;       // window of this lane's frame
                S2AG_DBG_ASSERT(woff + 15 < SEGN);
                const f32x4 z = NP == 1 ? conv1_tile_sp(spl + woff + 4 * g, w1r.h, w1r.l, bias1) : conv1_tile(seg + woff + g, w1r.f, bias1);
                const bool valid = FULL || S2 * (c_q0 + 16 * h + n) + r < p.L1;
                float a[4], du[4], zr[4];
                unsigned z0, z1;
                if (NP == 1) {                                      // z1 as bf16 mode stores it
                    z0 = bf_pack(z[0], z[1]);
                    z1 = bf_pack(z[2], z[3]);
                    zr[0] = bf_lo(z0); zr[1] = bf_hi(z0); zr[2] = bf_lo(z1); zr[3] = bf_hi(z1);
                } else {
#pragma unroll
                    for (int v = 0; v < 4; ++v) zr[v] = z[v];
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float t = fmaf(sc[v], zr[v], sh[v]);
                    a[v] = fmaxf(t, t * p.slope);                   // leaky, 0 <= slope <= 1
                    du[v] = da[v] * (t > 0.f ? 1.f : p.slope);
                    if (!FULL) {
                        a[v] = valid ? a[v] : 0.f;
                        du[v] = valid ? du[v] : 0.f;
                        zr[v] = valid ? zr[v] : 0.f;
                    }
                    s1[v] += du[v];
                    s2[v] = fmaf(du[v], zr[v], s2[v]);              // sum du1 z1; xhat = (z1 - mean) invstd is applied to the sums
                }
                const int off = (r * QT + 16 * h + n) * PA + 4 * g;
                S2AG_DBG_ASSERT(off + 4 <= S2 * QT * PA);
                const unsigned a0 = bf_pack(a[0], a[1]), a1 = bf_pack(a[2], a[3]);
                const unsigned u0 = bf_pack(du[0], du[1]), u1 = bf_pack(du[2], du[3]);
                if (NP != 1 || !FULL) {
                    z0 = bf_pack(zr[0], zr[1]);
                    z1 = bf_pack(zr[2], zr[3]);
                }
                *reinterpret_cast<u32x2*>(&aimg[0][off]) = u32x2{a0, a1};
                *reinterpret_cast<u32x2*>(&uimg[0][off]) = u32x2{u0, u1};
                *reinterpret_cast<u32x2*>(&zimg[0][off]) = u32x2{z0, z1};
                if (NP == 2) {
                    *reinterpret_cast<u32x2*>(&aimg[NP - 1][off]) = u32x2{bf_pack_rest(a[0], a[1], a0), bf_pack_rest(a[2], a[3], a1)};
                    *reinterpret_cast<u32x2*>(&uimg[NP - 1][off]) = u32x2{bf_pack_rest(du[0], du[1], u0), bf_pack_rest(du[2], du[3], u1)};
                    *reinterpret_cast<u32x2*>(&zimg[NP - 1][off]) = u32x2{bf_pack_rest(zr[0], zr[1], z0), bf_pack_rest(zr[2], zr[3], z1)};
                }
            }
        }
        W12_STAMP();
        __syncthreads();
        W12_STAMP();
        // ---- phase 3 -------------------------------------------------------------------------------------------------
        {
            bf16x8 af[NP][NT];
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int i = 0; i < NT; ++i) af[np][i] = tr_frag(dimg[buf][np], (tr_row + NT - 1 - i) * PG + 16 * cot + tr_col, PG);
            const bf16_t* mimg0 = mat == 0 ? uimg[0] : zimg[0];
            const bf16_t* mimg1 = mat == 0 ? uimg[NP - 1] : zimg[NP - 1];
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                const int r = 3 * ph + rr;
                const int ioff = (r * QT + tr_row) * PA + tr_col;
                S2AG_DBG_ASSERT(ioff + 4 * PA + 4 <= S2 * QT * PA && (S2 * S1) * (8 * g + 7) + S1 * r + n < SEGN);
                const bf16x8 bh = tr_frag(aimg[0], ioff, PA);
                if (NP == 2) {
                    const bf16x8 bl = tr_frag(aimg[NP - 1], ioff, PA);
#pragma unroll
                    for (int i = 0; i < NT; ++i)
                        if (r + S2 * i < KS) {
                            accw[rr][i] = mfma32(af[NP - 1][i], bh, accw[rr][i]);
                            accw[rr][i] = mfma32(af[0][i], bl, accw[rr][i]);
                        }
                }
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    if (r + S2 * i < KS) accw[rr][i] = mfma32(af[0][i], bh, accw[rr][i]);
                // conv1: M^T (16 c x 32 q) against the windows (32 q x 16 taps): lane (t = n, g) holds q = 8 g .. 8 g + 7
                unsigned xw[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) xw[j] = spl[(S2 * S1) * (8 * g + j) + S1 * r + n];
                unsigned xh[4], xl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh[j] = __builtin_amdgcn_perm(xw[2 * j + 1], xw[2 * j], 0x05040100u);
                    xl[j] = __builtin_amdgcn_perm(xw[2 * j + 1], xw[2 * j], 0x07060302u);
                }
                const bf16x8 bxh = __builtin_bit_cast(bf16x8, u32x4{xh[0], xh[1], xh[2], xh[3]});
                const bf16x8 bxl = __builtin_bit_cast(bf16x8, u32x4{xl[0], xl[1], xl[2], xl[3]});
                const bf16x8 mh = tr_frag(mimg0, ioff, PA);
                if (NP == 2) acc1 = mfma32(tr_frag(mimg1, ioff, PA), bxh, acc1);
                acc1 = mfma32(mh, bxl, acc1);
                acc1 = mfma32(mh, bxh, acc1);
                if (mat == 0) {                                     // wave-uniform: sum of the windows of the valid frames
                    u32x4 one = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
                    if (!FULL) {
                        unsigned m[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) m[j] = S2 * (c_q0 + 8 * g + j) + r < p.L1 ? 0x3f80u : 0u;
                        one = u32x4{m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16)};
                    }
                    const bf16x8 ones = __builtin_bit_cast(bf16x8, one);
                    accx = mfma32(ones, bxl, accx);
                    accx = mfma32(ones, bxh, accx);
                }
            }
        }
    };

#pragma unroll
    for (int r = 0; r < RING; ++r) fetch(s_beg + r, r);
    int buf = 0;
    for (int s = s_beg; s < s_end; s += RING) {
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            if (s + r < s_end) {                                    // uniform over the workgroup
                W12_STAMP();
                stash(r, buf);
                fetch(s + r + RING, r);
                W12_STAMP();
                __syncthreads();
                if (S2 * (c_q0 + QT) <= p.L1) step(buf, std::true_type{});
                else step(buf, std::false_type{});
                buf ^= 1;
                c_q0 += QT;
                if (c_q0 >= q_wrap) {
                    c_q0 = 0;
                    ++c_clip;
                }
            }
        }
    }

    W12_STAMP();
#undef W12_STAMP
    // ---- the workgroup's partial results ---------------------------------------------------------------------------------
    // dW2 tile: D[co][ci]: co = 16 cot + 4 g + v, ci = n; tap r + 6 i
    float* dst = p.part_w2 + (size_t)blockIdx.x * C2 * KS * C1;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int tap = 3 * ph + rr + S2 * i;
            if (tap < KS) {
#pragma unroll
                for (int v = 0; v < 4; ++v) dst[((size_t)(16 * cot + 4 * g + v) * KS + tap) * C1 + n] = accw[rr][i][v];
            }
        }
    // S[c][t] (c = 4 g + v, t = n) of this wave's phase half; S_x[t] = row 0 of the ones product
#pragma unroll
    for (int v = 0; v < 4; ++v) srow[ph][mat * 256 + (4 * g + v) * 16 + n] = acc1[v];
    if (mat == 0 && g == 0) srow[ph][512 + n] = accx[0];
    // BatchNorm sums: channels 4 g + v
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        double a1 = (double)s1[v], a2 = (double)s2[v];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            a1 += __shfl_xor(a1, m, 64);
            a2 += __shfl_xor(a2, m, 64);
        }
        if (n == 0) {
            red[wave][0][4 * g + v] = a1;
            red[wave][1][4 * g + v] = (a2 - (double)p.mean1[4 * g + v] * a1) * (double)p.inv1[4 * g + v];
        }
    }
    __syncthreads();
    for (int i = tid; i < SROW; i += 256) st_agent_f(p.part_s + (size_t)blockIdx.x * SROW + i, srow[0][i] + srow[1][i]);
    if (tid < 2 * C1) {
        const int which = tid / C1, c = tid - which * C1;
        st_agent(p.stats + ((size_t)which * gridDim.x + blockIdx.x) * C1 + c,
                 (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]));
    }
    // two-level fold (bn_fold_inl.h's two_level_done, with the conv1 rows riding along): the last finisher of a group of 16
    // workgroups sums the group's rows in a fixed order; the last group leader folds BatchNorm 1's backward
    const int R = gridDim.x, grp = blockIdx.x / FOLD_GROUP, ng = fold_groups(R);
    const int gsize = min(FOLD_GROUP, R - grp * FOLD_GROUP);
    if (!last_block_done(p.ticket + 1 + grp, gsize)) return;
    for (int i = tid; i < 2 * C1 + SROW; i += 256) {
        if (i < 2 * C1) {
            const int which = i / C1, c = i - which * C1;
            double v[FOLD_GROUP];
#pragma unroll
            for (int j = 0; j < FOLD_GROUP; ++j)
                v[j] = ld_agent(p.stats + ((size_t)which * R + grp * FOLD_GROUP + min(j, gsize - 1)) * C1 + c);
            double sum = 0.0;
#pragma unroll
            for (int j = 0; j < FOLD_GROUP; ++j)
                if (j < gsize) sum += v[j];
            st_agent(p.stats + (size_t)2 * R * C1 + ((size_t)which * ng + grp) * C1 + c, sum);
        } else {
            const int k = i - 2 * C1;
            float v[FOLD_GROUP];
#pragma unroll
            for (int j = 0; j < FOLD_GROUP; ++j) v[j] = ld_agent_f(p.part_s + (size_t)(grp * FOLD_GROUP + min(j, gsize - 1)) * SROW + k);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < FOLD_GROUP; ++j)
                if (j < gsize) sum += v[j];
            p.part_s[(size_t)(R + grp) * SROW + k] = sum;          // read by the next launch
        }
    }
    if (!last_block_done(p.ticket, ng)) return;
    __shared__ double fred[2][256];
    bn_bwd_fold_body<true>(p.stats + (size_t)2 * R * C1, ng, C1, p.inv_rows, p.gamma1, p.mean1, p.inv1, p.dgamma1, p.dbeta1, p.oca,
                           p.ocb, p.occ, fred);
}

// dW2 (32, 16, 15) += sum of the tiles (b, 32, 15, 16); dW1[c, t] += ca_c S_du[c, t] + cc_c S_z[c, t]
// + cb_c S_x[t] from the group rows.  A block owns 32 consecutive outputs; its 8 groups of 32 threads sum every 8th partial
// row with independent loads in flight and meet in LDS in a fixed order (one writer per element: bit-reproducible).
struct W12FinP {
    const float* part_w2;
    const float* group_s;     // (ngroups, SROW)
    const float* ca;
    const float* cb;
    const float* cc;
    float* dw2;               // nullable
    float* dw1;               // nullable
    int nparts, ngroups, nb2; // partial rows; group rows; blocks of the dW2 / db2 part
};

__device__ __forceinline__ float strided_sum(const float* src, size_t stride, int first, int n) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int b = first;
    for (; b + 56 < n; b += 64) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(b + 8 * j) * stride];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += v[j];
    }
    for (; b < n; b += 8) a[0] += src[(size_t)b * stride];
    return ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
}

__global__ __launch_bounds__(256) void wv12_finish_k(const W12FinP p) {
    __shared__ float red[8][32];
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
    if ((int)blockIdx.x < p.nb2) {
        constexpr int total = C2 * KS * C1;
        const int i = blockIdx.x * 32 + col;
        red[grp][col] = strided_sum(p.part_w2 + i, total, grp, p.nparts);
        __syncthreads();
        if (grp == 0) {
            float t = red[0][col];
#pragma unroll
            for (int k = 1; k < 8; ++k) t += red[k][col];
            const int ci = i % C1, tap = (i / C1) % KS, co = i / (C1 * KS);
            if (p.dw2) p.dw2[((size_t)co * C1 + ci) * KS + tap] += t;
        }
        return;
    }
    // conv1 (one block): output o = c * 16 + t (t = 15: the padding tap, dropped)
    const int o = threadIdx.x, c = o / 16, t = o % 16;
    float su = 0.f, sz = 0.f, sx = 0.f;
    for (int k = 0; k < p.ngroups; ++k) {
        su += p.group_s[(size_t)k * SROW + o];
        sz += p.group_s[(size_t)k * SROW + 256 + o];
        sx += p.group_s[(size_t)k * SROW + 512 + t];
    }
    if (t < KS && p.dw1) p.dw1[c * KS + t] += fmaf(p.ca[c], su, fmaf(p.cc[c], sz, p.cb[c] * sx));
}

FwdFold make_fold(const s2ag_bn_fold_args* f, long long rows) {
    return FwdFold{f->ticket, f->gamma, f->beta, f->running_mean, f->running_var, f->num_batches_tracked, f->eps, f->momentum,
                   f->repeat, rows, f->scale, f->shift, f->mean, f->invstd};
}
bool fold_ok(const s2ag_bn_fold_args* f) {
    return f->ticket && f->gamma && f->beta && f->running_mean && f->running_var && f->scale && f->shift && f->mean && f->invstd &&
           f->repeat >= 1;
}
int stats_blocks(int N, int L1) {
    const long long chunks = (long long)N * cdiv(L1, 128);
    const long long b = (chunks + 3) / 4;
    return (int)(b < 1024 ? b : 1024);
}
int fwd_chunks(int N, int L2, int* LC) {
    int per_clip = cdiv(512, N);
    if (per_clip < 1) per_clip = 1;
    *LC = cdiv(cdiv(L2, per_clip), 64) * 64;
    return cdiv(L2, *LC);
}
unsigned long long* g_w12_trace = nullptr;
int g_bwd_block_cap = 0;            // s2ag_wave12_set_bwd_block_cap (tests: few workgroups walk many steps and cross clips)
int bwd_blocks(int N, int L1, int dz_f32) {
    // two workgroups per CU in bf16 mode; the two-piece form holds twice the operands in registers: one per CU
    long long steps = (long long)N * cdiv(cdiv(L1, S2), 32), cap = dz_f32 ? 256 : 512;
    if (g_bwd_block_cap > 0 && g_bwd_block_cap < cap) cap = g_bwd_block_cap;
    return (int)(steps < cap ? steps : cap);
}
bool geom_ok(int N, int Lin, int L1, int L2, int pad) {
    return N > 0 && Lin > 0 && pad >= 0 && L1 == (Lin + 2 * pad - KS) / S1 + 1 && L1 >= KS && L2 == (L1 - KS) / S2 + 1;
}
}  // namespace

// ---- C ABI ---------------------------------------------------------------------------------------------------------
extern "C" int s2ag_wave12_pack_elems(void) { return W12_PACK; }

extern "C" int s2ag_wave12_pack(const float* w1, const float* w2, void* packed, void* stream) {
    if (!w1 || !w2 || !packed || ((uintptr_t)packed & 15)) return S2AG_E_BADARG;
    hipLaunchKernelGGL(wv12_pack_k, dim3(cdiv(W12_PACK_THREADS, 256)), dim3(256), 0, (hipStream_t)stream, w1, w2,
                       static_cast<bf16_t*>(packed));
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_wave12_stats_rows(int N, int L1) { return (N <= 0 || L1 <= 0) ? S2AG_E_BADARG : stats_blocks(N, L1); }

extern "C" int s2ag_wave12_stats(const float* x, const void* packed, const float* b1, double* partials,
                                 const s2ag_bn_fold_args* fold, int round_bf16, int N, int Lin, int L1, int pad, void* stream) {
    if (!x || !packed || !b1 || !partials || !fold || !fold_ok(fold) || N <= 0 || Lin <= 0 || L1 <= 0) return S2AG_E_BADARG;
    if (L1 != (Lin + 2 * pad - KS) / S1 + 1) return S2AG_E_BADARG;
    W12StatsP p{};
    p.x = x; p.wp = static_cast<const bf16_t*>(packed); p.b1 = b1; p.stats = partials;
    p.fold = make_fold(fold, (long long)N * L1);
    p.N = N; p.Lin = Lin; p.L1 = L1; p.pad = pad; p.cpc = cdiv(L1, 128); p.total = N * p.cpc;
    const int blocks = stats_blocks(N, L1);
    if (round_bf16) hipLaunchKernelGGL(wv12_stats_k<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(wv12_stats_k<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_wave12_act_signs(const float* x, const void* packed, const float* b1, const float* scale1,
                                     const float* shift1, int round_bf16, unsigned char* signs, int N, int Lin, int L1, int pad,
                                     void* stream) {
    if (!x || !packed || !b1 || !scale1 || !shift1 || !signs || N <= 0 || Lin <= 0 || L1 <= 0) return S2AG_E_BADARG;
    if (L1 != (Lin + 2 * pad - KS) / S1 + 1 || ((uintptr_t)signs & 3)) return S2AG_E_BADARG;
    W12SignP p{};
    p.x = x; p.wp = static_cast<const bf16_t*>(packed); p.b1 = b1; p.sc1 = scale1; p.sh1 = shift1; p.sign = signs;
    p.N = N; p.Lin = Lin; p.L1 = L1; p.pad = pad; p.cpc = cdiv(L1, 128); p.total = N * p.cpc;
    const int blocks = stats_blocks(N, L1);
    if (round_bf16) hipLaunchKernelGGL(wv12_signs_k<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(wv12_signs_k<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_wave12_fwd_rows(int N, int L2) {
    if (N <= 0 || L2 <= 0) return S2AG_E_BADARG;
    int LC;
    return N * fwd_chunks(N, L2, &LC);
}

extern "C" int s2ag_wave12_fwd(const float* x, const void* packed, const float* b1, const float* scale1, const float* shift1,
                               float slope, const float* b2, void* z2, int out_f32, double* partials,
                               const s2ag_bn_fold_args* fold, int N, int Lin, int L1, int L2, int pad, void* stream) {
    if (!x || !packed || !b1 || !scale1 || !shift1 || !z2 || !geom_ok(N, Lin, L1, L2, pad)) return S2AG_E_BADARG;
    if (!(slope >= 0.f && slope <= 1.f)) return S2AG_E_UNSUPPORTED;            // leaky(t) = max(t, slope t)
    if (fold && (!partials || !fold_ok(fold))) return S2AG_E_BADARG;
    if (((uintptr_t)z2 & 15) || ((uintptr_t)packed & 15)) return S2AG_E_BADARG;
    W12FwdP p{};
    p.x = x; p.wp = static_cast<const bf16_t*>(packed); p.b1 = b1; p.sc1 = scale1; p.sh1 = shift1; p.slope = slope;
    p.b2 = b2; p.z2 = z2; p.stats = partials;
    if (fold) p.fold = make_fold(fold, (long long)N * L2);
    p.N = N; p.Lin = Lin; p.L1 = L1; p.L2 = L2; p.pad = pad;
    p.chunks = fwd_chunks(N, L2, &p.LC);
    const dim3 grid(N * p.chunks);
    if (out_f32) hipLaunchKernelGGL(wv12_fwd_k<2>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(wv12_fwd_k<1>, grid, dim3(256), 0, (hipStream_t)stream, p);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_wave12_set_trace(void* buf) {
    g_w12_trace = static_cast<unsigned long long*>(buf);
    return 0;
}

extern "C" int s2ag_wave12_set_bwd_block_cap(int cap) {
    const int prev = g_bwd_block_cap;
    g_bwd_block_cap = cap > 0 ? cap : 0;
    return prev;
}

extern "C" int s2ag_wave12_bwd_blocks(int N, int L1, int dz_f32) {
    return (N <= 0 || L1 <= 0) ? S2AG_E_BADARG : bwd_blocks(N, L1, dz_f32);
}

extern "C" int s2ag_wave12_bwd(const s2ag_wave12_bwd_args* a, void* stream) {
    if (!a || !a->x || !a->packed || !a->b1 || !a->scale1 || !a->shift1 || !a->mean1 || !a->invstd1 || !a->gamma1 || !a->dz ||
        !a->part_w2 || !a->part_s || !a->stats || !a->ticket || !a->ca1 || !a->cb1 || !a->cc1 ||
        !geom_ok(a->N, a->Lin, a->L1, a->L2, a->pad))
        return S2AG_E_BADARG;
    if (!a->dz_f32 && (!a->z2 || !a->ca2 || !a->cb2 || !a->cc2)) return S2AG_E_BADARG;
    if (!(a->slope >= 0.f && a->slope <= 1.f)) return S2AG_E_UNSUPPORTED;      // leaky(t) = max(t, slope t)
    if (((uintptr_t)a->dz & 15) || ((uintptr_t)a->z2 & 15) || ((uintptr_t)a->packed & 15)) return S2AG_E_BADARG;
    W12BwdP p{};
    p.x = a->x; p.wp = static_cast<const bf16_t*>(a->packed); p.b1 = a->b1; p.sc1 = a->scale1; p.sh1 = a->shift1;
    p.mean1 = a->mean1; p.inv1 = a->invstd1; p.slope = a->slope; p.dz = a->dz; p.z2 = static_cast<const bf16_t*>(a->z2);
    p.ca = a->ca2; p.cb = a->cb2; p.cc = a->cc2; p.part_w2 = a->part_w2; p.part_s = a->part_s;
    p.stats = a->stats; p.ticket = a->ticket; p.gamma1 = a->gamma1; p.dgamma1 = a->dgamma1;
    p.dbeta1 = a->dbeta1; p.oca = a->ca1; p.ocb = a->cb1; p.occ = a->cc1;
    p.inv_rows = 1.0 / ((double)a->N * a->L1);
    p.N = a->N; p.Lin = a->Lin; p.L1 = a->L1; p.L2 = a->L2; p.pad = a->pad;
    p.QS = cdiv(cdiv(a->L1, S2), 32);
    p.trace = g_w12_trace;
    p.total_steps = a->N * p.QS;
    const int blocks = bwd_blocks(a->N, a->L1, a->dz_f32);
    hipStream_t s = (hipStream_t)stream;
    if (a->dz_f32) hipLaunchKernelGGL((wv12_bwd_k<2, false>), dim3(blocks), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wv12_bwd_k<1, true>), dim3(blocks), dim3(256), 0, s, p);
    W12FinP f{};
    f.part_w2 = a->part_w2; f.group_s = a->part_s + (size_t)blocks * SROW;
    f.ca = a->ca1; f.cb = a->cb1; f.cc = a->cc1; f.dw2 = a->dw2; f.dw1 = a->dw1;
    f.nparts = blocks; f.ngroups = fold_groups(blocks); f.nb2 = C2 * KS * C1 / 32;
    hipLaunchKernelGGL(wv12_finish_k, dim3(f.nb2 + 1), dim3(256), 0, s, f);
    S2AG_LAUNCH_CHECK();
    return 0;
}
