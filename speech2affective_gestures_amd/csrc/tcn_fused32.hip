// Clip-resident TemporalConvNet FORWARD for the default (fp32) step: the text encoder's four TemporalBlocks
// (net/tcn.py:16-64) in one launch, fp32 activations resident in LDS, every product formed from two bf16 pieces per operand
// (hi = rn(v), lo = rn(v - hi); lo*hi + hi*lo + hi*hi accumulated in fp32 by the MFMA: the 16-mantissa-bit products of
// conv_sp_k, the layer-by-layer kernel this replaces -- bench.py `matrix_products`).
//
// Same idea as csrc/tcn_fused.hip (bf16 mode): clips never mix along time, so a workgroup owns ONE clip (T <= 48 rows),
// keeps it as fp32 rows [T][320] in LDS (1 296-byte pitch: 16 consecutive rows tile the 64 banks for 16-byte reads) through
// all eight convs, and streams the weights from L2 in MFMA-fragment order -- two planes (hi, lo) per conv, prepared once
// per optimizer step.  The activation operand is split on the fly: a lane reads its 8 consecutive fp32 of a row and makes
// the hi and the lo fragment (8 packed converts + 8 subtractions).  Per K tile and wave: 10 weight fragments from L2,
// 3 x 2 activation fragments from LDS, 45 MFMAs.
// What the (layer-by-layer) backward pass needs goes to HBM once: h1, h2 (post-dropout conv outputs) and the block output
// y, as fp32 (clips*T, C) rows -- exactly the tensors the per-layer forward kernels would have left.
#include "s2ag_common.h"

namespace s2ag {                                        // csrc/tcn32p.hip: the opt-in two-clips-per-workgroup form (option TCN32_PAIR)
bool tcn32p_supported(int n_clips, int n_passes, int save_clips, int T, int ncl);
int tcn32p_fwd_launch(const void* params, int n_passes, const void* const* rngs, int ncl, hipStream_t st);
int tcn32p_bwd_launch(const void* params, int ncl, hipStream_t st);
}

namespace {
using namespace s2ag;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef unsigned short bf16_t;

#include "tcn_fused32_shared.h"

// keep bits of one pass in the epilogue's register layout: bit (i*MT + mt)*4 + c of thread (wave, lane)
__global__ __launch_bounds__(256) void tcn32_keep_k(const T32P p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = blockIdx.x, cv = blockIdx.y;
    const long long row0 = (long long)wg * p.T;
    const SiteKey key = site_key(p.rng, p.site[cv]);
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < CT_W; ++i) {
        const int co = (wave * CT_W + i) * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = mt * 16 + (lane & 15);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int b = (i * MT + mt) * 4 + c;
                bool k = false;
                if (m < p.T && co + c < p.C)
                    k = keep_scale(key, (unsigned long long)(row0 + m) * p.C + co + c, p.drop_p, p.inv_keep) != 0.f;
                w[b >> 5] |= k ? (1u << (b & 31)) : 0u;
            }
        }
    }
    p.keep[((size_t)cv * p.keep_total + p.keep_off + wg) * 256 + tid] = u32x4{w[0], w[1], w[2], w[3]};
}

// acc += conv over the fp32 LDS rows at `src`: K tile kt = tap kt / KT_TAP (rows q - d forward, q + d backward for tap 0;
// q for tap 1), channels (kt % KT_TAP)*32 .. +32.  wh / wl: this wave's hi / lo weight fragments (+ lane); a ring of three
// K tiles in flight.  The activation fragments are split here: hi = rn(v), lo = rn(v - hi).
template <bool BWD>
__device__ __forceinline__ void conv32_tile(const float* sm, int src, int Z, const u32x4* __restrict__ wh,
                                            const u32x4* __restrict__ wl, int d, int T, int lane, f32x4 (&acc)[CT_W][MT]) {
    int off0[MT], off1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 16 + (lane & 15);
        const bool ok0 = m < T && (BWD ? (m + d < T) : (m >= d));
        off1[mt] = (m < T ? src + m * PITCH : Z) + (lane >> 4) * 8;
        off0[mt] = (ok0 ? src + (BWD ? m + d : m - d) * PITCH : Z) + (lane >> 4) * 8;
    }
    u32x4 ah[3][CT_W], al[3][CT_W];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < CT_W; ++i) {
            ah[s][i] = wh[(i * NKT + s) * 64];
            al[s][i] = wl[(i * NKT + s) * 64];
        }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int s = kt % 3;
        const bool t0 = kt < KT_TAP;
        const int c0 = (t0 ? kt : kt - KT_TAP) * 32;
        bf16x8 bh[MT], bl[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float* r = sm + (t0 ? off0[mt] : off1[mt]) + c0;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(r), v1 = *reinterpret_cast<const f32x4*>(r + 4);
            const unsigned h0 = pk_bf16(v0[0], v0[1]), h1_ = pk_bf16(v0[2], v0[3]);
            const unsigned h2_ = pk_bf16(v1[0], v1[1]), h3 = pk_bf16(v1[2], v1[3]);
            const unsigned l0 = pk_bf16(v0[0] - __uint_as_float(h0 << 16), v0[1] - __uint_as_float(h0 & 0xffff0000u));
            const unsigned l1 = pk_bf16(v0[2] - __uint_as_float(h1_ << 16), v0[3] - __uint_as_float(h1_ & 0xffff0000u));
            const unsigned l2 = pk_bf16(v1[0] - __uint_as_float(h2_ << 16), v1[1] - __uint_as_float(h2_ & 0xffff0000u));
            const unsigned l3 = pk_bf16(v1[2] - __uint_as_float(h3 << 16), v1[3] - __uint_as_float(h3 & 0xffff0000u));
            bh[mt] = __builtin_bit_cast(bf16x8, u32x4{h0, h1_, h2_, h3});
            bl[mt] = __builtin_bit_cast(bf16x8, u32x4{l0, l1, l2, l3});
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CT_W; ++i) {
            const bf16x8 avh = __builtin_bit_cast(bf16x8, ah[s][i]), avl = __builtin_bit_cast(bf16x8, al[s][i]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(avl, bh[mt], acc[i][mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(avh, bl[mt], acc[i][mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(avh, bh[mt], acc[i][mt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 3 < NKT) {
#pragma unroll
            for (int i = 0; i < CT_W; ++i) {
                ah[s][i] = wh[(i * NKT + kt + 3) * 64];
                al[s][i] = wl[(i * NKT + kt + 3) * 64];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ __launch_bounds__(256) void tcn32_fwd_k(const T32P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, C = p.C;
    const int X = 0, H1 = T * PITCH, H2 = 2 * T * PITCH, Z = 3 * T * PITCH;
    const long long row0 = (long long)blockIdx.x * T;
    const int cpr = C / 4;                                       // 16-byte chunks of an HBM row

    // rows in: (row0 + m, 0..C) -> X[m][0..C), pad channels zero
    for (int idx = tid; idx < T * (CP / 4); idx += 256) {
        const int m = idx / (CP / 4), kc = idx - m * (CP / 4);
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kc < cpr) v = *reinterpret_cast<const f32x4*>(p.x + (row0 + m) * C + kc * 4);
        *reinterpret_cast<f32x4*>(sm + X + m * PITCH + kc * 4) = v;
        *reinterpret_cast<f32x4*>(sm + H1 + m * PITCH + kc * 4) = f32x4{0.f, 0.f, 0.f, 0.f};      // pad channels of H1 stay zero
    }
    for (int i = tid; i < PITCH; i += 256) sm[Z + i] = 0.f;
    __syncthreads();

    const bool drop = p.drop_p > 0.f;
    const float ik = p.inv_keep;
    f32x4 acc[CT_W][MT];
    for (int blk = 0; blk < p.n_blocks; ++blk) {
        const int d = p.dil[blk];
#pragma unroll 1
        for (int j = 0; j < 2; ++j) {
            const int cv = 2 * blk + j;
            const int src = j == 0 ? X : H1;
            const u32x4* wh = reinterpret_cast<const u32x4*>(p.wfrag + (long long)(4 * cv) * FRAG) + (wave * CT_W * NKT) * 64 + lane;
            const u32x4* wl = reinterpret_cast<const u32x4*>(p.wfrag + (long long)(4 * cv + 1) * FRAG) + (wave * CT_W * NKT) * 64 + lane;
            const float* bias = p.bias[cv];
            float bv[CT_W][4];
#pragma unroll
            for (int i = 0; i < CT_W; ++i) {
                const int co = (wave * CT_W + i) * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int c = 0; c < 4; ++c) bv[i][c] = (bias && co + c < C) ? bias[co + c] : 0.f;
            }
            u32x4 kv = u32x4{0u, 0u, 0u, 0u};
            if (drop) kv = p.keep[((size_t)cv * p.keep_total + blockIdx.x) * 256 + tid];
#pragma unroll
            for (int i = 0; i < CT_W; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[i][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            conv32_tile<false>(sm, src, Z, wh, wl, d, T, lane, acc);
            // epilogue: bias, ReLU, dropout; conv2 also adds the residual and writes the block output over the block input
            const unsigned kw[4] = {kv.x, kv.y, kv.z, kv.w};
            const int dst = j == 0 ? H1 : H2;
#pragma unroll
            for (int i = 0; i < CT_W; ++i) {
                const int co = (wave * CT_W + i) * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = mt * 16 + (lane & 15);
                    if (m >= T) continue;
                    f32x4 v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int b = (i * MT + mt) * 4 + c;
                        float t = fmaxf(acc[i][mt][c] + bv[i][c], 0.f);
                        if (drop) t = (kw[b >> 5] >> (b & 31)) & 1u ? t * ik : 0.f;
                        v[c] = t;
                    }
                    *reinterpret_cast<f32x4*>(sm + dst + m * PITCH + co) = v;
                    if (j == 1) {
                        f32x4* xp = reinterpret_cast<f32x4*>(sm + X + m * PITCH + co);
                        const f32x4 xv = *xp;
                        *xp = f32x4{fmaxf(v[0] + xv[0], 0.f), fmaxf(v[1] + xv[1], 0.f), fmaxf(v[2] + xv[2], 0.f), fmaxf(v[3] + xv[3], 0.f)};
                    }
                }
            }
            __syncthreads();
            // rows out (fp32 (rows, C), 16 bytes per thread and access)
            for (int idx = tid; idx < T * cpr; idx += 256) {
                const int m = idx / cpr, kc = idx - m * cpr;
                const long long go = (row0 + m) * C + kc * 4;
                const bool save = (int)blockIdx.x < p.save_clips;      // a no-grad pass of a lockstep batch keeps nothing
                if (j == 0) {
                    if (save)
                        *reinterpret_cast<f32x4*>(p.h1[blk] + go) = *reinterpret_cast<const f32x4*>(sm + H1 + m * PITCH + kc * 4);
                } else {
                    if (save)
                        *reinterpret_cast<f32x4*>(p.h2[blk] + go) = *reinterpret_cast<const f32x4*>(sm + H2 + m * PITCH + kc * 4);
                    if (save || blk == p.n_blocks - 1)
                        *reinterpret_cast<f32x4*>(p.y[blk] + go) = *reinterpret_cast<const f32x4*>(sm + X + m * PITCH + kc * 4);
                }
            }
        }
    }
}

struct Pack32 {
    const float* w[2 * S2AG_TCN_MAX_BLOCKS];
    int n, C;
    bf16_t* out;
};
// element e = (((((cv*2 + dir)*2 + plane)*NCT + ct)*NKT + kt)*64 + lane)*8 + i holds piece `plane` of A[row][k], row = ct*16 +
// (lane & 15), k = kt*32 + (lane >> 4)*8 + i = tap*320 + channel:  forward A[co][tap, ci] = w[co][tap][ci];
// data gradient A[ci][tap, co] = w[co][tap][ci]
__global__ __launch_bounds__(256) void tcn32_pack_k(const Pack32 p) {
    const long long total = (long long)p.n * 4 * FRAG;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int i = (int)(e & 7), lane = (int)((e >> 3) & 63);
        long long r = e >> 9;
        const int kt = (int)(r % NKT);
        r /= NKT;
        const int ct = (int)(r % NCT);
        r /= NCT;
        const int plane = (int)(r & 1), dir = (int)((r >> 1) & 1), cv = (int)(r >> 2);
        const int row = ct * 16 + (lane & 15), k = kt * 32 + (lane >> 4) * 8 + i;
        const int tap = k / CP, ch = k - tap * CP;
        const int co = dir == 0 ? row : ch, ci = dir == 0 ? ch : row;
        float v = 0.f;
        if (co < p.C && ci < p.C) v = p.w[cv][((long long)co * 2 + tap) * p.C + ci];
        const unsigned hi = pk_bf16(v, 0.f) & 0xffffu;
        const float rest = v - __uint_as_float(hi << 16);
        p.out[e] = (bf16_t)(plane == 0 ? hi : (pk_bf16(rest, 0.f) & 0xffffu));
    }
}

// The chain of data gradients of all blocks in one launch: G <- G * [y > 0]; P2 <- G * [h2 > 0] / keep (h2 = mask * relu(pre)
// is positive exactly where the element was kept and pre > 0); P1 <- dgrad_conv2(P2) * [h1 > 0] / keep; G <- dgrad_conv1(P1) + G.
// P2 / P1 go to HBM as gp2 / gp1: the `gy` operands of the eight weight gradients (s2ag_f32_wgrad_tr, one launch).
__global__ __launch_bounds__(256) void tcn32_bwd_k(const T32P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, C = p.C;
    const int G = 0, P2 = T * PITCH, P1 = 2 * T * PITCH, Z = 3 * T * PITCH;
    const long long row0 = (long long)blockIdx.x * T;
    const int cpr = C / 4;
    const float ik = p.inv_keep;
    for (int idx = tid; idx < T * (CP / 4); idx += 256) {
        const int m = idx / (CP / 4), kc = idx - m * (CP / 4);
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kc < cpr) v = *reinterpret_cast<const f32x4*>(p.gy + (row0 + m) * C + kc * 4);
        *reinterpret_cast<f32x4*>(sm + G + m * PITCH + kc * 4) = v;
        *reinterpret_cast<f32x4*>(sm + P2 + m * PITCH + kc * 4) = f32x4{0.f, 0.f, 0.f, 0.f};     // pad channels stay zero
        *reinterpret_cast<f32x4*>(sm + P1 + m * PITCH + kc * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int i = tid; i < PITCH; i += 256) sm[Z + i] = 0.f;
    __syncthreads();
    f32x4 acc[CT_W][MT];
    for (int blk = p.n_blocks - 1; blk >= 0; --blk) {
        const int d = p.dil[blk];
        // (a) element-wise; h1 parks in P1 for the epilogue of (b)
        for (int idx = tid; idx < T * cpr; idx += 256) {
            const int m = idx / cpr, kc = idx - m * cpr;
            const long long go = (row0 + m) * C + kc * 4;
            const int lo = m * PITCH + kc * 4;
            const f32x4 yv = *reinterpret_cast<const f32x4*>(p.y[blk] + go);
            const f32x4 hv = *reinterpret_cast<const f32x4*>(p.h2[blk] + go);
            const f32x4 h1v = *reinterpret_cast<const f32x4*>(p.h1[blk] + go);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(sm + G + lo);
            f32x4 gs, p2;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                gs[c] = yv[c] > 0.f ? gv[c] : 0.f;
                p2[c] = hv[c] > 0.f ? gs[c] * ik : 0.f;
            }
            *reinterpret_cast<f32x4*>(sm + G + lo) = gs;
            *reinterpret_cast<f32x4*>(sm + P2 + lo) = p2;
            *reinterpret_cast<f32x4*>(sm + P1 + lo) = h1v;
            *reinterpret_cast<f32x4*>(p.gp2[blk] + go) = p2;
        }
        __syncthreads();
        // (b) P1 <- dgrad_conv2(P2) * [h1 > 0] / keep
        {
            const int cv = 2 * blk + 1;
            const u32x4* wh = reinterpret_cast<const u32x4*>(p.wfrag + (long long)(4 * cv + 2) * FRAG) + (wave * CT_W * NKT) * 64 + lane;
            const u32x4* wl = reinterpret_cast<const u32x4*>(p.wfrag + (long long)(4 * cv + 3) * FRAG) + (wave * CT_W * NKT) * 64 + lane;
#pragma unroll
            for (int i = 0; i < CT_W; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[i][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            conv32_tile<true>(sm, P2, Z, wh, wl, d, T, lane, acc);
#pragma unroll
            for (int i = 0; i < CT_W; ++i) {
                const int co = (wave * CT_W + i) * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = mt * 16 + (lane & 15);
                    if (m >= T) continue;
                    f32x4* hp = reinterpret_cast<f32x4*>(sm + P1 + m * PITCH + co);
                    const f32x4 hv = *hp;
                    *hp = f32x4{hv[0] > 0.f ? acc[i][mt][0] * ik : 0.f, hv[1] > 0.f ? acc[i][mt][1] * ik : 0.f,
                                hv[2] > 0.f ? acc[i][mt][2] * ik : 0.f, hv[3] > 0.f ? acc[i][mt][3] * ik : 0.f};
                }
            }
        }
        __syncthreads();
        for (int idx = tid; idx < T * cpr; idx += 256) {
            const int m = idx / cpr, kc = idx - m * cpr;
            *reinterpret_cast<f32x4*>(p.gp1[blk] + (row0 + m) * C + kc * 4) = *reinterpret_cast<const f32x4*>(sm + P1 + m * PITCH + kc * 4);
        }
        // (c) G <- dgrad_conv1(P1) + G
        {
            const int cv = 2 * blk;
            const u32x4* wh = reinterpret_cast<const u32x4*>(p.wfrag + (long long)(4 * cv + 2) * FRAG) + (wave * CT_W * NKT) * 64 + lane;
            const u32x4* wl = reinterpret_cast<const u32x4*>(p.wfrag + (long long)(4 * cv + 3) * FRAG) + (wave * CT_W * NKT) * 64 + lane;
#pragma unroll
            for (int i = 0; i < CT_W; ++i)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[i][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            conv32_tile<true>(sm, P1, Z, wh, wl, d, T, lane, acc);
#pragma unroll
            for (int i = 0; i < CT_W; ++i) {
                const int co = (wave * CT_W + i) * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m = mt * 16 + (lane & 15);
                    if (m >= T) continue;
                    f32x4* gp = reinterpret_cast<f32x4*>(sm + G + m * PITCH + co);
                    const f32x4 gv = *gp;
                    *gp = f32x4{acc[i][mt][0] + gv[0], acc[i][mt][1] + gv[1], acc[i][mt][2] + gv[2], acc[i][mt][3] + gv[3]};
                }
            }
        }
        __syncthreads();
    }
    for (int idx = tid; idx < T * cpr; idx += 256) {
        const int m = idx / cpr, kc = idx - m * cpr;
        *reinterpret_cast<f32x4*>(p.gx + (row0 + m) * C + kc * 4) = *reinterpret_cast<const f32x4*>(sm + G + m * PITCH + kc * 4);
    }
}
}  // namespace

// T <= 40: three fp32 row buffers of T x 1 296 B + the zero row must fit the 160 KB of LDS
extern "C" int s2ag_tcn32_supported(int T, int C, int ksize) { return ksize == 2 && T >= 1 && T <= 40 && C > 256 && C <= CP && (C & 3) == 0; }
extern "C" long long s2ag_tcn32_pack_elems(int n_convs) { return (long long)n_convs * 4 * FRAG; }
extern "C" long long s2ag_tcn32_keep_bytes(int n_clips, int n_blocks) { return (long long)n_clips * 2 * n_blocks * 256 * 16; }

extern "C" int s2ag_tcn32_pack(const float* const* w, int n_convs, int C, void* wfrag, void* stream) {
    if (!w || !wfrag || n_convs < 1 || n_convs > 2 * S2AG_TCN_MAX_BLOCKS || C < 1 || C > CP) return S2AG_E_BADARG;
    Pack32 p{};
    for (int k = 0; k < n_convs; ++k) {
        if (!w[k]) return S2AG_E_BADARG;
        p.w[k] = w[k];
    }
    p.n = n_convs; p.C = C; p.out = static_cast<bf16_t*>(wfrag);
    hipLaunchKernelGGL(tcn32_pack_k, dim3(2048), dim3(256), 0, (hipStream_t)stream, p);
    S2AG_LAUNCH_CHECK();
    return 0;
}

static int tcn32_fwd_impl(const s2ag_tcn32_args* a, int n_passes, const void* const* rngs, int save_clips, void* stream) {
    if (!a || !a->x || !a->wfrag || a->n_blocks < 1 || a->n_blocks > S2AG_TCN_MAX_BLOCKS || a->n_clips <= 0) return S2AG_E_BADARG;
    if (!s2ag_tcn32_supported(a->T, a->C, 2)) return S2AG_E_UNSUPPORTED;
    if (n_passes < 1 || a->n_clips % n_passes != 0 || save_clips < 0 || save_clips > a->n_clips) return S2AG_E_BADARG;
    if (a->drop_p < 0.f || a->drop_p >= 1.f || (a->drop_p > 0.f && (!rngs || !a->keep))) return S2AG_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(a->x)) & 15) return S2AG_E_BADARG;
    T32P p{};
    p.x = a->x; p.wfrag = static_cast<const bf16_t*>(a->wfrag);
    for (int b = 0; b < a->n_blocks; ++b) {
        if (!a->h1[b] || !a->h2[b] || !a->y[b] || a->dil[b] < 1) return S2AG_E_BADARG;
        p.h1[b] = a->h1[b]; p.h2[b] = a->h2[b]; p.y[b] = a->y[b];
        p.dil[b] = a->dil[b];
        for (int j = 0; j < 2; ++j) {
            p.bias[2 * b + j] = a->bias[2 * b + j];
            p.site[2 * b + j] = a->site[2 * b + j];
        }
    }
    p.n_blocks = a->n_blocks; p.n_clips = a->n_clips; p.T = a->T; p.C = a->C;
    p.drop_p = a->drop_p;
    p.inv_keep = a->drop_p > 0.f ? 1.f / (1.f - a->drop_p) : 1.f;
    p.keep = static_cast<u32x4*>(a->keep);
    p.keep_total = a->n_clips; p.keep_off = 0; p.save_clips = save_clips;
    // opt-in variant (1: two clips per workgroup, 2: one): bit-identical h1 / h2 / y
    if (const int v = s2ag::option(s2ag::OPT_TCN32_PAIR); v && s2ag::tcn32p_supported(a->n_clips, n_passes, save_clips, a->T, v == 2 ? 1 : 2))
        return s2ag::tcn32p_fwd_launch(&p, n_passes, rngs, v == 2 ? 1 : 2, (hipStream_t)stream);
    const size_t lds = (size_t)(3 * p.T + 1) * PITCH * sizeof(float);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)tcn32_fwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return S2AG_E_UNSUPPORTED;
        attr = true;
    }
    if (p.drop_p > 0.f) {
        const int per = a->n_clips / n_passes;          // the keep bits of every pass from ITS noise snapshot, clip index
        for (int k = 0; k < n_passes; ++k) {            // relative to the pass: exactly the bits of a pass run alone
            if (!rngs[k]) return S2AG_E_BADARG;
            T32P q = p;
            q.rng = static_cast<const unsigned long long*>(rngs[k]);
            q.keep_off = k * per;
            hipLaunchKernelGGL(tcn32_keep_k, dim3(per, 2 * p.n_blocks), dim3(256), 0, (hipStream_t)stream, q);
        }
    }
    hipLaunchKernelGGL(tcn32_fwd_k, dim3(p.n_clips), dim3(256), lds, (hipStream_t)stream, p);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_tcn32_fwd(const s2ag_tcn32_args* a, void* stream) {
    if (!a) return S2AG_E_BADARG;
    const void* r[1] = {a->rng};
    return tcn32_fwd_impl(a, 1, r, a->n_clips, stream);
}

extern "C" int s2ag_tcn32_fwd_passes(const s2ag_tcn32_args* a, int n_passes, const void* const* rngs, int save_clips,
                                     void* stream) {
    return tcn32_fwd_impl(a, n_passes, rngs, save_clips, stream);
}

extern "C" int s2ag_tcn32_bwd(const s2ag_tcn32_args* a, void* stream) {
    if (!a || !a->wfrag || !a->gy || !a->gx || a->n_blocks < 1 || a->n_blocks > S2AG_TCN_MAX_BLOCKS || a->n_clips <= 0) return S2AG_E_BADARG;
    if (!s2ag_tcn32_supported(a->T, a->C, 2)) return S2AG_E_UNSUPPORTED;
    if (a->drop_p < 0.f || a->drop_p >= 1.f) return S2AG_E_BADARG;
    T32P p{};
    p.wfrag = static_cast<const bf16_t*>(a->wfrag);
    p.gy = a->gy; p.gx = a->gx;
    for (int b = 0; b < a->n_blocks; ++b) {
        if (!a->h1[b] || !a->h2[b] || !a->y[b] || !a->gp1[b] || !a->gp2[b] || a->dil[b] < 1) return S2AG_E_BADARG;
        p.h1[b] = a->h1[b]; p.h2[b] = a->h2[b]; p.y[b] = a->y[b];
        p.gp1[b] = a->gp1[b]; p.gp2[b] = a->gp2[b];
        p.dil[b] = a->dil[b];
    }
    p.n_blocks = a->n_blocks; p.n_clips = a->n_clips; p.T = a->T; p.C = a->C;
    p.drop_p = a->drop_p;
    p.inv_keep = a->drop_p > 0.f ? 1.f / (1.f - a->drop_p) : 1.f;
    // opt-in variant: bit-identical gp1 / gp2 / gx
    if (const int v = s2ag::option(s2ag::OPT_TCN32_PAIR); v && s2ag::tcn32p_supported(a->n_clips, 1, a->n_clips, a->T, v == 2 ? 1 : 2))
        return s2ag::tcn32p_bwd_launch(&p, v == 2 ? 1 : 2, (hipStream_t)stream);
    const size_t lds = (size_t)(3 * p.T + 1) * PITCH * sizeof(float);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)tcn32_bwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return S2AG_E_UNSUPPORTED;
        attr = true;
    }
    hipLaunchKernelGGL(tcn32_bwd_k, dim3(p.n_clips), dim3(256), lds, (hipStream_t)stream, p);
    S2AG_LAUNCH_CHECK();
    return 0;
}
