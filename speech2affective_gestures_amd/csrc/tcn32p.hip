// OPT-IN VARIANT of the clip-resident TemporalConvNet of the fp32 step (csrc/tcn_fused32.hip: tcn32_fwd_k / tcn32_bwd_k; the
// four TemporalBlocks of net/tcn.py:16-64 inside TextEncoderTCN, net/multimodal_context_net_v2.py:61-91): TWO clips per
// workgroup.  Config switch TCN32_PAIR (default OFF; tools/ab_variants.py times it against the default).
//
// Why.  The default kernel gives every clip a workgroup that streams ALL weights of the eight convs in MFMA-fragment order:
// 8 convs x 2 planes x 410 KB = 6.5 MB per workgroup, 1.7 GB per launch at B = 256 -- more than an XCD's 4 MB L2 holds, so
// it comes through the Infinity Cache at the rate the fabric allows (profiles/r03_cfg3_roofline_table.md: tcn32_fwd_k 143 us,
// tcn32_bwd_k 148 us, 33 % MFMA-busy; 48 us of MFMA work per workgroup), and a clip's 34 frames fill 34 of the 48 rows of its
// three 16-row MFMA tiles.  With two clips behind the same weight fragments
//   * every weight byte is streamed once per TWO clips (half the bytes per launch),
//   * 68 of 80 rows of five tiles are real (15 % padding instead of 29 %),
//   * a K tile is 75 MFMAs behind 10 weight-fragment loads instead of 45 (more matrix work per byte in flight),
//   * B = 128 x 3 lockstep passes = 384 clips are 192 workgroups: one round on 256 CUs instead of one and a half.
// Price: 128 workgroups at B = 256 (half of the CUs; each with ~80 us of MFMA work), and LDS.
//
// LDS.  Two clips x three fp32 row buffers do not fit 160 KB.  They are not needed: the four waves of a workgroup hold the
// COMPLETE output of a conv in their accumulators (5 channel tiles x 5 row tiles per wave) before a single element is
// stored, so a conv's output may overwrite its own input once every wave has finished reading it (one barrier), and what the
// default kernel keeps in a second / third buffer -- the block input for the residual, the running gradient G -- lives in
// registers in the accumulator layout (100 VGPRs).  ONE image of 2T rows (89 KB at T = 34) + the zero row.  Tensors the
// other pass needs (h1 / h2 / y forward; y / h2 / h1 masks and gp2 / gp1 backward) go to / come from HBM straight from that
// register layout (16 bytes per lane, 64 contiguous bytes per row and tile).
//
// SPLIT ONCE.  The image is two bf16 planes (hi, lo) instead of fp32 rows: the producer of a value splits it, the K loop of
// the convs reads ready MFMA fragments -- the default kernel's four waves each re-split every activation fragment for every
// K tile (see below).
//
// SAME MATH, SAME ORDER: every output element accumulates the same 20 K tiles x 3 piece products in the same order as in the
// default kernel, bias / ReLU / dropout / residual are the same fp32 operations, the keep bits are the same function of
// (seed, pass counter, site, element index) -- h1 / h2 / y, gp1 / gp2 and gx are bit-identical (tests/test_gpu_zy_variants.py).
#include "s2ag_common.h"

namespace {
using namespace s2ag;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef unsigned short bf16_t;

#include "tcn_fused32_shared.h"

// NCL clips per workgroup: 2 (TCN32_PAIR=1: five 16-row tiles for two clips of up to 40 frames) or 1 (TCN32_PAIR=2: the same
// kernel -- split-once planes, one LDS image, residual in registers -- on the default's one clip per workgroup and three row
// tiles, so that the A/B separates what the planes buy from what the pairing buys).
template <int NCL> struct Tiles { static constexpr int MT = NCL == 2 ? 5 : 3; };
static_assert(CT_W * 5 * 4 <= 128, "keep bits of a thread fit one u32x4");

// row m of the workgroup's rows: clip m / T, frame m % T; global row = first row + m (the clips are adjacent in memory)
__device__ __forceinline__ int frame_of(int m, int T) { return m >= T ? m - T : m; }

// keep bits of one pass in the pair epilogue's register layout: bit (i*MTP + mt)*4 + c of thread (wave, lane); blockIdx.x = pair
// index inside the pass (the element index a keep bit is drawn for is relative to the pass, as in tcn32_keep_k)
template <int NCL>
__global__ __launch_bounds__(256) void tcn32p_keep_k(const T32P p, int clips_in_pass) {
    constexpr int MTP = Tiles<NCL>::MT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = blockIdx.x, cv = blockIdx.y;
    const int rows = min(NCL, clips_in_pass - NCL * wg) * p.T;
    const long long row0 = (long long)NCL * wg * p.T;
    const SiteKey key = site_key(p.rng, p.site[cv]);
    unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < CT_W; ++i) {
        const int co = (wave * CT_W + i) * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int mt = 0; mt < MTP; ++mt) {
            const int m = mt * 16 + (lane & 15);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int b = (i * MTP + mt) * 4 + c;
                bool k = false;
                if (m < rows && co + c < p.C)
                    k = keep_scale(key, (unsigned long long)(row0 + m) * p.C + co + c, p.drop_p, p.inv_keep) != 0.f;
                w[b >> 5] |= k ? (1u << (b & 31)) : 0u;
            }
        }
    }
    // keep_total / keep_off count PAIRS here
    p.keep[((size_t)cv * p.keep_total + p.keep_off + wg) * 256 + tid] = u32x4{w[0], w[1], w[2], w[3]};
}

// ---- LDS image of the pair: TWO bf16 planes (hi = rn(v), lo = rn(v - hi): exactly the pieces the default kernel makes on
// the fly) of 2T rows x 320 channels.  The default kernel keeps fp32 rows and every one of its four waves splits every
// activation fragment again for every K tile (~120 vector-ALU instructions per K tile and wave, 3 600 per conv, between the
// MFMAs); here a value is split ONCE, by the thread that produces it, and the K loop reads ready fragments (16 bytes per
// lane and plane).  Same bytes as the fp32 image.  Row pitch 656 B = 164 dwords: 16 consecutive rows tile the 64 banks for
// 16-byte reads.
constexpr int PH = 328;                 // plane row pitch in bf16 elements

// 4 fp32 -> 4 hi + 4 lo bf16 at element offset `off` of the two planes (8-byte stores)
__device__ __forceinline__ void split_store4(bf16_t* hi, bf16_t* lo, int off, f32x4 v) {
    const unsigned h01 = pk_bf16(v[0], v[1]), h23 = pk_bf16(v[2], v[3]);
    const unsigned l01 = pk_bf16(v[0] - __uint_as_float(h01 << 16), v[1] - __uint_as_float(h01 & 0xffff0000u));
    const unsigned l23 = pk_bf16(v[2] - __uint_as_float(h23 << 16), v[3] - __uint_as_float(h23 & 0xffff0000u));
    *reinterpret_cast<uint2*>(hi + off) = make_uint2(h01, h23);
    *reinterpret_cast<uint2*>(lo + off) = make_uint2(l01, l23);
}

// acc += conv over the planes: K tile kt = tap kt / KT_TAP (rows q - d forward, q + d backward for tap 0, inside the SAME
// clip, else the zero row; q for tap 1), channels (kt % KT_TAP)*32 .. +32.  wh / wl: this wave's hi / lo weight fragments
// (+ lane), a ring of RW K tiles in flight (two clips: 2 -- at 75 MFMAs per K tile that is the ~2 400 cycles of cover the
// default's ring of three has at 45; one clip: 3).  Same K order and same order of the three piece products as conv32_tile: bit-identical sums.
template <bool BWD, int MTP>
__device__ __forceinline__ void conv32p_tile(const bf16_t* hi, const bf16_t* lo, int Zrow, const u32x4* __restrict__ wh,
                                             const u32x4* __restrict__ wl, int d, int T, int rows, int lane,
                                             f32x4 (&acc)[CT_W][MTP]) {
    constexpr int RW = MTP == 3 ? 3 : 2;                        // K tiles of weight fragments in flight (~2 200-2 400 cycles of MFMAs either way)
    int off0[MTP], off1[MTP];
#pragma unroll
    for (int mt = 0; mt < MTP; ++mt) {
        const int m = mt * 16 + (lane & 15);
        const int q = frame_of(m, T);
        const bool ok0 = m < rows && (BWD ? (q + d < T) : (q >= d));
        off1[mt] = (m < rows ? m : Zrow) * PH + (lane >> 4) * 8;
        off0[mt] = (ok0 ? (BWD ? m + d : m - d) : Zrow) * PH + (lane >> 4) * 8;
    }
    u32x4 ah[RW][CT_W], al[RW][CT_W];
#pragma unroll
    for (int s = 0; s < RW; ++s)
#pragma unroll
        for (int i = 0; i < CT_W; ++i) {
            ah[s][i] = wh[(i * NKT + s) * 64];
            al[s][i] = wl[(i * NKT + s) * 64];
        }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int s = kt % RW;
        const bool t0 = kt < KT_TAP;
        const int c0 = (t0 ? kt : kt - KT_TAP) * 32;
        bf16x8 bh[MTP], bl[MTP];
#pragma unroll
        for (int mt = 0; mt < MTP; ++mt) {
            const int o = (t0 ? off0[mt] : off1[mt]) + c0;
            bh[mt] = *reinterpret_cast<const bf16x8*>(hi + o);
            bl[mt] = *reinterpret_cast<const bf16x8*>(lo + o);
        }
#pragma unroll
        for (int i = 0; i < CT_W; ++i) {
            const bf16x8 avh = __builtin_bit_cast(bf16x8, ah[s][i]), avl = __builtin_bit_cast(bf16x8, al[s][i]);
#pragma unroll
            for (int mt = 0; mt < MTP; ++mt) acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(avl, bh[mt], acc[i][mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MTP; ++mt) acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(avh, bl[mt], acc[i][mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MTP; ++mt) acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(avh, bh[mt], acc[i][mt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);                      // (the ring slot is refilled only after its MFMAs were issued)
        if (kt + RW < NKT) {
#pragma unroll
            for (int i = 0; i < CT_W; ++i) {
                ah[s][i] = wh[(i * NKT + kt + RW) * 64];
                al[s][i] = wl[(i * NKT + kt + RW) * 64];
            }
        }
    }
}

template <int NCL>
__global__ __launch_bounds__(256) void tcn32p_fwd_k(const T32P p) {
    constexpr int MTP = Tiles<NCL>::MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* hi = reinterpret_cast<bf16_t*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, C = p.C;
    bf16_t* lo = hi + (NCL * T + 1) * PH;                        // each plane: NCL * T rows + the zero row (row index NCL * T)
    const int Zrow = NCL * T;
    const int clip0 = NCL * (int)blockIdx.x;
    const int rows = min(NCL, p.n_clips - clip0) * T;
    const long long row0 = (long long)clip0 * T;
    const int cpr = C / 4;                                       // 16-byte chunks of an HBM row
    const bool save = clip0 < p.save_clips;                      // a no-grad pass of a lockstep batch keeps nothing
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

    // rows in: (row0 + m, 0..C) -> planes[m][0..C), pad channels zero
    for (int idx = tid; idx < rows * (CP / 4); idx += 256) {
        const int m = idx / (CP / 4), kc = idx - m * (CP / 4);
        f32x4 v = zero4;
        if (kc < cpr) v = *reinterpret_cast<const f32x4*>(p.x + (row0 + m) * C + kc * 4);
        split_store4(hi, lo, m * PH + kc * 4, v);
    }
    for (int i = tid; i < PH / 2; i += 256) {
        reinterpret_cast<unsigned*>(hi + Zrow * PH)[i] = 0u;
        reinterpret_cast<unsigned*>(lo + Zrow * PH)[i] = 0u;
    }
    const bool drop = p.drop_p > 0.f;
    const float ik = p.inv_keep;
    f32x4 acc[CT_W][MTP], res[CT_W][MTP];                       // res: the block input in the accumulator layout (residual)
#pragma unroll
    for (int i = 0; i < CT_W; ++i) {
        const int co = (wave * CT_W + i) * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int mt = 0; mt < MTP; ++mt) {
            const int m = mt * 16 + (lane & 15);
            res[i][mt] = (m < rows && co < C) ? *reinterpret_cast<const f32x4*>(p.x + (row0 + m) * C + co) : zero4;
        }
    }
    __syncthreads();
    for (int blk = 0; blk < p.n_blocks; ++blk) {
        const int d = p.dil[blk];
#pragma unroll 1
        for (int j = 0; j < 2; ++j) {
            const int cv = 2 * blk + j;
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
                for (int mt = 0; mt < MTP; ++mt) acc[i][mt] = zero4;
            conv32p_tile<false, MTP>(hi, lo, Zrow, wh, wl, d, T, rows, lane, acc);
            __syncthreads();                                     // every wave has read its input rows: the planes may be overwritten
            // epilogue: bias, ReLU, dropout (conv1: h1; conv2: h2, + residual, ReLU = the block output).  What the backward
            // pass needs goes to HBM from the registers (64 contiguous bytes per row and channel tile); the next conv's
            // operand goes to the planes, split here, once.
            const unsigned kw[4] = {kv.x, kv.y, kv.z, kv.w};
            const bool last = blk == p.n_blocks - 1;
#pragma unroll
            for (int i = 0; i < CT_W; ++i) {
                const int co = (wave * CT_W + i) * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int mt = 0; mt < MTP; ++mt) {
                    const int m = mt * 16 + (lane & 15);
                    if (m >= rows) continue;
                    f32x4 v;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int b = (i * MTP + mt) * 4 + c;
                        float t = fmaxf(acc[i][mt][c] + bv[i][c], 0.f);
                        if (drop) t = (kw[b >> 5] >> (b & 31)) & 1u ? t * ik : 0.f;
                        v[c] = t;
                    }
                    const long long go = (row0 + m) * C + co;
                    if (j == 0) {
                        if (save && co < C) *reinterpret_cast<f32x4*>(p.h1[blk] + go) = v;
                    } else {
                        if (save && co < C) *reinterpret_cast<f32x4*>(p.h2[blk] + go) = v;
                        const f32x4 xv = res[i][mt];
                        v = f32x4{fmaxf(v[0] + xv[0], 0.f), fmaxf(v[1] + xv[1], 0.f), fmaxf(v[2] + xv[2], 0.f), fmaxf(v[3] + xv[3], 0.f)};
                        res[i][mt] = v;
                        if ((save || last) && co < C) *reinterpret_cast<f32x4*>(p.y[blk] + go) = v;
                    }
                    if (!(last && j == 1)) split_store4(hi, lo, m * PH + co, v);
                }
            }
            __syncthreads();
        }
    }
}

// The chain of data gradients (tcn32_bwd_k), two clips per workgroup: the running gradient G in registers (accumulator
// layout), the planes hold the operand of the next conv (P2, then P1).
template <int NCL>
__global__ __launch_bounds__(256) void tcn32p_bwd_k(const T32P p) {
    constexpr int MTP = Tiles<NCL>::MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* hi = reinterpret_cast<bf16_t*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = p.T, C = p.C;
    bf16_t* lo = hi + (NCL * T + 1) * PH;
    const int Zrow = NCL * T;
    const int clip0 = NCL * (int)blockIdx.x;
    const int rows = min(NCL, p.n_clips - clip0) * T;
    const long long row0 = (long long)clip0 * T;
    const float ik = p.inv_keep;
    for (int i = tid; i < PH / 2; i += 256) {
        reinterpret_cast<unsigned*>(hi + Zrow * PH)[i] = 0u;
        reinterpret_cast<unsigned*>(lo + Zrow * PH)[i] = 0u;
    }
    f32x4 acc[CT_W][MTP], G[CT_W][MTP];
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CT_W; ++i) {
        const int co = (wave * CT_W + i) * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int mt = 0; mt < MTP; ++mt) {
            const int m = mt * 16 + (lane & 15);
            G[i][mt] = (m < rows && co < C) ? *reinterpret_cast<const f32x4*>(p.gy + (row0 + m) * C + co) : zero4;
        }
    }
    for (int blk = p.n_blocks - 1; blk >= 0; --blk) {
        const int d = p.dil[blk];
        // (a) element-wise: G <- G * [y > 0]; P2 <- G * [h2 > 0] / keep -> planes (+ gp2 to HBM); pad channels stay zero
#pragma unroll
        for (int i = 0; i < CT_W; ++i) {
            const int co = (wave * CT_W + i) * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int mt = 0; mt < MTP; ++mt) {
                const int m = mt * 16 + (lane & 15);
                if (m >= rows) continue;
                f32x4 p2 = zero4;
                if (co < C) {
                    const long long go = (row0 + m) * C + co;
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(p.y[blk] + go);
                    const f32x4 hv = *reinterpret_cast<const f32x4*>(p.h2[blk] + go);
                    f32x4 gs;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        gs[c] = yv[c] > 0.f ? G[i][mt][c] : 0.f;
                        p2[c] = hv[c] > 0.f ? gs[c] * ik : 0.f;
                    }
                    G[i][mt] = gs;
                    *reinterpret_cast<f32x4*>(p.gp2[blk] + go) = p2;
                }
                split_store4(hi, lo, m * PH + co, p2);
            }
        }
        __syncthreads();
        // (b) P1 <- dgrad_conv2(P2) * [h1 > 0] / keep -> planes (+ gp1 to HBM)
        {
            const int cv = 2 * blk + 1;
            const u32x4* wh = reinterpret_cast<const u32x4*>(p.wfrag + (long long)(4 * cv + 2) * FRAG) + (wave * CT_W * NKT) * 64 + lane;
            const u32x4* wl = reinterpret_cast<const u32x4*>(p.wfrag + (long long)(4 * cv + 3) * FRAG) + (wave * CT_W * NKT) * 64 + lane;
#pragma unroll
            for (int i = 0; i < CT_W; ++i)
#pragma unroll
                for (int mt = 0; mt < MTP; ++mt) acc[i][mt] = zero4;
            conv32p_tile<true, MTP>(hi, lo, Zrow, wh, wl, d, T, rows, lane, acc);
            __syncthreads();                                     // every wave has read P2: the planes may take P1
#pragma unroll
            for (int i = 0; i < CT_W; ++i) {
                const int co = (wave * CT_W + i) * 16 + (lane >> 4) * 4;
#pragma unroll
                for (int mt = 0; mt < MTP; ++mt) {
                    const int m = mt * 16 + (lane & 15);
                    if (m >= rows) continue;
                    f32x4 p1 = zero4;
                    if (co < C) {
                        const long long go = (row0 + m) * C + co;
                        const f32x4 hv = *reinterpret_cast<const f32x4*>(p.h1[blk] + go);
                        p1 = f32x4{hv[0] > 0.f ? acc[i][mt][0] * ik : 0.f, hv[1] > 0.f ? acc[i][mt][1] * ik : 0.f,
                                   hv[2] > 0.f ? acc[i][mt][2] * ik : 0.f, hv[3] > 0.f ? acc[i][mt][3] * ik : 0.f};
                        *reinterpret_cast<f32x4*>(p.gp1[blk] + go) = p1;
                    }
                    split_store4(hi, lo, m * PH + co, p1);
                }
            }
        }
        __syncthreads();
        // (c) G <- dgrad_conv1(P1) + G
        {
            const int cv = 2 * blk;
            const u32x4* wh = reinterpret_cast<const u32x4*>(p.wfrag + (long long)(4 * cv + 2) * FRAG) + (wave * CT_W * NKT) * 64 + lane;
            const u32x4* wl = reinterpret_cast<const u32x4*>(p.wfrag + (long long)(4 * cv + 3) * FRAG) + (wave * CT_W * NKT) * 64 + lane;
#pragma unroll
            for (int i = 0; i < CT_W; ++i)
#pragma unroll
                for (int mt = 0; mt < MTP; ++mt) acc[i][mt] = zero4;
            conv32p_tile<true, MTP>(hi, lo, Zrow, wh, wl, d, T, rows, lane, acc);
#pragma unroll
            for (int i = 0; i < CT_W; ++i)
#pragma unroll
                for (int mt = 0; mt < MTP; ++mt) {
                    const f32x4 gv = G[i][mt];
                    G[i][mt] = f32x4{acc[i][mt][0] + gv[0], acc[i][mt][1] + gv[1], acc[i][mt][2] + gv[2], acc[i][mt][3] + gv[3]};
                }
        }
        __syncthreads();                                         // every wave has read P1: the next block's (a) overwrites the planes
    }
#pragma unroll
    for (int i = 0; i < CT_W; ++i) {
        const int co = (wave * CT_W + i) * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int mt = 0; mt < MTP; ++mt) {
            const int m = mt * 16 + (lane & 15);
            if (m < rows && co < C) *reinterpret_cast<f32x4*>(p.gx + (row0 + m) * C + co) = G[i][mt];
        }
    }
}
}  // namespace

namespace s2ag {
// ncl = clips per workgroup (2: TCN32_PAIR=1, 1: TCN32_PAIR=2).  Pairs must not straddle passes (the keep bits of a pass are
// drawn relative to it) nor the saved / unsaved boundary; single clips have no such condition.
bool tcn32p_supported(int n_clips, int n_passes, int save_clips, int T, int ncl) {
    if (n_passes < 1 || n_clips % n_passes) return false;
    if (ncl == 2) {
        const int per = n_clips / n_passes;
        if (n_passes > 1 && (per & 1)) return false;
        if (save_clips != n_clips && (save_clips & 1)) return false;
    } else if (ncl != 1) {
        return false;
    }
    return T >= 1 && T <= 40 && (size_t)2 * (ncl * T + 1) * PH * sizeof(bf16_t) <= 160 * 1024;
}

template <int NCL>
static int fwd_launch(T32P p, int n_passes, const void* const* rngs, hipStream_t st) {
    const int per = p.n_clips / n_passes, wg_per = (per + NCL - 1) / NCL, wgs = (p.n_clips + NCL - 1) / NCL;
    const size_t lds = (size_t)2 * (NCL * p.T + 1) * PH * sizeof(bf16_t);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)tcn32p_fwd_k<NCL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return S2AG_E_UNSUPPORTED;
        attr = true;
    }
    p.keep_total = n_passes * wg_per;             // keep_total / keep_off count WORKGROUPS of this layout
    p.keep_off = 0;
    if (p.drop_p > 0.f) {
        for (int k = 0; k < n_passes; ++k) {
            if (!rngs || !rngs[k]) return S2AG_E_BADARG;
            T32P q = p;
            q.rng = static_cast<const unsigned long long*>(rngs[k]);
            q.keep_off = k * wg_per;
            hipLaunchKernelGGL(tcn32p_keep_k<NCL>, dim3(wg_per, 2 * p.n_blocks), dim3(256), 0, st, q, per);
        }
    }
    hipLaunchKernelGGL(tcn32p_fwd_k<NCL>, dim3(wgs), dim3(256), lds, st, p);
    return (int)hipGetLastError();
}

// `params`: the T32P tcn32_fwd_impl filled (same layout: both files include tcn_fused32_shared.h); keep_total / keep_off are
// set here.  rngs: one noise snapshot per pass (drop_p > 0).
int tcn32p_fwd_launch(const void* params, int n_passes, const void* const* rngs, int ncl, hipStream_t st) {
    const T32P& p = *static_cast<const T32P*>(params);
    return ncl == 2 ? fwd_launch<2>(p, n_passes, rngs, st) : fwd_launch<1>(p, n_passes, rngs, st);
}

template <int NCL>
static int bwd_launch(const T32P& p, hipStream_t st) {
    const size_t lds = (size_t)2 * (NCL * p.T + 1) * PH * sizeof(bf16_t);
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)tcn32p_bwd_k<NCL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return S2AG_E_UNSUPPORTED;
        attr = true;
    }
    hipLaunchKernelGGL(tcn32p_bwd_k<NCL>, dim3((p.n_clips + NCL - 1) / NCL), dim3(256), lds, st, p);
    return (int)hipGetLastError();
}

int tcn32p_bwd_launch(const void* params, int ncl, hipStream_t st) {
    const T32P& p = *static_cast<const T32P*>(params);
    return ncl == 2 ? bwd_launch<2>(p, st) : bwd_launch<1>(p, st);
}
}  // namespace s2ag
