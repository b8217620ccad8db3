// bf16 mode of the Conv1d hot path (BASELINE configs[1] "bf16", configs[3] "fp32 and bf16"; SURVEY section 7, hard part 5):
// activations of the wave encoder (net/multimodal_context_net_v2.py:14-33) and of the text TCN (net/tcn.py:16-46) live in
// HBM as bf16, every product runs on v_mfma_f32_16x16x32_bf16 with fp32 accumulation, BatchNorm statistics / bias /
// activation / dropout are evaluated on the fp32 accumulators, weights keep fp32 masters (the arenas) and are converted
// once per optimizer step, weight gradients are accumulated in fp32.
//
// One implicit-GEMM kernel serves the forward pass, the stride-1 data gradient (same kernel, tap-flipped transposed
// weights) and the strided data gradient (poly-phase: output position s*q + r only receives taps t = r (mod s), so each
// phase r is a stride-1 convolution of gy with the weights w[:, r + s*i, :]; the phases are blockIdx.z):
//
//     y[n, q, co] = epi( b[co] + sum_{t, c} x[n, q*pos_mul + pos_off + t*pos_tap, c] * w[co, t, c] )
//
//   * channels-last rows, channels padded to a multiple of 32 with zeros (300 -> 320 in the TCN), so a K tile is 32
//     channels of one tap: 64 contiguous bytes per row, one 16-byte load per lane, no predicate inside a tile; where the
//     taps of a window are contiguous in memory (dilation 1, no padding: the wave encoder) the whole window is ONE "tap"
//     of ks*Cin channels;
//   * block = 128 output rows x BN output channels (BN = 16 / 32 / 64), 4 waves of 32 rows each, K tile 32, LDS double
//     buffered with a row pitch of 80 B (fragment reads of 16 consecutive rows tile the 64 banks), next tile's global
//     loads in flight behind the current tile's MFMAs, one barrier per K tile;
//   * operands swapped (A = weights, B = activations): an accumulator register holds 4 consecutive output CHANNELS of one
//     row, so the epilogue stores 8 bytes (4 bf16) per lane instead of four 2-byte scalars;
//   * epilogue: bias, ReLU / LeakyReLU, counter-based dropout (same mask index as the fp32 kernels), optional fp64 column
//     sums of the ROUNDED outputs for the BatchNorm behind the layer ((2, R, Cout) partial layout of the fp32 kernels,
//     folded by s2ag_bn_fold), bf16 or fp32 stores.
//
// The weight gradient  dw[co, t, c] += sum_m gy[m, co] * x[m, t, c]  contracts over ROWS, the slow axis of both operands:
// the loader transposes through registers (two rows per thread, packed pairs, 4-byte LDS stores into [column][row] images)
// so the fragments are again 16-byte reads; the contraction is split over blocks and merged with fp32 atomics, the bias
// gradient rides along.
#include <stdlib.h>

#include "s2ag_common.h"

namespace {
using namespace s2ag;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef unsigned short bf16_t;

constexpr int KT = 32;          // K tile (channels)
constexpr int PITCH = 40;       // LDS row pitch in bf16 (80 B)

__device__ __forceinline__ unsigned bf16_rn(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

struct CvP {
    const bf16_t* x;
    const bf16_t* w;            // (phases, Cout rows, ks, Cp) bf16, zero padded in Cp
    const float* bias;          // nullable
    void* y;
    int M, Lq;                  // rows of this launch = N * Lq; rows per clip
    int Lin;                    // valid source rows per clip
    long long x_clip;           // elements between clips of x
    int ldx;
    int pos_mul, pos_off, pos_tap;
    int ks, Cp, Cvalid;         // taps, padded channels per tap (multiple of 32), loadable channels per tap (multiple of 8)
    int Cout, CoutS;            // rows of w / bias entries;  channels stored (>= Cout: pad channels are written as zeros)
    long long y_clip;           // elements between clips of y
    int y_row, y_off;           // y element offset of (n, q, co) = n*y_clip + q*y_row + y_off + co
    long long w_phase;          // elements between the weight sets of consecutive phases
    int y_phase;                // y_off increment per phase
    int q_total, q_step;        // phase r stores rows q with q*q_step + r < q_total
    int act;
    float slope, drop_p, inv_keep;
    const unsigned long long* rng;
    unsigned site;
    int mask_cols;              // logical channel count of the dropout mask index (row * mask_cols + co)
    double* stats;              // nullable: (2, R, Cout), R = gridDim.x * 4
    // data-gradient launches: the result is the gradient w.r.t. the OUTPUT of the producing layer; multiplying it here by
    // that layer's act'(y) * dropout mask saves the separate epilogue-backward pass (y laid out like this launch's output)
    const bf16_t* post_y;       // nullable
    int post_act, post_cols;
    float post_slope, post_drop, post_inv;
    const unsigned long long* post_rng;
    unsigned post_site;
};

template <int BMT, int BN, bool OUT_F32>
__global__ __launch_bounds__(256) void conv_bf16_k(const CvP p) {
    constexpr int TN = BN / 16;             // weight (channel) tiles per wave
    constexpr int RH = BMT / 64;            // activation rows per loader thread
    __shared__ __attribute__((aligned(16))) bf16_t Xs[2][BMT * PITCH];
    __shared__ __attribute__((aligned(16))) bf16_t Ws[2][BN * PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx = blockIdx.x, by = blockIdx.y, phase = blockIdx.z;
    if (gridDim.z > 1) {
        // poly-phase launches: the phases of one row block read the same gy rows and write interleaved pieces (stride *
        // channels apart) of the same output lines.  In hardware order they are gridDim.x*gridDim.y blocks apart and on
        // different XCDs (block b runs on XCD b % 8); remapped, every XCD owns a contiguous range of virtual ids with the
        // phase running fastest, so a row block's phases run back to back behind one L2
        const int nwg = gridDim.x * gridDim.y * gridDim.z;
        const int hw = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int xcd = hw & 7, q8 = nwg >> 3, r8 = nwg & 7;
        const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (hw >> 3);
        phase = v % gridDim.z;
        const int t = v / gridDim.z;
        by = t % gridDim.y;
        bx = t / gridDim.y;
    }
    const int m0 = bx * BMT, n0 = by * BN;
    const bf16_t* wbase = p.w + (long long)phase * p.w_phase;

    // loader: rows lr (+ 64) of the activation tile, 16-byte chunk lc; row lr (< BN) of the weight tile
    const int lr = tid >> 2, lc = tid & 3;
    long long xb[RH];
    int xq[RH];
    bool xok[RH];
#pragma unroll
    for (int h = 0; h < RH; ++h) {
        const int m = m0 + lr + 64 * h;
        xok[h] = m < p.M;
        const int n = xok[h] ? m / p.Lq : 0;
        xq[h] = xok[h] ? m - n * p.Lq : 0;
        xb[h] = (long long)n * p.x_clip + lc * 8;
    }
    const bool wok = lr < BN && n0 + lr < p.Cout;
    const bf16_t* wsrc = wbase + (long long)(wok ? n0 + lr : 0) * p.ks * p.Cp + lc * 8;
    const int ct = p.Cp / KT, nkt = p.ks * ct;
    // two K tiles in flight in registers: a tile's operands are requested two tiles before they are multiplied (one
    // block per CU in the worst case: nothing else hides the global round trip)
    u32x4 rx[2][RH], rw[2];
    auto fetch = [&](int kt, int set) {
        const int tap = kt / ct, c0 = (kt - tap * ct) * KT;
        const bool cok = c0 + lc * 8 < p.Cvalid;
#pragma unroll
        for (int h = 0; h < RH; ++h) {
            const int row = xq[h] * p.pos_mul + p.pos_off + tap * p.pos_tap;
            const bool ok = xok[h] && cok && (unsigned)row < (unsigned)p.Lin;
            rx[set][h] = ok ? *reinterpret_cast<const u32x4*>(p.x + xb[h] + (long long)row * p.ldx + c0)
                            : u32x4{0u, 0u, 0u, 0u};
        }
        rw[set] = wok ? *reinterpret_cast<const u32x4*>(wsrc + (long long)tap * p.Cp + c0) : u32x4{0u, 0u, 0u, 0u};
    };
    const int l_off = lr * PITCH + lc * 8;
    auto stash = [&](int set, int buf) {
#pragma unroll
        for (int h = 0; h < RH; ++h) *reinterpret_cast<u32x4*>(&Xs[buf][l_off + h * 64 * PITCH]) = rx[set][h];
        if (lr < BN) *reinterpret_cast<u32x4*>(&Ws[buf][l_off]) = rw[set];
    };

    constexpr int TMT = BMT / 64;           // 16-row tiles per wave (a wave owns BMT/4 rows)
    f32x4 acc[TN][TMT];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TMT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int f_off = (lane & 15) * PITCH + (lane >> 4) * 8;
    auto mma = [&](int buf) {
        bf16x8 a[TN], b[TMT];
#pragma unroll
        for (int t = 0; t < TN; ++t)
            a[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&Ws[buf][t * 16 * PITCH + f_off]));
#pragma unroll
        for (int t = 0; t < TMT; ++t)
            b[t] = __builtin_bit_cast(
                bf16x8, *reinterpret_cast<const u32x4*>(&Xs[buf][(wave * (BMT / 4) + t * 16) * PITCH + f_off]));
#pragma unroll
        for (int ti = 0; ti < TN; ++ti)
#pragma unroll
            for (int tj = 0; tj < TMT; ++tj)
                acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
    };

    fetch(0, 0);
    if (nkt > 1) fetch(1, 1);
    stash(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nkt; kt += 2) {
        mma(0);
        if (kt + 1 < nkt) stash(1, 1);
        if (kt + 2 < nkt) fetch(kt + 2, 0);
        __syncthreads();
        if (kt + 1 >= nkt) break;
        mma(1);
        if (kt + 2 < nkt) stash(0, 0);
        if (kt + 3 < nkt) fetch(kt + 3, 1);
        __syncthreads();
    }

    // epilogue: lane holds channels co = n0 + ti*16 + (lane >> 4)*4 + {0..3} of row m = m0 + wave*(BMT/4) + tj*16 + (lane & 15)
    SiteKey key{0, 0};
    const bool drop = p.drop_p > 0.f;
    if (drop) key = site_key(p.rng, p.site);
    SiteKey pkey{0, 0};
    if (p.post_y && p.post_drop > 0.f) pkey = site_key(p.post_rng, p.post_site);
    const int q_lim = (p.q_total - phase + p.q_step - 1) / p.q_step;
    double s1[TN][4], s2[TN][4];
#pragma unroll
    for (int ti = 0; ti < TN; ++ti)
#pragma unroll
        for (int c = 0; c < 4; ++c) s1[ti][c] = s2[ti][c] = 0.0;
#pragma unroll
    for (int tj = 0; tj < TMT; ++tj) {
        const int m = m0 + wave * (BMT / 4) + tj * 16 + (lane & 15);
        const bool mok = m < p.M;
        const int n = mok ? m / p.Lq : 0, q = mok ? m - n * p.Lq : 0;
        const bool rok = mok && q < q_lim;
        const long long yb = (long long)n * p.y_clip + (long long)q * p.y_row + p.y_off + (long long)phase * p.y_phase;
#pragma unroll
        for (int ti = 0; ti < TN; ++ti) {
            const int co = n0 + ti * 16 + (lane >> 4) * 4;
            if (co >= p.CoutS) continue;
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float bv = (p.bias && co + c < p.Cout) ? p.bias[co + c] : 0.f;
                float t = apply_act(acc[ti][tj][c] + bv, p.act, p.slope);
                if (drop && co + c < p.mask_cols)
                    t *= keep_scale(key, (unsigned long long)m * p.mask_cols + co + c, p.drop_p, p.inv_keep);
                v[c] = t;
            }
            if (!rok) continue;
            if (p.post_y) {
                const uint2 yp2 = *reinterpret_cast<const uint2*>(p.post_y + yb + co);
                const unsigned yw[4] = {yp2.x & 0xffffu, yp2.x >> 16, yp2.y & 0xffffu, yp2.y >> 16};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float f = 1.f;
                    if (co + c >= p.post_cols) f = 0.f;
                    else {
                        if (p.post_drop > 0.f)
                            f = keep_scale(pkey, (unsigned long long)m * p.post_cols + co + c, p.post_drop, p.post_inv);
                        if (p.post_act == S2AG_ACT_LEAKY) f *= (bf16_f((bf16_t)yw[c]) > 0.f ? 1.f : p.post_slope);
                    }
                    v[c] *= f;
                }
            }
            if constexpr (OUT_F32) {
                float* yp = static_cast<float*>(p.y) + yb + co;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (co + c < p.CoutS) yp[c] = v[c];
            } else {
                unsigned h[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) h[c] = bf16_rn(v[c]);
                bf16_t* yp = static_cast<bf16_t*>(p.y) + yb + co;
                if (co + 3 < p.CoutS) {
                    *reinterpret_cast<uint2*>(yp) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (co + c < p.CoutS) yp[c] = (bf16_t)h[c];
                }
                if (p.stats) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const double r = (double)bf16_f((bf16_t)h[c]);
                        s1[ti][c] += r;
                        s2[ti][c] += r * r;
                    }
                }
            }
            if (OUT_F32 && p.stats) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    s1[ti][c] += (double)v[c];
                    s2[ti][c] += (double)v[c] * (double)v[c];
                }
            }
        }
    }
    if (p.stats) {
        // one partial row per wave: the 16 lanes that share (lane >> 4) hold 16 different rows of the same 4 channels
        const size_t R = (size_t)gridDim.x * 4, r = (size_t)bx * 4 + wave;
#pragma unroll
        for (int ti = 0; ti < TN; ++ti)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double a = s1[ti][c], b = s2[ti][c];
#pragma unroll
                for (int msk = 1; msk < 16; msk <<= 1) {
                    a += __shfl_xor(a, msk, 64);
                    b += __shfl_xor(b, msk, 64);
                }
                const int co = n0 + ti * 16 + (lane >> 4) * 4 + c;
                if ((lane & 15) == 0 && co < p.Cout) {
                    p.stats[r * p.Cout + co] = a;
                    p.stats[(R + r) * p.Cout + co] = b;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------------
struct WgP {
    const bf16_t* gy;           // (M, ldg) rows
    const bf16_t* x;
    float* dw;
    float* db;                  // nullable
    int M, Lq, Lin;
    long long x_clip;
    int ldx, ldg;
    int pos_mul, pos_off, pos_tap;
    int ks, Cp, Cvalid;         // as in CvP (Cp multiple of 64 here)
    int Cout, Cin;              // logical sizes: only dw[co < Cout][t][c < Cin] is written
    long long d_co;             // dw strides (elements)
    int d_t, d_c;
    int flat_cin;               // > 0: "flat" windows -- channel index k of the single tap is (t, c) = (k / flat_cin, k % flat_cin)
    int ks_out;                 // taps of dw (== ks unless flat)
    int m_chunk;                // rows per blockIdx.y (multiple of 32)
    // split launches (s2ag_bf16_conv_wgrad_split): every block stores its tile here instead of adding it to dw atomically
    float* part;                // (splits, tiles, 64, 64) fp32;  nullable
    float* part_b;              // (splits, co tiles, 64)
    int ntiles;
};

constexpr int WG_ROWS = 64;     // rows of the contraction per step (two MFMA K steps)
constexpr int WG_PITCH = 72;    // [column][row] image pitch in bf16 (144 B: 16 consecutive columns tile the 64 banks)

__device__ __forceinline__ void wgrad_body(const WgP& p, const int tile, const int split) {
    // [column][row] images of a 64-row step: 64 gy columns (channels co) and 64 x columns (channels c of tap t)
    __shared__ __attribute__((aligned(16))) bf16_t Gt[64 * WG_PITCH];
    __shared__ __attribute__((aligned(16))) bf16_t Xt[64 * WG_PITCH];
    __shared__ float bsum[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kct = p.Cp / 64;                                  // 64-channel tiles per tap
    const int nco = (p.Cout + 63) / 64;
    const int cot = tile % nco, kt = tile / nco;
    const int tap = kt / kct, c0 = (kt - tap * kct) * 64, co0 = cot * 64;
    const bool do_bias = p.db != nullptr && kt == 0;
    if (tid < 64) bsum[tid] = 0.f;

    // loader: every thread transposes the row pair (2*pr, 2*pr + 1) x the 8-column chunk ch of BOTH operands.  The chunk
    // index runs fastest along the lanes: a wave-level load touches 8 rows x 128 contiguous bytes (with the row pair along
    // the lanes it was 64 rows x 16 bytes: one cache line per lane, and the L1 line rate -- not LDS, MFMA or the atomics --
    // set the kernel's time).  The [column][row] images are XOR-swizzled in units of 4 words (8 rows) by the chunk index so
    // that the transposing 4-byte stores of a wave still spread over all 32 banks.
    const int ch = tid & 7, pr = tid >> 3;
    const int m_beg = split * p.m_chunk;
    const int m_end = min(p.M, m_beg + p.m_chunk);
    const bool g_col = co0 + ch * 8 < p.ldg, x_col = c0 + ch * 8 < p.Cvalid;
    float bacc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bacc[j] = 0.f;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wr = wave >> 1, wc = wave & 1;                    // wave's 32 x 32 quadrant: channels co, columns c
    // two steps in flight in registers (3 blocks per CU at best: nothing else hides the global round trip)
    u32x4 rg[2][2], rx[2][2];
    auto fetch = [&](int mb, int set) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int m = mb + 2 * pr + e;
            rg[set][e] = rx[set][e] = u32x4{0u, 0u, 0u, 0u};
            if (m < m_end) {
                if (g_col) rg[set][e] = *reinterpret_cast<const u32x4*>(p.gy + (long long)m * p.ldg + co0 + ch * 8);
                const int n = m / p.Lq, q = m - n * p.Lq;
                const int row = q * p.pos_mul + p.pos_off + tap * p.pos_tap;
                if (x_col && (unsigned)row < (unsigned)p.Lin)
                    rx[set][e] = *reinterpret_cast<const u32x4*>(p.x + (long long)n * p.x_clip + (long long)row * p.ldx + c0 + ch * 8);
            }
        }
    };
    auto transpose_store = [&](bf16_t* img, const u32x4& r0, const u32x4& r1) {
        const unsigned a[4] = {r0.x, r0.y, r0.z, r0.w}, b[4] = {r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned lo = (j & 1) ? (a[j >> 1] >> 16) : (a[j >> 1] & 0xffffu);
            const unsigned hi = (j & 1) ? (b[j >> 1] >> 16) : (b[j >> 1] & 0xffffu);
            *reinterpret_cast<unsigned*>(&img[(ch * 8 + j) * WG_PITCH + 2 * (pr ^ (ch << 2))]) = lo | (hi << 16);
        }
    };
    auto stash = [&](int set) {
        transpose_store(Gt, rg[set][0], rg[set][1]);
        transpose_store(Xt, rx[set][0], rx[set][1]);
        if (do_bias) {
            const unsigned a[4] = {rg[set][0].x, rg[set][0].y, rg[set][0].z, rg[set][0].w};
            const unsigned b[4] = {rg[set][1].x, rg[set][1].y, rg[set][1].z, rg[set][1].w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned lo = (j & 1) ? (a[j >> 1] >> 16) : (a[j >> 1] & 0xffffu);
                const unsigned hi = (j & 1) ? (b[j >> 1] >> 16) : (b[j >> 1] & 0xffffu);
                bacc[j] += bf16_f((bf16_t)lo) + bf16_f((bf16_t)hi);
            }
        }
    };
    auto mma = [&]() {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ca = wr * 32 + t * 16 + (lane & 15), cb = wc * 32 + t * 16 + (lane & 15);
                const int grp = kk * 4 + (lane >> 4);                   // 8-row group of this lane's K chunk
                a[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                                                      &Gt[ca * WG_PITCH + ((grp ^ ((ca >> 3) & 7)) << 3)]));
                b[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                                                      &Xt[cb * WG_PITCH + ((grp ^ ((cb >> 3) & 7)) << 3)]));
            }
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
        }
    };

    if (m_beg < m_end) fetch(m_beg, 0);
    if (m_beg + WG_ROWS < m_end) fetch(m_beg + WG_ROWS, 1);
    for (int mb = m_beg; mb < m_end; mb += 2 * WG_ROWS) {
        __syncthreads();                                        // the previous step's fragment reads are done
        stash(0);
        __syncthreads();
        if (mb + 2 * WG_ROWS < m_end) fetch(mb + 2 * WG_ROWS, 0);
        mma();
        if (mb + WG_ROWS >= m_end) break;
        __syncthreads();
        stash(1);
        __syncthreads();
        if (mb + 3 * WG_ROWS < m_end) fetch(mb + 3 * WG_ROWS, 1);
        mma();
    }
    // C layout: column (lane & 15) = x channel, rows (lane >> 4)*4 + {0..3} = output channel
    s2ag::det_enter();                    // deterministic mode: the splits of a tile add in workgroup-index order
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int k = c0 + wc * 32 + tj * 16 + (lane & 15);
            int t = tap, c = k;
            if (p.flat_cin > 0) {
                t = k / p.flat_cin;
                c = k - t * p.flat_cin;
            }
            if (c >= p.Cin || t >= p.ks_out) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = wr * 32 + ti * 16 + (lane >> 4) * 4 + q, co = co0 + col;
                if (co >= p.Cout) continue;
                if (p.part)
                    p.part[((long long)split * p.ntiles + tile) * 4096 + col * 64 + wc * 32 + tj * 16 + (lane & 15)] = acc[ti][tj][q];
                else
                    atomicAdd(p.dw + (long long)co * p.d_co + (long long)t * p.d_t + (long long)c * p.d_c, acc[ti][tj][q]);
            }
        }
    if (do_bias) {
        __syncthreads();
        S2AG_DET_WAVES_BEGIN
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicAdd(&bsum[ch * 8 + j], bacc[j]);
        S2AG_DET_WAVES_END
        __syncthreads();
        if (tid < 64 && co0 + tid < p.Cout) {
            if (p.part) p.part_b[((long long)split * nco + cot) * 64 + tid] = bsum[tid];
            else atomicAdd(p.db + co0 + tid, bsum[tid]);
        }
    }
    s2ag::det_leave();
}

// second half of a split launch: dw (+ db) += the sum over the splits of the stored tiles -- one thread per element, no
// atomics, and the scattered (Cout, Cin, ks) addresses of a reference-layout weight gradient are written once
__global__ __launch_bounds__(256) void conv_bf16_wgrad_reduce_k(const WgP p, const int splits) {
    // a block owns 32 consecutive elements; its 8 thread rows share the splits (many loads in flight, 128-byte segments)
    __shared__ float red[8][33];
    const int kct = p.Cp / 64, nco = (p.Cout + 63) / 64;
    const long long total = (long long)p.ntiles * 4096;
    const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const long long i = (long long)blockIdx.x * 32 + e;
    const bool is_bias = i >= total;
    const int j = (int)(i - total);
    float sum = 0.f;
    if (!is_bias) {
        const float* src = p.part + i;
        const int co_ = ((int)(i >> 12) % nco) * 64 + (((int)i & 4095) >> 6);
        if (co_ < p.Cout) {                                      // rows of a tile beyond Cout were never stored
#pragma unroll 8
            for (int sp = grp; sp < splits; sp += 8) sum += src[(long long)sp * total];
        }
    } else if (j < nco * 64) {
#pragma unroll 8
        for (int sp = grp; sp < splits; sp += 8) sum += p.part_b[(long long)sp * nco * 64 + j];
    }
    red[grp][e] = sum;
    __syncthreads();
    if (grp != 0) return;
#pragma unroll
    for (int g = 1; g < 8; ++g) sum += red[g][e];
    if (is_bias) {
        if (p.db && j < p.Cout) p.db[j] += sum;
        return;
    }
    const int tile = (int)(i >> 12), r = (int)(i & 4095), col = r >> 6, kl = r & 63;
    const int cot = tile % nco, kt = tile / nco;
    const int tap = kt / kct, k = (kt - tap * kct) * 64 + kl, co = cot * 64 + col;
    int t = tap, c = k;
    if (p.flat_cin > 0) {
        t = k / p.flat_cin;
        c = k - t * p.flat_cin;
    }
    if (co >= p.Cout || c >= p.Cin || t >= p.ks_out) return;
    p.dw[(long long)co * p.d_co + (long long)t * p.d_t + (long long)c * p.d_c] += sum;
}

__global__ __launch_bounds__(256) void conv_bf16_wgrad_k(const WgP p) { wgrad_body(p, blockIdx.x, blockIdx.y); }

// several layers' weight gradients in one launch: blockIdx.z = job; every job brings its own tile and split counts
struct WgJobs {
    WgP j[S2AG_BF16_MAX_WGRAD_JOBS];
    int tiles[S2AG_BF16_MAX_WGRAD_JOBS], splits[S2AG_BF16_MAX_WGRAD_JOBS];
};
__global__ __launch_bounds__(256) void conv_bf16_wgrad_multi_k(const WgJobs js) {
    const int job = blockIdx.z;
    if ((int)blockIdx.x >= js.tiles[job] || (int)blockIdx.y >= js.splits[job]) {
        s2ag::det_enter();                // (an idle workgroup of the padded grid still takes and passes on its turn)
        s2ag::det_leave();
        return;
    }
    wgrad_body(js.j[job], blockIdx.x, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------------------------------
// weights: fp32 master -> bf16 operand layouts, many tensors per launch
// ---------------------------------------------------------------------------------------------------------------------
struct PackJobs {
    s2ag_bf16_pack_job j[S2AG_BF16_MAX_PACK];
    long long start[S2AG_BF16_MAX_PACK + 1];
    int n;
};

__global__ __launch_bounds__(256) void pack_weights_k(const PackJobs js) {
    const long long total = js.start[js.n];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int k = 0;
        while (i >= js.start[k + 1]) ++k;
        const s2ag_bf16_pack_job& jb = js.j[k];
        const long long e = i - js.start[k];
        const int c = (int)(e % jb.Cp);
        const int t = (int)((e / jb.Cp) % jb.taps);
        const int o = (int)(e / ((long long)jb.Cp * jb.taps));
        int tap = jb.tap0 + t * jb.tap_step, cc = c;
        if (jb.flat_cin > 0) {                                  // one "tap" holding the whole window: k -> (tap, channel)
            tap = c / jb.flat_cin;
            cc = c - tap * jb.flat_cin;
        }
        float v = 0.f;
        if (cc < jb.cols && tap >= 0 && tap < jb.src_taps)
            v = jb.src[(long long)o * jb.s_o + (long long)tap * jb.s_t + (long long)cc * jb.s_c];
        static_cast<bf16_t*>(jb.dst)[e] = (bf16_t)bf16_rn(v);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// element-wise / reductions on bf16 activations
// ---------------------------------------------------------------------------------------------------------------------
// y (rows, ldy) bf16 <- x (rows, cols) fp32, zero in the pad columns [cols, ldy)
__global__ __launch_bounds__(256) void cast_to_bf16_k(const float* __restrict__ x, int ldx, long long rows, int cols,
                                                      bf16_t* __restrict__ y, int ldy) {
    const long long total = rows * ldy;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / ldy;
        const int c = (int)(i - r * ldy);
        y[i] = (bf16_t)(c < cols ? bf16_rn(x[r * ldx + c]) : 0u);
    }
}
__global__ __launch_bounds__(256) void cast_to_f32_k(const bf16_t* __restrict__ x, int ldx, long long rows, int cols,
                                                     float* __restrict__ y, int ldy) {
    const long long total = rows * cols;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cols;
        const int c = (int)(i - r * cols);
        y[r * ldy + c] = bf16_f(x[r * ldx + c]);
    }
}

// y = leaky(x * scale[c] + shift[c]); 8 channels per thread (cols % 8 == 0, ld % 8 == 0)
__global__ __launch_bounds__(256) void bn_apply_bf16_k(const bf16_t* __restrict__ x, long long rows, int cols, int ld,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       float slope, bf16_t* __restrict__ y) {
    const int cpr = cols / 8;
    const long long total = rows * cpr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cpr;
        const int c = (int)(i - r * cpr) * 8;
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + r * ld + c);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        unsigned o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = leaky(bf16_f((bf16_t)(w[j] & 0xffffu)) * scale[c + 2 * j] + shift[c + 2 * j], slope);
            const float b = leaky(bf16_f((bf16_t)(w[j] >> 16)) * scale[c + 2 * j + 1] + shift[c + 2 * j + 1], slope);
            o[j] = bf16_rn(a) | (bf16_rn(b) << 16);
        }
        *reinterpret_cast<u32x4*>(y + r * ld + c) = u32x4{o[0], o[1], o[2], o[3]};
    }
}

// BatchNorm backward, pass 1: per column  p = sum d,  q = sum d * xhat  with d = dy * leaky'(pre).  A thread owns an
// 8-channel chunk and a strided set of rows; the block folds its threads in LDS and adds one value per column to
// sums[2][cols] (fp32 atomics; zero on entry).
__global__ __launch_bounds__(256) void bn_bwd_sums_bf16_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                          long long rows, int cols, int ld, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, float slope,
                                                          float* __restrict__ sums) {
    extern __shared__ float sm[];                               // 2 * cols
    for (int i = threadIdx.x; i < 2 * cols; i += 256) sm[i] = 0.f;
    __syncthreads();
    const int cpr = cols / 8;                                   // chunks per row
    const int lanes_c = cpr < 256 ? cpr : 256;                  // threads along the chunk axis
    const int rstep = 256 / lanes_c;
    const int cc = threadIdx.x % lanes_c, rr = threadIdx.x / lanes_c;
    S2AG_DET_WAVES_BEGIN          // (deterministic mode: the LDS sums take the waves' terms one wave after the other)
    if (rr < rstep) {
        for (int cb = cc; cb < cpr; cb += lanes_c) {
            const int c = cb * 8;
            float sc[8], sh[8], mu[8], is[8], ps[8], qs[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sc[j] = scale[c + j]; sh[j] = shift[c + j]; mu[j] = mean[c + j]; is[j] = invstd[c + j];
                ps[j] = qs[j] = 0.f;
            }
            for (long long r = (long long)blockIdx.x * rstep + rr; r < rows; r += (long long)gridDim.x * rstep) {
                const u32x4 xv = *reinterpret_cast<const u32x4*>(x + r * ld + c);
                const u32x4 dv = *reinterpret_cast<const u32x4*>(dy + r * ld + c);
                const unsigned xw[4] = {xv.x, xv.y, xv.z, xv.w}, dw_[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float xf = bf16_f((bf16_t)((j & 1) ? xw[j >> 1] >> 16 : xw[j >> 1] & 0xffffu));
                    const float df = bf16_f((bf16_t)((j & 1) ? dw_[j >> 1] >> 16 : dw_[j >> 1] & 0xffffu));
                    const float d = df * ((xf * sc[j] + sh[j]) > 0.f ? 1.f : slope);
                    ps[j] += d;
                    qs[j] += d * (xf - mu[j]) * is[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                atomicAdd(&sm[c + j], ps[j]);
                atomicAdd(&sm[cols + c + j], qs[j]);
            }
        }
    }
    S2AG_DET_WAVES_END
    __syncthreads();
    s2ag::det_enter();
    for (int i = threadIdx.x; i < 2 * cols; i += 256) atomicAdd(sums + i, sm[i]);
    s2ag::det_leave();
}

// pass 2 (one block): dgamma += q, dbeta += p (atomically: several passes may share the parameters), c1 = p / rows,
// c2 = q / rows; sums is left zero for the next launch
__global__ __launch_bounds__(256) void bn_bwd_finish_bf16_k(float* __restrict__ sums, int cols, long long rows,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ c1, float* __restrict__ c2) {
    for (int c = threadIdx.x; c < cols; c += 256) {
        const float pv = sums[c], qv = sums[cols + c];
        sums[c] = 0.f;
        sums[cols + c] = 0.f;
        if (dgamma) atomicAdd(dgamma + c, qv);
        if (dbeta) atomicAdd(dbeta + c, pv);
        c1[c] = pv / (float)rows;
        c2[c] = qv / (float)rows;
    }
}

// pass 3: dx = scale * (d - c1 - xhat * c2)
__global__ __launch_bounds__(256) void bn_bwd_apply_bf16_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                           long long rows, int cols, int ld, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, float slope,
                                                           const float* __restrict__ c1, const float* __restrict__ c2,
                                                           bf16_t* __restrict__ dx) {
    const int cpr = cols / 8;
    const long long total = rows * cpr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cpr;
        const int c = (int)(i - r * cpr) * 8;
        const u32x4 xv = *reinterpret_cast<const u32x4*>(x + r * ld + c);
        const u32x4 dv = *reinterpret_cast<const u32x4*>(dy + r * ld + c);
        const unsigned xw[4] = {xv.x, xv.y, xv.z, xv.w}, dw_[4] = {dv.x, dv.y, dv.z, dv.w};
        unsigned o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xf = bf16_f((bf16_t)((j & 1) ? xw[j >> 1] >> 16 : xw[j >> 1] & 0xffffu));
            const float df = bf16_f((bf16_t)((j & 1) ? dw_[j >> 1] >> 16 : dw_[j >> 1] & 0xffffu));
            const float d = df * ((xf * scale[c + j] + shift[c + j]) > 0.f ? 1.f : slope);
            const float xh = (xf - mean[c + j]) * invstd[c + j];
            const unsigned h = bf16_rn(scale[c + j] * (d - c1[c + j] - xh * c2[c + j]));
            o[j >> 1] |= (j & 1) ? (h << 16) : h;
        }
        *reinterpret_cast<u32x4*>(dx + r * ld + c) = u32x4{o[0], o[1], o[2], o[3]};
    }
}

// y = leaky(a + b) (b nullable), all (rows, ld) bf16, 8 per thread over the padded width
__global__ __launch_bounds__(256) void add_act_bf16_k(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                      long long n8, float slope, bf16_t* __restrict__ y) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const u32x4 av = reinterpret_cast<const u32x4*>(a)[i];
        const u32x4 bv = b ? reinterpret_cast<const u32x4*>(b)[i] : u32x4{0u, 0u, 0u, 0u};
        const unsigned aw[4] = {av.x, av.y, av.z, av.w}, bw[4] = {bv.x, bv.y, bv.z, bv.w};
        unsigned o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = leaky(bf16_f((bf16_t)(aw[j] & 0xffffu)) + bf16_f((bf16_t)(bw[j] & 0xffffu)), slope);
            const float hi = leaky(bf16_f((bf16_t)(aw[j] >> 16)) + bf16_f((bf16_t)(bw[j] >> 16)), slope);
            o[j] = bf16_rn(lo) | (bf16_rn(hi) << 16);
        }
        reinterpret_cast<u32x4*>(y)[i] = u32x4{o[0], o[1], o[2], o[3]};
    }
}

// g = dy * act'(y) * dropout mask (regenerated); (rows, ld) bf16, logical cols, pad columns of g are zero
__global__ __launch_bounds__(256) void epilogue_bwd_bf16_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y,
                                                           long long rows, int cols, int ld, int act, float slope,
                                                           float drop_p, float inv_keep, const unsigned long long* rng,
                                                           unsigned site, bf16_t* __restrict__ g) {
    SiteKey key{0, 0};
    const bool drop = drop_p > 0.f;
    if (drop) key = site_key(rng, site);
    const int cpr = ld / 8;
    const long long total = rows * cpr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cpr;
        const int c = (int)(i - r * cpr) * 8;
        const u32x4 dv = *reinterpret_cast<const u32x4*>(dy + r * ld + c);
        const u32x4 yv = y ? *reinterpret_cast<const u32x4*>(y + r * ld + c) : u32x4{0u, 0u, 0u, 0u};
        const unsigned dw_[4] = {dv.x, dv.y, dv.z, dv.w}, yw[4] = {yv.x, yv.y, yv.z, yv.w};
        unsigned o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float d = bf16_f((bf16_t)((j & 1) ? dw_[j >> 1] >> 16 : dw_[j >> 1] & 0xffffu));
            const float yf = bf16_f((bf16_t)((j & 1) ? yw[j >> 1] >> 16 : yw[j >> 1] & 0xffffu));
            if (c + j >= cols) d = 0.f;
            else {
                // the forward stored y = act(pre) * mask: with the mask regenerated, act'(pre) follows from the sign of y
                // where the element was kept (ReLU / LeakyReLU), and a dropped element passes no gradient anyway
                if (drop) d *= keep_scale(key, (unsigned long long)r * cols + c + j, drop_p, inv_keep);
                if (act == S2AG_ACT_LEAKY) d *= (yf > 0.f ? 1.f : slope);
            }
            const unsigned h = bf16_rn(d);
            o[j >> 1] |= (j & 1) ? (h << 16) : h;
        }
        *reinterpret_cast<u32x4*>(g + r * ld + c) = u32x4{o[0], o[1], o[2], o[3]};
    }
}

// embedding rows -> bf16 (rows, ld) with dropout; pad columns zero
__global__ __launch_bounds__(256) void embedding_fwd_bf16_k(const long long* __restrict__ ids, const float* __restrict__ table,
                                                            long long rows, int dim, int n_entries, int ld, float drop_p,
                                                            float inv_keep, const unsigned long long* rng, unsigned site,
                                                            bf16_t* __restrict__ out) {
    SiteKey key{0, 0};
    const bool drop = drop_p > 0.f;
    if (drop) key = site_key(rng, site);
    const long long total = rows * ld;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / ld;
        const int c = (int)(i - r * ld);
        float v = 0.f;
        if (c < dim) {
            long long id = ids[r];
            if (id < 0 || id >= n_entries) id = 0;
            v = table[id * dim + c];
            if (drop) v *= keep_scale(key, (unsigned long long)r * dim + c, drop_p, inv_keep);
        }
        out[i] = (bf16_t)bf16_rn(v);
    }
}

// table gradient += dy (bf16) * mask.  A block owns 32 token rows, a thread one channel: the rows' loads are independent
// (unrolled, many in flight); contributions to the PAD row (id 0: most frames of a clip) are summed in a register and leave
// the block as ONE atomic per channel, the few word rows use direct atomics.
__global__ __launch_bounds__(320) void embedding_bwd_bf16_k(const long long* __restrict__ ids, const bf16_t* __restrict__ dy,
                                                            int ld, int dim, int n_entries, float drop_p, float inv_keep,
                                                            const unsigned long long* rng, unsigned site,
                                                            float* __restrict__ dtable, long long rows, int rows_per_block) {
    __shared__ long long sid[32];
    SiteKey key{0, 0};
    const bool drop = drop_p > 0.f;
    if (drop) key = site_key(rng, site);
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const int nr = (int)min((long long)rows_per_block, rows - r0);
    if (threadIdx.x < 32) {
        long long id = threadIdx.x < nr ? ids[r0 + threadIdx.x] : -1;
        if (threadIdx.x < nr && (id < 0 || id >= n_entries)) id = 0;
        sid[threadIdx.x] = id;
    }
    __syncthreads();
    s2ag::det_enter();                    // (det flavour: row blocks share table rows -- workgroups in index order; a thread owns its channel)
    for (int c = threadIdx.x; c < dim; c += blockDim.x) {
        float pad_acc = 0.f;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            if (i >= nr) break;
            const long long r = r0 + i;
            float d = bf16_f(dy[r * ld + c]);
            if (drop) d *= keep_scale(key, (unsigned long long)r * dim + c, drop_p, inv_keep);
            const long long id = sid[i];
            if (id == 0) pad_acc += d;
            else atomicAdd(dtable + id * dim + c, d);
        }
        if (pad_acc != 0.f) atomicAdd(dtable + c, pad_acc);
    }
    s2ag::det_leave();
}

inline int ew_blocks(long long n) {
    long long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    return b < 1 ? 1 : (int)b;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
static int launch_conv(const s2ag_bf16_conv_args* c, const s2ag_epilogue* e, double* partials, int* stat_rows, hipStream_t s) {
    if (stat_rows) *stat_rows = 0;
    if (!c || !c->x || !c->w || !c->y || c->N <= 0 || c->Lq <= 0 || c->ks <= 0 || c->Cout <= 0) return S2AG_E_BADARG;
    if ((c->Cp % KT) || (c->Cvalid & 7) || (c->ldx & 7) || c->Cvalid > c->Cp || c->phases < 1) return S2AG_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(c->x) | reinterpret_cast<uintptr_t>(c->w)) & 15) return S2AG_E_BADARG;
    if ((partials == nullptr) != (stat_rows == nullptr)) return S2AG_E_BADARG;
    if (e && e->drop_p > 0.f && !e->rng) return S2AG_E_BADARG;
    if (!c->out_f32 && ((c->y_row | c->y_off | c->y_phase) & 3 || (c->y_clip & 3))) return S2AG_E_BADARG;
    CvP p{};
    p.x = static_cast<const bf16_t*>(c->x); p.w = static_cast<const bf16_t*>(c->w); p.bias = c->bias; p.y = c->y;
    p.M = c->N * c->Lq; p.Lq = c->Lq; p.Lin = c->Lin; p.x_clip = c->x_clip; p.ldx = c->ldx;
    p.pos_mul = c->pos_mul; p.pos_off = c->pos_off; p.pos_tap = c->pos_tap;
    p.ks = c->ks; p.Cp = c->Cp; p.Cvalid = c->Cvalid; p.Cout = c->Cout; p.CoutS = c->CoutS > c->Cout ? c->CoutS : c->Cout;
    p.y_clip = c->y_clip; p.y_row = c->y_row; p.y_off = c->y_off;
    p.w_phase = c->w_phase; p.y_phase = c->y_phase;
    p.q_total = c->phases > 1 ? c->q_total : c->Lq; p.q_step = c->phases > 1 ? c->phases : 1;
    p.act = e ? e->act : S2AG_ACT_NONE; p.slope = e ? e->slope : 1.f; p.drop_p = e ? e->drop_p : 0.f;
    p.inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
    p.rng = e ? e->rng : nullptr; p.site = e ? e->site : 0u;
    p.mask_cols = c->mask_cols > 0 ? c->mask_cols : c->Cout;
    p.stats = partials;
    p.post_y = static_cast<const bf16_t*>(c->post_y); p.post_act = c->post_act; p.post_cols = c->post_cols;
    p.post_slope = c->post_slope; p.post_drop = c->post_drop;
    p.post_inv = c->post_drop > 0.f ? 1.f / (1.f - c->post_drop) : 1.f;
    p.post_rng = static_cast<const unsigned long long*>(c->post_rng); p.post_site = c->post_site;
    if (p.post_y && (c->out_f32 || c->phases > 1 || (p.post_drop > 0.f && !p.post_rng))) return S2AG_E_BADARG;
    const int bn = p.CoutS <= 16 ? 16 : (p.CoutS <= 32 ? 32 : 64);
    // 64-row tiles where 128-row ones would leave most CUs with a single block (the TCN: 68 x 5 blocks)
    const int bm = (long long)cdiv(p.M, 128) * cdiv(p.CoutS, bn) * c->phases >= 1024 ? 128 : 64;
    const dim3 grid(cdiv(p.M, bm), cdiv(p.CoutS, bn), c->phases);
#define S2AG_LAUNCH_CV(BN_)                                                                              \
    do {                                                                                                 \
        if (bm == 128) {                                                                                 \
            if (c->out_f32) hipLaunchKernelGGL((conv_bf16_k<128, BN_, true>), grid, dim3(256), 0, s, p); \
            else hipLaunchKernelGGL((conv_bf16_k<128, BN_, false>), grid, dim3(256), 0, s, p);           \
        } else {                                                                                         \
            if (c->out_f32) hipLaunchKernelGGL((conv_bf16_k<64, BN_, true>), grid, dim3(256), 0, s, p);  \
            else hipLaunchKernelGGL((conv_bf16_k<64, BN_, false>), grid, dim3(256), 0, s, p);            \
        }                                                                                                \
    } while (0)
    if (bn == 16) S2AG_LAUNCH_CV(16);
    else if (bn == 32) S2AG_LAUNCH_CV(32);
    else S2AG_LAUNCH_CV(64);
#undef S2AG_LAUNCH_CV
    if (stat_rows) *stat_rows = (int)grid.x * 4;
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bf16_conv_stats_rows(int rows) { return cdiv(rows, 64) * 4; }

extern "C" int s2ag_bf16_conv(const s2ag_bf16_conv_args* c, const s2ag_epilogue* e, double* partials, int* stat_rows,
                              void* stream) {
    return launch_conv(c, e, partials, stat_rows, (hipStream_t)stream);
}

// fills p and returns the tile count; *splits_out = blocks along the contraction for a launch aiming at `target` blocks
static int wgrad_plan(const s2ag_bf16_wgrad_args* g, WgP& p, int target, int* splits_out) {
    if (!g || !g->gy || !g->x || !g->dw || g->N <= 0 || g->Lq <= 0 || g->ks <= 0) return S2AG_E_BADARG;
    if ((g->Cp % 64) || (g->Cvalid & 7) || (g->ldx & 7) || (g->ldg & 7)) return S2AG_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(g->x) | reinterpret_cast<uintptr_t>(g->gy)) & 15) return S2AG_E_BADARG;
    p.gy = static_cast<const bf16_t*>(g->gy); p.x = static_cast<const bf16_t*>(g->x); p.dw = g->dw; p.db = g->db;
    p.M = g->N * g->Lq; p.Lq = g->Lq; p.Lin = g->Lin; p.x_clip = g->x_clip; p.ldx = g->ldx; p.ldg = g->ldg;
    p.pos_mul = g->pos_mul; p.pos_off = g->pos_off; p.pos_tap = g->pos_tap;
    p.ks = g->ks; p.Cp = g->Cp; p.Cvalid = g->Cvalid; p.Cout = g->Cout; p.Cin = g->Cin;
    p.d_co = g->d_co; p.d_t = g->d_t; p.d_c = g->d_c; p.flat_cin = g->flat_cin;
    p.ks_out = g->flat_cin > 0 ? g->ks_out : g->ks;
    const int tiles = cdiv(g->Cout, 64) * g->ks * (g->Cp / 64);
    int splits = cdiv(target, tiles);
    const int max_splits = cdiv(p.M, 512);                      // at least 8 steps of 64 rows per block
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.m_chunk = cdiv(cdiv(p.M, splits), WG_ROWS) * WG_ROWS;
    *splits_out = cdiv(p.M, p.m_chunk);
    return tiles;
}

// Split form: the contraction is cut into many short pieces (a block's loop is a chain of dependent global round trips:
// 66 steps per block made conv2's gradient 106 us), each block stores its tile, a second launch sums them.
static int split_count(const WgP& p, int tiles) {
    int splits = cdiv(1280, tiles);
    const int max_splits = cdiv(p.M, 4 * WG_ROWS);              // at least 4 steps of 64 rows per block
    if (splits > max_splits) splits = max_splits;
    return splits < 1 ? 1 : splits;
}

extern "C" long long s2ag_bf16_conv_wgrad_scratch_floats(const s2ag_bf16_wgrad_args* g) {
    WgP p{};
    int sp = 0;
    const int tiles = wgrad_plan(g, p, 1, &sp);
    if (tiles < 0) return tiles;
    const int splits = split_count(p, tiles);
    return (long long)splits * tiles * 4096 + (long long)splits * cdiv(p.Cout, 64) * 64;
}

extern "C" int s2ag_bf16_conv_wgrad_split(const s2ag_bf16_wgrad_args* g, float* scratch, long long scratch_floats, void* stream) {
    WgP p{};
    int sp = 0;
    const int tiles = wgrad_plan(g, p, 1, &sp);
    if (tiles < 0) return tiles;
    int splits = split_count(p, tiles);
    p.m_chunk = cdiv(cdiv(p.M, splits), WG_ROWS) * WG_ROWS;
    splits = cdiv(p.M, p.m_chunk);
    const long long need = (long long)splits * tiles * 4096 + (long long)splits * cdiv(p.Cout, 64) * 64;
    if (!scratch || scratch_floats < need) return S2AG_E_BADARG;
    p.part = scratch;
    p.part_b = scratch + (long long)splits * tiles * 4096;
    p.ntiles = tiles;
    hipLaunchKernelGGL(conv_bf16_wgrad_k, dim3(tiles, splits), dim3(256), 0, (hipStream_t)stream, p);
    hipLaunchKernelGGL(conv_bf16_wgrad_reduce_k, dim3(cdiv((long long)tiles * 4096 + cdiv(p.Cout, 64) * 64, 32)), dim3(256), 0,
                       (hipStream_t)stream, p, splits);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bf16_conv_wgrad_multi(const s2ag_bf16_wgrad_args* jobs, int njobs, void* stream) {
    if (!jobs || njobs < 1 || njobs > S2AG_BF16_MAX_WGRAD_JOBS) return S2AG_E_BADARG;
    constexpr int target = 768;
    WgJobs js{};
    int mt = 0, ms = 0;
    for (int k = 0; k < njobs; ++k) {
        const int t = wgrad_plan(jobs + k, js.j[k], cdiv(target, njobs), &js.splits[k]);
        if (t < 0) return t;
        js.tiles[k] = t;
        mt = t > mt ? t : mt;
        ms = js.splits[k] > ms ? js.splits[k] : ms;
    }
    hipLaunchKernelGGL(conv_bf16_wgrad_multi_k, dim3(mt, ms, njobs), dim3(256), 0, (hipStream_t)stream, js);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bf16_conv_wgrad(const s2ag_bf16_wgrad_args* g, void* stream) {
    if (!g || !g->gy || !g->x || !g->dw || g->N <= 0 || g->Lq <= 0 || g->ks <= 0) return S2AG_E_BADARG;
    if ((g->Cp % 64) || (g->Cvalid & 7) || (g->ldx & 7) || (g->ldg & 7)) return S2AG_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(g->x) | reinterpret_cast<uintptr_t>(g->gy)) & 15) return S2AG_E_BADARG;
    WgP p{};
    p.gy = static_cast<const bf16_t*>(g->gy); p.x = static_cast<const bf16_t*>(g->x); p.dw = g->dw; p.db = g->db;
    p.M = g->N * g->Lq; p.Lq = g->Lq; p.Lin = g->Lin; p.x_clip = g->x_clip; p.ldx = g->ldx; p.ldg = g->ldg;
    p.pos_mul = g->pos_mul; p.pos_off = g->pos_off; p.pos_tap = g->pos_tap;
    p.ks = g->ks; p.Cp = g->Cp; p.Cvalid = g->Cvalid; p.Cout = g->Cout; p.Cin = g->Cin;
    p.d_co = g->d_co; p.d_t = g->d_t; p.d_c = g->d_c; p.flat_cin = g->flat_cin;
    p.ks_out = g->flat_cin > 0 ? g->ks_out : g->ks;
    const int tiles = cdiv(g->Cout, 64) * g->ks * (g->Cp / 64);
    // every block ends with 4 096 fp32 atomics: the split count, not the loop, sets the time (768 blocks: 49 us per TCN
    // layer; 320 measured best)
    constexpr int target = 320;
    int splits = cdiv(target, tiles);
    const int max_splits = cdiv(p.M, 512);                      // at least 8 steps of 64 rows per block
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.m_chunk = cdiv(cdiv(p.M, splits), WG_ROWS) * WG_ROWS;
    splits = cdiv(p.M, p.m_chunk);
    hipLaunchKernelGGL(conv_bf16_wgrad_k, dim3(tiles, splits), dim3(256), 0, (hipStream_t)stream, p);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bf16_pack_weights(const s2ag_bf16_pack_job* jobs, int njobs, void* stream) {
    if (!jobs || njobs <= 0 || njobs > S2AG_BF16_MAX_PACK) return S2AG_E_BADARG;
    PackJobs js{};
    js.n = njobs;
    long long tot = 0;
    for (int k = 0; k < njobs; ++k) {
        if (!jobs[k].src || !jobs[k].dst || jobs[k].rows <= 0 || jobs[k].taps <= 0 || jobs[k].Cp <= 0) return S2AG_E_BADARG;
        js.j[k] = jobs[k];
        js.start[k] = tot;
        tot += (long long)jobs[k].rows * jobs[k].taps * jobs[k].Cp;
    }
    js.start[njobs] = tot;
    hipLaunchKernelGGL(pack_weights_k, dim3(ew_blocks(tot)), dim3(256), 0, (hipStream_t)stream, js);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bf16_cast(const void* x, int ldx, long long rows, int cols, void* y, int ldy, int to_bf16, void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0 || ldx < cols || ldy < cols) return S2AG_E_BADARG;
    if (to_bf16)
        hipLaunchKernelGGL(cast_to_bf16_k, dim3(ew_blocks(rows * ldy)), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const float*>(x), ldx, rows, cols, static_cast<bf16_t*>(y), ldy);
    else
        hipLaunchKernelGGL(cast_to_f32_k, dim3(ew_blocks(rows * cols)), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const bf16_t*>(x), ldx, rows, cols, static_cast<float*>(y), ldy);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bf16_bn_apply(const void* x, long long rows, int cols, int ld, const float* scale, const float* shift,
                                  float slope, void* y, void* stream) {
    if (!x || !y || !scale || !shift || rows <= 0 || cols <= 0 || (cols & 7) || (ld & 7) || ld < cols) return S2AG_E_BADARG;
    hipLaunchKernelGGL(bn_apply_bf16_k, dim3(ew_blocks(rows * (cols / 8))), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const bf16_t*>(x), rows, cols, ld, scale, shift, slope, static_cast<bf16_t*>(y));
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bf16_bn_bwd(const void* x, const void* dy, long long rows, int cols, int ld, const float* scale,
                                const float* shift, const float* mean, const float* invstd, float slope, float* dgamma,
                                float* dbeta, float* sums /*2*cols, zero on entry and exit*/, float* c1, float* c2, void* dx,
                                void* stream) {
    if (!x || !dy || !dx || !scale || !shift || !mean || !invstd || !sums || !c1 || !c2 || rows <= 0 || cols <= 0 ||
        (cols & 7) || (ld & 7) || ld < cols)
        return S2AG_E_BADARG;
    const int cpr = cols / 8, lanes_c = cpr < 256 ? cpr : 256, rstep = 256 / lanes_c;
    // ~16 rows per thread: with 64 the largest layer ran on 247 blocks -- one per CU, every thread a chain of 64 dependent
    // load pairs (47 us for 129 MB)
    long long nb = (rows + (long long)rstep * 16 - 1) / ((long long)rstep * 16);
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(bn_bwd_sums_bf16_k, dim3((unsigned)nb), dim3(256), sizeof(float) * 2 * cols, (hipStream_t)stream,
                       static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(dy), rows, cols, ld, scale, shift, mean,
                       invstd, slope, sums);
    hipLaunchKernelGGL(bn_bwd_finish_bf16_k, dim3(1), dim3(256), 0, (hipStream_t)stream, sums, cols, rows, dgamma, dbeta, c1,
                       c2);
    hipLaunchKernelGGL(bn_bwd_apply_bf16_k, dim3(ew_blocks(rows * cpr)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(dy), rows, cols, ld, scale, shift, mean,
                       invstd, slope, c1, c2, static_cast<bf16_t*>(dx));
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bf16_add_act(const void* a, const void* b, long long n, float slope, void* y, void* stream) {
    if (!a || !y || n <= 0 || (n & 7)) return S2AG_E_BADARG;
    hipLaunchKernelGGL(add_act_bf16_k, dim3(ew_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const bf16_t*>(a), static_cast<const bf16_t*>(b), n / 8, slope, static_cast<bf16_t*>(y));
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bf16_epilogue_bwd(const void* dy, const void* y, long long rows, int cols, int ld,
                                      const s2ag_epilogue* e, void* g, void* stream) {
    if (!dy || !g || !e || rows <= 0 || cols <= 0 || (ld & 7) || ld < cols) return S2AG_E_BADARG;
    if (e->drop_p > 0.f && !e->rng) return S2AG_E_BADARG;
    if (e->act == S2AG_ACT_LEAKY && !y) return S2AG_E_BADARG;
    const float ik = e->drop_p > 0.f ? 1.f / (1.f - e->drop_p) : 1.f;
    hipLaunchKernelGGL(epilogue_bwd_bf16_k, dim3(ew_blocks(rows * (ld / 8))), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const bf16_t*>(dy), static_cast<const bf16_t*>(y), rows, cols, ld, e->act, e->slope,
                       e->drop_p, ik, e->rng, e->site, static_cast<bf16_t*>(g));
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bf16_embedding_fwd(const long long* ids, const float* table, long long rows, int dim, int n_entries,
                                       void* out, int ld, const s2ag_epilogue* e, void* stream) {
    if (!ids || !table || !out || rows <= 0 || dim <= 0 || ld < dim) return S2AG_E_BADARG;
    const float p = e ? e->drop_p : 0.f;
    if (p > 0.f && !e->rng) return S2AG_E_BADARG;
    hipLaunchKernelGGL(embedding_fwd_bf16_k, dim3(ew_blocks(rows * ld)), dim3(256), 0, (hipStream_t)stream, ids, table, rows,
                       dim, n_entries, ld, p, p > 0.f ? 1.f / (1.f - p) : 1.f, e ? e->rng : nullptr, e ? e->site : 0u,
                       static_cast<bf16_t*>(out));
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bf16_embedding_bwd(const long long* ids, const void* dy, int ld, long long rows, int dim, int n_entries,
                                       float* dtable, const s2ag_epilogue* e, void* stream) {
    if (!ids || !dy || !dtable || rows <= 0 || dim <= 0 || ld < dim) return S2AG_E_BADARG;
    const float p = e ? e->drop_p : 0.f;
    if (p > 0.f && !e->rng) return S2AG_E_BADARG;
    const int rpb = 32;
    hipLaunchKernelGGL(embedding_bwd_bf16_k, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(320), 0, (hipStream_t)stream, ids,
                       static_cast<const bf16_t*>(dy), ld, dim, n_entries, p, p > 0.f ? 1.f / (1.f - p) : 1.f,
                       e ? e->rng : nullptr, e ? e->site : 0u, dtable, rows, rpb);
    S2AG_LAUNCH_CHECK();
    return 0;
}
S2AG_DET_HOOK(conv_bf16)
