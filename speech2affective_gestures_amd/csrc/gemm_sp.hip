// fp32 GEMM on the bf16 matrix pipe from PRE-SPLIT operands (the GRU input projections: M = clips*frames = 4 352,
// N = 2*3H = 1 800, K = 2H = 600 -- the one GEMM shape of the step that runs at 57 % of the f32 MFMA peak, i.e. that is
// bound by the f32 matrix pipe itself).
//
//   y[m, n] = sum_k a[m, k] * w[n, k] + bias[n]
//
// Every fp32 operand value is split EXACTLY into three bf16 pieces v = p0 + p1 + p2 (+ <= 2^-25 |v|; each piece is the
// bf16 rounding of what the previous ones left) and the six leading piece products a0w0 + a0w1 + a1w0 + a1w1 + a0w2 +
// a2w0 are accumulated in fp32 inside v_mfma_f32_16x16x32_bf16 (dropped terms <= 2^-24 |a w|: the result is as accurate
// as the f32 MFMA's, see gru_coop.hip and tools/diag_gru_split.py).  Six bf16 MFMAs (8 192 MACs each) do the work of
// eight f32 MFMAs (1 024 MACs in 32 cycles each).  Measured here (tools/bench_gemm_split.py): a bf16 16x16x32 MFMA
// issues every ~35-40 cycles per SIMD in this kernel whatever the accumulator order (rotating or pinned chains), register
// form (AGPR / VGPR), waves per SIMD or MFMA shape (a v_mfma_f32_32x32x16_bf16 variant with half as many instructions ran
// at the same speed: the time follows the MAC count at ~45 % of the nominal bf16 rate) -- so the gain on the matrix pipe
// is 8 x 32 / (6 x ~38) = 1.1-1.3x plus what the larger tile saves: 71-76 us against 113 us for the f32-MFMA kernel at
// M, N, K = 4 352, 1 800, 600.
//
// Splitting on the fly would cost as many vector-ALU cycles per K tile as it saves on the matrix pipe, so both operands
// arrive split: s2ag_split_bf16x3 writes the three planes [piece][rows][Kp] (Kp = K rounded up to 32, zero padded) --
// once per optimizer step for a weight, once per layer and pass for the activations (one streaming pass: 10 MB in, 16 MB
// out at M = 4 352, K = 600).
//
// Kernel: 128 x 128 output tile (the operand planes are 6 bytes per element: a 64 x 64 tile moved 880 MB through L2 at
// M, N, K = 4 352, 1 800, 600 and ran no faster than the f32 kernel), 4 waves of 64 x 64 (4 x 4 MFMA tiles, 64
// accumulator registers), K tile = 32 = one MFMA K; per K tile a thread moves two 16-byte chunks (8 bf16 along k) per
// operand plane through registers into LDS (row pitch 80 B: the 16-byte operand chunks of 16 consecutive rows tile the
// 64 banks), a wave issues 96 MFMAs (6 products x 16 tiles, consecutive MFMAs on different accumulators) against
// 24 ds_read_b128.
#include <stdlib.h>

#include "s2ag_common.h"

namespace {
using namespace s2ag;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int SPK = 32;                 // K tile
constexpr int SP_PITCH = 40;            // LDS row pitch in bf16 elements (80 B)

__device__ __forceinline__ unsigned bf16_rn(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

// planes[p][r][Kp]: piece p of x[r][k] (k < K), zero for K <= k < Kp
__global__ __launch_bounds__(256) void split_bf16x3_k(const float* __restrict__ x, int rows, int K, int ldx, int Kp,
                                                      unsigned short* __restrict__ planes) {
    const long long total = (long long)rows * (Kp / 2);               // two consecutive k per thread: one 4-byte store
    const size_t plane = (size_t)rows * Kp;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / (Kp / 2)), k = (int)(i - (long long)r * (Kp / 2)) * 2;
        unsigned pc[3] = {0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float v = (k + e < K) ? x[(long long)r * ldx + k + e] : 0.f;
            const unsigned p0 = bf16_rn(v);
            const float r1 = v - __uint_as_float(p0 << 16);
            const unsigned p1 = bf16_rn(r1);
            const float r2 = r1 - __uint_as_float(p1 << 16);
            const unsigned p2 = bf16_rn(r2);
            pc[0] |= p0 << (16 * e);
            pc[1] |= p1 << (16 * e);
            pc[2] |= p2 << (16 * e);
        }
        const size_t o = (size_t)r * Kp + k;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned*>(planes + p * plane + o) = pc[p];
    }
}

// Transposed planes for the weight gradients (contraction over the rows): planes[p][c][r] = piece p of x[src(r)][c],
// r < rows (zero for rows <= r < Rp), where src(r) = r + shift inside the clip of L frames the row belongs to (zero outside
// it: the GRU's dW_hh contracts d(gh)_t with h_{t-1}).  32 x 32 tiles through LDS: coalesced reads along the columns,
// coalesced 2-byte writes along the rows.  colsum (nullable): += column sums of x (the bias gradient rides along).
__global__ __launch_bounds__(256) void split_bf16x3_t_k(const float* __restrict__ x, int rows, int cols, int ldx, int Rp,
                                                        int shift, int L, unsigned short* __restrict__ planes,
                                                        float* __restrict__ colsum) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + ty + 8 * j, c = c0 + tx;
        float v = 0.f;
        if (r < rows && c < cols) {
            int src = r;
            bool ok = true;
            if (shift != 0) {
                const int l = r % L + shift;
                ok = l >= 0 && l < L;
                src = r + shift;
            }
            if (ok) v = x[(long long)src * ldx + c];
        }
        tile[ty + 8 * j][tx] = v;
        part += v;
    }
    if (colsum) {                                                    // (only used with shift == 0)
        __shared__ float cs[8][32];
        cs[ty][tx] = part;
        __syncthreads();
        if (ty == 0 && c0 + tx < cols) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += cs[j][tx];
            atomicAdd(colsum + c0 + tx, t);
        }
    }
    __syncthreads();
    const size_t plane = (size_t)cols * Rp;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c0 + ty + 8 * j, r = r0 + tx;
        if (c < cols && r < Rp) {
            const float v = tile[tx][ty + 8 * j];
            const unsigned p0 = bf16_rn(v);
            const float r1 = v - __uint_as_float(p0 << 16);
            const unsigned p1 = bf16_rn(r1);
            const float r2 = r1 - __uint_as_float(p1 << 16);
            const unsigned p2 = bf16_rn(r2);
            const size_t o = (size_t)c * Rp + r;
            planes[o] = (unsigned short)p0;
            planes[plane + o] = (unsigned short)p1;
            planes[2 * plane + o] = (unsigned short)p2;
        }
    }
}

struct SpP {
    const unsigned short* a;            // [3][M][Kp]
    const unsigned short* w;            // [3][N][Kp]
    const float* bias;                  // nullable
    float* y;
    int M, N, Kp, ldy;
    int kchunk;                         // K range per blockIdx.z (multiple of 32); == Kp without a K split
    int atomic_acc;                     // 1: y += (atomicAdd; weight gradients, K split over blockIdx.z), bias ignored
};

// NP: operand pieces used: 3 (six products, fp32-equivalent) or 2 (three products, 16 mantissa bits).
// SPT: tile rows = columns: 128 (4 waves of 64 x 64) where the grid still fills the chip, else 64 (4 waves of 32 x 32).
template <int NP, int SPT>
__global__ __launch_bounds__(256) void gemm_sp_k(const SpP p) {
    constexpr int SP_PLANE = SPT * SP_PITCH;
    constexpr int WT = SPT / 32;           // 16 x 16 MFMA tiles per wave along either axis
    constexpr int WS = SPT / 2;            // rows / columns per wave
    constexpr int NH = SPT / 64;           // loader passes of 64 rows
    // single-buffered LDS (61 KB at 128: two blocks per CU), the next K tile travels through registers meanwhile
    __shared__ __attribute__((aligned(16))) unsigned short As[NP][SP_PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[NP][SP_PLANE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * SPT, n0 = blockIdx.y * SPT;
    const size_t aplane = (size_t)p.M * p.Kp, wplane = (size_t)p.N * p.Kp;

    // loader: rows lr and lr + 64, k chunk lk (16 bytes = 8 bf16) of every plane of both operands per K tile
    const int lr = tid >> 2, lk = tid & 3;
    bool a_ok[NH], b_ok[NH];
    const unsigned short* a_src[NH];
    const unsigned short* b_src[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        a_ok[h] = m0 + lr + 64 * h < p.M;
        b_ok[h] = n0 + lr + 64 * h < p.N;
        a_src[h] = p.a + (size_t)(a_ok[h] ? m0 + lr + 64 * h : 0) * p.Kp + lk * 8;
        b_src[h] = p.w + (size_t)(b_ok[h] ? n0 + lr + 64 * h : 0) * p.Kp + lk * 8;
    }
    const int l_off = lr * SP_PITCH + lk * 8;
    u32x4 ra[NH][NP], rb[NH][NP];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
                ra[h][pc] = a_ok[h] ? *reinterpret_cast<const u32x4*>(a_src[h] + pc * aplane + k0) : u32x4{0u, 0u, 0u, 0u};
                rb[h][pc] = b_ok[h] ? *reinterpret_cast<const u32x4*>(b_src[h] + pc * wplane + k0) : u32x4{0u, 0u, 0u, 0u};
            }
    };
    auto stash = [&]() {
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
                *reinterpret_cast<u32x4*>(&As[pc][l_off + h * 64 * SP_PITCH]) = ra[h][pc];
                *reinterpret_cast<u32x4*>(&Bs[pc][l_off + h * 64 * SP_PITCH]) = rb[h][pc];
            }
    };

    f32x4 acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int f_off = (lane & 15) * SP_PITCH + (lane >> 4) * 8;       // fragment chunk of this lane inside a 16-row tile
    bf16x8 a[WT][NP], b[WT][NP];
    auto load_frags = [&]() {
#pragma unroll
        for (int t = 0; t < WT; ++t)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
                a[t][pc] = __builtin_bit_cast(
                    bf16x8, *reinterpret_cast<const u32x4*>(&As[pc][(wr * WS + t * 16) * SP_PITCH + f_off]));
                b[t][pc] = __builtin_bit_cast(
                    bf16x8, *reinterpret_cast<const u32x4*>(&Bs[pc][(wc * WS + t * 16) * SP_PITCH + f_off]));
            }
    };
    auto mma = [&]() {
        // every operand fragment is read from LDS once per K tile (24 ds_read_b128 for 96 MFMAs)
        load_frags();
        // all 24 reads in flight before the first MFMA: left to itself the scheduler sinks each read next to its first
        // use to save registers, and a wave then sits out one LDS latency per product group (measured 2.2 us per K tile
        // for a lone block instead of ~0.9)
        __builtin_amdgcn_sched_barrier(0);
        // the six products of a tile go back to back onto ITS accumulator (small terms first): a chain on one accumulator
        // runs at the pipe's full rate, while MFMAs that each read a different accumulator issued only every ~35-40 cycles
#pragma unroll
        for (int ti = 0; ti < WT; ++ti)
#pragma unroll
            for (int tj = 0; tj < WT; ++tj) {
                f32x4 c = acc[ti][tj];
                if constexpr (NP == 3) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti][NP - 1], b[tj][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti][0], b[tj][NP - 1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti][1], b[tj][1], c, 0, 0, 0);
                }
                if constexpr (NP >= 2) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti][NP - 1 ? 1 : 0], b[tj][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti][0], b[tj][NP - 1 ? 1 : 0], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti][0], b[tj][0], c, 0, 0, 0);
                acc[ti][tj] = c;
            }
    };

    const int kbeg = blockIdx.z * p.kchunk;
    const int nkt = (min(p.Kp, kbeg + p.kchunk) - kbeg) / SPK;
    fetch(kbeg);
    stash();
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) fetch(kbeg + (kt + 1) * SPK);
        mma();
        __syncthreads();                   // everyone is done reading this K tile
        if (kt + 1 < nkt) {
            stash();
            __syncthreads();
        }
    }
    // epilogue: C layout col = lane & 15, row = (lane >> 4) * 4 + q
#pragma unroll
    for (int ti = 0; ti < WT; ++ti)
#pragma unroll
        for (int tj = 0; tj < WT; ++tj) {
            const int col = n0 + wc * WS + tj * 16 + (lane & 15);
            if (col >= p.N) continue;
            const float bv = (p.bias && !p.atomic_acc) ? p.bias[col] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = m0 + wr * WS + ti * 16 + (lane >> 4) * 4 + q;
                if (row >= p.M) continue;
                if (p.atomic_acc)
                    atomicAdd(p.y + (long long)row * p.ldy + col, acc[ti][tj][q]);
                else
                    p.y[(long long)row * p.ldy + col] = acc[ti][tj][q] + bv;
            }
        }
}


// ---------------------------------------------------------------------------------------------------------------------
// Stride-1 convolution forward with tap-major weights on the same pipe:
//   y[(n,l), co] = epi( sum_{t,ci} x[(n, l - pad + t*dil), ci] * w[co, t, ci] + b[co] )
// The weight planes are split ahead of time ((3, Cout*ks, Kp) with Kp = Cin rounded up to 32: every tap padded on its own);
// the ACTIVATIONS are split on the fly by the loader (fp32 global -> registers -> bf16 pieces -> LDS): with two pieces that
// is ~8 vector-ALU operations per element against a third of the matrix-pipe time of the f32 kernel, and -- unlike the f32
// kernels, which read one k per ds_read_b32 -- a fragment read brings 8 k per lane, so the loop is no longer bound by LDS
// instruction issue.  32 x 64 tile (64 x 64 once that still gives >= 1 024 blocks), 4 waves, K tile = 32 channels of one
// tap, double-buffered LDS (one barrier per K tile), three K tiles in flight in registers.  At the TCN's shape (M = 4 352,
// 300 -> 300, 2 taps): 21.6 us against 27.5 us for the f32 straight-line kernel (the first version -- 64 x 64, single
// buffer, one tile in flight -- took 27 us: the block's dependent chain, not the matrix pipe, set the time).
struct SpcP {
    const float* x;
    const unsigned short* w;            // [3][Cout*ks][Kp]
    const float* bias;
    float* y;
    int M, L, Cin, Cout, ks, pad, dil, ldx, ldy, Kp;
    int act;
    float slope, drop_p, inv_keep;
    const unsigned long long* rng;
    unsigned site;
    double* stats;                      // nullable: (2, R, Cout) per-wave column sums of y and y^2, R = gridDim.x * 2
};

template <int NP, int BM>      // BM: tile rows, 64 (4 waves of 32 x 32) or 32 (4 waves of 16 x 32: twice the blocks)
__global__ __launch_bounds__(256) void conv_sp_k(const SpcP p) {
    constexpr int TM = BM / 32;                            // 16-row MFMA tiles per wave along M
    constexpr int AF = BM / 8;                             // activations per thread and K tile (8 or 4)
    constexpr int A_PLANE = BM * SP_PITCH, B_PLANE = 64 * SP_PITCH;
    __shared__ __attribute__((aligned(16))) unsigned short As[2][NP][A_PLANE];       // double buffered: one barrier per K tile
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][NP][B_PLANE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * 64;
    const size_t wplane = (size_t)p.Cout * p.ks * p.Kp;

    // A loader: row ar, AF consecutive channels starting at ak of the K tile;  B loader: column br, 8-wide k chunk bk
    const int ar = tid / (32 / AF), ak = (tid % (32 / AF)) * AF;
    const int m = m0 + ar;
    const bool a_row = m < p.M;
    const int nclip = a_row ? m / p.L : 0;
    const int l = a_row ? m - nclip * p.L : 0;
    const float* a_clip = p.x + (long long)nclip * p.L * p.ldx;
    const int br = tid >> 2, bk = (tid & 3) * 8;
    const bool b_ok = n0 + br < p.Cout;
    const unsigned short* b_src = p.w + (size_t)(b_ok ? n0 + br : 0) * p.ks * p.Kp + bk;
    const int a_off = ar * SP_PITCH + ak, b_off = br * SP_PITCH + bk;
    const int kpt = p.Kp / SPK;                            // K tiles per tap

    constexpr int PD = 3;                                  // K tiles in flight in registers: a tile's operands are
                                                           // requested three tiles before they are needed (a global round
                                                           // trip is ~1 us, a K tile of MFMAs ~0.2 us)
    float4 xq[PD][AF / 4];
    u32x4 rbq[PD][NP];
    auto fetch_to = [&](int kt, int set) {
        const int tap = kt / kpt, kb = (kt - tap * kpt) * SPK;
        const int pos = l - p.pad + tap * p.dil;
        const bool ok = a_row && (unsigned)pos < (unsigned)p.L;
        const float* src = a_clip + (long long)pos * p.ldx + kb + ak;
#pragma unroll
        for (int h = 0; h < AF / 4; ++h)
            xq[set][h] = (ok && kb + ak + 4 * h < p.Cin) ? *reinterpret_cast<const float4*>(src + 4 * h)
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int pc = 0; pc < NP; ++pc)
            rbq[set][pc] = b_ok ? *reinterpret_cast<const u32x4*>(b_src + pc * wplane + (size_t)tap * p.Kp + kb)
                                : u32x4{0u, 0u, 0u, 0u};
    };
    auto stash_from = [&](int set, int buf) {
        unsigned pa[NP][AF / 2];
#pragma unroll
        for (int d = 0; d < AF / 2; ++d) {
            unsigned q[2][3];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float4 x4 = xq[set][(2 * d + e) / 4];
                const int c = (2 * d + e) & 3;
                const float f = c == 0 ? x4.x : c == 1 ? x4.y : c == 2 ? x4.z : x4.w;
                q[e][0] = bf16_rn(f);
                const float r1 = f - __uint_as_float(q[e][0] << 16);
                q[e][1] = bf16_rn(r1);
                q[e][2] = NP == 3 ? bf16_rn(r1 - __uint_as_float(q[e][1] << 16)) : 0u;
            }
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) pa[pc][d] = q[0][pc] | (q[1][pc] << 16);
        }
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) {
            if constexpr (AF == 8)
                *reinterpret_cast<u32x4*>(&As[buf][pc][a_off]) = u32x4{pa[pc][0], pa[pc][1], pa[pc][2], pa[pc][3]};
            else
                *reinterpret_cast<uint2*>(&As[buf][pc][a_off]) = make_uint2(pa[pc][0], pa[pc][1]);
            *reinterpret_cast<u32x4*>(&Bs[buf][pc][b_off]) = rbq[set][pc];
        }
    };

    f32x4 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) acc[i][jj] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int f_off = (lane & 15) * SP_PITCH + (lane >> 4) * 8;
    auto mma = [&](int buf) {
        bf16x8 a[TM][NP], b[2][NP];
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
                a[t][pc] = __builtin_bit_cast(
                    bf16x8, *reinterpret_cast<const u32x4*>(&As[buf][pc][(wr * (BM / 2) + t * 16) * SP_PITCH + f_off]));
#pragma unroll
            for (int t = 0; t < 2; ++t)
                b[t][pc] = __builtin_bit_cast(
                    bf16x8, *reinterpret_cast<const u32x4*>(&Bs[buf][pc][(wc * 32 + t * 16) * SP_PITCH + f_off]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                f32x4 c = acc[ti][tj];
                if constexpr (NP == 3) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti][NP - 1], b[tj][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti][0], b[tj][NP - 1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti][1], b[tj][1], c, 0, 0, 0);
                }
                if constexpr (NP >= 2) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti][NP - 1 ? 1 : 0], b[tj][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti][0], b[tj][NP - 1 ? 1 : 0], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ti][0], b[tj][0], c, 0, 0, 0);
                acc[ti][tj] = c;
            }
    };

    const int nkt = p.ks * kpt;
#pragma unroll
    for (int d = 0; d < PD; ++d)
        if (d < nkt) fetch_to(d, d);
    stash_from(0, 0);
    __syncthreads();
    for (int base = 0; base < nkt; base += PD) {
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int kt = base + d;
            if (kt >= nkt) break;
            mma(kt & 1);
            if (kt + 1 < nkt) stash_from((d + 1) % PD, (kt + 1) & 1);
            if (kt + PD < nkt) fetch_to(kt + PD, d);
            __syncthreads();
        }
    }
    SiteKey key{0, 0};
    const bool drop = p.drop_p > 0.f;
    if (drop) key = site_key(p.rng, p.site);
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int col = n0 + wc * 32 + tj * 16 + (lane & 15);
            if (col >= p.Cout) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = m0 + wr * (BM / 2) + ti * 16 + (lane >> 4) * 4 + q;
                if (row >= p.M) continue;
                float v = apply_act(acc[ti][tj][q] + bv, p.act, p.slope);
                if (drop) v *= keep_scale(key, (unsigned long long)row * p.Cout + col, p.drop_p, p.inv_keep);
                p.y[(long long)row * p.ldy + col] = v;
                acc[ti][tj][q] = v;                        // what the statistics are taken of
            }
        }
    if (p.stats) {
        // fp64 column sums of this wave's rows for the BatchNorm behind the layer (as gemm_lin_k's epilogue): one partial row
        // per (row block, row wave); lanes that share a column (lane >> 4) meet in two shuffles
        const size_t R = (size_t)gridDim.x * 2, r = (size_t)blockIdx.x * 2 + wr;
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = m0 + wr * (BM / 2) + ti * 16 + (lane >> 4) * 4 + q;
                    if (row < p.M) {
                        const double v = (double)acc[ti][tj][q];
                        s1 += v;
                        s2 += v * v;
                    }
                }
            s1 += __shfl_xor(s1, 16, 64);
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64);
            s2 += __shfl_xor(s2, 32, 64);
            const int col = n0 + wc * 32 + tj * 16 + (lane & 15);
            if (lane < 16 && col < p.Cout) {
                p.stats[r * p.Cout + col] = s1;
                p.stats[(R + r) * p.Cout + col] = s2;
            }
        }
    }
}
}  // namespace

extern "C" int s2ag_gru_coop_split_pieces(void);

/* planes (3, rows, Kp) bf16, Kp = s2ag_split_k_padded(K): the exact 3-piece split of x (rows, K) with row pitch ldx */
extern "C" int s2ag_split_k_padded(int K) { return (K + SPK - 1) / SPK * SPK; }

extern "C" int s2ag_split_bf16x3(const float* x, int rows, int K, int ldx, void* planes, void* stream) {
    if (!x || !planes || rows <= 0 || K <= 0 || ldx < K) return S2AG_E_BADARG;
    const int Kp = s2ag_split_k_padded(K);
    const long long total = (long long)rows * (Kp / 2);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(split_bf16x3_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, rows, K, ldx, Kp,
                       static_cast<unsigned short*>(planes));
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_gemm_split_fwd(const void* a_planes, const void* w_planes, const float* bias, float* y, int M, int N,
                                   int K, int ldy, void* stream) {
    if (!a_planes || !w_planes || !y || M <= 0 || N <= 0 || K <= 0 || ldy < N) return S2AG_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(a_planes) | reinterpret_cast<uintptr_t>(w_planes)) & 15) return S2AG_E_BADARG;
    SpP p{static_cast<const unsigned short*>(a_planes), static_cast<const unsigned short*>(w_planes), bias, y, M, N,
          s2ag_split_k_padded(K), ldy, s2ag_split_k_padded(K), 0};
    // pieces: the setting shared with the cooperative GRU (S2AG_GRU_SPLIT; 0 there means "f32 MFMA": the caller then does
    // not come here); the planes always hold three pieces, two-piece products simply leave the third unread.
    // Tile: 128 x 128 halves the operand traffic but needs >= ~1.5 blocks per CU to fill the chip; else 64 x 64.
    const bool big = (long long)cdiv(M, 128) * cdiv(N, 128) >= 384;
    const int np = s2ag_gru_coop_split_pieces();      // 1: bf16 step mode (one piece, one product)
    const dim3 grid(cdiv(M, big ? 128 : 64), cdiv(N, big ? 128 : 64));
    if (big) {
        if (np == 1) hipLaunchKernelGGL((gemm_sp_k<1, 128>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else if (np == 2) hipLaunchKernelGGL((gemm_sp_k<2, 128>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((gemm_sp_k<3, 128>), grid, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        if (np == 1) hipLaunchKernelGGL((gemm_sp_k<1, 64>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else if (np == 2) hipLaunchKernelGGL((gemm_sp_k<2, 64>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((gemm_sp_k<3, 64>), grid, dim3(256), 0, (hipStream_t)stream, p);
    }
    S2AG_LAUNCH_CHECK();
    return 0;
}

/* planes (3, cols, Rp) bf16, Rp = s2ag_split_k_padded(rows): the TRANSPOSED split of x (rows, cols), optionally shifted by
 * `shift` frames inside clips of L frames (zero outside the clip); colsum (nullable) += column sums of x. */
extern "C" int s2ag_split_bf16x3_t(const float* x, int rows, int cols, int ldx, int shift, int L, void* planes,
                                   float* colsum, void* stream) {
    if (!x || !planes || rows <= 0 || cols <= 0 || ldx < cols || L <= 0 || (shift != 0 && colsum)) return S2AG_E_BADARG;
    const int Rp = s2ag_split_k_padded(rows);
    hipLaunchKernelGGL(split_bf16x3_t_k, dim3(cdiv(cols, 32), cdiv(Rp, 32)), dim3(256), 0, (hipStream_t)stream, x, rows,
                       cols, ldx, Rp, shift, L, static_cast<unsigned short*>(planes), colsum);
    S2AG_LAUNCH_CHECK();
    return 0;
}

/* y (M, N) += a w^T with a = planes (3, M, Kp), w = planes (3, N, Kp): the contraction is split over blocks and merged with
 * fp32 atomics (weight gradients: M, N = the weight's shape, K = clips * frames). */
extern "C" int s2ag_gemm_split_acc(const void* a_planes, const void* w_planes, float* y, int M, int N, int K, int ldy,
                                   void* stream) {
    if (!a_planes || !w_planes || !y || M <= 0 || N <= 0 || K <= 0 || ldy < N) return S2AG_E_BADARG;
    const int Kp = s2ag_split_k_padded(K);
    const int tiles = cdiv(M, 64) * cdiv(N, 64);
    int nsplit = cdiv(512, tiles);
    if (nsplit > Kp / (4 * SPK)) nsplit = Kp / (4 * SPK);
    if (nsplit < 1) nsplit = 1;
    const int kchunk = cdiv(cdiv(Kp, nsplit), SPK) * SPK;
    nsplit = cdiv(Kp, kchunk);
    SpP p{static_cast<const unsigned short*>(a_planes), static_cast<const unsigned short*>(w_planes), nullptr, y, M, N, Kp,
          ldy, kchunk, 1};
    const dim3 grid(cdiv(M, 64), cdiv(N, 64), nsplit);
    if (s2ag_gru_coop_split_pieces() == 1)
        hipLaunchKernelGGL((gemm_sp_k<1, 64>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (s2ag_gru_coop_split_pieces() == 2)
        hipLaunchKernelGGL((gemm_sp_k<2, 64>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((gemm_sp_k<3, 64>), grid, dim3(256), 0, (hipStream_t)stream, p);
    S2AG_LAUNCH_CHECK();
    return 0;
}

/* Forward of a stride-1 conv with tap-major weights (Cout, ks, Cin) from the weight's planes (3, Cout*ks, Kp), Kp =
 * s2ag_split_k_padded(Cin) (= s2ag_split_bf16x3 of the weight viewed as (Cout*ks, Cin)): the activations are split by the
 * loader.  Same epilogue as s2ag_conv1d_nlc_fwd (bias, activation, counter dropout with the same mask indexing);
 * partials / stat_rows (both or neither): BatchNorm column-sum partials as s2ag_conv1d_nlc_fwd_stats (same sizing).
 * S2AG_E_UNSUPPORTED (nothing launched) unless stride == 1, Lin == Lout, Cin % 4 == 0, ldx % 4 == 0 and x 16-byte aligned. */
extern "C" int s2ag_conv1d_nlc_fwd_split(const float* x, const void* w_planes, const float* bias, float* y,
                                         const s2ag_conv_geom* g, const s2ag_epilogue* e, double* partials,
                                         int* stat_rows, void* stream) {
    if (stat_rows) *stat_rows = 0;
    if ((partials == nullptr) != (stat_rows == nullptr)) return S2AG_E_BADARG;
    if (!x || !w_planes || !y || !g || g->N <= 0 || g->Lin <= 0 || g->Cin <= 0 || g->Cout <= 0 || g->ksize <= 0)
        return S2AG_E_BADARG;
    if (e && e->drop_p > 0.f && !e->rng) return S2AG_E_BADARG;
    if (g->stride != 1 || g->Lin != g->Lout || (g->Cin & 3) || (g->ldx & 3) || (reinterpret_cast<uintptr_t>(x) & 15) ||
        (reinterpret_cast<uintptr_t>(w_planes) & 15))
        return S2AG_E_UNSUPPORTED;
    SpcP p{};
    p.x = x; p.w = static_cast<const unsigned short*>(w_planes); p.bias = bias; p.y = y;
    p.M = g->N * g->Lout; p.L = g->Lin; p.Cin = g->Cin; p.Cout = g->Cout; p.ks = g->ksize; p.pad = g->pad; p.dil = g->dil;
    p.ldx = g->ldx; p.ldy = g->ldy; p.Kp = s2ag_split_k_padded(g->Cin);
    p.act = e ? e->act : S2AG_ACT_NONE; p.slope = e ? e->slope : 1.f; p.drop_p = e ? e->drop_p : 0.f;
    p.inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
    p.rng = e ? e->rng : nullptr; p.site = e ? e->site : 0u;
    p.stats = partials;
    const bool bm64 = (long long)cdiv(p.M, 64) * cdiv(p.Cout, 64) >= 1024;
    const dim3 grid(cdiv(p.M, bm64 ? 64 : 32), cdiv(p.Cout, 64));
    const int np = s2ag_gru_coop_split_pieces();
    if (bm64) {
        if (np == 1) hipLaunchKernelGGL((conv_sp_k<1, 64>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else if (np == 2) hipLaunchKernelGGL((conv_sp_k<2, 64>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((conv_sp_k<3, 64>), grid, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        if (np == 1) hipLaunchKernelGGL((conv_sp_k<1, 32>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else if (np == 2) hipLaunchKernelGGL((conv_sp_k<2, 32>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((conv_sp_k<3, 32>), grid, dim3(256), 0, (hipStream_t)stream, p);
    }
    if (stat_rows) *stat_rows = (int)grid.x * 2;
    S2AG_LAUNCH_CHECK();
    return 0;
}
