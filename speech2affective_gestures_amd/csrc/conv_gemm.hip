// Implicit-GEMM 1-D convolution family on channels-last fp32 matrices, on the f32 MFMA pipe
// (v_mfma_f32_16x16x4_f32: exact f32, bitwise an fmaf chain -- keeps the 1e-3 parity bar with margin).
//
// One tiling serves every contraction of the S2AG step (Conv1d, dilated causal TCN conv, Linear,
// folded ST-GCN Conv2d, GRU input projections) in its three forms:
//   fwd        out[(n,l),co]   = sum_{tap,ci} x[(n,pos),ci]   * w[co,ci,tap]     (+bias, act, dropout)
//   bwd_data   dx[(n,pos),ci]  = sum_{tap,co} gy[(n,l),co]    * w[co,ci,tap]
//   bwd_weight dw[co,ci,tap]  += sum_{(n,l)}  gy[(n,l),co]    * x[(n,pos),ci]    (split over rows, atomics)
//
// Block = 256 threads = 4 waves; block tile 64x64, K step 16; wave (wm,wn) owns a 32x32 quadrant as
// 2x2 MFMA 16x16 tiles.  Operands are staged k-major in LDS with an 80-float pitch so the four k-rows
// a wave reads in one ds_read_b32 fall on disjoint bank groups.  The next K tile is fetched into
// registers while the current one is multiplied.
#include <stdlib.h>

#include "s2ag_common.h"

namespace {
using namespace s2ag;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int BM = 64, BN = 64, BK = 16, PITCH = 80;

struct GemmP {
    const float* a;     // x (fwd) or gy (bwd_data)
    const float* w;
    const float* bias;
    float* out;
    int M, K, NC;       // output rows, contraction length, output cols
    int Lr, Lsrc;       // rows per clip of out / of a
    int CK;             // channels per tap in the contraction (Cin fwd, Cout bwd)
    int Cin;            // conv Cin (weight indexing)
    int ks, stride, pad, dil;
    int lda, ldo;
    int wtm;            // weight is tap-major (Cout, k, Cin)
    int rs;             // data-grad of a strided conv in residue mode (see conv_gemm2_k); rs_lq = ceil(Lin/stride)
    int rs_lq;
    int act;
    float slope, drop_p, inv_keep;
    const unsigned long long* rng;
    unsigned site;
    int accumulate;
};

template <bool BWD>
__device__ __forceinline__ bool src_pos(const GemmP& p, int l, int tap, int& pos) {
    if (!BWD) {
        pos = l * p.stride + tap * p.dil - p.pad;
    } else if (p.rs) {
        pos = l - tap;                 // l already holds i + q0; tap is the slot index
    } else {
        const int t = l + p.pad - tap * p.dil;
        if (t < 0) return false;
        if (p.stride == 1) {
            pos = t;
        } else {
            pos = t / p.stride;
            if (pos * p.stride != t) return false;
        }
    }
    return pos >= 0 && pos < p.Lsrc;
}

template <bool BWD, bool VEC>
__global__ __launch_bounds__(256) void conv_gemm_k(GemmP p) {
    __shared__ float As[BK][PITCH];
    __shared__ float Bs[BK][PITCH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int kq = tid & 3, r = tid >> 2;

    const int m = m0 + r;
    const bool mvalid = m < p.M;
    int nclip = 0, l = 0;
    if (mvalid) {
        nclip = m / p.Lr;
        l = m - nclip * p.Lr;
    }
    const long long arow0 = (long long)nclip * p.Lsrc;
    const int col = n0 + r;
    const bool cvalid = col < p.NC;

    float ra[4], rb[4];
    auto fetch = [&](int k0) {
        const int kk0 = k0 + kq * 4;
        if (VEC) {
            // CK % 4 == 0: the four k's share one tap and are contiguous channels
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int tap = 0, c = 0;
            const bool kval = kk0 < p.K;
            if (kval) {
                tap = kk0 / p.CK;
                c = kk0 - tap * p.CK;
                int pos;
                if (mvalid && src_pos<BWD>(p, l, tap, pos))
                    v = *reinterpret_cast<const float4*>(p.a + (arow0 + pos) * p.lda + c);
            }
            ra[0] = v.x; ra[1] = v.y; ra[2] = v.z; ra[3] = v.w;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float b = 0.f;
                if (kval && cvalid) {
                    const int cc = c + j;
                    b = BWD ? p.w[((long long)cc * p.Cin + col) * p.ks + tap]
                            : p.w[((long long)col * p.Cin + cc) * p.ks + tap];
                }
                rb[j] = b;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = kk0 + j;
                float a = 0.f, b = 0.f;
                if (kk < p.K) {
                    const int tap = kk / p.CK;
                    const int c = kk - tap * p.CK;
                    int pos;
                    if (mvalid && src_pos<BWD>(p, l, tap, pos)) a = p.a[(arow0 + pos) * p.lda + c];
                    if (cvalid)
                        b = BWD ? p.w[((long long)c * p.Cin + col) * p.ks + tap]
                                : p.w[((long long)col * p.Cin + c) * p.ks + tap];
                }
                ra[j] = a;
                rb[j] = b;
            }
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    // wave-uniform tile liveness: skip MFMAs of 16-wide tiles that lie outside the matrix
    bool rowlive[2], collive[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        rowlive[t] = (m0 + wm * 32 + t * 16) < p.M;
        collive[t] = (n0 + wn * 32 + t * 16) < p.NC;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    for (int k0 = 0; k0 < p.K; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            As[kq * 4 + j][r] = ra[j];
            Bs[kq * 4 + j][r] = rb[j];
        }
        __syncthreads();
        if (k0 + BK < p.K) fetch(k0 + BK);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kr = s * 4 + (lane >> 4);
            const int li = lane & 15;
            const float a0 = As[kr][wm * 32 + li], a1 = As[kr][wm * 32 + 16 + li];
            const float b0 = Bs[kr][wn * 32 + li], b1 = Bs[kr][wn * 32 + 16 + li];
            if (rowlive[0] && collive[0]) acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            if (rowlive[0] && collive[1]) acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            if (rowlive[1] && collive[0]) acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            if (rowlive[1] && collive[1]) acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }

    SiteKey key{0, 0};
    const bool drop = (!BWD) && p.drop_p > 0.f;
    if (drop) key = site_key(p.rng, p.site);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int c = n0 + wn * 32 + tj * 16 + (lane & 15);
            if (c >= p.NC) continue;
            const float bias = (!BWD && p.bias) ? p.bias[c] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = m0 + wm * 32 + ti * 16 + (lane >> 4) * 4 + q;
                if (row >= p.M) continue;
                float v = acc[ti][tj][q];
                float* dst = p.out + (long long)row * p.ldo + c;
                if (!BWD) {
                    v = apply_act(v + bias, p.act, p.slope);
                    if (drop) v *= keep_scale(key, (unsigned long long)row * p.NC + c, p.drop_p, p.inv_keep);
                    *dst = v;
                } else {
                    *dst = p.accumulate ? (*dst + v) : v;
                }
            }
        }
}


// ---- second-generation tile kernel ---------------------------------------------------------------------------
// K step 32 with double-buffered LDS (ONE barrier per K tile), the next tile's global loads in flight under the
// current tile's MFMAs, and an intra-block 2-way K split (KG = 2: two groups of waves take alternate halves of every
// K tile and are summed through LDS once at the end) -- twice the waves per CU to hide the L2/HBM latency that
// bounds these small-grid GEMMs.  BM in {64, 32}: 32-row tiles when 64-row tiles would leave CUs idle.
constexpr int BK2 = 32;

// LDS tiles are ROW-major [row][k] with a pitch of 34 floats: the loader's 8-byte stores (8 lanes cover one row's 32
// k's, rows 2 banks apart) and the MFMA fragment reads (16 rows x 2 k's per 32-lane group -> banks 2*row + k) are
// both bank-conflict free.  (A k-major image with an 80-float pitch made the loader's stores 8-way conflicted:
// 4*80 = 0 mod 32 -- LDS writes, not MFMA, bounded the first version.)
constexpr int PK = BK2 + 2;

// cycle accounting of one block for tools/diag_gemm_trace.py (compiled only with -DS2AG_GEMM_TRACE)
#ifdef S2AG_GEMM_TRACE
__device__ unsigned long long g_gemm_trace[8];
#define GEMM_TR_BEGIN() const unsigned long long tr0__ = clock64()
#define GEMM_TR_ADD(slot)                                                                      \
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {                              \
        const unsigned long long t__ = clock64();                                               \
        g_gemm_trace[slot] += t__ - trl__;                                                      \
        trl__ = t__;                                                                            \
    }
#else
#define GEMM_TR_ADD(slot)
#endif

template <bool BWD, bool VEC, int BM_, int WM, int WN, int KG>
__global__ __launch_bounds__(64 * WM * WN * KG) void conv_gemm2_k(GemmP p) {
    constexpr int NT = 64 * WM * WN * KG;
    constexpr int TM = BM_ / (16 * WM), TN = 64 / (16 * WN);
    constexpr int CA = (BM_ * (BK2 / 4) + NT - 1) / NT;          // 4-wide K chunks of A per thread
    constexpr int CB = (64 * (BK2 / 4) + NT - 1) / NT;
    constexpr int KSG = (BK2 / 4) / KG;                          // MFMA k-steps per wave group per tile
    __shared__ __attribute__((aligned(16))) float As[2][BM_][PK];
    __shared__ __attribute__((aligned(16))) float Bs[2][64][PK];
    static_assert(2 * BM_ * PK >= BM_ * 64 || KG == 1, "reduction scratch must fit in As");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / (WM * WN), wr = (wave % (WM * WN)) / WN, wc = wave % WN;
    const int n0 = blockIdx.y * 64;
    // Residue mode (data-grad of a conv with stride s > 1, dilation 1): an input position pos only receives taps
    // t = (pos + pad) mod s, t + s, ...  A block therefore owns BM_ positions of ONE residue class of one clip
    // (pos = rr + s*i), and the contraction runs over ceil(k/s) tap slots instead of k taps (15 -> 3 for the wave
    // encoder): "tap" in the loaders below is the slot j, the real tap is rs_t0 + j*s and l = i + rs_q0 - j.
    const bool rs = BWD && p.rs;
    int m0 = blockIdx.x * BM_;
    int rs_clip = 0, rs_rr = 0, rs_i0 = 0, rs_t0 = 0, rs_q0 = 0;
    if (rs) {
        const int nib = (p.rs_lq + BM_ - 1) / BM_;
        const int bx = blockIdx.x;
        const int ib = bx % nib;
        rs_rr = (bx / nib) % p.stride;
        rs_clip = bx / (nib * p.stride);
        rs_i0 = ib * BM_;
        rs_t0 = (rs_rr + p.pad) % p.stride;
        rs_q0 = (rs_rr + p.pad - rs_t0) / p.stride;
        m0 = 0;
    }
    // global output row of tile row rl, or -1
    auto out_row = [&](int rl) -> long long {
        if (rs) {
            const int i = rs_i0 + rl;
            const int pos = rs_rr + p.stride * i;
            return (i < p.rs_lq && pos < p.Lr) ? (long long)rs_clip * p.Lr + pos : -1;
        }
        const int m = m0 + rl;
        return m < p.M ? (long long)m : -1;
    };

    // per-thread loader coordinates (fixed across the K loop)
    int a_r[CA], a_kq[CA], a_l[CA];
    long long a_row0[CA];
    bool a_ok[CA];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
        const int c = tid + i * NT;
        a_kq[i] = c % (BK2 / 4);
        a_r[i] = c / (BK2 / 4);
        a_ok[i] = (a_r[i] < BM_) && out_row(a_r[i]) >= 0;
        int nclip = 0, l = 0;
        if (a_ok[i]) {
            if (rs) {
                nclip = rs_clip;
                l = rs_i0 + a_r[i] + rs_q0;
            } else {
                const int m = m0 + a_r[i];
                nclip = m / p.Lr;
                l = m - nclip * p.Lr;
            }
        }
        a_l[i] = l;
        a_row0[i] = (long long)nclip * p.Lsrc;
    }
    int b_c[CB], b_kq[CB];
    bool b_ok[CB];
#pragma unroll
    for (int i = 0; i < CB; ++i) {
        const int c = tid + i * NT;
        if (BWD) {
            // data gradient: the weight element of (k = co, n = ci) lies ci-contiguous, so consecutive lanes take
            // consecutive COLUMNS (256-byte runs per k); with the forward mapping (lanes along k) every lane touched its
            // own cache line -- the GRU input-gradient GEMM ran at 27 TFLOP/s
            b_c[i] = c % 64;
            b_kq[i] = c / 64;
        } else {
            b_kq[i] = c % (BK2 / 4);
            b_c[i] = c / (BK2 / 4);
        }
        b_ok[i] = (b_c[i] < 64) && (b_kq[i] < BK2 / 4) && (n0 + b_c[i] < p.NC);
    }

    float ra0[CA][4], rb0[CB][4], ra1[CA][4], rb1[CB][4];   // two tiles in flight
    // (tap, channel) of each chunk's first k, advanced by BK2 per tile instead of dividing every time
    int a_tap[CA], a_c[CA], b_tap[CB], b_cc[CB];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
        a_tap[i] = (a_kq[i] * 4) / p.CK;
        a_c[i] = a_kq[i] * 4 - a_tap[i] * p.CK;
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
        b_tap[i] = (b_kq[i] * 4) / p.CK;
        b_cc[i] = b_kq[i] * 4 - b_tap[i] * p.CK;
    }
    const bool b_contig = (!BWD) && VEC && (p.ks == 1 || p.wtm) && ((reinterpret_cast<uintptr_t>(p.w) & 15) == 0);
    const bool b_lin_bwd = BWD && p.ks == 1 && !rs;     // data gradient of a Linear / GRU projection: w[k = co][n = ci]
    // Operand pointers are carried from tile to tile (+= BK2 per tile) and re-derived only when a chunk crosses into the
    // next tap: computing (row, position, channel) -> address from scratch for every chunk of every tile was a third
    // of the block's cycles (tools/diag_gemm_trace.py: "load issue" 31-46 %).
    const float* a_ptr[CA];
    auto a_setup = [&](int i) {
        int pos;
        a_ptr[i] = (a_ok[i] && src_pos<BWD>(p, a_l[i], a_tap[i], pos)) ? p.a + (a_row0[i] + pos) * p.lda + a_c[i]
                                                                        : nullptr;
    };
    const float* b_ptr[CB];
#pragma unroll
    for (int i = 0; i < CA; ++i) a_setup(i);
#pragma unroll
    for (int i = 0; i < CB; ++i) {
        const long long col = n0 + b_c[i];
        b_ptr[i] = !b_ok[i] ? nullptr
                            : (b_contig ? p.w + col * p.K + b_kq[i] * 4
                                        : (b_lin_bwd ? p.w + (long long)(b_kq[i] * 4) * p.Cin + col : p.w));
    }
    auto fetch = [&](float (&ra)[CA][4], float (&rb)[CB][4], int k0) {   // call with k0 = 0, BK2, 2*BK2, ... in order
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            const int kk0 = k0 + a_kq[i] * 4;
            if (VEC) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a_ptr[i] && kk0 < p.K) v = *reinterpret_cast<const float4*>(a_ptr[i]);
                ra[i][0] = v.x; ra[i][1] = v.y; ra[i][2] = v.z; ra[i][3] = v.w;
            } else {
                int tap = a_tap[i], c = a_c[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = 0.f;
                    if (a_ok[i] && kk0 + j < p.K) {
                        int pos;
                        if (src_pos<BWD>(p, a_l[i], tap, pos)) a = p.a[(a_row0[i] + pos) * p.lda + c];
                    }
                    ra[i][j] = a;
                    if (++c == p.CK) { c = 0; ++tap; }
                }
            }
            a_c[i] += BK2;
            if (a_c[i] >= p.CK) {
                do { a_c[i] -= p.CK; ++a_tap[i]; } while (a_c[i] >= p.CK);
                if (VEC) a_setup(i);
            } else if (VEC && a_ptr[i]) {
                a_ptr[i] += BK2;
            }
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int kk0 = k0 + b_kq[i] * 4;
            const int col = n0 + b_c[i];
            if (b_contig) {             // Linear / GRU projection: the weight row is K-contiguous -> one 16-byte load
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (b_ptr[i] && kk0 < p.K) v = *reinterpret_cast<const float4*>(b_ptr[i]);
                rb[i][0] = v.x; rb[i][1] = v.y; rb[i][2] = v.z; rb[i][3] = v.w;
                if (b_ptr[i]) b_ptr[i] += BK2;
            } else if (b_lin_bwd) {     // four k rows of W, lanes along the contiguous input-channel axis
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rb[i][j] = (b_ptr[i] && kk0 + j < p.K) ? b_ptr[i][(long long)j * p.Cin] : 0.f;
                if (b_ptr[i]) b_ptr[i] += (long long)BK2 * p.Cin;
            } else {
                int tap = b_tap[i], c = b_cc[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float b = 0.f;
                    if (b_ok[i] && kk0 + j < p.K) {
                        const int tp = rs ? rs_t0 + tap * p.stride : tap;      // residue mode: slot -> real tap
                        if (tp < p.ks) {
                            if (p.wtm)
                                b = BWD ? p.w[((long long)c * p.ks + tp) * p.Cin + col]
                                        : p.w[((long long)col * p.ks + tp) * p.Cin + c];
                            else
                                b = BWD ? p.w[((long long)c * p.Cin + col) * p.ks + tp]
                                        : p.w[((long long)col * p.Cin + c) * p.ks + tp];
                        }
                    }
                    rb[i][j] = b;
                    if (++c == p.CK) { c = 0; ++tap; }
                }
            }
            b_cc[i] += BK2;
            while (b_cc[i] >= p.CK) { b_cc[i] -= p.CK; ++b_tap[i]; }
        }
    };
    auto stash = [&](const float (&ra)[CA][4], const float (&rb)[CB][4], int buf) {
#pragma unroll
        for (int i = 0; i < CA; ++i)
            if (a_r[i] < BM_) {
                float* d = &As[buf][a_r[i]][a_kq[i] * 4];
                *reinterpret_cast<float2*>(d) = make_float2(ra[i][0], ra[i][1]);
                *reinterpret_cast<float2*>(d + 2) = make_float2(ra[i][2], ra[i][3]);
            }
#pragma unroll
        for (int i = 0; i < CB; ++i)
            if (b_c[i] < 64) {
                float* d = &Bs[buf][b_c[i]][b_kq[i] * 4];
                *reinterpret_cast<float2*>(d) = make_float2(rb[i][0], rb[i][1]);
                *reinterpret_cast<float2*>(d + 2) = make_float2(rb[i][2], rb[i][3]);
            }
    };

    bool rowlive[TM], collive[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) rowlive[t] = rs ? (rs_i0 + (wr * TM + t) * 16) < p.rs_lq : (m0 + (wr * TM + t) * 16) < p.M;
#pragma unroll
    for (int t = 0; t < TN; ++t) collive[t] = (n0 + (wc * TN + t) * 16) < p.NC;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nkt = (p.K + BK2 - 1) / BK2;
    auto mma = [&](int cur) {
#pragma unroll
        for (int s = 0; s < KSG; ++s) {
            const int kr = (kg * KSG + s) * 4 + (lane >> 4);
            const int li = lane & 15;
            float a[TM], b[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) a[t] = As[cur][(wr * TM + t) * 16 + li][kr];
#pragma unroll
            for (int t = 0; t < TN; ++t) b[t] = Bs[cur][(wc * TN + t) * 16 + li][kr];
#pragma unroll
            for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                for (int tj = 0; tj < TN; ++tj)
                    if (rowlive[ti] && collive[tj])
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti], b[tj], acc[ti][tj], 0, 0, 0);
        }
    };
    // software pipeline, two tiles deep: tile kt+2 is requested while tile kt is multiplied and tile kt+1 (requested
    // one iteration earlier, so it has had a full iteration to land) is moved from registers into the idle LDS buffer
#ifdef S2AG_GEMM_TRACE
    unsigned long long trl__ = clock64();
#endif
    fetch(ra0, rb0, 0);
    stash(ra0, rb0, 0);
    if (nkt > 1) fetch(ra0, rb0, BK2);
    __syncthreads();
    GEMM_TR_ADD(0);                                   // prologue
    for (int kt = 0; kt < nkt; kt += 2) {
        if (kt + 2 < nkt) fetch(ra1, rb1, (kt + 2) * BK2);
        GEMM_TR_ADD(1);                               // issue of the global loads
        mma(0);
        GEMM_TR_ADD(2);                               // LDS reads + MFMAs
        if (kt + 1 < nkt) stash(ra0, rb0, 1);
        GEMM_TR_ADD(3);                               // wait for the tile in flight + LDS stores
        __syncthreads();
        GEMM_TR_ADD(4);                               // barrier
        if (kt + 1 >= nkt) break;
        if (kt + 3 < nkt) fetch(ra0, rb0, (kt + 3) * BK2);
        GEMM_TR_ADD(1);
        mma(1);
        GEMM_TR_ADD(2);
        if (kt + 2 < nkt) stash(ra1, rb1, 0);
        GEMM_TR_ADD(3);
        __syncthreads();
        GEMM_TR_ADD(4);
    }

    if (KG == 2) {       // sum the two K halves: group 1 parks its accumulators in LDS, group 0 finishes
        float* red = &As[0][0][0];
        if (kg == 1) {
#pragma unroll
            for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                for (int tj = 0; tj < TN; ++tj)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        red[((wr * TM + ti) * 16 + (lane >> 4) * 4 + q) * 64 + (wc * TN + tj) * 16 + (lane & 15)] =
                            acc[ti][tj][q];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
            for (int tj = 0; tj < TN; ++tj)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[ti][tj][q] +=
                        red[((wr * TM + ti) * 16 + (lane >> 4) * 4 + q) * 64 + (wc * TN + tj) * 16 + (lane & 15)];
    }

    SiteKey key{0, 0};
    const bool drop = (!BWD) && p.drop_p > 0.f;
    if (drop) key = site_key(p.rng, p.site);
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < TN; ++tj) {
            const int c = n0 + (wc * TN + tj) * 16 + (lane & 15);
            if (c >= p.NC) continue;
            const float bias = (!BWD && p.bias) ? p.bias[c] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const long long row = out_row((wr * TM + ti) * 16 + (lane >> 4) * 4 + q);
                if (row < 0) continue;
                float v = acc[ti][tj][q];
                float* dst = p.out + row * p.ldo + c;
                if (!BWD) {
                    v = apply_act(v + bias, p.act, p.slope);
                    if (drop) v *= keep_scale(key, (unsigned long long)row * p.NC + c, p.drop_p, p.inv_keep);
                    *dst = v;
                } else {
                    *dst = p.accumulate ? (*dst + v) : v;
                }
            }
        }
    GEMM_TR_ADD(5);                                   // K-group merge + epilogue
}

template <bool BWD, bool VEC>
void launch_gemm2(const GemmP& p, hipStream_t stream) {
    const long long colb = cdiv(p.NC, 64);
    // row blocks: plain mode tiles the M rows; residue mode tiles (clip, residue, ceil(Lin/stride)) separately
    const long long nclips = p.rs ? p.M / p.Lr : 1;
    auto rowblocks = [&](int bm) -> long long {
        return p.rs ? nclips * p.stride * cdiv(p.rs_lq, bm) : (long long)cdiv(p.M, bm);
    };
    // largest row tile that still gives >= 512 blocks (2 per CU)
    if (rowblocks(64) * colb >= 512) {
        hipLaunchKernelGGL((conv_gemm2_k<BWD, VEC, 64, 2, 2, 2>), dim3((unsigned)rowblocks(64), (unsigned)colb), dim3(512), 0, stream, p);
    } else {
        hipLaunchKernelGGL((conv_gemm2_k<BWD, VEC, 32, 2, 2, 2>), dim3((unsigned)rowblocks(32), (unsigned)colb), dim3(512), 0, stream, p);
    }
}

// ---- weight gradient: dw[co, ci, tap] += sum_m gy[m, co] * xwin[m, (tap, ci)] -----------------------
struct WgradP {
    const float* gy;
    const float* x;
    float* dw;
    float* db;   // nullable: bias gradient, db[co] += sum_m gy[m, co] (folded into the blocks of the first column tile)
    int Mtot, Lin, Lout, Cin, Cout, ks, stride, pad, dil, ldx, ldg, wtm;
    int chunk;   // rows per z-slice, multiple of BK
};

__global__ __launch_bounds__(256) void conv_wgrad_k(WgradP p) {
    __shared__ float As[BK][PITCH];   // [m][co]
    __shared__ float Bs[BK][PITCH];   // [m][j = tap*Cin+ci]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = blockIdx.x * BM, j0 = blockIdx.y * BN;
    const int mbeg = blockIdx.z * p.chunk;
    const int mend = min(p.Mtot, mbeg + p.chunk);
    const int cidx = tid & 63, mq = tid >> 6;
    const int NCW = p.ks * p.Cin;

    const int co = co0 + cidx;
    const bool covalid = co < p.Cout;
    const int jcol = j0 + cidx;
    const bool jvalid = jcol < NCW;
    int tap = 0, ci = 0;
    if (jvalid) {
        tap = jcol / p.Cin;
        ci = jcol - tap * p.Cin;
    }
    const int tapoff = tap * p.dil - p.pad;

    float ra[4], rb[4];
    const bool want_db = p.db != nullptr && blockIdx.y == 0;     // the gy tile passes through this thread anyway
    float bsum = 0.f;
    auto fetch = [&](int mb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mm = mb + mq * 4 + j;
            float a = 0.f, b = 0.f;
            if (mm < mend) {
                if (covalid) a = p.gy[(long long)mm * p.ldg + co];
                if (jvalid) {
                    const int nclip = mm / p.Lout;
                    const int l = mm - nclip * p.Lout;
                    const int pos = l * p.stride + tapoff;
                    if (pos >= 0 && pos < p.Lin) b = p.x[((long long)nclip * p.Lin + pos) * p.ldx + ci];
                }
            }
            ra[j] = a;
            rb[j] = b;
            bsum += a;
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    bool rowlive[2], collive[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        rowlive[t] = (co0 + wm * 32 + t * 16) < p.Cout;
        collive[t] = (j0 + wn * 32 + t * 16) < NCW;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (mbeg < mend) fetch(mbeg);
    for (int mb = mbeg; mb < mend; mb += BK) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            As[mq * 4 + j][cidx] = ra[j];
            Bs[mq * 4 + j][cidx] = rb[j];
        }
        __syncthreads();
        if (mb + BK < mend) fetch(mb + BK);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kr = s * 4 + (lane >> 4);
            const int li = lane & 15;
            const float a0 = As[kr][wm * 32 + li], a1 = As[kr][wm * 32 + 16 + li];
            const float b0 = Bs[kr][wn * 32 + li], b1 = Bs[kr][wn * 32 + 16 + li];
            if (rowlive[0] && collive[0]) acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            if (rowlive[0] && collive[1]) acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            if (rowlive[1] && collive[0]) acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            if (rowlive[1] && collive[1]) acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    s2ag::det_enter();                                // deterministic mode: the z-slices of a tile add in index order
    if (want_db) {                                    // block-uniform: the four row groups of a column meet in LDS
        __syncthreads();
        As[mq][cidx] = bsum;
        __syncthreads();
        if (mq == 0 && covalid) atomicAdd(p.db + co, As[0][cidx] + As[1][cidx] + As[2][cidx] + As[3][cidx]);
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int jc = j0 + wn * 32 + tj * 16 + (lane & 15);
            if (jc >= NCW) continue;
            const int t2 = jc / p.Cin, c2 = jc - t2 * p.Cin;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = co0 + wm * 32 + ti * 16 + (lane >> 4) * 4 + q;
                if (row >= p.Cout) continue;
                atomicAdd(p.dw + (p.wtm ? ((long long)row * p.ks + t2) * p.Cin + c2
                                        : ((long long)row * p.Cin + c2) * p.ks + t2), acc[ti][tj][q]);
            }
        }
    s2ag::det_leave();
}

// ---- weight gradient, second generation --------------------------------------------------------------------
// Same pipeline as conv_gemm2_k: 32 contraction rows per tile, double-buffered LDS (one barrier per tile), next-but-
// one tile in flight, 8 waves with a 2-way in-block split of every tile.  Both operands are read along their
// contiguous channel axis (gy rows over co, x rows over ci), so every global load is coalesced and the k-major LDS
// image is written conflict-free.  Blocks split the clips*frames axis (grid.z) and merge with fp32 atomics.
__global__ __launch_bounds__(512) void conv_wgrad2_k(WgradP p) {
    constexpr int PW = 64 + 16;                       // k-major pitch: fragment reads of 4 k-rows hit disjoint banks
    __shared__ float As[2][BK2][PW];                  // [m][co]
    __shared__ float Bs[2][BK2][PW];                  // [m][j = tap*Cin + ci]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int co0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const int mbeg = blockIdx.z * p.chunk;
    const int mend = min(p.Mtot, mbeg + p.chunk);
    const int cidx = tid & 63, mq = tid >> 6;         // mq 0..7 -> rows mq*4 .. mq*4+3 of the 32-row tile
    const int NCW = p.ks * p.Cin;
    const int co = co0 + cidx;
    const bool covalid = co < p.Cout;
    const int jcol = j0 + cidx;
    const bool jvalid = jcol < NCW;
    int tap = 0, ci = 0;
    if (jvalid) {
        tap = jcol / p.Cin;
        ci = jcol - tap * p.Cin;
    }
    const int tapoff = tap * p.dil - p.pad;

    float ra0[4], rb0[4], ra1[4], rb1[4];
    const bool want_db = p.db != nullptr && blockIdx.y == 0;
    float bsum = 0.f;
    auto fetch = [&](float (&ra)[4], float (&rb)[4], int mb) {
        int mm = mb + mq * 4;
        int nclip = mm / p.Lout;
        int l = mm - nclip * p.Lout;
#pragma unroll
        for (int j = 0; j < 4; ++j, ++mm) {
            float a = 0.f, b = 0.f;
            if (mm < mend) {
                if (covalid) a = p.gy[(long long)mm * p.ldg + co];
                if (jvalid) {
                    const int pos = l * p.stride + tapoff;
                    if (pos >= 0 && pos < p.Lin) b = p.x[((long long)nclip * p.Lin + pos) * p.ldx + ci];
                }
            }
            ra[j] = a;
            rb[j] = b;
            bsum += a;
            if (++l == p.Lout) { l = 0; ++nclip; }
        }
    };
    auto stash = [&](const float (&ra)[4], const float (&rb)[4], int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            As[buf][mq * 4 + j][cidx] = ra[j];
            Bs[buf][mq * 4 + j][cidx] = rb[j];
        }
    };
    bool rowlive[2], collive[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        rowlive[t] = (co0 + wm * 32 + t * 16) < p.Cout;
        collive[t] = (j0 + wn * 32 + t * 16) < NCW;
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](int cur) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int kr = (kg * 4 + s) * 4 + (lane >> 4);
            const int li = lane & 15;
            const float a0 = As[cur][kr][wm * 32 + li], a1 = As[cur][kr][wm * 32 + 16 + li];
            const float b0 = Bs[cur][kr][wn * 32 + li], b1 = Bs[cur][kr][wn * 32 + 16 + li];
            if (rowlive[0] && collive[0]) acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
            if (rowlive[0] && collive[1]) acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
            if (rowlive[1] && collive[0]) acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
            if (rowlive[1] && collive[1]) acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    };
    const int nkt = (mend - mbeg + BK2 - 1) / BK2;
    if (nkt <= 0) {
        s2ag::det_enter();                            // an empty slice still takes (and passes on) its turn
        s2ag::det_leave();
        return;
    }
    fetch(ra0, rb0, mbeg);
    stash(ra0, rb0, 0);
    if (nkt > 1) fetch(ra0, rb0, mbeg + BK2);
    __syncthreads();
    for (int kt = 0; kt < nkt; kt += 2) {
        if (kt + 2 < nkt) fetch(ra1, rb1, mbeg + (kt + 2) * BK2);
        mma(0);
        if (kt + 1 < nkt) stash(ra0, rb0, 1);
        __syncthreads();
        if (kt + 1 >= nkt) break;
        if (kt + 3 < nkt) fetch(ra0, rb0, mbeg + (kt + 3) * BK2);
        mma(1);
        if (kt + 2 < nkt) stash(ra1, rb1, 0);
        __syncthreads();
    }
    // merge the two wave groups through LDS so only one of them issues the (cross-block) atomics
    if (want_db) Bs[0][mq][cidx] = bsum;              // Bs is idle after the last tile; read behind the barrier below
    {
        float* redw = &As[0][0][0];                   // 2*32*80 floats >= 64*64
        if (kg == 1) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        redw[(wm * 32 + ti * 16 + (lane >> 4) * 4 + q) * 64 + wn * 32 + tj * 16 + (lane & 15)] =
                            acc[ti][tj][q];
        }
        __syncthreads();
        s2ag::det_enter();                            // (all eight waves: the second wave group leaves right below)
        if (want_db && mq == 0 && covalid) {          // wave 0 (kg = 0): eight row groups per column
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += Bs[0][q][cidx];
            atomicAdd(p.db + co, t);
        }
        if (kg == 1) return;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[ti][tj][q] +=
                        redw[(wm * 32 + ti * 16 + (lane >> 4) * 4 + q) * 64 + wn * 32 + tj * 16 + (lane & 15)];
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int jc = j0 + wn * 32 + tj * 16 + (lane & 15);
            if (jc >= NCW) continue;
            const int t2 = jc / p.Cin, c2 = jc - t2 * p.Cin;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = co0 + wm * 32 + ti * 16 + (lane >> 4) * 4 + q;
                if (row >= p.Cout) continue;
                atomicAdd(p.dw + (p.wtm ? ((long long)row * p.ks + t2) * p.Cin + c2
                                        : ((long long)row * p.Cin + c2) * p.ks + t2), acc[ti][tj][q]);
            }
        }
    s2ag::det_leave();
}

// ---- column sums (bias gradients, BatchNorm batch statistics) ---------------------------------------
__global__ __launch_bounds__(256) void colsum_k(const float* __restrict__ x, int rows, int cols, int ld,
                                                int rows_per_block, float* out, float* sq) {
    __shared__ float s1[4][64], s2[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ry = threadIdx.x >> 6;
    const int rbeg = blockIdx.y * rows_per_block;
    const int rend = min(rows, rbeg + rows_per_block);
    float a = 0.f, b = 0.f;
    if (c < cols)
        for (int r = rbeg + ry; r < rend; r += 4) {
            const float v = x[(long long)r * ld + c];
            a += v;
            b += v * v;
        }
    s1[ry][threadIdx.x & 63] = a;
    s2[ry][threadIdx.x & 63] = b;
    __syncthreads();
    s2ag::det_enter();
    if (ry == 0 && c < cols) {
        const int i = threadIdx.x;
        atomicAdd(out + c, s1[0][i] + s1[1][i] + s1[2][i] + s1[3][i]);
        if (sq) atomicAdd(sq + c, s2[0][i] + s2[1][i] + s2[2][i] + s2[3][i]);
    }
    s2ag::det_leave();
}

__global__ __launch_bounds__(256) void colstats_f64_k(const float* __restrict__ x, int rows, int cols, int ld,
                                                      int rows_per_block, double* out, double* sq) {
    __shared__ double s1[4][64], s2[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ry = threadIdx.x >> 6;
    const int rbeg = blockIdx.y * rows_per_block;
    const int rend = min(rows, rbeg + rows_per_block);
    double a = 0.0, b = 0.0;
    if (c < cols)
        for (int r = rbeg + ry; r < rend; r += 4) {
            const double v = (double)x[(long long)r * ld + c];
            a += v;
            b += v * v;
        }
    s1[ry][threadIdx.x & 63] = a;
    s2[ry][threadIdx.x & 63] = b;
    __syncthreads();
    s2ag::det_enter();
    if (ry == 0 && c < cols) {
        const int i = threadIdx.x;
        atomicAdd(out + c, s1[0][i] + s1[1][i] + s1[2][i] + s1[3][i]);
        atomicAdd(sq + c, s2[0][i] + s2[1][i] + s2[2][i] + s2[3][i]);
    }
    s2ag::det_leave();
}

// ---- lane-dense reductions for narrow contiguous matrices (cols in {1,2,4,...,32}, ld == cols) ----------------------
// The matrix is walked as a flat array with a stride that is a multiple of 256, so a thread always sees the same
// column (tid % cols) and every lane is busy -- the column-per-lane kernels above use 16 of 64 lanes on the wave
// encoder's 16-channel activations (1 M rows).
template <typename T>
__device__ __forceinline__ void block_col_merge(T v, int cols, T* sm, T* out) {
    sm[threadIdx.x] = v;
    __syncthreads();
    if ((int)threadIdx.x < cols) {
        T s = 0;
        for (int i = threadIdx.x; i < 256; i += cols) s += sm[i];
        atomicAdd(out + threadIdx.x, s);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void colsum_flat_k(const float* __restrict__ x, long long total, int cols, float* out,
                                                     float* sq) {
    __shared__ float sm[256];
    float a = 0.f, b = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const float v = x[i];
        a += v;
        b += v * v;
    }
    s2ag::det_enter();
    block_col_merge(a, cols, sm, out);
    if (sq) block_col_merge(b, cols, sm, sq);
    s2ag::det_leave();
}

__global__ __launch_bounds__(256) void colstats_flat_k(const float* __restrict__ x, long long total, int cols,
                                                       double* out, double* sq) {
    __shared__ double sm[256];
    double a = 0.0, b = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const double v = (double)x[i];
        a += v;
        b += v * v;
    }
    s2ag::det_enter();
    block_col_merge(a, cols, sm, out);
    block_col_merge(b, cols, sm, sq);
    s2ag::det_leave();
}

inline bool narrow_ok(int cols, int ld) { return ld == cols && cols <= 32 && (cols & (cols - 1)) == 0; }
inline int flat_blocks(long long total) {
    long long b = (total + 256 * 16 - 1) / (256 * 16);
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool bad_geom(const s2ag_conv_geom* g) {
    return !g || g->N <= 0 || g->Lin <= 0 || g->Lout <= 0 || g->Cin <= 0 || g->Cout <= 0 || g->ksize <= 0 ||
           g->stride <= 0 || g->dil <= 0 || g->ldx < g->Cin || g->ldy < g->Cout;
}
}  // namespace

// 1-tap / tap-major stride-1 layers go through the straight-line kernels of gemm_lin.hip (the A/B switch that routed them
// through the general kernel is gone: +60 % on those layers, r01)
static constexpr bool use_gemm_lin() { return true; }

int s2ag_bwd_pair(const float* gy, const float* w, const float* x, float* dx, float* dw, float* db, int nclips, int L,
                  int Cin, int Cout, int ks, int pad, int dil, int ldx, int ldg, int wtm, int chunk, int nsplit,
                  hipStream_t stream);
struct s2ag_wg_job_i {
    const float* gy;
    const float* x;
    float* dw;
    float* db;
    int nclips, Lin, Lout, Cin, Cout, ks, stride, pad, dil, ldx, ldg, wtm, chunk, nsplit;
};
int s2ag_wgrad_multi(const s2ag_wg_job_i* jb, int n, hipStream_t stream);
// conv_c1.hip
int s2ag_conv_c1_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Lin, int Lout, int Cin,
                     int Cout, int ks, int stride, int pad, int dil, int ldx, int ldy, double* stats, int stats_cap_rows,
                     hipStream_t stream);
int s2ag_conv_c1_wgrad(const float* gy, const float* x, float* dw, float* db, int N, int Lin, int Lout, int Cin, int Cout,
                       int ks, int stride, int pad, int dil, int ldx, int ldg, hipStream_t stream);
// gemm_lin.hip
int s2ag_gemm_lin_fwd(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int ldx, int ldy,
                      int act, float slope, float drop_p, const unsigned long long* rng, unsigned site, double* stats,
                      hipStream_t stream);
int s2ag_gemm_lin_bwd_data(const float* gy, const float* w, float* dx, int M, int Cout, int Cin, int ldg, int ldx,
                           int accumulate, hipStream_t stream);
int s2ag_wgrad_lin(const float* gy, const float* x, float* dw, float* db, int nclips, int Lin, int Lout, int Cin,
                   int Cout, int ks, int stride, int pad, int dil, int ldx, int ldg, int wtm, int chunk, int nsplit,
                   hipStream_t stream);
int s2ag_gemm_conv_tm_fwd(const float* x, const float* w, const float* bias, float* y, int nclips, int Lin, int Lout,
                          int Cin, int Cout, int ks, int stride, int pad, int dil, int ldx, int ldy, int act, float slope,
                          float drop_p, const unsigned long long* rng, unsigned site, double* stats, hipStream_t stream);
int s2ag_gemm_conv_tm_bwd_data(const float* gy, const float* w, float* dx, int nclips, int L, int Cin, int Cout, int ks,
                               int pad, int dil, int ldg, int ldx, int accumulate, hipStream_t stream);
// conv_pp.hip
int s2ag_conv_fwd_fw(const float* x, const float* w, const float* bias, float* y, int N, int Lin, int Lout, int Cin,
                     int Cout, int ks, int stride, int pad, int dil, int ldx, int ldy, int wtm, int act, float slope,
                     float drop_p, double* stats, hipStream_t stream);
int s2ag_conv_dgrad_pp(const float* gy, const float* w, float* dx, int N, int Lin, int Lout, int Cin, int Cout, int ks,
                       int stride, int pad, int dil, int ldg, int ldx, int wtm, int accumulate, hipStream_t stream);

extern "C" int s2ag_abi_version(void) { return S2AG_ABI_VERSION; }

static int conv1d_nlc_fwd_impl(const float* x, const float* w, const float* bias, float* y, const s2ag_conv_geom* g,
                               const s2ag_epilogue* e, double* stats, int* stat_rows, void* stream) {
    if (stat_rows) *stat_rows = 0;
    if (bad_geom(g) || !x || !w || !y) return S2AG_E_BADARG;
    if (e && e->drop_p > 0.f && !e->rng) return S2AG_E_BADARG;
    if (e && (e->drop_p < 0.f || e->drop_p >= 1.f)) return S2AG_E_BADARG;
    GemmP p{};
    p.a = x; p.w = w; p.bias = bias; p.out = y;
    p.M = g->N * g->Lout; p.K = g->ksize * g->Cin; p.NC = g->Cout;
    p.Lr = g->Lout; p.Lsrc = g->Lin; p.CK = g->Cin; p.Cin = g->Cin;
    p.ks = g->ksize; p.stride = g->stride; p.pad = g->pad; p.dil = g->dil;
    p.lda = g->ldx; p.ldo = g->ldy; p.wtm = g->w_tap_major;
    p.act = e ? e->act : S2AG_ACT_NONE;
    p.slope = e ? e->slope : 1.f;
    p.drop_p = e ? e->drop_p : 0.f;
    p.inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
    p.rng = e ? e->rng : nullptr;
    p.site = e ? e->site : 0;
    p.accumulate = 0;
    int rows_ = 0;
    // 1-tap layers (Linear, GRU projections, 1x1 convs): the straight-line kernel of gemm_lin.hip
    if (use_gemm_lin() && g->ksize == 1 && g->stride == 1 && g->pad == 0 && g->Lin == g->Lout &&
        (rows_ = s2ag_gemm_lin_fwd(x, w, bias, y, p.M, g->Cin, g->Cout, g->ldx, g->ldy, p.act, p.slope, p.drop_p, p.rng,
                                   p.site, stats, (hipStream_t)stream))) {
        S2AG_LAUNCH_CHECK();
        if (stat_rows && stats) *stat_rows = rows_;
        return 0;
    }
    // the wave encoder's Conv1d(16, 32, 15, stride 6): flat-window kernel with the weights in registers (conv_pp.hip)
    if (g->stride > 1 &&
        (rows_ = s2ag_conv_fwd_fw(x, w, bias, y, g->N, g->Lin, g->Lout, g->Cin, g->Cout, g->ksize, g->stride, g->pad, g->dil,
                                  g->ldx, g->ldy, g->w_tap_major, p.act, p.slope, p.drop_p, stats, (hipStream_t)stream))) {
        S2AG_LAUNCH_CHECK();
        if (stat_rows && stats) *stat_rows = rows_;
        return 0;
    }
    // stride-1 convs with tap-major weights (TCN, folded ST-GCN): the same straight-line kernel with (tap, channel) tracking
    if (use_gemm_lin() && g->ksize > 1 && g->w_tap_major &&
        (rows_ = s2ag_gemm_conv_tm_fwd(x, w, bias, y, g->N, g->Lin, g->Lout, g->Cin, g->Cout, g->ksize, g->stride, g->pad,
                                       g->dil, g->ldx, g->ldy, p.act, p.slope, p.drop_p, p.rng, p.site, stats,
                                       (hipStream_t)stream))) {
        S2AG_LAUNCH_CHECK();
        if (stat_rows && stats) *stat_rows = rows_;
        return 0;
    }
    // the one-channel waveform conv: direct vector-ALU kernel (conv_c1.hip)
    if (use_gemm_lin() && g->Cin == 1 && p.act == S2AG_ACT_NONE && p.drop_p == 0.f &&
        (rows_ = s2ag_conv_c1_fwd(x, w, bias, y, g->N, g->Lin, g->Lout, g->Cin, g->Cout, g->ksize, g->stride, g->pad,
                                  g->dil, g->ldx, g->ldy, stats, 2 * cdiv((long long)g->N * g->Lout, 32),
                                  (hipStream_t)stream))) {
        S2AG_LAUNCH_CHECK();
        if (stat_rows && stats) *stat_rows = rows_;
        return 0;
    }
    const bool vec = (g->Cin % 4 == 0) && (g->ldx % 4 == 0) && aligned16(x);
    if (vec)
        launch_gemm2<false, true>(p, (hipStream_t)stream);
    else
        launch_gemm2<false, false>(p, (hipStream_t)stream);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_conv1d_nlc_fwd(const float* x, const float* w, const float* bias, float* y,
                                   const s2ag_conv_geom* g, const s2ag_epilogue* e, void* stream) {
    return conv1d_nlc_fwd_impl(x, w, bias, y, g, e, nullptr, nullptr, stream);
}

extern "C" int s2ag_conv_stats_rows(const s2ag_conv_geom* g) {
    if (bad_geom(g)) return S2AG_E_BADARG;
    return 2 * cdiv((long long)g->N * g->Lout, 32);
}

extern "C" int s2ag_conv1d_nlc_fwd_stats(const float* x, const float* w, const float* bias, float* y,
                                         const s2ag_conv_geom* g, const s2ag_epilogue* e, double* partials,
                                         int* stat_rows, void* stream) {
    if (!partials || !stat_rows) return S2AG_E_BADARG;
    return conv1d_nlc_fwd_impl(x, w, bias, y, g, e, partials, stat_rows, stream);
}

extern "C" int s2ag_conv1d_nlc_bwd_data(const float* gy, const float* w, float* dx, const s2ag_conv_geom* g,
                                        int accumulate, void* stream) {
    if (bad_geom(g) || !gy || !w || !dx) return S2AG_E_BADARG;
    GemmP p{};
    p.a = gy; p.w = w; p.bias = nullptr; p.out = dx;
    p.M = g->N * g->Lin; p.K = g->ksize * g->Cout; p.NC = g->Cin;
    p.Lr = g->Lin; p.Lsrc = g->Lout; p.CK = g->Cout; p.Cin = g->Cin;
    p.ks = g->ksize; p.stride = g->stride; p.pad = g->pad; p.dil = g->dil;
    p.lda = g->ldy; p.ldo = g->ldx; p.wtm = g->w_tap_major;
    if (use_gemm_lin() && g->ksize == 1 && g->stride == 1 && g->pad == 0 && g->Lin == g->Lout &&
        s2ag_gemm_lin_bwd_data(gy, w, dx, p.M, g->Cout, g->Cin, g->ldy, g->ldx, accumulate, (hipStream_t)stream)) {
        S2AG_LAUNCH_CHECK();
        return 0;
    }
    if (use_gemm_lin() && g->ksize > 1 && g->w_tap_major && g->stride == 1 && g->Lin == g->Lout &&
        s2ag_gemm_conv_tm_bwd_data(gy, w, dx, g->N, g->Lin, g->Cin, g->Cout, g->ksize, g->pad, g->dil, g->ldy, g->ldx,
                                   accumulate, (hipStream_t)stream)) {
        S2AG_LAUNCH_CHECK();
        return 0;
    }
    // the wave encoder's strided convs: poly-phase kernel (other strided geometries: the general kernel's residue mode)
    if (g->stride > 1 &&
        s2ag_conv_dgrad_pp(gy, w, dx, g->N, g->Lin, g->Lout, g->Cin, g->Cout, g->ksize, g->stride, g->pad, g->dil, g->ldy,
                           g->ldx, g->w_tap_major, accumulate, (hipStream_t)stream)) {
        S2AG_LAUNCH_CHECK();
        return 0;
    }
    if (g->stride > 1 && g->dil == 1 && g->pad >= 0) {      // residue mode: contract over ceil(k/stride) tap slots
        p.rs = 1;
        p.rs_lq = cdiv(g->Lin, g->stride);
        p.K = cdiv(g->ksize, g->stride) * g->Cout;
    }
    p.act = S2AG_ACT_NONE; p.slope = 1.f; p.drop_p = 0.f; p.inv_keep = 1.f; p.rng = nullptr; p.site = 0;
    p.accumulate = accumulate;
    const bool vec = (g->Cout % 4 == 0) && (g->ldy % 4 == 0) && aligned16(gy);
    if (vec)
        launch_gemm2<true, true>(p, (hipStream_t)stream);
    else
        launch_gemm2<true, false>(p, (hipStream_t)stream);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_conv1d_nlc_bwd_weight(const float* gy, const float* x, float* dw, float* dbias,
                                          const s2ag_conv_geom* g, int accumulate, void* stream) {
    if (bad_geom(g) || !gy || !x || !dw) return S2AG_E_BADARG;
    WgradP p{};
    p.gy = gy; p.x = x; p.dw = dw; p.db = dbias;
    p.Mtot = g->N * g->Lout; p.Lin = g->Lin; p.Lout = g->Lout; p.Cin = g->Cin; p.Cout = g->Cout;
    p.ks = g->ksize; p.stride = g->stride; p.pad = g->pad; p.dil = g->dil; p.ldx = g->ldx; p.ldg = g->ldy;
    p.wtm = g->w_tap_major;
    const int tiles = cdiv(g->Cout, BM) * cdiv(g->ksize * g->Cin, BN);
    // split of the clips*frames axis: the straight-line kernel (stride-1 layers) likes ~384 blocks -- longer K loops, a
    // third of the merge atomics (sweep 256...1024 in the full step); the general kernels keep ~1024
    const bool lin_ok = use_gemm_lin() && ((g->Lin == 1 && g->Lout == 1 && g->pad == 0 && g->ksize == 1 && g->stride == 1) ||
                                           g->Lout >= BK2);
    int nsplit = cdiv(lin_ok ? 384 : 1024, tiles);
    const int max_split = cdiv(p.Mtot, 4 * BK);
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    p.chunk = cdiv(cdiv(p.Mtot, nsplit), BK2) * BK2;
    nsplit = cdiv(p.Mtot, p.chunk);
    // the 8-wave pipelined kernel pays off when a block contracts many rows (GRU / Linear weight gradients);
    // short contractions into big outputs are bound by the merge atomics and keep the lighter 4-wave kernel
    const bool v2 = g->ksize == 1 && p.chunk >= 256;
    if (!accumulate) {
        hipError_t me = zero_async(dw, sizeof(float) * (size_t)g->Cout * g->Cin * g->ksize, (hipStream_t)stream);
        if (me != hipSuccess) return (int)me;
        if (dbias) {
            me = zero_async(dbias, sizeof(float) * (size_t)g->Cout, (hipStream_t)stream);
            if (me != hipSuccess) return (int)me;
        }
    }
    if (use_gemm_lin() && g->Cin == 1 &&
        s2ag_conv_c1_wgrad(gy, x, dw, dbias, g->N, g->Lin, g->Lout, g->Cin, g->Cout, g->ksize, g->stride, g->pad, g->dil,
                           g->ldx, g->ldy, (hipStream_t)stream)) {
        S2AG_LAUNCH_CHECK();
        return 0;
    }
    if (lin_ok && p.chunk % BK2 == 0 &&
        s2ag_wgrad_lin(gy, x, dw, dbias, g->N, g->Lin, g->Lout, g->Cin, g->Cout, g->ksize, g->stride, g->pad, g->dil,
                       g->ldx, g->ldy, g->w_tap_major, p.chunk, nsplit, (hipStream_t)stream)) {
        S2AG_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid(cdiv(g->Cout, BM), cdiv(g->ksize * g->Cin, BN), nsplit);
    if (v2)
        hipLaunchKernelGGL(conv_wgrad2_k, grid, dim3(512), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(conv_wgrad_k, grid, dim3(256), 0, (hipStream_t)stream, p);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_conv1d_nlc_bwd_weight_multi(const s2ag_wgrad_job* jobs, int njobs, void* stream) {
    if (!jobs || njobs < 1 || njobs > S2AG_MAX_WGRAD_JOBS) return S2AG_E_BADARG;
    if (!use_gemm_lin()) return S2AG_E_UNSUPPORTED;
    s2ag_wg_job_i jb[S2AG_MAX_WGRAD_JOBS];
    for (int i = 0; i < njobs; ++i) {
        const s2ag_conv_geom* g = &jobs[i].geom;
        if (bad_geom(g) || !jobs[i].gy || !jobs[i].x || !jobs[i].dw) return S2AG_E_BADARG;
        if (g->Cin == 1) return S2AG_E_UNSUPPORTED;
        const int Mtot = g->N * g->Lout;
        const int tiles = cdiv(g->Cout, BM) * cdiv(g->ksize * g->Cin, BN);
        int nsplit = cdiv(384, tiles);
        const int max_split = cdiv(Mtot, 4 * BK);
        if (nsplit > max_split) nsplit = max_split;
        if (nsplit < 1) nsplit = 1;
        const int chunk = cdiv(cdiv(Mtot, nsplit), BK2) * BK2;
        nsplit = cdiv(Mtot, chunk);
        jb[i] = s2ag_wg_job_i{jobs[i].gy, jobs[i].x, jobs[i].dw, jobs[i].dbias, g->N, g->Lin, g->Lout, g->Cin, g->Cout,
                              g->ksize, g->stride, g->pad, g->dil, g->ldx, g->ldy, g->w_tap_major, chunk, nsplit};
    }
    if (!s2ag_wgrad_multi(jb, njobs, (hipStream_t)stream)) return S2AG_E_UNSUPPORTED;
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_conv1d_nlc_bwd_pair(const float* gy, const float* w, const float* x, float* dx, float* dw,
                                        float* dbias, const s2ag_conv_geom* g, void* stream) {
    if (bad_geom(g) || !gy || !w || !x || !dx || !dw) return S2AG_E_BADARG;
    if (!use_gemm_lin() || g->stride != 1 || g->Lin != g->Lout) return S2AG_E_UNSUPPORTED;
    const int Mtot = g->N * g->Lout;
    const int tiles = cdiv(g->Cout, BM) * cdiv(g->ksize * g->Cin, BN);
    int nsplit = cdiv(384, tiles);                       // as s2ag_conv1d_nlc_bwd_weight on the straight-line path
    const int max_split = cdiv(Mtot, 4 * BK);
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    const int chunk = cdiv(cdiv(Mtot, nsplit), BK2) * BK2;
    nsplit = cdiv(Mtot, chunk);
    if (!s2ag_bwd_pair(gy, w, x, dx, dw, dbias, g->N, g->Lin, g->Cin, g->Cout, g->ksize, g->pad, g->dil, g->ldx, g->ldy,
                       g->w_tap_major, chunk, nsplit, (hipStream_t)stream))
        return S2AG_E_UNSUPPORTED;
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_colsum(const float* x, int rows, int cols, int ld, float* out, float* sq, int accumulate,
                           void* stream) {
    if (!x || !out || rows <= 0 || cols <= 0 || ld < cols) return S2AG_E_BADARG;
    if (!accumulate) {
        hipError_t me = zero_async(out, sizeof(float) * cols, (hipStream_t)stream);
        if (me != hipSuccess) return (int)me;
        if (sq) {
            me = zero_async(sq, sizeof(float) * cols, (hipStream_t)stream);
            if (me != hipSuccess) return (int)me;
        }
    }
    if (narrow_ok(cols, ld)) {
        const long long total = (long long)rows * cols;
        hipLaunchKernelGGL(colsum_flat_k, dim3(flat_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, total, cols, out, sq);
        S2AG_LAUNCH_CHECK();
        return 0;
    }
    int rpb = 256;
    const int colblocks = cdiv(cols, 64);
    // aim for >= ~1024 blocks on long matrices, but at least 64 rows each
    while (rpb > 64 && (long long)cdiv(rows, rpb) * colblocks < 1024) rpb >>= 1;
    dim3 grid(colblocks, cdiv(rows, rpb));
    hipLaunchKernelGGL(colsum_k, grid, dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, rpb, out, sq);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_colstats_f64(const float* x, int rows, int cols, int ld, double* sum, double* sq, void* stream) {
    if (!x || !sum || !sq || rows <= 0 || cols <= 0 || ld < cols) return S2AG_E_BADARG;
    hipError_t me;
    if (sq == sum + cols) {      // one contiguous (2, cols) buffer: a single clear
        me = zero_async(sum, sizeof(double) * 2 * cols, (hipStream_t)stream);
        if (me != hipSuccess) return (int)me;
    } else {
        me = zero_async(sum, sizeof(double) * cols, (hipStream_t)stream);
        if (me != hipSuccess) return (int)me;
        me = zero_async(sq, sizeof(double) * cols, (hipStream_t)stream);
        if (me != hipSuccess) return (int)me;
    }
    if (narrow_ok(cols, ld)) {
        const long long total = (long long)rows * cols;
        hipLaunchKernelGGL(colstats_flat_k, dim3(flat_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, total, cols, sum, sq);
        S2AG_LAUNCH_CHECK();
        return 0;
    }
    int rpb = 256;
    const int colblocks = cdiv(cols, 64);
    while (rpb > 64 && (long long)cdiv(rows, rpb) * colblocks < 1024) rpb >>= 1;
    dim3 grid(colblocks, cdiv(rows, rpb));
    hipLaunchKernelGGL(colstats_f64_k, grid, dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, rpb, sum, sq);
    S2AG_LAUNCH_CHECK();
    return 0;
}

#ifdef S2AG_GEMM_TRACE
extern "C" int s2ag_gemm_trace_read(unsigned long long* host8, int reset) {
    hipError_t e = hipMemcpyFromSymbol(host8, HIP_SYMBOL(g_gemm_trace), sizeof(unsigned long long) * 8);
    if (e == hipSuccess && reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace), z, sizeof(z));
    }
    return (int)e;
}
#endif
S2AG_DET_HOOK(conv_gemm)
