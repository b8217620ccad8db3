// Straight-line fp32-MFMA GEMM for the 1-tap layers (Linear, GRU input projections, 1x1 convs):
//   fwd  out[m, n] = act( sum_k a[m, k] * w[n, k] + bias[n] ) (x dropout)        a = x,  w = (N, K) row-major
//   bwd  out[m, n] (+)= sum_k a[m, k] * w[k, n]                                   a = gy, w = (K, N) row-major
//
// Same tiling and pipeline as conv_gemm2_k (conv_gemm.hip): BM x 64 block tile, K step 32, double-buffered row-major
// LDS with a 34-float pitch, two K tiles in flight in registers, 8 waves with a 2-way in-block K split.  What differs is
// what is NOT there: the general kernel resolves (tap, position, channel), padding, residue mode, weight layouts and
// dead-tile tests at run time inside the K loop -- hardware counters showed 14 vector-ALU and 8 scalar instructions per
// MFMA (SQ_INSTS_VALU / SQ_INSTS_MFMA), i.e. the loop was bound by instruction issue, not by the matrix pipe or memory.
// Here every operand chunk is one pointer that advances by a constant per tile, the only predicate in the loop is the
// K tail, and every MFMA is unconditional (rows / columns outside the matrix are zero in LDS; the epilogue masks them).
// Requirements (checked by the host, else the general kernel runs): K % 4 == 0, lda % 4 == 0, 16-byte aligned a and w,
// and for bwd N-contiguous W rows.
#include <stdlib.h>

#include "s2ag_common.h"

namespace {
using namespace s2ag;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int LBK = 32;          // K step
constexpr int LPK = LBK + 2;     // LDS row pitch (see conv_gemm.hip)

struct LinP {
    const float* a;
    const float* w;
    const float* bias;
    float* out;
    int M, K, N;
    int lda, ldo, ldw;           // ldw: fwd = K (row n of W), bwd = N (row k of W)
    // CONV (tap-major weights (Cout, ks, Cin), stride 1): K = ks * CK, the contraction index is (tap, channel)
    int CK;                      // channels per tap of the contraction (Cin fwd, Cout bwd), >= 32, % 4 == 0
    int ks, dil;
    int L;                       // output frames per clip
    int Ls;                      // source frames per clip (== L unless the forward conv is strided)
    int stride;                  // forward only (the data gradient of a strided conv stays with the general kernel)
    int off;                     // source frame of tap 0 for output frame 0: fwd -pad, bwd +pad (bwd steps by -dil)
    double* stats;               // fwd, nullable: per-row-block column sums of the output and of its square,
                                 // layout (2, gridDim.x, N) -- the BatchNorm that follows folds them (s2ag_bn_fold)
    int act;
    float slope, drop_p, inv_keep;
    const unsigned long long* rng;
    unsigned site;
    int accumulate;
};

template <bool BWD, int BM_, bool CONV>
__device__ __forceinline__ void gemm_lin_body(const LinP& p, float* smem, const int bx, const int by, const int gdx) {
    constexpr int NT = 512;
    constexpr int WM = 2, WN = 2, KG = 2;
    constexpr int TM = BM_ / (16 * WM), TN = 64 / (16 * WN);
    constexpr int CA = (BM_ * (LBK / 4)) / NT;                    // float4 chunks of A per thread (1 at BM=64)
    constexpr int KSG = (LBK / 4) / KG;
    static_assert(BM_ == 64 || BM_ == 32, "row tile");
    float (*As)[BM_][LPK] = reinterpret_cast<float (*)[BM_][LPK]>(smem);                       // [2][BM_][LPK]
    float (*Bs)[64][LPK] = reinterpret_cast<float (*)[64][LPK]>(smem + 2 * BM_ * LPK);          // [2][64][LPK]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / (WM * WN), wr = (wave % (WM * WN)) / WN, wc = wave % WN;
    const int m0 = bx * BM_, n0 = by * 64;

    // ---- loader coordinates: A chunk = (row a_r, k quad a_kq); at BM = 32 only the first 256 threads carry one
    const bool has_a = CA > 0 || tid < BM_ * (LBK / 4);
    const int a_kq = tid % (LBK / 4), a_r = tid / (LBK / 4);
    const float* a_ptr = (has_a && m0 + a_r < p.M) ? p.a + (long long)(m0 + a_r) * p.lda + a_kq * 4 : nullptr;
    // CONV: the A row of tap t is the clip's frame (l + off +- t*dil); (a_tap, a_c) = (tap, channel) of this thread's
    // chunk, advanced by LBK per tile (CK >= 32: at most one wrap per tile)
    int a_tap = 0, a_c = a_kq * 4, a_l = 0;
    if (CONV) while (a_c >= p.CK) { a_c -= p.CK; ++a_tap; }        // fewer than 32 channels per tap
    const float* a_clip = nullptr;              // row 0 of the clip
    if (CONV && has_a && m0 + a_r < p.M) {
        const int m = m0 + a_r;
        const int nclip = m / p.L;
        a_l = (m - nclip * p.L) * (BWD ? 1 : p.stride) + p.off;
        a_clip = p.a + (long long)nclip * p.Ls * p.lda;
    }
    // B chunk: fwd = (col b_c, k quad b_kq) one float4 along k;  bwd = (col b_c, k quad b_kq) four k rows of W
    int b_c, b_kq;
    if (BWD) {
        b_c = tid % 64;          // lanes along the contiguous N axis of W
        b_kq = tid / 64;
    } else {
        b_kq = tid % (LBK / 4);
        b_c = tid / (LBK / 4);
    }
    const bool b_ok = n0 + b_c < p.N;
    const float* b_ptr = !b_ok ? nullptr
                               : (BWD ? p.w + (long long)(b_kq * 4) * p.ldw + n0 + b_c
                                      : p.w + (long long)(n0 + b_c) * p.ldw + b_kq * 4);
    const long long b_step = BWD ? (long long)LBK * p.ldw : LBK;
    // CONV bwd: k = (tap, co) lives in W row co*ks + tap, i.e. rows of one tap are ks*Cin floats apart
    int b_tap = 0, b_co = b_kq * 4;
    if (CONV && BWD) while (b_co >= p.CK) { b_co -= p.CK; ++b_tap; }
    const long long b_row = (long long)p.ks * p.ldw;     // bwd CONV: distance of consecutive co at a fixed tap

    float4 ra0, rb0, ra1, rb1;
    auto fetch = [&](float4& ra, float4& rb, int k0) {        // call with k0 = 0, LBK, 2*LBK, ... in order
        const int ka = k0 + a_kq * 4, kb = k0 + b_kq * 4;
        ra = make_float4(0.f, 0.f, 0.f, 0.f);
        rb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (CONV) {
            if (a_clip && ka < p.K) {
                const int pos = BWD ? a_l - a_tap * p.dil : a_l + a_tap * p.dil;
                if ((unsigned)pos < (unsigned)p.Ls)
                    ra = *reinterpret_cast<const float4*>(a_clip + (long long)pos * p.lda + a_c);
            }
            a_c += LBK;
            while (a_c >= p.CK) { a_c -= p.CK; ++a_tap; }
        } else {
            if (a_ptr && ka < p.K) ra = *reinterpret_cast<const float4*>(a_ptr + k0);
        }
        if (BWD) {
            if (CONV) {
                if (b_ptr && kb < p.K) {          // CK % 4 == 0: the four k of a chunk share the tap
                    const float* q = p.w + ((long long)b_co * p.ks + b_tap) * p.ldw + n0 + b_c;
                    rb.x = q[0];
                    rb.y = q[b_row];
                    rb.z = q[2 * b_row];
                    rb.w = q[3 * b_row];
                }
                b_co += LBK;
                while (b_co >= p.CK) { b_co -= p.CK; ++b_tap; }
            } else if (b_ptr) {
                const float* q = b_ptr + (long long)(k0 / LBK) * b_step;
                if (kb < p.K) rb.x = q[0];
                if (kb + 1 < p.K) rb.y = q[p.ldw];
                if (kb + 2 < p.K) rb.z = q[2 * (long long)p.ldw];
                if (kb + 3 < p.K) rb.w = q[3 * (long long)p.ldw];
            }
        } else {
            if (b_ptr && kb < p.K) rb = *reinterpret_cast<const float4*>(b_ptr + k0);
        }
    };
    float* const a_dst = &As[0][has_a ? a_r : 0][a_kq * 4];
    float* const b_dst = &Bs[0][b_c][b_kq * 4];
    auto stash = [&](const float4& ra, const float4& rb, int buf) {
        if (has_a) {
            float* d = a_dst + buf * (BM_ * LPK);
            *reinterpret_cast<float2*>(d) = make_float2(ra.x, ra.y);
            *reinterpret_cast<float2*>(d + 2) = make_float2(ra.z, ra.w);
        }
        float* d = b_dst + buf * (64 * LPK);
        *reinterpret_cast<float2*>(d) = make_float2(rb.x, rb.y);
        *reinterpret_cast<float2*>(d + 2) = make_float2(rb.z, rb.w);
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int li = lane & 15;
    const float* a_frag = &As[0][wr * TM * 16 + li][kg * KSG * 4 + (lane >> 4)];
    const float* b_frag = &Bs[0][wc * TN * 16 + li][kg * KSG * 4 + (lane >> 4)];
    auto mma = [&](int cur) {
        const float* af = a_frag + cur * (BM_ * LPK);
        const float* bf = b_frag + cur * (64 * LPK);
        float a[KSG][TM], b[KSG][TN];
#pragma unroll
        for (int s = 0; s < KSG; ++s) {
#pragma unroll
            for (int t = 0; t < TM; ++t) a[s][t] = af[t * 16 * LPK + s * 4];
#pragma unroll
            for (int t = 0; t < TN; ++t) b[s][t] = bf[t * 16 * LPK + s * 4];
        }
#pragma unroll
        for (int s = 0; s < KSG; ++s)
#pragma unroll
            for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                for (int tj = 0; tj < TN; ++tj)
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][ti], b[s][tj], acc[ti][tj], 0, 0, 0);
    };

    const int nkt = (p.K + LBK - 1) / LBK;
    fetch(ra0, rb0, 0);
    stash(ra0, rb0, 0);
    if (nkt > 1) fetch(ra0, rb0, LBK);
    __syncthreads();
    for (int kt = 0; kt < nkt; kt += 2) {
        if (kt + 2 < nkt) fetch(ra1, rb1, (kt + 2) * LBK);
        mma(0);
        if (kt + 1 < nkt) stash(ra0, rb0, 1);
        __syncthreads();
        if (kt + 1 >= nkt) break;
        if (kt + 3 < nkt) fetch(ra0, rb0, (kt + 3) * LBK);
        mma(1);
        if (kt + 2 < nkt) stash(ra1, rb1, 0);
        __syncthreads();
    }

    // sum the two K halves: group 1 parks its accumulators in LDS, group 0 finishes
    {
        float* red = &As[0][0][0];
        static_assert(2 * BM_ * LPK >= BM_ * 64, "reduction scratch must fit in As");
        if (kg == 1) {
#pragma unroll
            for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                for (int tj = 0; tj < TN; ++tj)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        red[((wr * TM + ti) * 16 + (lane >> 4) * 4 + q) * 64 + (wc * TN + tj) * 16 + li] = acc[ti][tj][q];
        }
        __syncthreads();
        if (kg == 1) return;
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
            for (int tj = 0; tj < TN; ++tj)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[ti][tj][q] += red[((wr * TM + ti) * 16 + (lane >> 4) * 4 + q) * 64 + (wc * TN + tj) * 16 + li];
    }

    SiteKey key{0, 0};
    const bool drop = (!BWD) && p.drop_p > 0.f;
    if (drop) key = site_key(p.rng, p.site);
    const bool want_stats = (!BWD) && p.stats != nullptr;       // block-uniform
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < TN; ++tj) {
            const int c = n0 + (wc * TN + tj) * 16 + li;
            if (c >= p.N) continue;
            const float bias = (!BWD && p.bias) ? p.bias[c] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = m0 + (wr * TM + ti) * 16 + (lane >> 4) * 4 + q;
                if (row >= p.M) continue;
                float v = acc[ti][tj][q];
                float* dst = p.out + (long long)row * p.ldo + c;
                if (!BWD) {
                    v = apply_act(v + bias, p.act, p.slope);
                    if (drop) v *= keep_scale(key, (unsigned long long)row * p.N + c, p.drop_p, p.inv_keep);
                    *dst = v;
                    acc[ti][tj][q] = v;                          // what the statistics are taken of
                } else {
                    *dst = p.accumulate ? (*dst + v) : v;
                }
            }
        }
    if (want_stats) {
        // Column sums of this wave's output rows (fp64 from the first add: E[x^2] - E[x]^2 cancels in fp32) for the
        // BatchNorm that follows the layer -- saves it a pass over the matrix.  One partial row per (row block, row
        // wave); lanes that share a column (lane >> 4 = 0..3) meet in two DPP steps.
        const size_t R = (size_t)gdx * WM, r = (size_t)bx * WM + wr;
#pragma unroll
        for (int tj = 0; tj < TN; ++tj) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = m0 + (wr * TM + ti) * 16 + (lane >> 4) * 4 + q;
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
            const int c = n0 + (wc * TN + tj) * 16 + lane;
            if (lane < 16 && c < p.N) {
                p.stats[r * p.N + c] = s1;
                p.stats[(R + r) * p.N + c] = s2;
            }
        }
    }
}

constexpr int gemm_lin_smem_floats(int bm) { return 2 * LPK * (bm + 64); }

template <bool BWD, int BM_, bool CONV>
__global__ __launch_bounds__(512) void gemm_lin_k(LinP p) {
    __shared__ __attribute__((aligned(16))) float smem[gemm_lin_smem_floats(BM_)];
    gemm_lin_body<BWD, BM_, CONV>(p, smem, blockIdx.x, blockIdx.y, gridDim.x);
}

inline bool al16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }
inline long long bm64_min_blocks() {
    return 512LL;
}
}  // namespace

// Internal entry points used by s2ag_conv1d_nlc_fwd / s2ag_conv1d_nlc_bwd_data (conv_gemm.hip) for 1-tap geometries.
// Return 1 if the launch was taken, 0 if the shape / alignment is not covered (caller falls back to the general kernel).
// `stats` (nullable): (2, rows, N) doubles, rows = the return value: per-partial-row column sums of y and y^2.
int s2ag_gemm_lin_fwd(const float* x, const float* w, const float* bias, float* y, int M, int K, int N, int ldx, int ldy,
                      int act, float slope, float drop_p, const unsigned long long* rng, unsigned site, double* stats,
                      hipStream_t stream) {
    if ((K & 3) || (ldx & 3) || !al16(x) || !al16(w)) return 0;
    LinP p{};
    p.a = x; p.w = w; p.bias = bias; p.out = y; p.M = M; p.K = K; p.N = N; p.lda = ldx; p.ldo = ldy; p.ldw = K;
    p.act = act; p.slope = slope; p.drop_p = drop_p; p.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    p.rng = rng; p.site = site; p.accumulate = 0; p.stats = stats;
    const int colb = cdiv(N, 64);
    if ((long long)cdiv(M, 64) * colb >= bm64_min_blocks()) {
        hipLaunchKernelGGL((gemm_lin_k<false, 64, false>), dim3(cdiv(M, 64), colb), dim3(512), 0, stream, p);
        return 2 * cdiv(M, 64);
    }
    hipLaunchKernelGGL((gemm_lin_k<false, 32, false>), dim3(cdiv(M, 32), colb), dim3(512), 0, stream, p);
    return 2 * cdiv(M, 32);
}

int s2ag_gemm_lin_bwd_data(const float* gy, const float* w, float* dx, int M, int Cout, int Cin, int ldg, int ldx,
                           int accumulate, hipStream_t stream) {
    // dx[m, ci] = sum_co gy[m, co] * w[co, ci]: K = Cout, N = Cin, W rows are N-contiguous
    if ((Cout & 3) || (ldg & 3) || !al16(gy)) return 0;
    LinP p{};
    p.a = gy; p.w = w; p.bias = nullptr; p.out = dx; p.M = M; p.K = Cout; p.N = Cin; p.lda = ldg; p.ldo = ldx; p.ldw = Cin;
    p.act = 0; p.slope = 1.f; p.drop_p = 0.f; p.inv_keep = 1.f; p.rng = nullptr; p.site = 0; p.accumulate = accumulate;
    const int colb = cdiv(Cin, 64);
    if ((long long)cdiv(M, 64) * colb >= bm64_min_blocks())
        hipLaunchKernelGGL((gemm_lin_k<true, 64, false>), dim3(cdiv(M, 64), colb), dim3(512), 0, stream, p);
    else
        hipLaunchKernelGGL((gemm_lin_k<true, 32, false>), dim3(cdiv(M, 32), colb), dim3(512), 0, stream, p);
    return 1;
}

// Convolutions with TAP-MAJOR weights (Cout, ks, Cin) -- the weight-normed TCN convs, the folded ST-GCN convs and the
// tap-major copies of the wave / MFCC encoder weights (all produced by derived-parameter kernels), Cin % 4 == 0:
//   fwd  y[(n,l), co]   = epi( sum_{t,ci} x[(n, l*stride - pad + t*dil), ci] w[co, t, ci] + b[co] )      (any stride)
//   bwd  dx[(n,p), ci] (+)= sum_{t,co} gy[(n, p + pad - t*dil), co] w[co, t, ci]
int s2ag_gemm_conv_tm_fwd(const float* x, const float* w, const float* bias, float* y, int nclips, int Lin, int Lout,
                          int Cin, int Cout, int ks, int stride, int pad, int dil, int ldx, int ldy, int act, float slope,
                          float drop_p, const unsigned long long* rng, unsigned site, double* stats, hipStream_t stream) {
    if ((Cin & 3) || (ldx & 3) || !al16(x) || !al16(w)) return 0;
    LinP p{};
    p.a = x; p.w = w; p.bias = bias; p.out = y; p.M = nclips * Lout; p.K = ks * Cin; p.N = Cout;
    p.lda = ldx; p.ldo = ldy; p.ldw = ks * Cin;
    p.CK = Cin; p.ks = ks; p.dil = dil; p.L = Lout; p.Ls = Lin; p.stride = stride; p.off = -pad;
    p.act = act; p.slope = slope; p.drop_p = drop_p; p.inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    p.rng = rng; p.site = site; p.accumulate = 0; p.stats = stats;
    const int colb = cdiv(Cout, 64);
    if ((long long)cdiv(p.M, 64) * colb >= bm64_min_blocks()) {
        hipLaunchKernelGGL((gemm_lin_k<false, 64, true>), dim3(cdiv(p.M, 64), colb), dim3(512), 0, stream, p);
        return 2 * cdiv(p.M, 64);
    }
    hipLaunchKernelGGL((gemm_lin_k<false, 32, true>), dim3(cdiv(p.M, 32), colb), dim3(512), 0, stream, p);
    return 2 * cdiv(p.M, 32);
}

int s2ag_gemm_conv_tm_bwd_data(const float* gy, const float* w, float* dx, int nclips, int L, int Cin, int Cout, int ks,
                               int pad, int dil, int ldg, int ldx, int accumulate, hipStream_t stream) {
    if ((Cout & 3) || (ldg & 3) || !al16(gy)) return 0;
    LinP p{};
    p.a = gy; p.w = w; p.bias = nullptr; p.out = dx; p.M = nclips * L; p.K = ks * Cout; p.N = Cin;
    p.lda = ldg; p.ldo = ldx; p.ldw = Cin;
    p.CK = Cout; p.ks = ks; p.dil = dil; p.L = L; p.Ls = L; p.stride = 1; p.off = pad;
    p.act = 0; p.slope = 1.f; p.drop_p = 0.f; p.inv_keep = 1.f; p.rng = nullptr; p.site = 0; p.accumulate = accumulate;
    const int colb = cdiv(Cin, 64);
    if ((long long)cdiv(p.M, 64) * colb >= bm64_min_blocks())
        hipLaunchKernelGGL((gemm_lin_k<true, 64, true>), dim3(cdiv(p.M, 64), colb), dim3(512), 0, stream, p);
    else
        hipLaunchKernelGGL((gemm_lin_k<true, 32, true>), dim3(cdiv(p.M, 32), colb), dim3(512), 0, stream, p);
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// Straight-line weight gradient:  dw[co, (tap, ci)] += sum_{n,l} gy[(n,l), co] * x[(n, l*stride + tap*dil - pad), ci]
// (frames of one clip only), db[co] += sum_m gy[m, co].  Same tiling as conv_wgrad2_k (64 x 64 output tile, 32 contraction
// rows per K tile, double-buffered k-major LDS, 8 waves with a 2-way in-block split, clips*frames split over grid.z and
// merged with fp32 atomics) without the per-row clip/position arithmetic of the general kernel (10-15 vector-ALU
// instructions per MFMA there): a thread owns one output column for the whole launch, so its operand rows are two
// pointers that advance by 32 rows per tile; the only per-row work is the frame-boundary test of shifted taps.
namespace {
struct WgP {
    const float* gy;
    const float* x;
    float* dw;
    float* db;
    int M, L, Ls, stride, Cin, Cout, ks, pad, dil, ldx, ldg, wtm;     // L: output frames per clip, Ls: source frames
    int chunk;
};

constexpr int WPW = 64 + 16;          // k-major LDS pitch (fragment reads of 4 k-rows hit disjoint banks)

constexpr int WG_SMEM_FLOATS = 4 * LBK * WPW;

template <bool SHIFT>
__device__ __forceinline__ void wgrad_lin_body(const WgP& p, float* smem, const int bx, const int by, const int bz) {
    float (*As)[LBK][WPW] = reinterpret_cast<float (*)[LBK][WPW]>(smem);                        // [2][m][co]
    float (*Bs)[LBK][WPW] = reinterpret_cast<float (*)[LBK][WPW]>(smem + 2 * LBK * WPW);        // [2][m][j = tap*Cin + ci]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int co0 = bx * 64, j0 = by * 64;
    const int mbeg = bz * p.chunk;
    const int mend = min(p.M, mbeg + p.chunk);
    const int cidx = tid & 63, mq = tid >> 6;         // rows mq*4 .. mq*4+3 of the 32-row tile
    const int NCW = p.ks * p.Cin;
    const int co = co0 + cidx, jcol = j0 + cidx;
    int tap = 0, ci = 0;
    if (jcol < NCW) {
        tap = jcol / p.Cin;
        ci = jcol - tap * p.Cin;
    }
    const int tapoff = tap * p.dil - p.pad;
    const long long r0 = mbeg + mq * 4;
    const float* g_ptr = co < p.Cout ? p.gy + r0 * p.ldg + co : nullptr;
    // linear mode: the x row of output row m is m itself.  SHIFT mode: row m = (clip n, frame l) reads source frame
    // l*stride + tapoff of clip n (zero outside the clip): x_ptr is the clip's frame 0, advanced at clip boundaries
    int l = SHIFT ? (int)(r0 % p.L) : 0;              // frame of this thread's first row inside its clip
    const float* x_ptr = jcol < NCW ? (SHIFT ? p.x + (r0 / p.L) * (long long)p.Ls * p.ldx + ci : p.x + r0 * p.ldx + ci)
                                    : nullptr;
    const bool want_db = p.db != nullptr && by == 0;
    float bsum = 0.f;

    float ra0[4], rb0[4], ra1[4], rb1[4];
    auto fetch = [&](float (&ra)[4], float (&rb)[4], int mb) {       // call with mb = mbeg, mbeg+32, ... in order
        const bool full = mb + LBK <= mend;           // block-uniform: only the last tile of the last chunk is ragged
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool okrow = full || (mb + mq * 4 + j < mend);
            ra[j] = (g_ptr && okrow) ? g_ptr[(long long)j * p.ldg] : 0.f;
            if (SHIFT) {
                int lj = l + j;
                const float* xc = x_ptr;
                if (lj >= p.L) {                      // this row already belongs to the next clip
                    lj -= p.L;
                    xc += (long long)p.Ls * p.ldx;
                }
                const int pos = lj * p.stride + tapoff;
                rb[j] = (x_ptr && okrow && (unsigned)pos < (unsigned)p.Ls) ? xc[(long long)pos * p.ldx] : 0.f;
            } else {
                rb[j] = (x_ptr && okrow) ? x_ptr[(long long)j * p.ldx] : 0.f;
            }
            bsum += ra[j];
        }
        if (g_ptr) g_ptr += (long long)LBK * p.ldg;
        if (SHIFT) {
            l += LBK;
            if (l >= p.L) {                           // host guarantees L >= 32: at most one clip boundary per tile
                l -= p.L;
                if (x_ptr) x_ptr += (long long)p.Ls * p.ldx;
            }
        } else if (x_ptr) {
            x_ptr += (long long)LBK * p.ldx;
        }
    };
    float* const a_dst = &As[0][mq * 4][cidx];
    float* const b_dst = &Bs[0][mq * 4][cidx];
    auto stash = [&](const float (&ra)[4], const float (&rb)[4], int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a_dst[buf * (LBK * WPW) + j * WPW] = ra[j];
            b_dst[buf * (LBK * WPW) + j * WPW] = rb[j];
        }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15;
    const float* a_frag = &As[0][kg * 16 + (lane >> 4)][wm * 32 + li];
    const float* b_frag = &Bs[0][kg * 16 + (lane >> 4)][wn * 32 + li];
    auto mma = [&](int cur) {                          // every MFMA unconditional: rows / columns outside hold zeros
        const float* af = a_frag + cur * (LBK * WPW);
        const float* bf = b_frag + cur * (LBK * WPW);
        float a0[4], a1[4], b0[4], b1[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a0[s] = af[s * 4 * WPW];
            a1[s] = af[s * 4 * WPW + 16];
            b0[s] = bf[s * 4 * WPW];
            b1[s] = bf[s * 4 * WPW + 16];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b0[s], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b1[s], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b0[s], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b1[s], acc[1][1], 0, 0, 0);
        }
    };
    const int nkt = (mend - mbeg + LBK - 1) / LBK;
    if (nkt <= 0) {
        s2ag::det_enter();                            // deterministic mode: an empty slice still takes and passes on its turn
        s2ag::det_leave();
        return;
    }
    fetch(ra0, rb0, mbeg);
    stash(ra0, rb0, 0);
    if (nkt > 1) fetch(ra0, rb0, mbeg + LBK);
    __syncthreads();
    for (int kt = 0; kt < nkt; kt += 2) {
        if (kt + 2 < nkt) fetch(ra1, rb1, mbeg + (kt + 2) * LBK);
        mma(0);
        if (kt + 1 < nkt) stash(ra0, rb0, 1);
        __syncthreads();
        if (kt + 1 >= nkt) break;
        if (kt + 3 < nkt) fetch(ra0, rb0, mbeg + (kt + 3) * LBK);
        mma(1);
        if (kt + 2 < nkt) stash(ra1, rb1, 0);
        __syncthreads();
    }
    if (want_db) Bs[0][mq][cidx] = bsum;              // Bs is idle after the last tile; read behind the barrier below
    {   // merge the two wave groups through LDS so only one of them issues the (cross-block) atomics
        float* redw = &As[0][0][0];                   // 2*32*80 floats >= 64*64
        if (kg == 1) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        redw[(wm * 32 + ti * 16 + (lane >> 4) * 4 + q) * 64 + wn * 32 + tj * 16 + li] = acc[ti][tj][q];
        }
        __syncthreads();
        s2ag::det_enter();                            // deterministic mode: workgroups add in index order (all eight waves here)
        if (want_db && mq == 0 && co < p.Cout) {      // wave 0 (kg = 0): eight row groups per column
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
                    acc[ti][tj][q] += redw[(wm * 32 + ti * 16 + (lane >> 4) * 4 + q) * 64 + wn * 32 + tj * 16 + li];
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int jc = j0 + wn * 32 + tj * 16 + li;
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

template <bool SHIFT>
__global__ __launch_bounds__(512) void wgrad_lin_k(WgP p) {
    __shared__ __attribute__((aligned(16))) float smem[WG_SMEM_FLOATS];
    wgrad_lin_body<SHIFT>(p, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

// One launch for a layer's two backward GEMMs (both read gy): blocks [0, n2) run the weight-gradient tiles, the rest the
// data-gradient tiles.  At M = clips*frames = 4 352 either kernel alone leaves CUs idle (400 / 680 blocks of uneven
// length over 256 CUs) and the backward pass of the TCN / ST-GCN stacks is a serial chain of such 25-40 us launches.
template <int BM_, bool CONV, bool SHIFT>
__global__ __launch_bounds__(512) void bwd_pair_k(LinP p1, WgP p2, int n2, int g2x, int g2y, int g1x) {
    constexpr int SM = gemm_lin_smem_floats(BM_) > WG_SMEM_FLOATS ? gemm_lin_smem_floats(BM_) : WG_SMEM_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[SM];
    const int b = blockIdx.x;
    if (b < n2) {
        const int x = b % g2x, r = b / g2x;
        wgrad_lin_body<SHIFT>(p2, smem, x, r % g2y, r / g2y);
    } else {
        const int b1 = b - n2;
        gemm_lin_body<true, BM_, CONV>(p1, smem, b1 % g1x, b1 / g1x, g1x);
        s2ag::det_enter();                            // the data-gradient tiles accumulate nothing: they only pass the turn on
        s2ag::det_leave();
    }
}

// Several weight gradients in ONE launch (a GRU layer's dW_ih and the two directions' dW_hh: 450-580 blocks each, every
// one of them alone leaves most of the 1 024 block slots of the chip empty)
constexpr int WG_MAX_JOBS = 4;
struct WgJobs {
    WgP p[WG_MAX_JOBS];
    int end[WG_MAX_JOBS];            // exclusive prefix sums of the jobs' block counts
    int gx[WG_MAX_JOBS], gy[WG_MAX_JOBS];
    int shift[WG_MAX_JOBS];
    int n;
};
__global__ __launch_bounds__(512) void wgrad_multi_k(WgJobs jobs) {
    __shared__ __attribute__((aligned(16))) float smem[WG_SMEM_FLOATS];
    const int b = blockIdx.x;
    int j = 0;
    while (j + 1 < jobs.n && b >= jobs.end[j]) ++j;
    const int b0 = b - (j ? jobs.end[j - 1] : 0);
    const int x = b0 % jobs.gx[j], r = b0 / jobs.gx[j];
    if (jobs.shift[j])
        wgrad_lin_body<true>(jobs.p[j], smem, x, r % jobs.gy[j], r / jobs.gy[j]);
    else
        wgrad_lin_body<false>(jobs.p[j], smem, x, r % jobs.gy[j], r / jobs.gy[j]);
}

template <int BM_, bool CONV>
void launch_pair(const LinP& p1, const WgP& p2, bool shift, dim3 g1, dim3 g2, hipStream_t stream) {
    const int n2 = g2.x * g2.y * g2.z, n1 = g1.x * g1.y;
    if (shift)
        hipLaunchKernelGGL((bwd_pair_k<BM_, CONV, true>), dim3(n1 + n2), dim3(512), 0, stream, p1, p2, n2, (int)g2.x,
                           (int)g2.y, (int)g1.x);
    else
        hipLaunchKernelGGL((bwd_pair_k<BM_, CONV, false>), dim3(n1 + n2), dim3(512), 0, stream, p1, p2, n2, (int)g2.x,
                           (int)g2.y, (int)g1.x);
}
}  // namespace

// Returns 1 if the launch was taken (one frame per clip without padding -- Linear -- or at least 32 output frames per
// clip, any stride / padding / dilation); dw / db must already hold the values to accumulate into.
int s2ag_wgrad_lin(const float* gy, const float* x, float* dw, float* db, int nclips, int Lin, int Lout, int Cin,
                   int Cout, int ks, int stride, int pad, int dil, int ldx, int ldg, int wtm, int chunk, int nsplit,
                   hipStream_t stream) {
    const bool linear = (Lin == 1 && Lout == 1 && pad == 0 && ks == 1 && stride == 1);
    if (!linear && Lout < LBK) return 0;
    WgP p{};
    p.gy = gy; p.x = x; p.dw = dw; p.db = db; p.M = nclips * Lout; p.L = Lout; p.Ls = Lin; p.stride = stride;
    p.Cin = Cin; p.Cout = Cout; p.ks = ks;
    p.pad = pad; p.dil = dil; p.ldx = ldx; p.ldg = ldg; p.wtm = wtm; p.chunk = chunk;
    dim3 grid(cdiv(Cout, 64), cdiv(ks * Cin, 64), nsplit);
    if (linear)
        hipLaunchKernelGGL(wgrad_lin_k<false>, grid, dim3(512), 0, stream, p);
    else
        hipLaunchKernelGGL(wgrad_lin_k<true>, grid, dim3(512), 0, stream, p);
    return 1;
}

// Data gradient + weight (+ bias) gradient of one stride-1 layer in ONE launch (bwd_pair_k).  Returns 1 if taken: the
// geometry must be covered by both straight-line kernels (1-tap layer, or tap-major weights with Lin == Lout).
int s2ag_bwd_pair(const float* gy, const float* w, const float* x, float* dx, float* dw, float* db, int nclips, int L,
                  int Cin, int Cout, int ks, int pad, int dil, int ldx, int ldg, int wtm, int chunk, int nsplit,
                  hipStream_t stream) {
    if ((Cout & 3) || (ldg & 3) || !al16(gy)) return 0;
    const bool one_tap = (ks == 1 && pad == 0);
    if (!one_tap && !wtm) return 0;
    const bool linear = (L == 1 && one_tap);
    if (!linear && L < LBK) return 0;
    LinP p1{};
    p1.a = gy; p1.w = w; p1.bias = nullptr; p1.out = dx; p1.M = nclips * L; p1.K = ks * Cout; p1.N = Cin;
    p1.lda = ldg; p1.ldo = ldx; p1.ldw = Cin;
    p1.CK = Cout; p1.ks = ks; p1.dil = dil; p1.L = L; p1.Ls = L; p1.stride = 1; p1.off = pad;
    p1.act = 0; p1.slope = 1.f; p1.drop_p = 0.f; p1.inv_keep = 1.f; p1.rng = nullptr; p1.site = 0; p1.accumulate = 0;
    WgP p2{};
    p2.gy = gy; p2.x = x; p2.dw = dw; p2.db = db; p2.M = nclips * L; p2.L = L; p2.Ls = L; p2.stride = 1;
    p2.Cin = Cin; p2.Cout = Cout; p2.ks = ks;
    p2.pad = pad; p2.dil = dil; p2.ldx = ldx; p2.ldg = ldg; p2.wtm = wtm; p2.chunk = chunk;
    const dim3 g2(cdiv(Cout, 64), cdiv(ks * Cin, 64), nsplit);
    const int colb = cdiv(Cin, 64);
    const bool bm64 = (long long)cdiv(p1.M, 64) * colb >= bm64_min_blocks();
    const dim3 g1(cdiv(p1.M, bm64 ? 64 : 32), colb);
    if (one_tap) {
        if (bm64) launch_pair<64, false>(p1, p2, !linear, g1, g2, stream);
        else launch_pair<32, false>(p1, p2, !linear, g1, g2, stream);
    } else {
        if (bm64) launch_pair<64, true>(p1, p2, true, g1, g2, stream);
        else launch_pair<32, true>(p1, p2, true, g1, g2, stream);
    }
    return 1;
}

// Up to 4 accumulating weight (+ bias) gradients in one launch; every job must be covered by the straight-line kernel
// (see s2ag_wgrad_lin).  Returns 1 if launched.
struct s2ag_wg_job_i {
    const float* gy;
    const float* x;
    float* dw;
    float* db;
    int nclips, Lin, Lout, Cin, Cout, ks, stride, pad, dil, ldx, ldg, wtm, chunk, nsplit;
};
int s2ag_wgrad_multi(const s2ag_wg_job_i* jb, int n, hipStream_t stream) {
    if (n < 1 || n > WG_MAX_JOBS) return 0;
    WgJobs J{};
    J.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const s2ag_wg_job_i& q = jb[i];
        const bool linear = (q.Lin == 1 && q.Lout == 1 && q.pad == 0 && q.ks == 1 && q.stride == 1);
        if (!linear && q.Lout < LBK) return 0;
        WgP& p = J.p[i];
        p.gy = q.gy; p.x = q.x; p.dw = q.dw; p.db = q.db; p.M = q.nclips * q.Lout; p.L = q.Lout; p.Ls = q.Lin;
        p.stride = q.stride; p.Cin = q.Cin; p.Cout = q.Cout; p.ks = q.ks; p.pad = q.pad; p.dil = q.dil; p.ldx = q.ldx;
        p.ldg = q.ldg; p.wtm = q.wtm; p.chunk = q.chunk;
        J.gx[i] = cdiv(q.Cout, 64);
        J.gy[i] = cdiv(q.ks * q.Cin, 64);
        J.shift[i] = linear ? 0 : 1;
        total += J.gx[i] * J.gy[i] * q.nsplit;
        J.end[i] = total;
    }
    hipLaunchKernelGGL(wgrad_multi_k, dim3(total), dim3(512), 0, stream, J);
    return 1;
}
S2AG_DET_HOOK(gemm_lin)
