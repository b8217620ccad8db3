// BatchNorm folds done by "the workgroup that finishes last" (shared by wave_fused.hip and conv_c1.hip): every workgroup
// publishes its partial column sums with agent-scope stores, takes a ticket, and the one that draws the last ticket turns the
// (2, R, C) partials into coefficients -- no launch of its own, no host round trip.
#pragma once
#include "s2ag_common.h"

namespace s2ag_fold {

struct FwdFold {                  // forward: batch statistics -> running estimates + scale / shift / mean / invstd
    int* ticket;                  // null: no in-kernel fold (the caller launches s2ag_bn_fold)
    const float* gamma;
    const float* beta;
    float* rmean;
    float* rvar;
    long long* nbt;
    float eps, momentum;
    int repeat;
    long long rows;               // rows of the normalised tensor (clips * frames)
    float* scale;
    float* shift;
    float* mean;
    float* invstd;
};

__device__ __forceinline__ void st_agent(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every workgroup has published its partial sums with st_agent; the one that draws the last ticket re-arms the ticket word
// (the next launch on it needs no clearing kernel) and returns true
__device__ __forceinline__ bool last_block_done(int* ticket, int nblocks) {
    __shared__ int is_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (t == nblocks - 1);
        if (is_last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return is_last != 0;
}

// Two-level form for hundreds of partial rows (one workgroup reading 2 x 512 x 64 doubles with L1-bypassing loads is a 10 us
// serial tail): groups of FOLD_GROUP consecutive workgroups have a ticket each; the last finisher of a group sums the group's
// rows (fixed order: reproducible) into one group row, then takes the global ticket; the workgroup that draws the last
// global ticket returns true and folds the <= R / 16 group rows.  tickets: 1 + cdiv(R, 16) zero words (left zero);
// part: (2, R, C) followed by the group rows (2, cdiv(R, 16), C).
constexpr int FOLD_GROUP = 16;
__host__ __device__ inline int fold_groups(int R) { return (R + FOLD_GROUP - 1) / FOLD_GROUP; }

__device__ __forceinline__ bool two_level_done(double* part, int R, int C, int* tickets) {
    const int grp = blockIdx.x / FOLD_GROUP, ng = fold_groups(R);
    const int gsize = min(FOLD_GROUP, R - grp * FOLD_GROUP);
    if (!last_block_done(tickets + 1 + grp, gsize)) return false;
    double* grows = part + (size_t)2 * R * C;
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        const int which = i / C, c = i - which * C;
        double v[FOLD_GROUP];
#pragma unroll
        for (int j = 0; j < FOLD_GROUP; ++j)
            v[j] = ld_agent(part + ((size_t)which * R + grp * FOLD_GROUP + min(j, gsize - 1)) * C + c);
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < FOLD_GROUP; ++j)
            if (j < gsize) s += v[j];
        st_agent(grows + ((size_t)which * ng + grp) * C + c, s);
    }
    return last_block_done(tickets, ng);
}

// The fold of the BatchNorm backward (256 threads): partial column sums of dz and dz * xhat (2, R, C) -> gradients of
// gamma / beta (added to their slots) and the coefficients of dy = A dz + C y + B.  16 partial rows in flight per thread.
template <bool AGENT>
__device__ __forceinline__ void bn_bwd_fold_body(const double* part, int R, int C, double inv_rows, const float* gamma,
                                                 const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                                 float* ca, float* cb, float* cc, double (*red)[256]) {
    const int per = 256 / C;                      // threads per column (C <= 64, power of two)
    const int c = threadIdx.x % C, k = threadIdx.x / C;
    double a = 0.0, b = 0.0;
    for (int r = k; r < R; r += 16 * per) {
        double va[16], vb[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int rr = min(r + j * per, R - 1);
            va[j] = AGENT ? ld_agent(part + (size_t)rr * C + c) : part[(size_t)rr * C + c];
            vb[j] = AGENT ? ld_agent(part + ((size_t)R + rr) * C + c) : part[((size_t)R + rr) * C + c];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (r + j * per < R) {
                a += va[j];
                b += vb[j];
            }
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x < C) {
        for (int j = 1; j < per; ++j) {
            a += red[0][j * C + c];
            b += red[1][j * C + c];
        }
        if (dbeta) atomicAdd(dbeta + c, (float)a);
        if (dgamma) atomicAdd(dgamma + c, (float)b);
        const double m1 = a * inv_rows, m2 = b * inv_rows;
        const double g = gamma[c], r = invstd[c], mu = mean[c];
        ca[c] = (float)(g * r);
        cc[c] = (float)(-g * r * r * m2);
        cb[c] = (float)(g * r * (r * mu * m2 - m1));
    }
}


// forward fold (256 threads): (2, R, C) partial sums of y and y^2 -> the arithmetic of norm_elementwise.hip's
// bn_finish_coeffs (fp64 statistics, `repeat` running-estimate updates rounded to fp32 each, fp32 scale / shift)
__device__ __forceinline__ void bn_fwd_fold_body(const double* part, int R, int C, const FwdFold& f, double (*red)[256]) {
    const int per = 256 / C;
    const int c = threadIdx.x % C, k = threadIdx.x / C;
    double a = 0.0, b = 0.0;
    for (int r = k; r < R; r += 16 * per) {
        double va[16], vb[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int rr = min(r + j * per, R - 1);
            va[j] = ld_agent(part + (size_t)rr * C + c);
            vb[j] = ld_agent(part + ((size_t)R + rr) * C + c);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (r + j * per < R) {
                a += va[j];
                b += vb[j];
            }
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x < C) {
        for (int j = 1; j < per; ++j) {
            a += red[0][j * C + c];
            b += red[1][j * C + c];
        }
        const double n = (double)f.rows;
        const double mean = a / n;
        double var = b / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
        float rm = f.rmean[c], rv = f.rvar[c];
        for (int it = 0; it < f.repeat; ++it) {
            rm = (float)((1.0 - (double)f.momentum) * (double)rm + (double)f.momentum * mean);
            rv = (float)((1.0 - (double)f.momentum) * (double)rv + (double)f.momentum * unbiased);
        }
        f.rmean[c] = rm;
        f.rvar[c] = rv;
        const float meanf = (float)mean, invstd = (float)(1.0 / sqrt(var + (double)f.eps));
        const float sc = f.gamma[c] * invstd;
        f.scale[c] = sc;
        f.shift[c] = f.beta[c] - meanf * sc;
        f.mean[c] = meanf;
        f.invstd[c] = invstd;
    }
    if (threadIdx.x == 0 && f.nbt) *f.nbt += f.repeat;
}
}  // namespace s2ag_fold
