// BatchNorm (training + eval), residual/activation glue and their backward passes on channels-last
// fp32 matrices.  All of these are HBM/latency-bound streaming kernels: consecutive lanes walk
// consecutive columns of a row (coalesced), grid-stride over rows.
#include <stdlib.h>

#include "s2ag_common.h"

namespace {
using namespace s2ag;

// ---- agent-scope publish / consume (write-through stores, L1-bypassing loads), as in gru_coop.hip -------------
__device__ __forceinline__ void st_agent(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// "Last block finalises": every block has published its partial sums with st_agent.  Drain them (vmcnt(0) per wave),
// barrier, then one lane takes a ticket; the block that draws the last ticket re-arms the ticket word (so the NEXT
// launch on this word needs no clearing kernel) and returns true.  No fences: the payload is write-through and the
// consumer reads it with agent-scope loads (CDNA4 guide, publish/consume recipe).
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

// ---- grid-wide wait for the ONE-launch BatchNorm (statistics + apply in the same kernel) ----------------------------------
// w[0] counts arrivals, w[1] is the "coefficients are published" flag, w[2] counts workgroups that have passed the wait.
// The last arriver folds the partial sums, publishes the coefficients (write-through) and raises the flag; every other
// workgroup polls the flag with agent-scope loads, then applies the coefficients to the rows it has just read (L2-hot).
// The workgroup that passes last re-arms all three words, so the next launch on them (stream ordered) needs no clear.
// Co-residency: the waiters occupy their CUs until the last workgroup has ARRIVED, so the launch must fit on the chip
// beside whatever else runs -- the host caps these launches at 64 workgroups of 1 024 threads (1 024 of the chip's 8 192
// wave slots; a few such launches on forked streams cannot starve each other) and takes the two-launch path for bigger
// tensors.  The poll is bounded: on a time-out the kernel goes on (wrong values) and ORs 4 into the sticky error word the
// trainer reads back every step (ops.check_coop_flag), exactly like the cooperative GRU's exchange.
__device__ int* g_bn_err = nullptr;
constexpr int BN_SPIN_LIMIT = 4000000;

__device__ __forceinline__ bool bar_arrive(int* w, int nblocks) {
    __shared__ int is_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (t == nblocks - 1);
    }
    __syncthreads();
    return is_last != 0;
}
__device__ __forceinline__ void bar_release(int* w) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's write-through coefficient stores are out
    __syncthreads();                                             // ... and every other wave's
    if (threadIdx.x == 0) __hip_atomic_store(w + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void bar_wait(int* w) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > BN_SPIN_LIMIT) {
                if (g_bn_err) atomicOr(g_bn_err, 4);
                break;
            }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void bar_leave(int* w, int nblocks) {
    if (threadIdx.x == 0) {
        const int e = __hip_atomic_fetch_add(w + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (e == nblocks - 1) {
            __hip_atomic_store(w, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(w + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(w + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Per-column totals from the (2, nrb, cols) partial sums, folded straight into per-channel LDS accumulators.
// NT threads: G = NT/cols groups share the partial rows of a column, one thread per column if cols >= NT.
template <typename T, typename A>
__device__ __forceinline__ void fold_column(const T* part, int nrb, int cols, int c, int r0, int rstep, A* a_out,
                                            A* b_out) {
    // 16 partial rows = 32 independent L1-bypassing loads in flight per thread, then the adds (a load-use chain per row
    // costs one fabric round trip per row)
    A a = 0, b = 0;
    for (int r = r0; r < nrb; r += 16 * rstep) {
        T va[16], vb[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int rr = min(r + j * rstep, nrb - 1);
            va[j] = ld_agent(part + (size_t)rr * cols + c);
            vb[j] = ld_agent(part + (size_t)(nrb + rr) * cols + c);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (r + j * rstep < nrb) {
                a += (A)va[j];
                b += (A)vb[j];
            }
    }
    *a_out = a;
    *b_out = b;
}

template <typename T, typename A, int NT>
__device__ __forceinline__ void fold_partials(const T* part, int nrb, int cols, const int* chan_of_col, A* c0, A* c1,
                                              A* cn) {
    const int t = threadIdx.x;
    const int G = cols >= NT ? 1 : NT / cols;
    // (deterministic mode: the LDS accumulators take their terms wavefront by wavefront; called by a whole workgroup)
    S2AG_DET_WAVES_BEGIN
    if (G == 1) {
        for (int c = t; c < cols; c += NT) {
            A a, b;
            fold_column<T, A>(part, nrb, cols, c, 0, 1, &a, &b);
            const int ch = chan_of_col ? chan_of_col[c] : c;
            atomicAdd(&c0[ch], a);
            atomicAdd(&c1[ch], b);
            atomicAdd(&cn[ch], (A)1);
        }
    } else if (t < G * cols) {
        const int c = t % cols, sub = t / cols;
        A a, b;
        fold_column<T, A>(part, nrb, cols, c, sub, G, &a, &b);
        const int ch = chan_of_col ? chan_of_col[c] : c;
        atomicAdd(&c0[ch], a);
        atomicAdd(&c1[ch], b);
        if (sub == 0) atomicAdd(&cn[ch], (A)1);
    }
    S2AG_DET_WAVES_END
}

// Channel statistics (already summed into cs / cq / cn in LDS) -> running estimates and per-COLUMN coefficients.
__device__ __forceinline__ void bn_finish_coeffs(double* cs, double* cq, const double* cn, const int* chan_of_col,
                                                 int ncols, int nchan, int rows, const float* gamma, const float* beta,
                                                 float* rmean, float* rvar, long long* nbt, float eps, float momentum,
                                                 int training, float* scale_col, float* shift_col, float* mean_col,
                                                 float* invstd_col, int repeat = 1) {
    if (training) {
        for (int ch = threadIdx.x; ch < nchan; ch += blockDim.x) {
            const double n = cn[ch] * (double)rows;
            const double mean = cs[ch] / n;
            double var = cq[ch] / n - mean * mean;      // fp64: safe against cancellation
            var = var < 0.0 ? 0.0 : var;
            const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
            // `repeat` forward passes of one step saw this very batch (same input, same weights): one statistics pass,
            // the running estimates advance once per pass, rounded to fp32 each time exactly as separate passes would
            float rm = rmean[ch], rv = rvar[ch];
            for (int it = 0; it < repeat; ++it) {
                rm = (float)((1.0 - (double)momentum) * (double)rm + (double)momentum * mean);
                rv = (float)((1.0 - (double)momentum) * (double)rv + (double)momentum * unbiased);
            }
            rmean[ch] = rm;
            rvar[ch] = rv;
            cs[ch] = mean;
            cq[ch] = 1.0 / sqrt(var + (double)eps);
        }
        if (threadIdx.x == 0 && nbt) *nbt += repeat;
    } else {
        for (int ch = threadIdx.x; ch < nchan; ch += blockDim.x) {
            cs[ch] = (double)rmean[ch];
            cq[ch] = 1.0 / sqrt((double)rvar[ch] + (double)eps);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < ncols; c += blockDim.x) {
        const int ch = chan_of_col ? chan_of_col[c] : c;
        const float mean = (float)cs[ch], invstd = (float)cq[ch];
        const float sc = gamma[ch] * invstd;
        st_agent(scale_col + c, sc);                            // write-through: the one-launch form reads them in this kernel
        st_agent(shift_col + c, beta[ch] - mean * sc);
        st_agent(mean_col + c, mean);
        st_agent(invstd_col + c, invstd);
    }
}

// One block.  Folds per-COLUMN sums into per-CHANNEL statistics through chan_of_col (LDS atomics),
// updates the running estimates, then scatters the per-channel coefficients back to columns.
__global__ __launch_bounds__(256) void bn_coeffs_k(const double* colsum, const double* colsq, const int* chan_of_col,
                                                   int ncols, int nchan, int rows, const float* gamma,
                                                   const float* beta, float* rmean, float* rvar, long long* nbt,
                                                   float eps, float momentum, int training, float* scale_col,
                                                   float* shift_col, float* mean_col, float* invstd_col) {
    extern __shared__ double smd[];
    double* cs = smd;               // nchan: sum  -> mean
    double* cq = smd + nchan;       // nchan: sumsq -> invstd
    double* cn = smd + 2 * nchan;   // nchan: columns per channel
    for (int i = threadIdx.x; i < 3 * nchan; i += blockDim.x) smd[i] = 0.0;
    __syncthreads();
    if (training) {
        S2AG_DET_WAVES_BEGIN
            for (int c = threadIdx.x; c < ncols; c += blockDim.x) {
                const int ch = chan_of_col ? chan_of_col[c] : c;
                atomicAdd(&cs[ch], colsum[c]);
                atomicAdd(&cq[ch], colsq[c]);
                atomicAdd(&cn[ch], 1.0);
            }
        S2AG_DET_WAVES_END
        __syncthreads();
    }
    bn_finish_coeffs(cs, cq, cn, chan_of_col, ncols, nchan, rows, gamma, beta, rmean, rvar, nbt, eps, momentum, training,
                     scale_col, shift_col, mean_col, invstd_col);
}

// Training-mode BatchNorm statistics in ONE launch: fp64 partial column sums per row block -> (2, nrb, cols) scratch,
// and the block that finishes last folds them into channel statistics, updates the running estimates and writes the
// per-column coefficients.  No accumulator to clear, no separate single-block kernel: BN forward = this + bn_apply.
// FLAT: narrow contiguous matrices (cols a power of two <= 32): lane-dense grid-stride walk, column = tid % cols.
template <bool FLAT, int NT>
__global__ __launch_bounds__(NT) void bn_fwd_stats_k(const float* __restrict__ x, int rows, int cols, int ldx, int rpb,
                                                      double* part, int* ticket, const int* chan_of_col, int nchan,
                                                      const float* gamma, const float* beta, float* rmean,
                                                      float* rvar, long long* nbt, float eps, float momentum,
                                                      float* scale_col, float* shift_col, float* mean_col,
                                                      float* invstd_col, int repeat,
                                                      float* __restrict__ y, int ldy, float act_slope, int* bar) {
    extern __shared__ double smd[];           // 3 * nchan doubles (finalising block)
    constexpr int RL = NT / 64;               // row lanes of the column-per-lane walk
    __shared__ double s1[RL][64], s2[RL][64];
    const int nrb = gridDim.y;
    if (FLAT) {
        static_assert(!FLAT || NT == 256, "flat walk: 256 threads");
        const long long total = (long long)rows * cols;
        double a = 0.0, b = 0.0;
#pragma unroll 8
        for (long long i = (long long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long long)nrb * 256) {
            const double v = (double)x[i];
            a += v;
            b += v * v;
        }
        double* f1 = &s1[0][0];
        double* f2 = &s2[0][0];
        f1[threadIdx.x] = a;
        f2[threadIdx.x] = b;
        __syncthreads();
        if ((int)threadIdx.x < cols) {
            double p = 0.0, q = 0.0;
            for (int i = threadIdx.x; i < 256; i += cols) {
                p += f1[i];
                q += f2[i];
            }
            st_agent(part + (size_t)blockIdx.y * cols + threadIdx.x, p);
            st_agent(part + (size_t)(nrb + blockIdx.y) * cols + threadIdx.x, q);
        }
    } else {
        const int c = blockIdx.x * 64 + (threadIdx.x & 63);
        const int ry = threadIdx.x >> 6;
        const int rbeg = blockIdx.y * rpb;
        const int rend = min(rows, rbeg + rpb);
        double a = 0.0, b = 0.0;
        if (c < cols) {
#pragma unroll 8
            for (int r = rbeg + ry; r < rend; r += RL) {
                const double v = (double)x[(long long)r * ldx + c];
                a += v;
                b += v * v;
            }
        }
        s1[ry][threadIdx.x & 63] = a;
        s2[ry][threadIdx.x & 63] = b;
        __syncthreads();
        if (ry == 0 && c < cols) {
            const int i = threadIdx.x;
            double p = 0.0, q = 0.0;
#pragma unroll
            for (int j = 0; j < RL; ++j) {
                p += s1[j][i];
                q += s2[j][i];
            }
            st_agent(part + (size_t)blockIdx.y * cols + c, p);
            st_agent(part + (size_t)(nrb + blockIdx.y) * cols + c, q);
        }
    }
    const int nblocks = gridDim.x * gridDim.y;
    const bool last = bar ? bar_arrive(bar, nblocks) : last_block_done(ticket, nblocks);
    if (last) {
        double* cs = smd;
        double* cq = smd + nchan;
        double* cn = smd + 2 * nchan;
        for (int i = threadIdx.x; i < 3 * nchan; i += blockDim.x) smd[i] = 0.0;
        __syncthreads();
        fold_partials<double, double, NT>(part, nrb, cols, chan_of_col, cs, cq, cn);
        __syncthreads();
        bn_finish_coeffs(cs, cq, cn, chan_of_col, cols, nchan, rows, gamma, beta, rmean, rvar, nbt, eps, momentum, 1,
                         scale_col, shift_col, mean_col, invstd_col, repeat);
        if (bar) bar_release(bar);
    }
    if (!bar) return;
    if (!last) bar_wait(bar);
    bar_leave(bar, nblocks);
    // apply to the rows this workgroup has just read
    if (FLAT) {
        const long long total = (long long)rows * cols;
        const int c = threadIdx.x % cols;
        const float sc = ld_agent(scale_col + c), sh = ld_agent(shift_col + c);
#pragma unroll 8
        for (long long i = (long long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long long)nrb * 256)
            y[i] = leaky(x[i] * sc + sh, act_slope);
    } else {
        const int c = blockIdx.x * 64 + (threadIdx.x & 63);
        const int ry = threadIdx.x >> 6;
        const int rbeg = blockIdx.y * rpb;
        const int rend = min(rows, rbeg + rpb);
        if (c < cols) {
            const float sc = ld_agent(scale_col + c), sh = ld_agent(shift_col + c);
#pragma unroll 8
            for (int r = rbeg + ry; r < rend; r += RL) y[(long long)r * ldy + c] = leaky(x[(long long)r * ldx + c] * sc + sh, act_slope);
        }
    }
}

__global__ __launch_bounds__(256) void bn_apply_k(const float* __restrict__ x, int rows, int cols, int ldx,
                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                  float slope, float* __restrict__ y, int ldy) {
    const long long total = (long long)rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
        y[(long long)r * ldy + c] = leaky(x[(long long)r * ldx + c] * scale[c] + shift[c], slope);
    }
}

// column sums of dpre and dpre * xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                       int rows, int cols, int ldx, int lddy, int rows_per_block,
                                                       const float* scale, const float* shift, const float* mean,
                                                       const float* invstd, float slope, float* s1, float* s2) {
    __shared__ float a1[4][64], a2[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ry = threadIdx.x >> 6;
    const int rbeg = blockIdx.y * rows_per_block;
    const int rend = min(rows, rbeg + rows_per_block);
    float p = 0.f, q = 0.f;
    if (c < cols) {
        const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
        for (int r = rbeg + ry; r < rend; r += 4) {
            const float xv = x[(long long)r * ldx + c];
            const float pre = xv * sc + sh;
            const float d = dy[(long long)r * lddy + c] * (pre > 0.f ? 1.f : slope);
            p += d;
            q += d * (xv - mu) * is;
        }
    }
    a1[ry][threadIdx.x & 63] = p;
    a2[ry][threadIdx.x & 63] = q;
    __syncthreads();
    s2ag::det_enter();
    if (ry == 0 && c < cols) {
        const int i = threadIdx.x;
        atomicAdd(s1 + c, a1[0][i] + a1[1][i] + a1[2][i] + a1[3][i]);
        atomicAdd(s2 + c, a2[0][i] + a2[1][i] + a2[2][i] + a2[3][i]);
    }
    s2ag::det_leave();
}

// lane-dense variant for narrow contiguous matrices (cols a power of two <= 32, ldx == lddy == cols)
__global__ __launch_bounds__(256) void bn_bwd_reduce_flat_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                            long long total, int cols, const float* scale,
                                                            const float* shift, const float* mean, const float* invstd,
                                                            float slope, float* s1, float* s2) {
    __shared__ float sm[256];
    const int c = threadIdx.x % cols;
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    float p = 0.f, q = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const float xv = x[i];
        const float pre = xv * sc + sh;
        const float d = dy[i] * (pre > 0.f ? 1.f : slope);
        p += d;
        q += d * (xv - mu) * is;
    }
    s2ag::det_enter();
    for (int pass = 0; pass < 2; ++pass) {
        sm[threadIdx.x] = pass ? q : p;
        __syncthreads();
        if ((int)threadIdx.x < cols) {
            float s = 0.f;
            for (int i = threadIdx.x; i < 256; i += cols) s += sm[i];
            atomicAdd((pass ? s2 : s1) + threadIdx.x, s);
        }
        __syncthreads();
    }
    s2ag::det_leave();
}

__device__ __forceinline__ void bn_bwd_finish(const float* t1, const float* t2, const float* cn,
                                              const int* chan_of_col, int ncols, int nchan, int rows, float* dgamma,
                                              float* dbeta, int accumulate, float* c1, float* c2) {
    for (int ch = threadIdx.x; ch < nchan; ch += blockDim.x) {
        if (accumulate) {
            // atomic: the backward passes of D(real) and D(fake) run on two streams and meet in the same gradient slots
            // (a plain += lost updates now and then -- graph replay only, where the two chains really overlap)
            atomicAdd(dgamma + ch, t2[ch]);
            atomicAdd(dbeta + ch, t1[ch]);
        } else {
            dgamma[ch] = t2[ch];
            dbeta[ch] = t1[ch];
        }
    }
    for (int c = threadIdx.x; c < ncols; c += blockDim.x) {
        const int ch = chan_of_col ? chan_of_col[c] : c;
        const float n = cn[ch] * (float)rows;
        st_agent(c1 + c, t1[ch] / n);
        st_agent(c2 + c, t2[ch] / n);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_coeffs_k(const float* s1, const float* s2, const int* chan_of_col,
                                                       int ncols, int nchan, int rows, float* dgamma, float* dbeta,
                                                       int accumulate, float* c1, float* c2) {
    extern __shared__ float sm[];
    float* t1 = sm;
    float* t2 = sm + nchan;
    float* cn = sm + 2 * nchan;
    for (int i = threadIdx.x; i < 3 * nchan; i += blockDim.x) sm[i] = 0.f;
    __syncthreads();
    S2AG_DET_WAVES_BEGIN
        for (int c = threadIdx.x; c < ncols; c += blockDim.x) {
            const int ch = chan_of_col ? chan_of_col[c] : c;
            atomicAdd(&t1[ch], s1[c]);
            atomicAdd(&t2[ch], s2[c]);
            atomicAdd(&cn[ch], 1.0f);
        }
    S2AG_DET_WAVES_END
    __syncthreads();
    bn_bwd_finish(t1, t2, cn, chan_of_col, ncols, nchan, rows, dgamma, dbeta, accumulate, c1, c2);
}

// BatchNorm backward statistics in ONE launch (see bn_fwd_stats_k): partial column sums of dpre and dpre*xhat per row
// block, folded by the last block into dgamma / dbeta (+=) and the per-column c1, c2 of the dx formula.
// The producing layer already left per-row-block column sums (conv_gemm / gemm_lin epilogue): only the fold is left.
// A big layer leaves thousands of partial rows (one per 32 output rows): one block folding them alone is a 60 us serial
// tail.  This pre-pass folds slices of FOLD_SLICE rows in place -- block g leaves the sums of rows [g*S, (g+1)*S) in row
// g*S of either half -- and bn_fold_k then reads every S-th row.  Fixed order: bit-reproducible.
constexpr int FOLD_SLICE = 64;
__global__ __launch_bounds__(256) void bn_fold_pre_k(double* part, int prow, int cols) {
    // block (slice of FOLD_SLICE rows, block of up to 256 columns)
    __shared__ double red[2][256];
    const int beg = blockIdx.x * FOLD_SLICE, end = min(prow, beg + FOLD_SLICE);
    const int c0 = blockIdx.y * 256, cw = min(256, cols - c0);       // this block's columns
    const int nsub = 256 / cw;
    const int cl = threadIdx.x % cw, sub = threadIdx.x / cw;
    const int c = c0 + cl;
    double a = 0.0, b = 0.0;
    if (sub < nsub)
        for (int r = beg + sub; r < end; r += nsub) {
            a += part[(size_t)r * cols + c];
            b += part[(size_t)(prow + r) * cols + c];
        }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    if ((int)threadIdx.x < cw) {
        for (int j = 1; j < nsub; ++j) {
            a += red[0][j * cw + cl];
            b += red[1][j * cw + cl];
        }
        part[(size_t)beg * cols + c] = a;
        part[(size_t)(prow + beg) * cols + c] = b;
    }
}

__global__ __launch_bounds__(1024) void bn_fold_k(const double* part, int prow, int rows, int cols, const int* chan_of_col,
                                                  int nchan, const float* gamma, const float* beta, float* rmean,
                                                  float* rvar, long long* nbt, float eps, float momentum, int repeat,
                                                  float* scale_col, float* shift_col, float* mean_col,
                                                  float* invstd_col, int pstep) {
    extern __shared__ double smd[];
    double* cs = smd;
    double* cq = smd + nchan;
    double* cn = smd + 2 * nchan;
    for (int i = threadIdx.x; i < 3 * nchan; i += blockDim.x) smd[i] = 0.0;
    __syncthreads();
    if (pstep == 1) {
        fold_partials<double, double, 1024>(part, prow, cols, chan_of_col, cs, cq, cn);
    } else {
        // rows 0, pstep, 2*pstep, ... of either half (left by bn_fold_pre_k)
        const int nsub = cols >= 1024 ? 1 : 1024 / cols;
        const int sub = cols >= 1024 ? 0 : (int)threadIdx.x / cols;
        S2AG_DET_WAVES_BEGIN
        for (int c = cols >= 1024 ? (int)threadIdx.x : (int)threadIdx.x % cols; c < cols && sub < nsub; c += 1024) {
            double a = 0.0, b = 0.0;
            for (int r = sub * pstep; r < prow; r += nsub * pstep) {
                a += part[(size_t)r * cols + c];
                b += part[(size_t)(prow + r) * cols + c];
            }
            const int ch = chan_of_col ? chan_of_col[c] : c;
            atomicAdd(&cs[ch], a);
            atomicAdd(&cq[ch], b);
            if (sub == 0) atomicAdd(&cn[ch], 1.0);
            if (cols < 1024) break;
        }
        S2AG_DET_WAVES_END
    }
    __syncthreads();
    bn_finish_coeffs(cs, cq, cn, chan_of_col, cols, nchan, rows, gamma, beta, rmean, rvar, nbt, eps, momentum, 1,
                     scale_col, shift_col, mean_col, invstd_col, repeat);
}

template <bool FLAT, int NT>
__global__ __launch_bounds__(NT) void bn_bwd_stats_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                      int rows, int cols, int ldx, int lddy, int rpb,
                                                      const float* scale, const float* shift, const float* mean,
                                                      const float* invstd, float slope, float* part, int* ticket,
                                                      const int* chan_of_col, int nchan, float* dgamma, float* dbeta,
                                                      int accumulate, float* c1, float* c2, float* __restrict__ dx,
                                                      int lddx, int* bar) {
    extern __shared__ float sm[];            // 3 * nchan floats (finalising block)
    constexpr int RL = NT / 64;
    __shared__ float a1[RL][64], a2[RL][64];
    const int nrb = gridDim.y;
    if (FLAT) {
        const long long total = (long long)rows * cols;
        const int c = threadIdx.x % cols;
        const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
        float p = 0.f, q = 0.f;
#pragma unroll 8
        for (long long i = (long long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long long)nrb * 256) {
            const float xv = x[i];
            const float pre = xv * sc + sh;
            const float d = dy[i] * (pre > 0.f ? 1.f : slope);
            p += d;
            q += d * (xv - mu) * is;
        }
        float* f1 = &a1[0][0];
        float* f2 = &a2[0][0];
        f1[threadIdx.x] = p;
        f2[threadIdx.x] = q;
        __syncthreads();
        if ((int)threadIdx.x < cols) {
            float u = 0.f, v = 0.f;
            for (int i = threadIdx.x; i < 256; i += cols) {
                u += f1[i];
                v += f2[i];
            }
            st_agent(part + (size_t)blockIdx.y * cols + threadIdx.x, u);
            st_agent(part + (size_t)(nrb + blockIdx.y) * cols + threadIdx.x, v);
        }
    } else {
        const int c = blockIdx.x * 64 + (threadIdx.x & 63);
        const int ry = threadIdx.x >> 6;
        const int rbeg = blockIdx.y * rpb;
        const int rend = min(rows, rbeg + rpb);
        float p = 0.f, q = 0.f;
        if (c < cols) {
            const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
#pragma unroll 8
            for (int r = rbeg + ry; r < rend; r += RL) {
                const float xv = x[(long long)r * ldx + c];
                const float pre = xv * sc + sh;
                const float d = dy[(long long)r * lddy + c] * (pre > 0.f ? 1.f : slope);
                p += d;
                q += d * (xv - mu) * is;
            }
        }
        a1[ry][threadIdx.x & 63] = p;
        a2[ry][threadIdx.x & 63] = q;
        __syncthreads();
        if (ry == 0 && c < cols) {
            const int i = threadIdx.x;
            float u = 0.f, v = 0.f;
#pragma unroll
            for (int j = 0; j < RL; ++j) {
                u += a1[j][i];
                v += a2[j][i];
            }
            st_agent(part + (size_t)blockIdx.y * cols + c, u);
            st_agent(part + (size_t)(nrb + blockIdx.y) * cols + c, v);
        }
    }
    const int nblocks = gridDim.x * gridDim.y;
    const bool last = bar ? bar_arrive(bar, nblocks) : last_block_done(ticket, nblocks);
    if (last) {
        float* t1 = sm;
        float* t2 = sm + nchan;
        float* cn = sm + 2 * nchan;
        for (int i = threadIdx.x; i < 3 * nchan; i += blockDim.x) sm[i] = 0.f;
        __syncthreads();
        fold_partials<float, float, NT>(part, nrb, cols, chan_of_col, t1, t2, cn);
        __syncthreads();
        bn_bwd_finish(t1, t2, cn, chan_of_col, cols, nchan, rows, dgamma, dbeta, accumulate, c1, c2);
        if (bar) bar_release(bar);
    }
    if (!bar) return;
    if (!last) bar_wait(bar);
    bar_leave(bar, nblocks);
    // dx = scale * (d - c1 - xhat * c2) for the rows this workgroup has just read
    if (FLAT) {
        const long long total = (long long)rows * cols;
        const int c = threadIdx.x % cols;
        const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
        const float k1 = ld_agent(c1 + c), k2 = ld_agent(c2 + c);
#pragma unroll 8
        for (long long i = (long long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long long)nrb * 256) {
            const float xv = x[i];
            const float d = dy[i] * (xv * sc + sh > 0.f ? 1.f : slope);
            dx[i] = sc * (d - k1 - (xv - mu) * is * k2);
        }
    } else {
        const int c = blockIdx.x * 64 + (threadIdx.x & 63);
        const int ry = threadIdx.x >> 6;
        const int rbeg = blockIdx.y * rpb;
        const int rend = min(rows, rbeg + rpb);
        if (c < cols) {
            const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
            const float k1 = ld_agent(c1 + c), k2 = ld_agent(c2 + c);
#pragma unroll 8
            for (int r = rbeg + ry; r < rend; r += RL) {
                const float xv = x[(long long)r * ldx + c];
                const float d = dy[(long long)r * lddy + c] * (xv * sc + sh > 0.f ? 1.f : slope);
                dx[(long long)r * lddx + c] = sc * (d - k1 - (xv - mu) * is * k2);
            }
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                      int rows, int cols, int ldx, int lddy, const float* scale,
                                                      const float* shift, const float* mean, const float* invstd,
                                                      float slope, const float* c1, const float* c2,
                                                      float* __restrict__ dx, int lddx) {
    const long long total = (long long)rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
        const float xv = x[(long long)r * ldx + c];
        const float pre = xv * scale[c] + shift[c];
        const float d = dy[(long long)r * lddy + c] * (pre > 0.f ? 1.f : slope);
        const float xhat = (xv - mean[c]) * invstd[c];
        dx[(long long)r * lddx + c] = scale[c] * (d - c1[c] - xhat * c2[c]);
    }
}

__global__ __launch_bounds__(256) void add_act_k(const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                 int ldb, float* __restrict__ y, int ldy, int rows, int cols,
                                                 float slope) {
    const long long total = (long long)rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
        float v = a[(long long)r * lda + c];
        if (b) v += b[(long long)r * ldb + c];
        y[(long long)r * ldy + c] = leaky(v, slope);
    }
}

__global__ __launch_bounds__(256) void epilogue_bwd_k(const float* __restrict__ dy, int lddy,
                                                      const float* __restrict__ y, int ldy, float* __restrict__ g,
                                                      int ldg, int rows, int cols, int act, float slope, float drop_p,
                                                      float inv_keep, const unsigned long long* rng, unsigned site) {
    const long long total = (long long)rows * cols;
    SiteKey key{0, 0};
    if (drop_p > 0.f) key = site_key(rng, site);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
        float d = dy[(long long)r * lddy + c];
        float ks = 1.f;
        if (drop_p > 0.f) ks = keep_scale(key, (unsigned long long)i, drop_p, inv_keep);
        d *= ks;
        if (act == S2AG_ACT_LEAKY) {
            if (slope != 1.f) {
                // y is the post-dropout output: its sign is that of the activation where the unit was kept
                const float yv = y[(long long)r * ldy + c];
                d *= (yv > 0.f ? 1.f : slope);
            }
        } else if (act == S2AG_ACT_SIGMOID) {
            const float yv = (ks != 0.f) ? y[(long long)r * ldy + c] / ks : 0.f;
            d *= yv * (1.f - yv);
        }
        g[(long long)r * ldg + c] = d;
    }
}

inline int ew_grid(long long total) {
    long long b = (total + 255) / 256;
    if (b > 4096) b = 4096;   // 16 blocks per CU, grid-stride beyond
    if (b < 1) b = 1;
    return (int)b;
}
}  // namespace

extern "C" int s2ag_bn_coeffs(const double* colsum, const double* colsq, const int* chan_of_col, int ncols, int nchan,
                              int rows, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, long long* nbt, float eps, float momentum, int training,
                              float* scale_col, float* shift_col, float* mean_col, float* invstd_col, void* stream) {
    if (ncols <= 0 || nchan <= 0 || rows <= 0 || !gamma || !beta || !running_mean || !running_var || !scale_col ||
        !shift_col || !mean_col || !invstd_col)
        return S2AG_E_BADARG;
    if (training && (!colsum || !colsq)) return S2AG_E_BADARG;
    hipLaunchKernelGGL(bn_coeffs_k, dim3(1), dim3(256), sizeof(double) * 3 * nchan, (hipStream_t)stream, colsum, colsq,
                       chan_of_col, ncols, nchan, rows, gamma, beta, running_mean, running_var, nbt, eps, momentum,
                       training, scale_col, shift_col, mean_col, invstd_col);
    S2AG_LAUNCH_CHECK();
    return 0;
}

// row-block plan shared by the fused statistics kernels and the scratch sizing query
static inline bool bn_flat(int cols, int ldx, int lddy) {
    return ldx == cols && lddy == cols && cols <= 32 && (cols & (cols - 1)) == 0;
}
static inline void bn_plan(int rows, int cols, bool flat, int* colblocks, int* nrb, int* rpb, int max_blocks = 0) {
    // The finalising block reads 2 * nrb * cols partial sums with 256 threads: cap that at ~128 loads per thread
    // (4 batches of 32 in flight), i.e. nrb <= 16384 / cols -- which still gives ~256 workgroups at every width.
    int cap = 16384 / cols;
    cap = cap < 1 ? 1 : (cap > 1024 ? 1024 : cap);
    if (flat) {
        long long nb = ((long long)rows * cols + 256 * 16 - 1) / (256 * 16);
        *colblocks = 1;
        if (max_blocks > 0 && cap > 4 * max_blocks) cap = 4 * max_blocks;      // 256-thread workgroups: 4 per 1024 threads
        *nrb = (int)(nb < 1 ? 1 : (nb > cap ? cap : nb));
        *rpb = 0;
        return;
    }
    *colblocks = cdiv(cols, 64);
    int r = 64;      // 1024 threads = 16 row lanes per block: 4+ rows per lane, more while that still leaves ~256 blocks
    while (cdiv(rows, r) > cap || (long long)cdiv(rows, 2 * r) * *colblocks >= 256) r *= 2;
    while (max_blocks > 0 && r < rows && (long long)cdiv(rows, r) * *colblocks > max_blocks) r *= 2;
    *rpb = r;
    *nrb = cdiv(rows, r);
}

extern "C" int s2ag_bn_partial_rows(int rows, int cols, int ld) {
    if (rows <= 0 || cols <= 0 || ld < cols) return S2AG_E_BADARG;
    // the forward pass plans with ldx, the backward pass with (ldx, lddy): size for whichever plan is larger
    int cb, nrb, rpb, cb2, nrb2, rpb2;
    bn_plan(rows, cols, bn_flat(cols, ld, ld), &cb, &nrb, &rpb);
    bn_plan(rows, cols, false, &cb2, &nrb2, &rpb2);
    return nrb > nrb2 ? nrb : nrb2;
}

extern "C" int s2ag_bn_fwd_stats(const float* x, int rows, int cols, int ldx, const int* chan_of_col, int nchan,
                                 const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 long long* nbt, float eps, float momentum, int repeat, double* partials, int* ticket,
                                 float* scale_col, float* shift_col, float* mean_col, float* invstd_col, void* stream) {
    if (repeat < 1) return S2AG_E_BADARG;
    if (!x || rows <= 0 || cols <= 0 || ldx < cols || nchan <= 0 || !gamma || !beta || !running_mean || !running_var ||
        !partials || !ticket || !scale_col || !shift_col || !mean_col || !invstd_col)
        return S2AG_E_BADARG;
    const bool flat = bn_flat(cols, ldx, ldx);
    int cb, nrb, rpb;
    bn_plan(rows, cols, flat, &cb, &nrb, &rpb);
    const size_t smem = sizeof(double) * 3 * nchan;
    if (flat)
        hipLaunchKernelGGL((bn_fwd_stats_k<true, 256>), dim3(cb, nrb), dim3(256), smem, (hipStream_t)stream, x, rows, cols, ldx,
                           rpb, partials, ticket, chan_of_col, nchan, gamma, beta, running_mean, running_var, nbt, eps,
                           momentum, scale_col, shift_col, mean_col, invstd_col, repeat, nullptr, 0, 1.f, nullptr);
    else
        hipLaunchKernelGGL((bn_fwd_stats_k<false, 1024>), dim3(cb, nrb), dim3(1024), smem, (hipStream_t)stream, x, rows, cols,
                           ldx, rpb, partials, ticket, chan_of_col, nchan, gamma, beta, running_mean, running_var, nbt,
                           eps, momentum, scale_col, shift_col, mean_col, invstd_col, repeat, nullptr, 0, 1.f, nullptr);
    S2AG_LAUNCH_CHECK();
    return 0;
}

// ---- one-launch forms (statistics + apply; see bar_arrive) ---------------------------------------------------------------
constexpr int BN_FUSED_MAX_BLOCKS = 64;                 // workgroups of 1024 threads (4x as many of 256 in the flat form)
constexpr long long BN_FUSED_MAX_ELEMS = 4ll << 20;     // bigger tensors keep the two-launch path (full grids)

extern "C" int s2ag_bn_set_error_flag(int* flag) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_bn_err), &flag, sizeof(flag));
}

extern "C" int s2ag_bn_fused_supported(int rows, int cols) {
    // wider than 64 column blocks cannot fit BN_FUSED_MAX_BLOCKS whatever the row blocking: two-launch path
    return rows > 0 && cols > 0 && (long long)rows * cols <= BN_FUSED_MAX_ELEMS && cdiv(cols, 64) <= BN_FUSED_MAX_BLOCKS;
}

extern "C" int s2ag_bn_fused_partial_rows(int rows, int cols, int ld) {
    if (rows <= 0 || cols <= 0 || ld < cols) return S2AG_E_BADARG;
    int cb, nrb, rpb, cb2, nrb2, rpb2;
    bn_plan(rows, cols, bn_flat(cols, ld, ld), &cb, &nrb, &rpb, BN_FUSED_MAX_BLOCKS);
    bn_plan(rows, cols, false, &cb2, &nrb2, &rpb2, BN_FUSED_MAX_BLOCKS);
    return nrb > nrb2 ? nrb : nrb2;
}

extern "C" int s2ag_bn_fwd_fused(const float* x, int rows, int cols, int ldx, const int* chan_of_col, int nchan,
                                 const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 long long* nbt, float eps, float momentum, int repeat, double* partials, int* bar,
                                 float* scale_col, float* shift_col, float* mean_col, float* invstd_col, float slope,
                                 float* y, int ldy, void* stream) {
    if (repeat < 1) return S2AG_E_BADARG;
    if (!x || !y || rows <= 0 || cols <= 0 || ldx < cols || ldy < cols || nchan <= 0 || !gamma || !beta || !running_mean ||
        !running_var || !partials || !bar || !scale_col || !shift_col || !mean_col || !invstd_col)
        return S2AG_E_BADARG;
    if (!s2ag_bn_fused_supported(rows, cols)) return S2AG_E_UNSUPPORTED;
    const bool flat = bn_flat(cols, ldx, ldy);
    int cb, nrb, rpb;
    bn_plan(rows, cols, flat, &cb, &nrb, &rpb, BN_FUSED_MAX_BLOCKS);
    const size_t smem = sizeof(double) * 3 * nchan;
    if (flat)
        hipLaunchKernelGGL((bn_fwd_stats_k<true, 256>), dim3(cb, nrb), dim3(256), smem, (hipStream_t)stream, x, rows, cols, ldx,
                           rpb, partials, nullptr, chan_of_col, nchan, gamma, beta, running_mean, running_var, nbt, eps,
                           momentum, scale_col, shift_col, mean_col, invstd_col, repeat, y, ldy, slope, bar);
    else
        hipLaunchKernelGGL((bn_fwd_stats_k<false, 1024>), dim3(cb, nrb), dim3(1024), smem, (hipStream_t)stream, x, rows, cols,
                           ldx, rpb, partials, nullptr, chan_of_col, nchan, gamma, beta, running_mean, running_var, nbt,
                           eps, momentum, scale_col, shift_col, mean_col, invstd_col, repeat, y, ldy, slope, bar);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bn_bwd_fused(const float* x, const float* dy, int rows, int cols, int ldx, int lddy,
                                 const float* scale_col, const float* shift_col, const float* mean_col,
                                 const float* invstd_col, float slope, const int* chan_of_col, int nchan, float* dgamma,
                                 float* dbeta, int accumulate, float* partials, int* bar, float* c1_col, float* c2_col,
                                 float* dx, int lddx, void* stream) {
    if (!x || !dy || !dx || rows <= 0 || cols <= 0 || ldx < cols || lddy < cols || lddx < cols || nchan <= 0 || !dgamma ||
        !dbeta || !partials || !bar || !c1_col || !c2_col)
        return S2AG_E_BADARG;
    if (!s2ag_bn_fused_supported(rows, cols)) return S2AG_E_UNSUPPORTED;
    const bool flat = bn_flat(cols, ldx, lddy) && lddx == cols;
    int cb, nrb, rpb;
    bn_plan(rows, cols, flat, &cb, &nrb, &rpb, BN_FUSED_MAX_BLOCKS);
    const size_t smem = sizeof(float) * 3 * nchan;
    if (flat)
        hipLaunchKernelGGL((bn_bwd_stats_k<true, 256>), dim3(cb, nrb), dim3(256), smem, (hipStream_t)stream, x, dy, rows, cols,
                           ldx, lddy, rpb, scale_col, shift_col, mean_col, invstd_col, slope, partials, nullptr,
                           chan_of_col, nchan, dgamma, dbeta, accumulate, c1_col, c2_col, dx, lddx, bar);
    else
        hipLaunchKernelGGL((bn_bwd_stats_k<false, 1024>), dim3(cb, nrb), dim3(1024), smem, (hipStream_t)stream, x, dy, rows, cols,
                           ldx, lddy, rpb, scale_col, shift_col, mean_col, invstd_col, slope, partials, nullptr,
                           chan_of_col, nchan, dgamma, dbeta, accumulate, c1_col, c2_col, dx, lddx, bar);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bn_fold(double* partials, int partial_rows, int rows, int cols, const int* chan_of_col,
                            int nchan, const float* gamma, const float* beta, float* running_mean, float* running_var,
                            long long* nbt, float eps, float momentum, int repeat, float* scale_col, float* shift_col,
                            float* mean_col, float* invstd_col, void* stream) {
    if (!partials || partial_rows <= 0 || rows <= 0 || cols <= 0 || nchan <= 0 || repeat < 1 || !gamma || !beta ||
        !running_mean || !running_var || !scale_col || !shift_col || !mean_col || !invstd_col)
        return S2AG_E_BADARG;
    int pstep = 1;
    constexpr int pre_min = 1024;
    if (partial_rows >= pre_min) {
        pstep = FOLD_SLICE;
        hipLaunchKernelGGL(bn_fold_pre_k, dim3(cdiv(partial_rows, FOLD_SLICE), cdiv(cols, 256)), dim3(256), 0,
                           (hipStream_t)stream, partials, partial_rows, cols);
    }
    hipLaunchKernelGGL(bn_fold_k, dim3(1), dim3(1024), sizeof(double) * 3 * nchan, (hipStream_t)stream, partials,
                       partial_rows, rows, cols, chan_of_col, nchan, gamma, beta, running_mean, running_var, nbt, eps,
                       momentum, repeat, scale_col, shift_col, mean_col, invstd_col, pstep);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bn_bwd_stats(const float* x, const float* dy, int rows, int cols, int ldx, int lddy,
                                 const float* scale_col, const float* shift_col, const float* mean_col,
                                 const float* invstd_col, float slope, const int* chan_of_col, int nchan,
                                 float* dgamma, float* dbeta, int accumulate, float* partials, int* ticket,
                                 float* c1_col, float* c2_col, void* stream) {
    if (!x || !dy || rows <= 0 || cols <= 0 || ldx < cols || lddy < cols || nchan <= 0 || !dgamma || !dbeta ||
        !partials || !ticket || !c1_col || !c2_col)
        return S2AG_E_BADARG;
    const bool flat = bn_flat(cols, ldx, lddy);
    int cb, nrb, rpb;
    bn_plan(rows, cols, flat, &cb, &nrb, &rpb);
    const size_t smem = sizeof(float) * 3 * nchan;
    if (flat)
        hipLaunchKernelGGL((bn_bwd_stats_k<true, 256>), dim3(cb, nrb), dim3(256), smem, (hipStream_t)stream, x, dy, rows, cols,
                           ldx, lddy, rpb, scale_col, shift_col, mean_col, invstd_col, slope, partials, ticket,
                           chan_of_col, nchan, dgamma, dbeta, accumulate, c1_col, c2_col, nullptr, 0, nullptr);
    else
        hipLaunchKernelGGL((bn_bwd_stats_k<false, 1024>), dim3(cb, nrb), dim3(1024), smem, (hipStream_t)stream, x, dy, rows, cols,
                           ldx, lddy, rpb, scale_col, shift_col, mean_col, invstd_col, slope, partials, ticket,
                           chan_of_col, nchan, dgamma, dbeta, accumulate, c1_col, c2_col, nullptr, 0, nullptr);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bn_apply(const float* x, int rows, int cols, int ldx, const float* scale_col,
                             const float* shift_col, float slope, float* y, int ldy, void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0 || ldx < cols || ldy < cols) return S2AG_E_BADARG;
    hipLaunchKernelGGL(bn_apply_k, dim3(ew_grid((long long)rows * cols)), dim3(256), 0, (hipStream_t)stream, x, rows,
                       cols, ldx, scale_col, shift_col, slope, y, ldy);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bn_bwd_reduce(const float* x, const float* dy, int rows, int cols, int ldx, int lddy,
                                  const float* scale_col, const float* shift_col, const float* mean_col,
                                  const float* invstd_col, float slope, float* s1_col, float* s2_col, void* stream) {
    if (!x || !dy || rows <= 0 || cols <= 0 || !s1_col || !s2_col) return S2AG_E_BADARG;
    hipError_t me;
    if (s2_col == s1_col + cols) {
        me = zero_async(s1_col, sizeof(float) * 2 * cols, (hipStream_t)stream);
        if (me != hipSuccess) return (int)me;
    } else {
        me = zero_async(s1_col, sizeof(float) * cols, (hipStream_t)stream);
        if (me != hipSuccess) return (int)me;
        me = zero_async(s2_col, sizeof(float) * cols, (hipStream_t)stream);
        if (me != hipSuccess) return (int)me;
    }
    if (ldx == cols && lddy == cols && cols <= 32 && (cols & (cols - 1)) == 0) {
        const long long total = (long long)rows * cols;
        long long nb = (total + 256 * 16 - 1) / (256 * 16);
        nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
        hipLaunchKernelGGL(bn_bwd_reduce_flat_k, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, dy, total, cols,
                           scale_col, shift_col, mean_col, invstd_col, slope, s1_col, s2_col);
        S2AG_LAUNCH_CHECK();
        return 0;
    }
    int rpb = 256;
    const int colblocks = cdiv(cols, 64);
    while (rpb > 64 && (long long)cdiv(rows, rpb) * colblocks < 1024) rpb >>= 1;
    hipLaunchKernelGGL(bn_bwd_reduce_k, dim3(colblocks, cdiv(rows, rpb)), dim3(256), 0, (hipStream_t)stream, x, dy,
                       rows, cols, ldx, lddy, rpb, scale_col, shift_col, mean_col, invstd_col, slope, s1_col, s2_col);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bn_bwd_coeffs(const float* s1_col, const float* s2_col, const int* chan_of_col, int ncols,
                                  int nchan, int rows, float* dgamma, float* dbeta, int accumulate, float* c1_col,
                                  float* c2_col, void* stream) {
    if (!s1_col || !s2_col || ncols <= 0 || nchan <= 0 || rows <= 0 || !dgamma || !dbeta || !c1_col || !c2_col)
        return S2AG_E_BADARG;
    hipLaunchKernelGGL(bn_bwd_coeffs_k, dim3(1), dim3(256), sizeof(float) * 3 * nchan, (hipStream_t)stream, s1_col,
                       s2_col, chan_of_col, ncols, nchan, rows, dgamma, dbeta, accumulate, c1_col, c2_col);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bn_bwd_apply(const float* x, const float* dy, int rows, int cols, int ldx, int lddy,
                                 const float* scale_col, const float* shift_col, const float* mean_col,
                                 const float* invstd_col, float slope, const float* c1_col, const float* c2_col,
                                 float* dx, int lddx, void* stream) {
    if (!x || !dy || !dx || rows <= 0 || cols <= 0) return S2AG_E_BADARG;
    hipLaunchKernelGGL(bn_bwd_apply_k, dim3(ew_grid((long long)rows * cols)), dim3(256), 0, (hipStream_t)stream, x, dy,
                       rows, cols, ldx, lddy, scale_col, shift_col, mean_col, invstd_col, slope, c1_col, c2_col, dx,
                       lddx);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_add_act(const float* a, int lda, const float* b, int ldb, float* y, int ldy, int rows, int cols,
                            float slope, void* stream) {
    if (!a || !y || rows <= 0 || cols <= 0 || lda < cols || ldy < cols || (b && ldb < cols)) return S2AG_E_BADARG;
    hipLaunchKernelGGL(add_act_k, dim3(ew_grid((long long)rows * cols)), dim3(256), 0, (hipStream_t)stream, a, lda, b,
                       ldb, y, ldy, rows, cols, slope);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_epilogue_bwd(const float* dy, int lddy, const float* y, int ldy, float* g, int ldg, int rows,
                                 int cols, const s2ag_epilogue* e, void* stream) {
    if (!dy || !g || !e || rows <= 0 || cols <= 0) return S2AG_E_BADARG;
    const bool needs_y = (e->act == S2AG_ACT_LEAKY && e->slope != 1.f) || e->act == S2AG_ACT_SIGMOID;
    if (needs_y && !y) return S2AG_E_BADARG;
    if (e->drop_p > 0.f && !e->rng) return S2AG_E_BADARG;
    const float inv_keep = e->drop_p > 0.f ? 1.f / (1.f - e->drop_p) : 1.f;
    hipLaunchKernelGGL(epilogue_bwd_k, dim3(ew_grid((long long)rows * cols)), dim3(256), 0, (hipStream_t)stream, dy,
                       lddy, y, ldy, g, ldg, rows, cols, e->act, e->slope, e->drop_p, inv_keep, e->rng, e->site);
    S2AG_LAUNCH_CHECK();
    return 0;
}
S2AG_DET_HOOK(norm_elementwise)
