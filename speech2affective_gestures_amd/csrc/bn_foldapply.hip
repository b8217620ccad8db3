// OPT-IN VARIANT (config switch BN_FOLD_APPLY, default OFF): the fold of a producing layer's BatchNorm statistics and the
// normalise + LeakyReLU pass in ONE launch instead of bn_fold_k (one workgroup) followed by bn_apply_k
// (csrc/norm_elementwise.hip).  Replaces, for a training-mode nn.BatchNorm1d + LeakyReLU whose input conv left its column
// sums behind (s2ag_conv1d_nlc_fwd_stats; MFCCEncoder net/multimodal_context_net_v2.py:39-48, WavEncoder :18-27,
// ConvDiscriminator.pre_conv :397-403, EmbeddingNet), native_batch_norm's statistics / running-estimate update and the
// elementwise apply.  21 such pairs per GAN step (tools/count_launches_emu.py): 21 launches and 21 single-workgroup links in
// the step's dependency chains less.
//
// No grid-wide wait (the one-launch BatchNorm of norm_elementwise.hip needs one because its workgroups PRODUCE the sums): the
// sums exist before the launch, so EVERY workgroup folds them itself -- (2, R, C) doubles, a few tens of KB from L2 -- in a
// FIXED order (per column: partial rows ascending, combined through LDS in a fixed order; per channel: columns ascending), so
// all workgroups hold the same coefficients to the bit, and workgroup 0 alone writes what outlives the launch (running
// estimates, batch counter, the four coefficient vectors the backward pass reads).  The host only takes this path when the
// re-read is small (s2ag_bn_fold_apply_supported: R * C <= 16 384) and caps the grid at 64 workgroups.
//
// Arithmetic = bn_finish_coeffs (fp64 statistics, biased variance for the normalisation, unbiased for the running estimate,
// `repeat` running-estimate updates rounded to fp32 each, fp32 scale / shift); the default's bn_fold_k adds the partial rows
// through LDS atomics in arrival order, so the two agree to fp64 rounding of the sums, not to the bit
// (tests/test_gpu_zy_variants.py: 1e-6 on y and on every coefficient, exact on the batch counter).
#include "s2ag_common.h"

namespace {
using namespace s2ag;

__global__ __launch_bounds__(256) void bn_fold_apply_k(const double* __restrict__ part, int prow, int rows, int cols,
                                                       const int* __restrict__ chan_of_col, int nchan,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* rmean, float* rvar, long long* nbt, float eps, float momentum,
                                                       int repeat, float* scale_col, float* shift_col, float* mean_col,
                                                       float* invstd_col, const float* __restrict__ x, int ldx, float slope,
                                                       float* __restrict__ y, int ldy) {
    extern __shared__ double smd[];
    double* colA = smd;                         // cols: column sums           -> (per channel) mean
    double* colB = smd + cols;                  // cols: column sums of squares -> (per channel) 1 / sqrt(var + eps)
    double* red = smd + 2 * cols;               // 2 * 256: partial sums of the thread groups that share a column
    float* sc = reinterpret_cast<float*>(red + 512);     // cols: scale per column
    float* sh = sc + cols;                               // cols: shift per column
    const int t = threadIdx.x;
    // (1) column totals, fixed order
    if (cols >= 256) {
        for (int c = t; c < cols; c += 256) {
            double a = 0.0, b = 0.0;
            for (int r = 0; r < prow; ++r) {
                a += part[(size_t)r * cols + c];
                b += part[(size_t)(prow + r) * cols + c];
            }
            colA[c] = a;
            colB[c] = b;
        }
    } else {
        const int G = 256 / cols;               // thread groups per column: group `sub` takes partial rows sub, sub + G, ...
        const int c = t % cols, sub = t / cols;
        double a = 0.0, b = 0.0;
        if (sub < G)
            for (int r = sub; r < prow; r += G) {
                a += part[(size_t)r * cols + c];
                b += part[(size_t)(prow + r) * cols + c];
            }
        red[t] = a;
        red[256 + t] = b;
        __syncthreads();
        if (t < cols) {
            for (int j = 1; j < G; ++j) {       // ascending group order
                a += red[j * cols + t];
                b += red[256 + j * cols + t];
            }
            colA[t] = a;
            colB[t] = b;
        }
    }
    __syncthreads();
    // (2) per channel: statistics (columns of a channel in ascending order), running estimates by workgroup 0
    double* chm = red;                          // nchan <= 256: mean, then invstd
    double* chr = red + 256;
    for (int ch = t; ch < nchan; ch += 256) {
        double a = 0.0, b = 0.0, n_cols = 0.0;
        if (chan_of_col) {
            for (int c = 0; c < cols; ++c)
                if (chan_of_col[c] == ch) {
                    a += colA[c];
                    b += colB[c];
                    n_cols += 1.0;
                }
        } else {
            a = colA[ch];
            b = colB[ch];
            n_cols = 1.0;
        }
        const double n = n_cols * (double)rows;
        const double mean = a / n;
        double var = b / n - mean * mean;       // fp64: safe against cancellation
        var = var < 0.0 ? 0.0 : var;
        if (blockIdx.x == 0) {
            const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
            float rm = rmean[ch], rv = rvar[ch];
            for (int it = 0; it < repeat; ++it) {
                rm = (float)((1.0 - (double)momentum) * (double)rm + (double)momentum * mean);
                rv = (float)((1.0 - (double)momentum) * (double)rv + (double)momentum * unbiased);
            }
            rmean[ch] = rm;
            rvar[ch] = rv;
        }
        chm[ch] = mean;
        chr[ch] = 1.0 / sqrt(var + (double)eps);
    }
    if (blockIdx.x == 0 && t == 0 && nbt) *nbt += repeat;
    __syncthreads();
    for (int c = t; c < cols; c += 256) {
        const int ch = chan_of_col ? chan_of_col[c] : c;
        const float mean = (float)chm[ch], invstd = (float)chr[ch];
        const float s = gamma[ch] * invstd;
        const float h = beta[ch] - mean * s;
        sc[c] = s;
        sh[c] = h;
        if (blockIdx.x == 0) {                  // what the backward pass reads (_BNAct.backward: coef[0..3])
            scale_col[c] = s;
            shift_col[c] = h;
            mean_col[c] = mean;
            invstd_col[c] = invstd;
        }
    }
    __syncthreads();
    // (3) y = leaky(x * scale + shift), grid-stride
    const long long total = (long long)rows * cols;
    for (long long i = (long long)blockIdx.x * 256 + t; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
        y[(long long)r * ldy + c] = leaky(x[(long long)r * ldx + c] * sc[c] + sh[c], slope);
    }
}
}  // namespace

extern "C" int s2ag_bn_fold_apply_supported(int partial_rows, int cols, int nchan) {
    return partial_rows > 0 && cols > 0 && cols <= 1024 && nchan > 0 && nchan <= 256 && (long long)partial_rows * cols <= 16384;
}

extern "C" int s2ag_bn_fold_apply(const double* partials, int partial_rows, int rows, int cols, const int* chan_of_col, int nchan,
                                  const float* gamma, const float* beta, float* running_mean, float* running_var,
                                  long long* nbt, float eps, float momentum, int repeat, float* scale_col, float* shift_col,
                                  float* mean_col, float* invstd_col, const float* x, int ldx, float slope, float* y, int ldy,
                                  void* stream) {
    if (!partials || rows <= 0 || repeat < 1 || !gamma || !beta || !running_mean || !running_var || !scale_col || !shift_col ||
        !mean_col || !invstd_col || !x || !y || ldx < cols || ldy < cols)
        return S2AG_E_BADARG;
    if (!s2ag_bn_fold_apply_supported(partial_rows, cols, nchan)) return S2AG_E_UNSUPPORTED;
    const long long total = (long long)rows * cols;
    long long blocks = (total + 256 * 8 - 1) / (256 * 8);       // >= 8 elements per thread, at most 64 workgroups re-read the sums
    if (blocks > 64) blocks = 64;
    if (blocks < 1) blocks = 1;
    const size_t smem = sizeof(double) * (2 * (size_t)cols + 512) + sizeof(float) * 2 * (size_t)cols;
    hipLaunchKernelGGL(bn_fold_apply_k, dim3((unsigned)blocks), dim3(256), smem, (hipStream_t)stream, partials, partial_rows,
                       rows, cols, chan_of_col, nchan, gamma, beta, running_mean, running_var, nbt, eps, momentum, repeat,
                       scale_col, shift_col, mean_col, invstd_col, x, ldx, slope, y, ldy);
    S2AG_LAUNCH_CHECK();
    return 0;
}
