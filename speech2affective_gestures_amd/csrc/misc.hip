// Embedding gather/scatter, weight-norm, CSR fold, transpose, re-parametrisation, fused GAN losses,
// flat-arena Adam and the noise materialisers.
#include <stdlib.h>

#include "s2ag_common.h"
#include <string.h>

namespace s2ag {                                        // csrc/emb_rows.hip: the opt-in 256-row embedding backward (option EMB_BWD_ROWS)
int embedding_bwd_rows_launch(const long long* ids, const float* g, int ldg, int rows, int dim, int n_entries, float* dtable,
                              float drop_p, const unsigned long long* rng, unsigned site, hipStream_t st);
}

namespace {
using namespace s2ag;

inline int ew_grid(long long total) {
    long long b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- embedding -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embedding_fwd_k(const long long* ids, const float* __restrict__ table, int rows,
                                                       int dim, int n_entries, float* __restrict__ out, int ldo,
                                                       float drop_p, float inv_keep, const unsigned long long* rng,
                                                       unsigned site) {
    const long long total = (long long)rows * dim;
    SiteKey key{0, 0};
    if (drop_p > 0.f) key = site_key(rng, site);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / dim), c = (int)(i - (long long)r * dim);
        long long id = ids[r];
        S2AG_DBG_ASSERT(id >= 0 && id < n_entries);      // nn.Embedding raises here; the release build clamps
        id = id < 0 ? 0 : (id >= n_entries ? n_entries - 1 : id);
        float v = table[id * dim + c];
        if (drop_p > 0.f) v *= keep_scale(key, (unsigned long long)i, drop_p, inv_keep);
        out[(long long)r * ldo + c] = v;
    }
}

// Rows that repeat an id (the PAD token fills most of every transcript) would hammer the same `dim` addresses with
// atomics -- 110 us per step.  A block walks RB consecutive rows with one thread per column and merges RUNS of equal
// ids in a register: one atomic per (run, column) instead of one per (row, column).  Exact for any id pattern.
constexpr int EMB_RB = 32;
__global__ __launch_bounds__(256) void embedding_bwd_k(const long long* ids, const float* __restrict__ g, int ldg,
                                                       int rows, int dim, int n_entries, float* dtable, float drop_p,
                                                       float inv_keep, const unsigned long long* rng, unsigned site) {
    SiteKey key{0, 0};
    if (drop_p > 0.f) key = site_key(rng, site);
    const int r0 = blockIdx.x * EMB_RB;
    const int r1 = min(rows, r0 + EMB_RB);
    s2ag::det_enter();          // (det flavour: row blocks share table rows -- workgroups in index order; a thread owns its column)
    for (int c = blockIdx.y * 256 + threadIdx.x; c < dim; c += gridDim.y * 256) {
        long long run_id = -1;
        float acc = 0.f;
        for (int r = r0; r < r1; ++r) {
            long long id = ids[r];
            id = id < 0 ? 0 : (id >= n_entries ? n_entries - 1 : id);
            float v = g[(long long)r * ldg + c];
            if (drop_p > 0.f) v *= keep_scale(key, (unsigned long long)r * dim + c, drop_p, inv_keep);
            if (id != run_id) {
                if (run_id >= 0) atomicAdd(dtable + run_id * dim + c, acc);
                run_id = id;
                acc = 0.f;
            }
            acc += v;
        }
        if (run_id >= 0) atomicAdd(dtable + run_id * dim + c, acc);
    }
    s2ag::det_leave();
}

// ---- weight norm: one wave per output row ------------------------------------------------------------
// column c of v's row (reference order: channel-major, tap fastest) -> column of the tap-major image
__device__ __forceinline__ int tm_col(int c, int cin, int ks) {
    if (ks <= 1) return c;
    const int ci = c / ks, tap = c - ci * ks;
    return tap * cin + ci;
}

__global__ __launch_bounds__(256) void weight_norm_fwd_k(const float* __restrict__ v, const float* __restrict__ g,
                                                         int rows, int cols, int ks, float* __restrict__ w,
                                                         float* norm) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* vr = v + (long long)row * cols;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += vr[c] * vr[c];
    s = wave_sum(s);
    const float nrm = sqrtf(s);
    const float f = g[row] / nrm;
    const int cin = ks > 1 ? cols / ks : cols;
    for (int c = lane; c < cols; c += 64) w[(long long)row * cols + tm_col(c, cin, ks)] = vr[c] * f;
    if (lane == 0) norm[row] = nrm;
}

__global__ __launch_bounds__(256) void weight_norm_bwd_k(const float* __restrict__ dw, const float* __restrict__ v,
                                                         const float* __restrict__ g, const float* __restrict__ norm,
                                                         int rows, int cols, int ks, float* __restrict__ dv,
                                                         float* dg) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* vr = v + (long long)row * cols;
    const float* dr = dw + (long long)row * cols;
    const int cin = ks > 1 ? cols / ks : cols;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += dr[tm_col(c, cin, ks)] * vr[c];
    s = wave_sum(s);
    const float nrm = norm[row], gg = g[row];
    const float dgv = s / nrm;
    const float a = gg / nrm, b = gg * s / (nrm * nrm * nrm);
    for (int c = lane; c < cols; c += 64) dv[(long long)row * cols + c] = a * dr[tm_col(c, cin, ks)] - b * vr[c];
    if (lane == 0) dg[row] = dgv;
}

// ---- CSR sparse linear map -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spmv_k(const int* __restrict__ rowptr, const int* __restrict__ col,
                                              const float* __restrict__ val, const float* __restrict__ x, float* y,
                                              int nrows, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    float s = 0.f;
    for (int j = rowptr[i]; j < rowptr[i + 1]; ++j) s += val[j] * x[col[j]];
    y[i] = accumulate ? y[i] + s : s;
}

__global__ void transpose_k(const float* __restrict__ src, int rows, int cols, float* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: ty 0..7
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? src[(long long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (r < rows && c < cols) dst[(long long)c * rows + r] = tile[tx][j];
    }
}

// ---- derived parameters: several maps per launch ------------------------------------------------------------
// The ST-GCN blocks and the weight-normed TCN convs consume tensors DERIVED from the trainable ones (folded / normalised
// weights).  They are recomputed once per optimizer step (not once per forward pass) and a whole block's worth per
// launch; their gradients are staged in persistent zero-on-entry buffers that the conv weight-gradient kernels
// accumulate into, and one "flush" launch routes the staged gradients to the trainable tensors and leaves the stage
// zeroed for the next step.
constexpr int MAX_JOBS = S2AG_MAX_JOBS;
struct SpmvJobs {
    s2ag_spmv_job j[MAX_JOBS];
    int first_block[MAX_JOBS + 1];
    int n;
};

__device__ __forceinline__ int find_job(const int* first_block, int n) {
    int k = 0;
    while (k + 1 < n && (int)blockIdx.x >= first_block[k + 1]) ++k;
    return k;
}

__global__ __launch_bounds__(256) void spmv_multi_k(SpmvJobs js) {
    const int k = find_job(js.first_block, js.n);
    const s2ag_spmv_job& J = js.j[k];
    const int i = ((int)blockIdx.x - js.first_block[k]) * 256 + threadIdx.x;
    if (i >= J.nrows) return;
    float s = 0.f;
    for (int q = J.rowptr[i]; q < J.rowptr[i + 1]; ++q) s += J.val[q] * J.x[J.col[q]];
    J.y[i] = s;
}

// transpose of spmv_multi_k in scatter form: row i of the map takes the staged gradient y[i], clears it, and adds
// val * y[i] to the gradient of every source element of the row (few per row; atomics because rows share sources)
__global__ __launch_bounds__(256) void spmv_multi_flush_k(SpmvJobs js) {
    const int k = find_job(js.first_block, js.n);
    const s2ag_spmv_job& J = js.j[k];
    const int i = ((int)blockIdx.x - js.first_block[k]) * 256 + threadIdx.x;
#if defined(S2AG_DET) && S2AG_DET      // det flavour: no early return (every thread reaches the workgroup-wide ordering points)
    const float g = i < J.nrows ? J.y[i] : 0.f;
    if (g != 0.f) J.y[i] = 0.f;
    s2ag::det_enter();                 // rows share sources -- workgroups in index order, waves in order
    S2AG_DET_WAVES_BEGIN
        if (g != 0.f)
            for (int q = J.rowptr[i]; q < J.rowptr[i + 1]; ++q) atomicAdd(J.x + J.col[q], J.val[q] * g);
    S2AG_DET_WAVES_END
    s2ag::det_leave();
#else
    if (i >= J.nrows) return;
    const float g = J.y[i];
    if (g == 0.f) return;
    J.y[i] = 0.f;
    for (int q = J.rowptr[i]; q < J.rowptr[i + 1]; ++q) atomicAdd(J.x + J.col[q], J.val[q] * g);
#endif
}

struct WnJobs {
    s2ag_wn_job j[MAX_JOBS];
    int first_block[MAX_JOBS + 1];
    int n;
};

__global__ __launch_bounds__(256) void weight_norm_multi_fwd_k(WnJobs js) {
    const int k = find_job(js.first_block, js.n);
    const s2ag_wn_job& J = js.j[k];
    const int row = ((int)blockIdx.x - js.first_block[k]) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= J.rows) return;
    const float* vr = J.v + (long long)row * J.cols;
    float s = 0.f;
    for (int c = lane; c < J.cols; c += 64) s += vr[c] * vr[c];
    s = wave_sum(s);
    const float nrm = sqrtf(s);
    const float f = J.g[row] / nrm;
    const int cin = J.ksize > 1 ? J.cols / J.ksize : J.cols;
    for (int c = lane; c < J.cols; c += 64) J.w[(long long)row * J.cols + tm_col(c, cin, J.ksize)] = vr[c] * f;
    if (lane == 0) J.norm[row] = nrm;
}

// dv += ..., dg += ... from the staged dw, which is cleared behind the second read (a row belongs to one wave and every
// lane re-reads exactly the elements it read first)
__global__ __launch_bounds__(256) void weight_norm_multi_flush_k(WnJobs js) {
    const int k = find_job(js.first_block, js.n);
    const s2ag_wn_job& J = js.j[k];
    const int row = ((int)blockIdx.x - js.first_block[k]) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= J.rows) return;
    const float* vr = J.v + (long long)row * J.cols;
    float* dr = J.dw + (long long)row * J.cols;
    const int cin = J.ksize > 1 ? J.cols / J.ksize : J.cols;
    float s = 0.f;
    for (int c = lane; c < J.cols; c += 64) s += dr[tm_col(c, cin, J.ksize)] * vr[c];
    s = wave_sum(s);
    const float nrm = J.norm[row], gg = J.g[row];
    const float a = gg / nrm, b = gg * s / (nrm * nrm * nrm);
    for (int c = lane; c < J.cols; c += 64) {
        const int t = tm_col(c, cin, J.ksize);
        J.dv[(long long)row * J.cols + c] += a * dr[t] - b * vr[c];
        dr[t] = 0.f;
    }
    if (lane == 0) J.dg[row] += s / nrm;
}

// ---- re-parametrisation ----------------------------------------------------------------------------------
__global__ void reparam_fwd_k(const float* mu, const float* lv, int n, const unsigned long long* rng, unsigned site,
                              float* z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SiteKey key = site_key(rng, site);
    z[i] = mu[i] + normal_dev(key, i) * expf(0.5f * lv[i]);
}

__global__ void reparam_bwd_k(const float* dz, const float* lv, int n, const unsigned long long* rng, unsigned site,
                              float* dmu, float* dlv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SiteKey key = site_key(rng, site);
    const float d = dz[i];
    dmu[i] += d;
    dlv[i] += d * normal_dev(key, i) * 0.5f * expf(0.5f * lv[i]);
}

// ---- losses ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += sm[i];
    return t;
}

__global__ __launch_bounds__(256) void dis_loss_k(const float* dr, const float* df, int B, float* loss, float* gr,
                                                  float* gf) {
    __shared__ float sm[4];
    float s = 0.f;
    const float invB = 1.f / (float)B;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {      // either half may be absent (the two terms are separable)
        if (dr) {
            const float a = dr[i] + 1e-8f;
            s += logf(a);
            gr[i] = -invB / a;
        }
        if (df) {
            const float b = 1.f - df[i] + 1e-8f;
            s += logf(b);
            gf[i] = invB / b;
        }
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) loss[0] = -s * invB;
}

__device__ __forceinline__ float smooth_l1(float d) {
    const float a = fabsf(d);
    return a < 1.f ? 0.5f * d * d : a - 0.5f;
}

// one block per sample: partial sums into scratch[b*8 + {huber, pose_l1, z_l1, div, kld, gen, l1, l1tri}]
__global__ __launch_bounds__(256) void gen_loss_partial_k(const float* out, const float* target, const float* out_tri,
                                                          const float* dis_out, const float* out_rand, const float* z,
                                                          const float* z_rand, const float* mu, const float* lv,
                                                          int TP, int ZD, float* scratch) {
    __shared__ float sm[4];
    const int b = blockIdx.x;
    const float* o = out + (long long)b * TP;
    const float* t = target + (long long)b * TP;
    // out_rand == nullptr: the branch without the regulariser (processor_v2.py:933-934) -- the divergence and KLD terms are
    // not evaluated at all (the reference never forms exp(z_log_var) there), z / z_rand / mu / lv are not read
    const float* rr = out_rand ? out_rand + (long long)b * TP : nullptr;
    const float* tri = out_tri ? out_tri + (long long)b * TP : nullptr;
    float hub = 0.f, pl1 = 0.f, l1 = 0.f, l1t = 0.f;
    for (int i = threadIdx.x; i < TP; i += blockDim.x) {
        const float ov = o[i], tv = t[i];
        hub += smooth_l1((ov - tv) * 10.f);
        if (rr) pl1 += smooth_l1((ov - rr[i]) * 20.f) * 0.05f;
        l1 += fabsf(ov - tv);
        if (tri) l1t += fabsf(tri[i] - tv);
    }
    float zl = 0.f, kl = 0.f;
    if (rr)
        for (int i = threadIdx.x; i < ZD; i += blockDim.x) {
            zl += fabsf(z[b * ZD + i] - z_rand[b * ZD + i]);
            const float m = mu[b * ZD + i], l = lv[b * ZD + i];
            kl += 1.f + l - m * m - expf(l);
        }
    hub = block_sum(hub, sm);
    pl1 = block_sum(pl1, sm);
    l1 = block_sum(l1, sm);
    l1t = block_sum(l1t, sm);
    zl = block_sum(zl, sm);
    kl = block_sum(kl, sm);
    if (threadIdx.x == 0) {
        float* s = scratch + (long long)b * 8;
        const float zmean = zl / (float)ZD;
        float div = rr ? -(pl1 / (zmean + 1.0e-5f)) : 0.f;
        s[0] = hub;
        s[1] = pl1;
        s[2] = zmean;
        s[3] = div;
        s[4] = kl;
        s[5] = logf(dis_out[b] + 1e-8f);
        s[6] = l1;
        s[7] = l1t;
    }
}

__global__ __launch_bounds__(256) void gen_loss_final_k(const float* scratch, int B, int TP, int ZD, float w_reg,
                                                        float w_gan, float w_div, float w_kld, float* comps) {
    __shared__ float sm[4];
    float hub = 0.f, div = 0.f, kl = 0.f, gen = 0.f, l1 = 0.f, l1t = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float* s = scratch + (long long)b * 8;
        hub += s[0];
        div += fmaxf(s[3], -1000.f);
        kl += s[4];
        gen += s[5];
        l1 += s[6];
        l1t += s[7];
    }
    hub = block_sum(hub, sm);
    div = block_sum(div, sm);
    kl = block_sum(kl, sm);
    gen = block_sum(gen, sm);
    l1 = block_sum(l1, sm);
    l1t = block_sum(l1t, sm);
    if (threadIdx.x == 0) {
        const float n = (float)B * (float)TP;
        const float huber = 0.1f * hub / n;
        const float gen_error = -gen / (float)B;
        const float div_reg = div / (float)B;
        const float kld = -0.5f * kl / ((float)B * (float)ZD);
        comps[0] = w_reg * huber + w_kld * kld + w_div * div_reg + w_gan * gen_error;
        comps[1] = huber;
        comps[2] = gen_error;
        comps[3] = div_reg;
        comps[4] = kld;
        comps[5] = l1 / n;
        comps[6] = l1t / n;
        comps[7] = 0.f;
    }
}

__global__ __launch_bounds__(256) void gen_loss_grad_k(const float* out, const float* target, const float* dis_out,
                                                       const float* out_rand, const float* mu, const float* lv,
                                                       const float* scratch, int B, int TP, int ZD, float w_reg,
                                                       float w_gan, float w_div, float w_kld, float* g_out,
                                                       float* g_dis, float* g_mu, float* g_lv) {
    const int b = blockIdx.x;
    const float* s = scratch + (long long)b * 8;
    const float n = (float)B * (float)TP;
    // d div_b / d out = -(1/(zmean+1e-5)) * clamp((out-rand)/0.05, -1, 1), unless clamped at -1000
    const float dcoef = (out_rand && s[3] >= -1000.f) ? -(w_div / (float)B) / (s[2] + 1.0e-5f) : 0.f;
    for (int i = threadIdx.x; i < TP; i += blockDim.x) {
        const long long k = (long long)b * TP + i;
        const float ov = out[k];
        const float h = fminf(fmaxf((ov - target[k]) * 10.f, -1.f), 1.f);
        const float d = out_rand ? fminf(fmaxf((ov - out_rand[k]) * 20.f, -1.f), 1.f) : 0.f;
        g_out[k] = w_reg * h / n + dcoef * d;
    }
    if (out_rand)      // no regulariser: no KLD term, mu / log_var receive no gradient from the loss (g_mu / g_lv may be NULL)
        for (int i = threadIdx.x; i < ZD; i += blockDim.x) {
            const float m = mu[b * ZD + i], l = lv[b * ZD + i];
            const float c = w_kld * (-0.5f) / ((float)B * (float)ZD);
            g_mu[b * ZD + i] = c * (-2.f * m);
            g_lv[b * ZD + i] = c * (1.f - expf(l));
        }
    if (threadIdx.x == 0) g_dis[b] = -w_gan / ((float)B * (dis_out[b] + 1e-8f));
}

// ---- Adam over a flat arena ------------------------------------------------------------------------------------
// Sticky error word of the step (cooperative GRU time-outs, one-launch BatchNorm time-outs, touched-row overflow): when it
// is non-zero the gradients of this step were formed from wrong activations, so the optimizer leaves weights, moments and
// step count alone -- on the device, without a host round trip; the trainer raises at its next read-back
// (ops.check_coop_flag) with the weights still those of the last good step.
__device__ const int* g_adam_guard = nullptr;

__global__ __launch_bounds__(256) void adam_k(float* __restrict__ p, const float* __restrict__ g,
                                              float* __restrict__ m, float* __restrict__ v, long long n, float lr,
                                              float b1, float b2, float eps, const int* step, float gscale) {
    if (g_adam_guard && __hip_atomic_load(g_adam_guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    const float t = (float)(*step);
    const float bc1 = 1.f - powf(b1, t);
    const float bc2s = sqrtf(1.f - powf(b2, t));
    const float step_size = lr / bc1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const float gv = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gv;
        const float vi = b2 * v[i] + (1.f - b2) * gv * gv;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) / bc2s + eps);
    }
}

__global__ void counter_inc_k(int* counter, unsigned long long* rng) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        // an optimizer's step count (counter without rng) stands still while the step guard is raised, like adam_k
        const bool held = counter && !rng && g_adam_guard &&
                          __hip_atomic_load(g_adam_guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        if (counter && !held) *counter += 1;
        if (rng) rng[1] += 1ULL;
    }
}

// snapshot of the noise state for one forward pass + advance of the pass counter, in ONE one-thread launch (was: a
// 16-byte ATen clone + counter_inc_k, ten times per step)
__global__ void rng_snapshot_k(unsigned long long* rng, unsigned long long* snap) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned long long s = rng[0], c = rng[1];
        snap[0] = s;
        snap[1] = c;
        rng[1] = c + 1ULL;
    }
}

// the snapshots of SEVERAL consecutive passes (and of passes further down the step's pass order: snapshot 0 with the
// counter advanced by off[j]) in one launch -- a trainer phase drew them with 3-4 launches plus clone + add pairs
struct SnapOff { int off[8]; };
__global__ void rng_snapshots_k(unsigned long long* rng, unsigned long long* out, int n, int n_extra, SnapOff o) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned long long s = rng[0], c = rng[1];
        for (int i = 0; i < n; ++i) {
            out[2 * i] = s;
            out[2 * i + 1] = c + (unsigned long long)i;
        }
        for (int j = 0; j < n_extra; ++j) {
            out[2 * (n + j)] = s;
            out[2 * (n + j) + 1] = c + (unsigned long long)o.off[j];
        }
        rng[1] = c + (unsigned long long)n;
    }
}

// pre_seq (B, T, D + 1): the first n_pre frames of the target poses with a 1 in the extra column, zero elsewhere
__global__ __launch_bounds__(256) void pre_seq_k(const float* target, float* pre, long long rows, int T, int D, int n_pre) {
    const long long n = rows * (D + 1);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / (D + 1);
        const int c = (int)(i - row * (D + 1));
        const bool on = (int)(row % T) < n_pre;
        pre[i] = on ? (c < D ? target[row * D + c] : 1.f) : 0.f;
    }
}

// out[row, :] = [src0[row] | src1[row] | ...]; a source with per_clip != 0 has one row per clip (row / T): the speaker
// code z broadcast over the frames of its clip
struct CatSrc { const float* p; int cols, ld, per_clip; };
struct CatP { CatSrc s[4]; int n; };
__global__ __launch_bounds__(256) void concat_cols_k(CatP c, float* out, long long rows, int T, int total) {
    const long long n = rows * total;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / total;
        int col = (int)(i - row * total);
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < c.n) {
                if (col >= 0 && col < c.s[k].cols) v = c.s[k].p[(c.s[k].per_clip ? row / T : row) * c.s[k].ld + col];
                col -= c.s[k].cols;
            }
        }
        out[i] = v;
    }
}

// dz[b, c] = sum_t g[(b, t), col0 + c]: gradient of the broadcast source above
__global__ __launch_bounds__(64) void sum_frames_k(const float* g, int ldg, int col0, int cols, int T, float* dz) {
    const int b = blockIdx.x, c = threadIdx.x;
    if (c >= cols) return;
    const float* q = g + (long long)b * T * ldg + col0 + c;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += q[(long long)t * ldg];
    dz[(long long)b * cols + c] = s;
}

__global__ __launch_bounds__(256) void dropout_mask_k(const unsigned long long* rng, unsigned site, float p,
                                                      float inv_keep, long long n, float* mask) {
    const SiteKey key = site_key(rng, site);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        mask[i] = keep_scale(key, (unsigned long long)i, p, inv_keep);
}

__global__ __launch_bounds__(256) void normal_noise_k(const unsigned long long* rng, unsigned site, long long n,
                                                      float* eps) {
    const SiteKey key = site_key(rng, site);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x)
        eps[i] = normal_dev(key, (unsigned long long)i);
}
}  // namespace

extern "C" int s2ag_embedding_fwd(const long long* ids, const float* table, int rows, int dim, int n_entries,
                                  float* out, int ldo, const s2ag_epilogue* e, void* stream) {
    if (!ids || !table || !out || rows <= 0 || dim <= 0 || n_entries <= 0 || ldo < dim) return S2AG_E_BADARG;
    const float p = e ? e->drop_p : 0.f;
    if (p > 0.f && !e->rng) return S2AG_E_BADARG;
    hipLaunchKernelGGL(embedding_fwd_k, dim3(ew_grid((long long)rows * dim)), dim3(256), 0, (hipStream_t)stream, ids,
                       table, rows, dim, n_entries, out, ldo, p, p > 0.f ? 1.f / (1.f - p) : 1.f,
                       e ? e->rng : nullptr, e ? e->site : 0u);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_embedding_bwd(const long long* ids, const float* g, int ldg, int rows, int dim, int n_entries,
                                  float* dtable, int accumulate, const s2ag_epilogue* e, void* stream) {
    if (!ids || !g || !dtable || rows <= 0 || dim <= 0 || n_entries <= 0 || ldg < dim) return S2AG_E_BADARG;
    const float p = e ? e->drop_p : 0.f;
    if (p > 0.f && !e->rng) return S2AG_E_BADARG;
    if (!accumulate) {
        hipError_t me = zero_async(dtable, sizeof(float) * (size_t)n_entries * dim, (hipStream_t)stream);
        if (me != hipSuccess) return (int)me;
    }
    if (s2ag::option(s2ag::OPT_EMB_BWD_ROWS))       // opt-in variant (csrc/emb_rows.hip): 256-row blocks, the PAD row summed in registers
        return s2ag::embedding_bwd_rows_launch(ids, g, ldg, rows, dim, n_entries, dtable, p, e ? e->rng : nullptr,
                                               e ? e->site : 0u, (hipStream_t)stream);
    hipLaunchKernelGGL(embedding_bwd_k, dim3(cdiv(rows, EMB_RB), cdiv(dim, 256)), dim3(256), 0, (hipStream_t)stream, ids,
                       g, ldg, rows, dim, n_entries, dtable, p, p > 0.f ? 1.f / (1.f - p) : 1.f,
                       e ? e->rng : nullptr, e ? e->site : 0u);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_weight_norm_fwd(const float* v, const float* g, int rows, int cols, int ksize, float* w, float* norm,
                                    void* stream) {
    if (!v || !g || !w || !norm || rows <= 0 || cols <= 0 || (ksize > 1 && cols % ksize)) return S2AG_E_BADARG;
    hipLaunchKernelGGL(weight_norm_fwd_k, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, v, g, rows, cols,
                       ksize, w, norm);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_weight_norm_bwd(const float* dw, const float* v, const float* g, const float* norm, int rows,
                                    int cols, int ksize, float* dv, float* dg, void* stream) {
    if (!dw || !v || !g || !norm || !dv || !dg || rows <= 0 || cols <= 0 || (ksize > 1 && cols % ksize))
        return S2AG_E_BADARG;
    hipLaunchKernelGGL(weight_norm_bwd_k, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, dw, v, g, norm, rows,
                       cols, ksize, dv, dg);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_spmv(const int* rowptr, const int* col, const float* val, const float* x, float* y, int nrows,
                         int accumulate, void* stream) {
    if (!rowptr || !col || !val || !x || !y || nrows <= 0) return S2AG_E_BADARG;
    hipLaunchKernelGGL(spmv_k, dim3(cdiv(nrows, 256)), dim3(256), 0, (hipStream_t)stream, rowptr, col, val, x, y,
                       nrows, accumulate);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_spmv_multi(const s2ag_spmv_job* jobs, int njobs, int flush, void* stream) {
    if (!jobs || njobs <= 0 || njobs > MAX_JOBS) return S2AG_E_BADARG;
    SpmvJobs js;
    js.n = njobs;
    int nb = 0;
    for (int k = 0; k < njobs; ++k) {
        if (!jobs[k].rowptr || !jobs[k].col || !jobs[k].val || !jobs[k].x || !jobs[k].y || jobs[k].nrows <= 0)
            return S2AG_E_BADARG;
        js.j[k] = jobs[k];
        js.first_block[k] = nb;
        nb += cdiv(jobs[k].nrows, 256);
    }
    js.first_block[njobs] = nb;
    if (flush)
        hipLaunchKernelGGL(spmv_multi_flush_k, dim3(nb), dim3(256), 0, (hipStream_t)stream, js);
    else
        hipLaunchKernelGGL(spmv_multi_k, dim3(nb), dim3(256), 0, (hipStream_t)stream, js);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_weight_norm_multi(const s2ag_wn_job* jobs, int njobs, int flush, void* stream) {
    if (!jobs || njobs <= 0 || njobs > MAX_JOBS) return S2AG_E_BADARG;
    WnJobs js;
    js.n = njobs;
    int nb = 0;
    for (int k = 0; k < njobs; ++k) {
        const s2ag_wn_job& J = jobs[k];
        if (!J.v || !J.g || !J.norm || J.rows <= 0 || J.cols <= 0 || (J.ksize > 1 && J.cols % J.ksize))
            return S2AG_E_BADARG;
        if (flush ? (!J.dw || !J.dv || !J.dg) : !J.w) return S2AG_E_BADARG;
        js.j[k] = J;
        js.first_block[k] = nb;
        nb += cdiv(J.rows, 4);
    }
    js.first_block[njobs] = nb;
    if (flush)
        hipLaunchKernelGGL(weight_norm_multi_flush_k, dim3(nb), dim3(256), 0, (hipStream_t)stream, js);
    else
        hipLaunchKernelGGL(weight_norm_multi_fwd_k, dim3(nb), dim3(256), 0, (hipStream_t)stream, js);
    S2AG_LAUNCH_CHECK();
    return 0;
}

// ---- device-side wall clock (100 MHz) for scheduling diagnostics inside captured graphs ------------------------
namespace {
// Batch decode on the device (processor_v2.py:603-614 do it on the host before the copy): the raw int16 waveform and
// its per-clip peak cross PCIe (half the bytes of the decoded fp32), out = float(double(a) * peak / 32767) -- the
// reference's float64 arithmetic, so the result is bit-identical to its host path.
__global__ __launch_bounds__(256) void audio_decode_k(const short* __restrict__ a, const double* __restrict__ peak,
                                                      float* __restrict__ out, int rows, int cols) {
    const long long total = (long long)rows * cols;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / cols);
        out[i] = (float)((double)a[i] * peak[r] / 32767.0);
    }
}
// vec_seq float64 -> float32 (.float()), mfcc float16 -> float32
__global__ __launch_bounds__(256) void f64_to_f32_k(const double* __restrict__ x, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = (float)x[i];
}
__global__ __launch_bounds__(256) void f16_to_f32_k(const _Float16* __restrict__ x, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        out[i] = (float)x[i];
}

__global__ void timestamp_k(unsigned long long* out) { *out = wall_clock64(); }
}  // namespace

extern "C" int s2ag_audio_decode(const short* audio_i16, const double* peak, float* out, int rows, int cols,
                                 void* stream) {
    if (!audio_i16 || !peak || !out || rows <= 0 || cols <= 0) return S2AG_E_BADARG;
    hipLaunchKernelGGL(audio_decode_k, dim3(ew_grid((long long)rows * cols)), dim3(256), 0, (hipStream_t)stream, audio_i16,
                       peak, out, rows, cols);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_to_f32(const void* x, int src_is_f16, float* out, long long n, void* stream) {
    if (!x || !out || n <= 0) return S2AG_E_BADARG;
    if (src_is_f16)
        hipLaunchKernelGGL(f16_to_f32_k, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const _Float16*>(x), out, n);
    else
        hipLaunchKernelGGL(f64_to_f32_k, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const double*>(x), out, n);
    S2AG_LAUNCH_CHECK();
    return 0;
}

// ---- measurement aid: known-byte-count access patterns for calibrating the FETCH_SIZE / WRITE_SIZE counters -----------
namespace {
__global__ __launch_bounds__(256) void calib_read16_k(const uint4* __restrict__ p, size_t n, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) *sink = acc;
}
// the cooperative GRU's polling load: 8 bytes per lane, agent scope (bypasses the non-coherent caches)
__global__ __launch_bounds__(256) void calib_read8_agent_k(const unsigned long long* p, size_t n, unsigned* sink) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc ^= __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (acc == 0x9e3779b97f4a7c15ull) *sink = 1u;
}
__global__ __launch_bounds__(256) void calib_write16_k(uint4* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        p[i] = uint4{(unsigned)i, 1u, 2u, 3u};
}
__global__ __launch_bounds__(256) void calib_write8_agent_k(unsigned long long* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        __hip_atomic_store(p + i, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace

extern "C" int s2ag_calib_traffic(void* buf, long long bytes, int pattern, void* stream) {
    if (!buf || bytes < 4096 || pattern < 0 || pattern > 3) return S2AG_E_BADARG;
    unsigned* sink = static_cast<unsigned*>(buf);
    const dim3 grid(2048), block(256);
    if (pattern == 0)
        hipLaunchKernelGGL(calib_read16_k, grid, block, 0, (hipStream_t)stream, static_cast<const uint4*>(buf) + 1,
                           (size_t)(bytes - 16) / 16, sink);
    else if (pattern == 1)
        hipLaunchKernelGGL(calib_read8_agent_k, grid, block, 0, (hipStream_t)stream,
                           static_cast<const unsigned long long*>(buf) + 1, (size_t)(bytes - 8) / 8, sink);
    else if (pattern == 2)
        hipLaunchKernelGGL(calib_write16_k, grid, block, 0, (hipStream_t)stream, static_cast<uint4*>(buf), (size_t)bytes / 16);
    else
        hipLaunchKernelGGL(calib_write8_agent_k, grid, block, 0, (hipStream_t)stream,
                           static_cast<unsigned long long*>(buf), (size_t)bytes / 8);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_timestamp(unsigned long long* out, void* stream) {
    if (!out) return S2AG_E_BADARG;
    hipLaunchKernelGGL(timestamp_k, dim3(1), dim3(1), 0, (hipStream_t)stream, out);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_transpose(const float* src, int rows, int cols, float* dst, void* stream) {
    if (!src || !dst || rows <= 0 || cols <= 0) return S2AG_E_BADARG;
    hipLaunchKernelGGL(transpose_k, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, (hipStream_t)stream, src, rows,
                       cols, dst);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_reparam_fwd(const float* mu, const float* log_var, int n, const unsigned long long* rng,
                                unsigned site, float* z, void* stream) {
    if (!mu || !log_var || !rng || !z || n <= 0) return S2AG_E_BADARG;
    hipLaunchKernelGGL(reparam_fwd_k, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, mu, log_var, n, rng, site,
                       z);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_reparam_bwd(const float* dz, const float* log_var, int n, const unsigned long long* rng,
                                unsigned site, float* dmu, float* dlog_var, void* stream) {
    if (!dz || !log_var || !rng || !dmu || !dlog_var || n <= 0) return S2AG_E_BADARG;
    hipLaunchKernelGGL(reparam_bwd_k, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dz, log_var, n, rng, site,
                       dmu, dlog_var);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_dis_loss(const float* d_real, const float* d_fake, int B, float* loss, float* g_real,
                             float* g_fake, void* stream) {
    if ((!d_real && !d_fake) || (d_real && !g_real) || (d_fake && !g_fake) || !loss || B <= 0) return S2AG_E_BADARG;
    hipLaunchKernelGGL(dis_loss_k, dim3(1), dim3(256), 0, (hipStream_t)stream, d_real, d_fake, B, loss, g_real, g_fake);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_gen_loss(const float* out, const float* target, const float* out_tri, const float* dis_out,
                             const float* out_rand, const float* z, const float* z_rand, const float* mu,
                             const float* log_var, int B, int TP, int ZD, const float* weights, float* scratch,
                             float* comps, float* g_out, float* g_dis, float* g_mu, float* g_logvar, void* stream) {
    if (!out || !target || !dis_out || !weights || !scratch || !comps || !g_out || !g_dis || B <= 0 || TP <= 0 || ZD <= 0)
        return S2AG_E_BADARG;
    if (out_rand && (!z || !z_rand || !mu || !log_var || !g_mu || !g_logvar)) return S2AG_E_BADARG;
    // without the regulariser the two weights the reference never reads are not read here either
    const float w_reg = weights[0], w_gan = weights[1], w_div = out_rand ? weights[2] : 0.f, w_kld = out_rand ? weights[3] : 0.f;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gen_loss_partial_k, dim3(B), dim3(256), 0, s, out, target, out_tri, dis_out, out_rand, z, z_rand,
                       mu, log_var, TP, ZD, scratch);
    hipLaunchKernelGGL(gen_loss_final_k, dim3(1), dim3(256), 0, s, scratch, B, TP, ZD, w_reg, w_gan, w_div, w_kld,
                       comps);
    hipLaunchKernelGGL(gen_loss_grad_k, dim3(B), dim3(256), 0, s, out, target, dis_out, out_rand, mu, log_var, scratch,
                       B, TP, ZD, w_reg, w_gan, w_div, w_kld, g_out, g_dis, g_mu, g_logvar);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                              float beta2, float eps, const int* step, float grad_scale, void* stream) {
    if (!p || !g || !m || !v || !step || n <= 0) return S2AG_E_BADARG;
    hipLaunchKernelGGL(adam_k, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2,
                       eps, step, grad_scale);
    S2AG_LAUNCH_CHECK();
    return 0;
}

// ---- evaluation metrics of forward_pass_s2ag(calculate_metrics=True) (processor_v2.py:738-774 push_samples) -------------
// One thread per (clip, frame): the joint positions of frames t, t+1, t+2 of the generated and the target sequence are
// rebuilt from the direction vectors (utils/ted_db_utils.py:81-102 convert_dir_vec_to_pose: joint[child] = joint[parent]
// + length * dir, ten joints, root at the origin) exactly as numpy does it -- dir + mean formed in float64 and rounded to
// float32, length * dir rounded to float32, the chain and everything behind it in float64 -- and three sums leave the
// launch: sum |out - target| (the L1 metric), sum |joint_out - joint_tgt| over frames >= n_pre (joint MAE) and
// sum |acc_tgt - acc_out| of the second differences over time (acceleration difference).
__device__ __forceinline__ void pose_of(const float* __restrict__ v, const double* __restrict__ mean, double (*jp)[3]) {
    constexpr int par[9] = {0, 1, 2, 1, 4, 5, 1, 7, 8}, chi[9] = {1, 2, 3, 4, 5, 6, 7, 8, 9};
    constexpr float len[9] = {0.26f, 0.18f, 0.14f, 0.22f, 0.36f, 0.33f, 0.22f, 0.36f, 0.33f};
    jp[0][0] = jp[0][1] = jp[0][2] = 0.0;
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = (float)((double)v[3 * j + c] + mean[3 * j + c]);
            jp[chi[j]][c] = jp[par[j]][c] + (double)(len[j] * d);
        }
}

__global__ __launch_bounds__(256) void pose_metrics_k(const float* __restrict__ out, const float* __restrict__ tgt,
                                                      const double* __restrict__ mean, int B, int T, int n_pre,
                                                      double* __restrict__ sums) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double s_l1 = 0.0, s_mae = 0.0, s_acc = 0.0;
    if (i < B * T) {
        const int t = i % T;
        const float* o = out + (size_t)i * 27;
        const float* g = tgt + (size_t)i * 27;
        for (int k = 0; k < 27; ++k) s_l1 += (double)fabsf(o[k] - g[k]);
        double po[3][10][3], pg[3][10][3];
        const int nf = t + 2 < T ? 3 : 1;
        for (int f = 0; f < nf; ++f) {
            pose_of(o + 27 * f, mean, po[f]);
            pose_of(g + 27 * f, mean, pg[f]);
        }
        if (t >= n_pre)
            for (int j = 0; j < 10; ++j)
                for (int c = 0; c < 3; ++c) s_mae += fabs(po[0][j][c] - pg[0][j][c]);
        if (nf == 3)
            for (int j = 0; j < 10; ++j)
                for (int c = 0; c < 3; ++c) {
                    // np.diff(n=2): (p[t+2] - p[t+1]) - (p[t+1] - p[t])
                    const double ag = (pg[2][j][c] - pg[1][j][c]) - (pg[1][j][c] - pg[0][j][c]);
                    const double ao = (po[2][j][c] - po[1][j][c]) - (po[1][j][c] - po[0][j][c]);
                    s_acc += fabs(ag - ao);
                }
    }
    __shared__ double red[3][256];
    red[0][threadIdx.x] = s_l1;
    red[1][threadIdx.x] = s_mae;
    red[2][threadIdx.x] = s_acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (threadIdx.x < h)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + h];
        __syncthreads();
    }
    s2ag::det_enter();
    if (threadIdx.x < 3) atomicAdd(&sums[threadIdx.x], red[threadIdx.x][0]);
    s2ag::det_leave();
}

extern "C" int s2ag_pose_metrics(const float* out, const float* target, const double* mean_dir_vec, int B, int T,
                                 int n_pre, double* sums, void* stream) {
    if (!out || !target || !mean_dir_vec || !sums || B < 1 || T < 3 || n_pre < 0 || n_pre >= T) return S2AG_E_BADARG;
    (void)s2ag::zero_async(sums, 3 * sizeof(double), (hipStream_t)stream);
    hipLaunchKernelGGL(pose_metrics_k, dim3(cdiv(B * T, 256)), dim3(256), 0, (hipStream_t)stream, out, target,
                       mean_dir_vec, B, T, n_pre, sums);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_adam_set_guard(const int* flag) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_adam_guard), &flag, sizeof(flag));
}

extern "C" int s2ag_counter_inc(int* counter, unsigned long long* rng, void* stream) {
    hipLaunchKernelGGL(counter_inc_k, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, rng);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_rng_snapshot(unsigned long long* rng, unsigned long long* snap, void* stream) {
    if (!rng || !snap) return S2AG_E_BADARG;
    hipLaunchKernelGGL(rng_snapshot_k, dim3(1), dim3(64), 0, (hipStream_t)stream, rng, snap);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_rng_snapshots(unsigned long long* rng, unsigned long long* out, int n, const int* extra_offsets,
                                  int n_extra, void* stream) {
    if (!rng || !out || n < 1 || n_extra < 0 || n_extra > 8 || (n_extra > 0 && !extra_offsets)) return S2AG_E_BADARG;
    SnapOff o{};
    for (int j = 0; j < n_extra; ++j) o.off[j] = extra_offsets[j];
    hipLaunchKernelGGL(rng_snapshots_k, dim3(1), dim3(64), 0, (hipStream_t)stream, rng, out, n, n_extra, o);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_concat_cols(const float* const* src, const int* cols, const int* ld, const int* per_clip, int n,
                                float* out, long long rows, int T, void* stream) {
    if (!src || !cols || !ld || !per_clip || n < 1 || n > 4 || !out || rows < 1 || T < 1) return S2AG_E_BADARG;
    CatP c{};
    c.n = n;
    int total = 0;
    for (int k = 0; k < n; ++k) {
        if (!src[k] || cols[k] < 1) return S2AG_E_BADARG;
        c.s[k] = CatSrc{src[k], cols[k], ld[k], per_clip[k]};
        total += cols[k];
    }
    hipLaunchKernelGGL(concat_cols_k, dim3(ew_grid(rows * total)), dim3(256), 0, (hipStream_t)stream, c, out, rows, T, total);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_sum_frames(const float* g, int ldg, int col0, int cols, int B, int T, float* dz, void* stream) {
    if (!g || !dz || B < 1 || T < 1 || cols < 1 || cols > 64) return S2AG_E_BADARG;
    hipLaunchKernelGGL(sum_frames_k, dim3(B), dim3(64), 0, (hipStream_t)stream, g, ldg, col0, cols, T, dz);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_make_pre_seq(const float* target, float* pre, int B, int T, int D, int n_pre, void* stream) {
    if (!target || !pre || B < 1 || T < 1 || D < 1 || n_pre < 0) return S2AG_E_BADARG;
    const long long rows = (long long)B * T;
    hipLaunchKernelGGL(pre_seq_k, dim3(ew_grid(rows * (D + 1))), dim3(256), 0, (hipStream_t)stream, target, pre, rows, T, D,
                       n_pre);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_dropout_mask(const unsigned long long* rng, unsigned site, float p, long long n, float* mask,
                                 void* stream) {
    if (!rng || !mask || n <= 0 || p < 0.f || p >= 1.f) return S2AG_E_BADARG;
    hipLaunchKernelGGL(dropout_mask_k, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, rng, site, p,
                       1.f / (1.f - p), n, mask);
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_normal_noise(const unsigned long long* rng, unsigned site, long long n, float* eps, void* stream) {
    if (!rng || !eps || n <= 0) return S2AG_E_BADARG;
    hipLaunchKernelGGL(normal_noise_k, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, rng, site, n, eps);
    S2AG_LAUNCH_CHECK();
    return 0;
}


// ---- run-time options (the registry is speech2affective_gestures_amd/config.py) ------------------------------------------
namespace s2ag {
namespace {
int g_options[OPT_COUNT] = {2, 0, 0, 0};
const char* const g_option_names[OPT_COUNT] = {"GRU_SPLIT", "WGRAD32_PIPE", "TCN32_PAIR", "EMB_BWD_ROWS"};
int option_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (strcmp(name, g_option_names[i]) == 0) return i;
    return -1;
}
}  // namespace
int option(Option o) { return g_options[o]; }
}  // namespace s2ag

extern "C" int s2ag_set_option(const char* name, int value) {
    const int i = s2ag::option_index(name);
    if (i < 0 || value < 0) return S2AG_E_BADARG;
    if (i == s2ag::OPT_GRU_SPLIT && value > 3) return S2AG_E_BADARG;
    const int prev = s2ag::g_options[i];
    s2ag::g_options[i] = value;
    return prev;
}
extern "C" int s2ag_get_option(const char* name) {
    const int i = s2ag::option_index(name);
    return i < 0 ? S2AG_E_BADARG : s2ag::g_options[i];
}

// ---- deterministic mode (s2ag_common.h: det_enter / det_leave / S2AG_DET_WAVES_BEGIN..END) ------------------------------------------
S2AG_DET_HOOK(misc)
extern "C" int s2ag_det_hook_conv_gemm(int*, unsigned*);
extern "C" int s2ag_det_hook_gemm_lin(int*, unsigned*);
extern "C" int s2ag_det_hook_norm_elementwise(int*, unsigned*);
extern "C" int s2ag_det_hook_wgrad_tr(int*, unsigned*);
extern "C" int s2ag_det_hook_conv_bf16(int*, unsigned*);
extern "C" int s2ag_det_hook_conv_c1(int*, unsigned*);
extern "C" int s2ag_det_hook_wgrad_tr32p(int*, unsigned*);
extern "C" int s2ag_det_hook_emb_rows(int*, unsigned*);
extern "C" int s2ag_det_flavour(void) {
#if defined(S2AG_DET) && S2AG_DET
    return 1;
#else
    return 0;
#endif
}
extern "C" int s2ag_set_deterministic(int* zero_device_word, int* error_word) {
    unsigned* e = reinterpret_cast<unsigned*>(error_word);
    int rc = s2ag_det_hook_misc(zero_device_word, e);
    if (!rc) rc = s2ag_det_hook_conv_gemm(zero_device_word, e);
    if (!rc) rc = s2ag_det_hook_gemm_lin(zero_device_word, e);
    if (!rc) rc = s2ag_det_hook_norm_elementwise(zero_device_word, e);
    if (!rc) rc = s2ag_det_hook_wgrad_tr(zero_device_word, e);
    if (!rc) rc = s2ag_det_hook_conv_bf16(zero_device_word, e);
    if (!rc) rc = s2ag_det_hook_conv_c1(zero_device_word, e);
    if (!rc) rc = s2ag_det_hook_wgrad_tr32p(zero_device_word, e);
    if (!rc) rc = s2ag_det_hook_emb_rows(zero_device_word, e);
    return rc;
}
