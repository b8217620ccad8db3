// Direct (vector-ALU) kernels for the ONE-input-channel convolution at the head of the wave encoder
// (net/multimodal_context_net_v2.py:18, nn.Conv1d(1, 16, 15, stride=5, padding=1600)):
//   fwd    y[(n,l), co]  = b[co] + sum_t x[n, l*stride - pad + t] * w[co, t]
//   wgrad  dw[co, t]    += sum_{n,l} gy[(n,l), co] * x[n, l*stride - pad + t],      db[co] += sum_{n,l} gy[(n,l), co]
// As an implicit GEMM this layer has K = 15 and N = 16: a 64 x 64 x 32 MFMA tile is 6 % occupied and the kernel spends its
// time gathering a (row, tap) operand matrix that is just the waveform read 15 times.  It is a memory-bound layer -- it
// writes (fwd) or reads (wgrad) 16 floats per output frame and needs 240 FMAs for them -- so:
//   * a block owns a run of output frames of ONE clip; the waveform segment under that run is staged in LDS once
//     (coalesced, zero where the padding is), so the taps cost LDS reads, not global gathers;
//   * a thread owns (frame, quad of output channels): its 4 x KS weights (fwd) or weight-gradient accumulators (wgrad)
//     stay in registers for all its frames, and the float4 it stores / loads per frame is contiguous with its
//     neighbours' -- every wave-level access of y / gy is one contiguous 1 KB run;
//   * fwd: the fp64 column sums the following BatchNorm needs ride along (same (2, R, Cout) partial layout as the GEMM
//     epilogues, one partial row per wave);
//   * wgrad: per-thread accumulators are folded across the wave by shuffles, across waves in LDS, and one wave adds the
//     block's 16 x KS (+16) sums to dw / db with contiguous atomics; blocks loop over several runs to keep that count low.
// With Cin == 1 the reference (Cout, 1, k) and tap-major (Cout, k, 1) weight layouts coincide.
#include "s2ag_common.h"
#include "bn_fold_inl.h"

namespace {
using namespace s2ag;

constexpr int C1_NT = 256;        // threads per block
constexpr int C1_RUN = 512;       // output frames per run (a block-iteration): 64 frames x 8 per thread

struct C1P {
    const float* x;               // (N, Lin) waveform
    const float* w;               // (CO, KS)
    const float* bias;            // fwd, nullable
    float* y;                     // fwd: (N*Lout, ldy) out;  wgrad: gy (read)
    float* dw;                    // wgrad
    float* db;                    // wgrad, nullable
    double* stats;                // fwd, nullable: (2, R, CO), R = blocks * 4
    int N, Lin, Lout, stride, pad, ldx, ldy;
    int runs_per_clip;            // cdiv(Lout, C1_RUN)
    int total_runs;               // N * runs_per_clip
    // wgrad with the BatchNorm backward folded into the loader (wave_fused.hip): gy = ca * dz + cc * y1 + cb per channel,
    // `y` then holds dz (bf16) and y1 the raw conv output (bf16)
    const unsigned short* y1;
    const float* ca;
    const float* cb;
    const float* cc;
    float* part;                  // wgrad, nullable: (gridDim.x, CO * KS + CO) per-block sums instead of atomics into dw / db
    s2ag_fold::FwdFold fold;      // fwd: fold.ticket != null: ONE partial row per block, folded by the block that finishes last
};

__device__ __forceinline__ unsigned c1_bf16_rn(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

// The waveform under output frames [l0, l0 + C1_RUN) of clip n: seg[i] = x[n, l0*stride - pad + i].  Two halves so that
// the NEXT run's segment is in flight (registers) while the current one is consumed from LDS: a block is a chain of runs,
// and with the load, the barrier and the compute in sequence every run paid a full memory round trip (fwd 48 us, wgrad
// 86 us at B = 256 against ~25 us of traffic).
constexpr int C1_SEG_REGS = 17;   // ceil(((C1_RUN - 1) * stride + KS) / C1_NT) for stride <= 8, KS <= 15 -- checked by the host

__device__ __forceinline__ void fetch_segment(const C1P& p, float (&r)[C1_SEG_REGS], int run, int seg_len) {
    const int n = run / p.runs_per_clip, l0 = (run - n * p.runs_per_clip) * C1_RUN;
    const long long base = (long long)l0 * p.stride - p.pad;
    const float* xc = p.x + (long long)n * p.Lin * p.ldx;
#pragma unroll
    for (int k = 0; k < C1_SEG_REGS; ++k) {
        const int i = threadIdx.x + k * C1_NT;
        const long long pos = base + i;
        r[k] = (i < seg_len && pos >= 0 && pos < p.Lin) ? xc[pos * p.ldx] : 0.f;
    }
}
__device__ __forceinline__ void store_segment(float* seg, const float (&r)[C1_SEG_REGS], int seg_len) {
    S2AG_DBG_ASSERT(seg_len <= C1_SEG_REGS * C1_NT);
#pragma unroll
    for (int k = 0; k < C1_SEG_REGS; ++k) {
        const int i = threadIdx.x + k * C1_NT;
        if (i < seg_len) seg[i] = r[k];
    }
}

template <int CO, int KS, bool BF>
__global__ __launch_bounds__(C1_NT) void conv_c1_fwd_k(const C1P p) {
    static_assert(CO == 16, "thread mapping: 4 channel quads");
    extern __shared__ float seg[];
    const int tid = threadIdx.x, q = tid & 3, fr = tid >> 2;          // channel quad, frame within a 64-frame group
    const int lane = tid & 63, wave = tid >> 6;
    float wr[4][KS], br[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        br[c] = p.bias ? p.bias[q * 4 + c] : 0.f;
#pragma unroll
        for (int t = 0; t < KS; ++t) wr[c][t] = p.w[(q * 4 + c) * KS + t];
    }
    const int seg_len = (C1_RUN - 1) * p.stride + KS;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    float sr[C1_SEG_REGS];
    if ((int)blockIdx.x < p.total_runs) fetch_segment(p, sr, blockIdx.x, seg_len);
    for (int run = blockIdx.x; run < p.total_runs; run += gridDim.x) {
        const int n = run / p.runs_per_clip, l0 = (run - n * p.runs_per_clip) * C1_RUN;
        __syncthreads();                                              // the previous run's readers are done
        store_segment(seg, sr, seg_len);
        __syncthreads();
        if (run + (int)gridDim.x < p.total_runs) fetch_segment(p, sr, run + gridDim.x, seg_len);
#pragma unroll 2
        for (int i = 0; i < C1_RUN / 64; ++i) {
            const int f = i * 64 + fr, l = l0 + f;
            if (l < p.Lout) {
                S2AG_DBG_ASSERT(f * p.stride + KS <= seg_len);
                const float* sx = seg + f * p.stride;
                float xv[KS];
#pragma unroll
                for (int t = 0; t < KS; ++t) xv[t] = sx[t];
                float4 o;
                float* op = &o.x;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float a = br[c];
#pragma unroll
                    for (int t = 0; t < KS; ++t) a = fmaf(xv[t], wr[c][t], a);
                    if (BF) a = __uint_as_float(c1_bf16_rn(a) << 16);        // what is stored (and normalised later)
                    op[c] = a;
                    if (p.stats) {
                        s1[c] += (double)a;
                        s2[c] += (double)a * (double)a;
                    }
                }
                if (BF)
                    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.y) + ((long long)n * p.Lout + l) * p.ldy + q * 4) =
                        make_uint2((__float_as_uint(o.x) >> 16) | (__float_as_uint(o.y) & 0xffff0000u),
                                   (__float_as_uint(o.z) >> 16) | (__float_as_uint(o.w) & 0xffff0000u));
                else
                    *reinterpret_cast<float4*>(p.y + ((long long)n * p.Lout + l) * p.ldy + q * 4) = o;
            }
        }
    }
    if (p.stats) {
        // one partial row per wave for all the block's runs; lanes that share a channel quad differ in lane bits 2..5
        const size_t R = (size_t)gridDim.x * 4, r = (size_t)blockIdx.x * 4 + wave;
        __shared__ double wred[4][2][CO];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int m = 4; m < 64; m <<= 1) {
                s1[c] += __shfl_xor(s1[c], m, 64);
                s2[c] += __shfl_xor(s2[c], m, 64);
            }
            if (lane < 4) {
                if (p.fold.ticket) {
                    wred[wave][0][q * 4 + c] = s1[c];
                    wred[wave][1][q * 4 + c] = s2[c];
                } else {
                    p.stats[r * CO + q * 4 + c] = s1[c];
                    p.stats[(R + r) * CO + q * 4 + c] = s2[c];
                }
            }
        }
        if (p.fold.ticket) {        // (2, gridDim.x, CO): one row per block, published for the block that finishes last
            __syncthreads();
            if (tid < 2 * CO) {
                const int which = tid / CO, c = tid - which * CO;
                s2ag_fold::st_agent(p.stats + ((size_t)which * gridDim.x + blockIdx.x) * CO + c,
                                    (wred[0][which][c] + wred[1][which][c]) + (wred[2][which][c] + wred[3][which][c]));
            }
            if (s2ag_fold::two_level_done(p.stats, gridDim.x, CO, p.fold.ticket)) {
                __shared__ double fred[2][256];
                s2ag_fold::bn_fwd_fold_body(p.stats + (size_t)2 * gridDim.x * CO, s2ag_fold::fold_groups(gridDim.x), CO, p.fold,
                                            fred);
            }
        }
    }
}

template <int CO, int KS, bool BF, bool XF = false>
__global__ __launch_bounds__(C1_NT) void conv_c1_wgrad_k(const C1P p) {
    static_assert(CO == 16, "thread mapping: 4 channel quads");
    static_assert(BF || !XF, "the folded BatchNorm backward reads bf16 rows");
    extern __shared__ float seg[];
    __shared__ float red[4][CO * KS + CO];                           // per-wave sums: dw (CO x KS) then db (CO)
    const int tid = threadIdx.x, q = tid & 3, fr = tid >> 2;
    const int lane = tid & 63, wave = tid >> 6;
    float xa[4] = {0.f, 0.f, 0.f, 0.f}, xb[4] = {0.f, 0.f, 0.f, 0.f}, xc[4] = {0.f, 0.f, 0.f, 0.f};
    if (XF) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            xa[c] = p.ca[q * 4 + c];
            xb[c] = p.cb[q * 4 + c];
            xc[c] = p.cc[q * 4 + c];
        }
    }
    float acc[4][KS], bsum[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bsum[c] = 0.f;
#pragma unroll
        for (int t = 0; t < KS; ++t) acc[c][t] = 0.f;
    }
    const int seg_len = (C1_RUN - 1) * p.stride + KS;
    float sr[C1_SEG_REGS];
    if ((int)blockIdx.x < p.total_runs) fetch_segment(p, sr, blockIdx.x, seg_len);
    // XF: the raw rows (dz, y1) of the NEXT run are in flight while the current run is multiplied (a block is a chain of
    // runs at two blocks per CU: with the loads of a run requested when the run begins it was a chain of memory round trips)
    uint2 nz[XF ? C1_RUN / 64 : 1], ny[XF ? C1_RUN / 64 : 1];
    auto fetch_rows = [&](int run) {
        if constexpr (XF) {
            const int n = run / p.runs_per_clip, l0 = (run - n * p.runs_per_clip) * C1_RUN;
#pragma unroll
            for (int i = 0; i < C1_RUN / 64; ++i) {
                const int l = l0 + i * 64 + fr;
                nz[i] = ny[i] = make_uint2(0u, 0u);
                if (run < p.total_runs && l < p.Lout) {
                    const long long off = ((long long)n * p.Lout + l) * p.ldy + q * 4;
                    nz[i] = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.y) + off);
                    ny[i] = *reinterpret_cast<const uint2*>(p.y1 + off);
                }
            }
        }
    };
    fetch_rows(blockIdx.x);
    for (int run = blockIdx.x; run < p.total_runs; run += gridDim.x) {
        const int n = run / p.runs_per_clip, l0 = (run - n * p.runs_per_clip) * C1_RUN;
        // all of the run's output-gradient rows of this thread are requested up front (one round trip per run, not four)
        float4 gq[C1_RUN / 64];
#pragma unroll
        for (int i = 0; i < C1_RUN / 64; ++i) {
            const int l = l0 + i * 64 + fr;
            gq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (l < p.Lout) {
                if (XF) {
                    const uint2 h = nz[i], v = ny[i];
                    gq[i].x = fmaf(xa[0], __uint_as_float(h.x << 16), fmaf(xc[0], __uint_as_float(v.x << 16), xb[0]));
                    gq[i].y = fmaf(xa[1], __uint_as_float(h.x & 0xffff0000u), fmaf(xc[1], __uint_as_float(v.x & 0xffff0000u), xb[1]));
                    gq[i].z = fmaf(xa[2], __uint_as_float(h.y << 16), fmaf(xc[2], __uint_as_float(v.y << 16), xb[2]));
                    gq[i].w = fmaf(xa[3], __uint_as_float(h.y & 0xffff0000u), fmaf(xc[3], __uint_as_float(v.y & 0xffff0000u), xb[3]));
                } else if (BF) {
                    const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p.y) +
                                                                    ((long long)n * p.Lout + l) * p.ldy + q * 4);
                    gq[i] = make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u),
                                        __uint_as_float(h.y << 16), __uint_as_float(h.y & 0xffff0000u));
                } else {
                    gq[i] = *reinterpret_cast<const float4*>(p.y + ((long long)n * p.Lout + l) * p.ldy + q * 4);
                }
            }
        }
        __syncthreads();
        store_segment(seg, sr, seg_len);
        __syncthreads();
        if (run + (int)gridDim.x < p.total_runs) fetch_segment(p, sr, run + gridDim.x, seg_len);
        fetch_rows(run + gridDim.x);
#pragma unroll
        for (int i = 0; i < C1_RUN / 64; ++i) {
            const int f = i * 64 + fr, l = l0 + f;
            if (l < p.Lout) {
                const float4 g4 = gq[i];
                const float* gp = &g4.x;
                const float* sx = seg + f * p.stride;
                float xv[KS];
#pragma unroll
                for (int t = 0; t < KS; ++t) xv[t] = sx[t];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    bsum[c] += gp[c];
#pragma unroll
                    for (int t = 0; t < KS; ++t) acc[c][t] = fmaf(gp[c], xv[t], acc[c][t]);
                }
            }
        }
    }
    // fold: across the 16 lanes of each channel quad (lane bits 2..5), then across the 4 waves through LDS
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int m = 4; m < 64; m <<= 1) bsum[c] += __shfl_xor(bsum[c], m, 64);
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            float v = acc[c][t];
#pragma unroll
            for (int m = 4; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
            acc[c][t] = v;
        }
    }
    if (lane < 4) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            red[wave][CO * KS + q * 4 + c] = bsum[c];
#pragma unroll
            for (int t = 0; t < KS; ++t) red[wave][(q * 4 + c) * KS + t] = acc[c][t];
        }
    }
    __syncthreads();
    s2ag::det_enter();
    for (int i = tid; i < CO * KS + CO; i += C1_NT) {
        const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
        if (p.part)             // 1 024 blocks x 256 atomics on the same 256 addresses were most of this kernel's time
            p.part[(size_t)blockIdx.x * (CO * KS + CO) + i] = v;
        else if (i < CO * KS)
            atomicAdd(p.dw + i, v);
        else if (p.db)
            atomicAdd(p.db + (i - CO * KS), v);
    }
    s2ag::det_leave();
}

// dw[i] += sum_b part[b][i] (i < CO * KS), db[c] += sum_b part[b][CO * KS + c]: one block, 256 = CO * KS + CO threads,
// 16 rows in flight per thread
__global__ __launch_bounds__(256) void conv_c1_wgrad_fold_k(const float* __restrict__ part, int nblocks, float* __restrict__ dw,
                                                           float* __restrict__ db, int n_w, int n_all) {
    __shared__ float red[4][64];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    float s = 0.f;
    if (i < n_all) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        int b = grp;
        for (; b + 60 < nblocks; b += 64) {          // 16 rows in flight per thread
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = part[(size_t)(b + 4 * j) * n_all + i];
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j & 3] += v[j];
        }
        for (; b < nblocks; b += 4) a[0] += part[(size_t)b * n_all + i];
        s = (a[0] + a[1]) + (a[2] + a[3]);
    }
    red[grp][threadIdx.x & 63] = s;
    __syncthreads();
    if (grp == 0 && i < n_all) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (i < n_w) dw[i] += v;
        else if (db) db[i - n_w] += v;
    }
}

inline bool c1_shape_ok(int Cin, int Cout, int ks, int dil, int ldx, int ldy, int stride) {
    return Cin == 1 && Cout == 16 && ks == 15 && dil == 1 && ldx == 1 && (ldy & 3) == 0 && stride >= 1 &&
           (C1_RUN - 1) * stride + 15 <= C1_SEG_REGS * C1_NT;      // the register-staged segment (stride <= 8)
}
inline size_t c1_seg_bytes(int stride) { return sizeof(float) * ((size_t)(C1_RUN - 1) * stride + 15); }
}  // namespace

// Returns the number of statistics partial rows written (> 0) if the launch was taken, 0 if the geometry is not this
// kernel's (the caller falls back to the implicit-GEMM path).
int s2ag_conv_c1_fwd(const float* x, const float* w, const float* bias, float* y, int N, int Lin, int Lout, int Cin, int Cout,
                int ks, int stride, int pad, int dil, int ldx, int ldy, double* stats, int stats_cap_rows,
                hipStream_t stream) {
    if (!c1_shape_ok(Cin, Cout, ks, dil, ldx, ldy, stride) || (reinterpret_cast<uintptr_t>(y) & 15)) return 0;
    C1P p{};
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.stats = stats;
    p.N = N; p.Lin = Lin; p.Lout = Lout; p.stride = stride; p.pad = pad; p.ldx = ldx; p.ldy = ldy;
    p.runs_per_clip = cdiv(Lout, C1_RUN);
    p.total_runs = N * p.runs_per_clip;
    // persistent blocks (4 per CU): a thread's 60 weights are loaded once and serve all its runs
    const int blocks = p.total_runs < 1024 ? p.total_runs : 1024;
    if (stats && blocks * 4 > stats_cap_rows) return 0;
    hipLaunchKernelGGL((conv_c1_fwd_k<16, 15, false>), dim3(blocks), dim3(C1_NT), c1_seg_bytes(stride), stream, p);
    return blocks * 4;
}

int s2ag_conv_c1_wgrad(const float* gy, const float* x, float* dw, float* db, int N, int Lin, int Lout, int Cin, int Cout,
                  int ks, int stride, int pad, int dil, int ldx, int ldg, hipStream_t stream) {
    if (!c1_shape_ok(Cin, Cout, ks, dil, ldx, ldg, stride) || (reinterpret_cast<uintptr_t>(gy) & 15)) return 0;
    C1P p{};
    p.x = x; p.y = const_cast<float*>(gy); p.dw = dw; p.db = db;
    p.N = N; p.Lin = Lin; p.Lout = Lout; p.stride = stride; p.pad = pad; p.ldx = ldx; p.ldy = ldg;
    p.runs_per_clip = cdiv(Lout, C1_RUN);
    p.total_runs = N * p.runs_per_clip;
    const int blocks = p.total_runs < 1024 ? p.total_runs : 1024;    // 4 blocks per CU; each adds 256 sums at the end
    hipLaunchKernelGGL((conv_c1_wgrad_k<16, 15, false>), dim3(blocks), dim3(C1_NT), c1_seg_bytes(stride), stream, p);
    return 1;
}

// bf16 mode (conv_bf16.hip): the same two kernels storing / loading 4 bf16 (8 bytes) per thread and frame
extern "C" int s2ag_bf16_conv_c1_fwd(const float* x, const float* w, const float* bias, void* y, const s2ag_conv_geom* g,
                                     double* partials, int* stat_rows, void* stream) {
    if (stat_rows) *stat_rows = 0;
    if (!x || !w || !y || !g || (partials == nullptr) != (stat_rows == nullptr)) return S2AG_E_BADARG;
    if (!c1_shape_ok(g->Cin, g->Cout, g->ksize, g->dil, g->ldx, g->ldy, g->stride) || (reinterpret_cast<uintptr_t>(y) & 7))
        return S2AG_E_UNSUPPORTED;
    C1P p{};
    p.x = x; p.w = w; p.bias = bias; p.y = static_cast<float*>(y); p.stats = partials;
    p.N = g->N; p.Lin = g->Lin; p.Lout = g->Lout; p.stride = g->stride; p.pad = g->pad; p.ldx = g->ldx; p.ldy = g->ldy;
    p.runs_per_clip = cdiv(g->Lout, C1_RUN);
    p.total_runs = g->N * p.runs_per_clip;
    const int blocks = p.total_runs < 1024 ? p.total_runs : 1024;
    hipLaunchKernelGGL((conv_c1_fwd_k<16, 15, true>), dim3(blocks), dim3(C1_NT), c1_seg_bytes(g->stride), (hipStream_t)stream, p);
    if (stat_rows) *stat_rows = blocks * 4;
    S2AG_LAUNCH_CHECK();
    return 0;
}

extern "C" int s2ag_bf16_conv_c1_rows(const s2ag_conv_geom* g) {
    if (!g) return S2AG_E_BADARG;
    const int runs = g->N * cdiv(g->Lout, C1_RUN);
    return (runs < 1024 ? runs : 1024) * 4;
}

extern "C" int s2ag_bf16_conv_c1_wgrad(const void* gy, const float* x, float* dw, float* db, const s2ag_conv_geom* g,
                                       void* stream) {
    if (!gy || !x || !dw || !g) return S2AG_E_BADARG;
    if (!c1_shape_ok(g->Cin, g->Cout, g->ksize, g->dil, g->ldx, g->ldy, g->stride) || (reinterpret_cast<uintptr_t>(gy) & 7))
        return S2AG_E_UNSUPPORTED;
    C1P p{};
    p.x = x; p.y = static_cast<float*>(const_cast<void*>(gy)); p.dw = dw; p.db = db;
    p.N = g->N; p.Lin = g->Lin; p.Lout = g->Lout; p.stride = g->stride; p.pad = g->pad; p.ldx = g->ldx; p.ldy = g->ldy;
    p.runs_per_clip = cdiv(g->Lout, C1_RUN);
    p.total_runs = g->N * p.runs_per_clip;
    const int blocks = p.total_runs < 1024 ? p.total_runs : 1024;
    hipLaunchKernelGGL((conv_c1_wgrad_k<16, 15, true>), dim3(blocks), dim3(C1_NT), c1_seg_bytes(g->stride), (hipStream_t)stream, p);
    S2AG_LAUNCH_CHECK();
    return 0;
}

// conv1's weight gradient with the BatchNorm backward folded into the loader (see wave_fused.hip): gy = ca dz + cc y1 + cb
extern "C" int s2ag_wave_conv1_wgrad_blocks(const s2ag_conv_geom* g) {
    if (!g) return S2AG_E_BADARG;
    const int runs = g->N * cdiv(g->Lout, C1_RUN);
    return runs < 1024 ? runs : 1024;
}

extern "C" int s2ag_wave_conv1_wgrad(const void* dz, const void* y1, const float* ca, const float* cb, const float* cc,
                                     const float* x, float* partials, float* dw, float* db, const s2ag_conv_geom* g,
                                     void* stream) {
    if (!dz || !y1 || !ca || !cb || !cc || !x || !partials || !dw || !g) return S2AG_E_BADARG;
    if (!c1_shape_ok(g->Cin, g->Cout, g->ksize, g->dil, g->ldx, g->ldy, g->stride) || (reinterpret_cast<uintptr_t>(dz) & 7) ||
        (reinterpret_cast<uintptr_t>(y1) & 7))
        return S2AG_E_UNSUPPORTED;
    C1P p{};
    p.x = x; p.y = static_cast<float*>(const_cast<void*>(dz)); p.dw = dw; p.db = db;
    p.y1 = static_cast<const unsigned short*>(y1); p.ca = ca; p.cb = cb; p.cc = cc; p.part = partials;
    p.N = g->N; p.Lin = g->Lin; p.Lout = g->Lout; p.stride = g->stride; p.pad = g->pad; p.ldx = g->ldx; p.ldy = g->ldy;
    p.runs_per_clip = cdiv(g->Lout, C1_RUN);
    p.total_runs = g->N * p.runs_per_clip;
    const int blocks = p.total_runs < 1024 ? p.total_runs : 1024;
    hipLaunchKernelGGL((conv_c1_wgrad_k<16, 15, true, true>), dim3(blocks), dim3(C1_NT), c1_seg_bytes(g->stride),
                       (hipStream_t)stream, p);
    const int n_w = 16 * 15, n_all = n_w + 16;
    hipLaunchKernelGGL(conv_c1_wgrad_fold_k, dim3(cdiv(n_all, 64)), dim3(256), 0, (hipStream_t)stream, partials, blocks, dw, db,
                       n_w, n_all);
    S2AG_LAUNCH_CHECK();
    return 0;
}

// conv1 forward of the fused wave encoder: bf16 output + ONE statistics row per block, folded into the BatchNorm
// coefficients by the block that finishes last (no s2ag_bn_fold launch).  partials: (2, s2ag_wave_conv1_fwd_rows, 16) doubles.
extern "C" int s2ag_wave_conv1_fwd_rows(const s2ag_conv_geom* g) {
    if (!g) return S2AG_E_BADARG;
    const int runs = g->N * cdiv(g->Lout, C1_RUN);
    return runs < 1024 ? runs : 1024;
}

extern "C" int s2ag_wave_conv1_fwd(const float* x, const float* w, const float* bias, void* y, const s2ag_conv_geom* g,
                                   double* partials, const s2ag_bn_fold_args* fold, void* stream) {
    if (!x || !w || !y || !g || !partials || !fold || !fold->ticket || !fold->gamma || !fold->beta || !fold->running_mean ||
        !fold->running_var || !fold->scale || !fold->shift || !fold->mean || !fold->invstd || fold->repeat < 1)
        return S2AG_E_BADARG;
    if (!c1_shape_ok(g->Cin, g->Cout, g->ksize, g->dil, g->ldx, g->ldy, g->stride) || (reinterpret_cast<uintptr_t>(y) & 7))
        return S2AG_E_UNSUPPORTED;
    C1P p{};
    p.x = x; p.w = w; p.bias = bias; p.y = static_cast<float*>(y); p.stats = partials;
    p.N = g->N; p.Lin = g->Lin; p.Lout = g->Lout; p.stride = g->stride; p.pad = g->pad; p.ldx = g->ldx; p.ldy = g->ldy;
    p.runs_per_clip = cdiv(g->Lout, C1_RUN);
    p.total_runs = g->N * p.runs_per_clip;
    p.fold = s2ag_fold::FwdFold{fold->ticket, fold->gamma, fold->beta, fold->running_mean, fold->running_var,
                                fold->num_batches_tracked, fold->eps, fold->momentum, fold->repeat, (long long)g->N * g->Lout,
                                fold->scale, fold->shift, fold->mean, fold->invstd};
    const int blocks = p.total_runs < 1024 ? p.total_runs : 1024;
    hipLaunchKernelGGL((conv_c1_fwd_k<16, 15, true>), dim3(blocks), dim3(C1_NT), c1_seg_bytes(g->stride), (hipStream_t)stream, p);
    S2AG_LAUNCH_CHECK();
    return 0;
}
S2AG_DET_HOOK(conv_c1)
