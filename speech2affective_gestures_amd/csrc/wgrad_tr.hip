// Weight gradients of the bf16 mode without a transposing loader (include/s2ag_hip.h, s2ag_bf16_conv_wgrad_tr).
//
//   dw[co, t, c] += sum_m gy[m, co] * x[row(m, t), c]
//
// contracts over ROWS, the slow axis of both operands, while an MFMA operand register holds 8 consecutive K values of one
// row / column.  The first kernel (conv_bf16_wgrad_k) transposed in its loader -- row pairs packed into 4-byte LDS stores --
// and was bound by exactly that: ~250 vector-ALU instructions per 64-row step against 8 MFMAs per wave (150 us for the
// TCN's eight gradients, 25 GFLOP).  gfx950's LDS transpose read does the transposition for free: the operand tiles are
// stored row-major as they arrive (16-byte global loads -> 16-byte LDS stores) and ds_read_b64_tr_b16 hands lane
// (g = lane >> 4, t = lane & 15) the four elements img[8g + 0..3][col0 + t] when it points at &img[8g + t/4][col0 + 4*(t%4)]
// (semantics probed on the hardware: tools/probe/tr16_probe.hip) -- two such reads are one MFMA operand of the 16x16x32
// instruction, for gy^T (A) and for x (B) alike.
//
//   * block tile TCO output channels x TK columns of one tap, 4 waves as 2 x 2, 32 rows (one MFMA K) per step, LDS double
//     buffered (one barrier per step), the global loads of step s + 2 in flight behind the MFMAs of step s;
//   * <160, 160> for the TCN (300 -> 320 channels = 2 tiles either way: every operand element is read 2 / 4 times
//     instead of 5 / 10 with 64 x 64 tiles), <64, 64> for the wave encoder's small weights;
//   * no atomics: the contraction is split over blockIdx.y, every block stores its tile to a scratch buffer and a second
//     launch sums the splits into dw / db (one thread per element: the scattered (Cout, Cin, ks) addresses of a
//     reference-layout weight are written once, and 10^7 contended fp32 atomics are gone);
//   * up to 8 layers per launch (blockIdx.z).
#include <stdlib.h>

#include "s2ag_common.h"

namespace {
using namespace s2ag;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef unsigned short bf16_t;

struct TrP {
    const bf16_t* gy;
    const bf16_t* x;
    float* dw;
    float* db;                  // nullable
    int M, Lq, Lin;
    long long x_clip;
    int ldx, ldg;
    int pos_mul, pos_off, pos_tap;
    int ks, Cp, Cvalid;
    int Cout, Cin;
    long long d_co;
    int d_t, d_c;
    int flat_cin, ks_out;
    int m_chunk, splits, ntiles, nco, kct;
    float* part;                // (splits, ntiles, TCO, TK)
    float* part_b;              // (splits, nco, TCO)
};

struct TrJobs {
    TrP j[S2AG_BF16_MAX_WGRAD_JOBS];
    int xcd_remap;
};

__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

template <int TCO, int TK>
__global__ __launch_bounds__(256) void wgrad_tr_k(const TrJobs js) {
    // XCD-aware block order: the dispatcher places hardware block b on XCD b % 8, each XCD has its own L2, and the tiles of
    // one (layer, split) read the same operand rows (the k tiles share gy, the co tiles share x).  In hardware order
    // (tile fastest) the 8 tiles of a group land on 8 different XCDs and every L2 fetches its own copy through the
    // Infinity Cache (356 MB for 89 MB of operands: the TCN launch ran at that fabric's ~3.7 TB/s).  The bijective remap
    // gives every XCD a contiguous range of virtual ids, so a group's tiles share one L2.
    const int nwg = gridDim.x * gridDim.y * gridDim.z;
    const int hw = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = hw & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int v = js.xcd_remap ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (hw >> 3) : hw;
    const int tile = v % gridDim.x, split = (v / gridDim.x) % gridDim.y;
    const TrP& p = js.j[v / (gridDim.x * gridDim.y)];
    if (tile >= p.ntiles || split >= p.splits) return;
    constexpr int PA = TCO + 8, PB = TK + 8;                    // LDS row pitches (bf16): rows stay 16-byte aligned
    constexpr int CA = TCO / 8, CB = TK / 8;                    // 16-byte chunks per row
    constexpr int NA = (32 * CA + 255) / 256, NB = (32 * CB + 255) / 256;
    constexpr int WA = TCO / 32, WB = TK / 32;                  // 16-wide tiles per wave along co / along k
    __shared__ __attribute__((aligned(16))) bf16_t Gs[2][32 * PA];
    __shared__ __attribute__((aligned(16))) bf16_t Xs[2][32 * PB];
    __shared__ float bsum[TCO];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int cot = tile % p.nco, kt = tile / p.nco;
    const int tap = kt / p.kct, c0 = (kt - tap * p.kct) * TK, co0 = cot * TCO;
    const bool do_bias = p.db != nullptr && kt == 0;
    for (int i = tid; i < TCO; i += 256) bsum[i] = 0.f;

    const int m_beg = split * p.m_chunk;
    const int m_end = min(p.M, m_beg + p.m_chunk);
    int ra[NA], ca[NA], rb[NB], cb[NB];
    bool oka[NA], okb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int id = tid + 256 * i;
        ra[i] = id / CA;
        ca[i] = id - ra[i] * CA;
        oka[i] = id < 32 * CA && co0 + ca[i] * 8 < p.ldg;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int id = tid + 256 * i;
        rb[i] = id / CB;
        cb[i] = id - rb[i] * CB;
        okb[i] = id < 32 * CB && c0 + cb[i] * 8 < p.Cvalid;
    }
    // Running pointers: a chunk's source address advances by a constant per 32-row step (plus a constant at a clip
    // boundary for x), so a fetch costs an add and a select per chunk -- recomputing clip, frame and the 64-bit products
    // per load was ~100 of the ~360 instructions of a step, and with one wave per SIMD a step is its instruction count.
    const bf16_t* gptr[NA];
    const bf16_t* xptr[NB];
    int xq[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) gptr[i] = p.gy + (long long)(m_beg + ra[i]) * p.ldg + co0 + ca[i] * 8;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int m = min(m_beg + rb[i], p.M - 1);
        const int n = m / p.Lq;
        xq[i] = m - n * p.Lq;
        xptr[i] = p.x + (long long)n * p.x_clip + (long long)(xq[i] * p.pos_mul + p.pos_off + tap * p.pos_tap) * p.ldx + c0 + cb[i] * 8;
    }
    const long long g_step = 32ll * p.ldg, x_step = 32ll * p.pos_mul * p.ldx;
    const long long x_wrap = p.x_clip - (long long)p.Lq * p.pos_mul * p.ldx;      // extra advance across a clip boundary
    const int row_off = p.pos_off + tap * p.pos_tap;
    int next_mb = m_beg;                                         // fetches are issued for consecutive steps
    constexpr int RING = 4;                                      // register sets = steps in flight (+ the one being stored)
    u32x4 rg[RING][NA], rx[RING][NB];
    unsigned vmask[RING];                                           // bit i: G chunk i valid, bit 8 + i: X chunk i valid
    // branch-free: an invalid chunk loads from a safe address and is zeroed when it is stored to LDS
    auto fetch = [&](int set) {
        const int mb = next_mb;
        unsigned vm = 0u;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool v = oka[i] && mb + ra[i] < m_end;
            vm |= v ? (1u << i) : 0u;
            rg[set][i] = *reinterpret_cast<const u32x4*>(v ? gptr[i] : p.gy);
            gptr[i] += g_step;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = xq[i] * p.pos_mul + row_off;
            const bool v = okb[i] && mb + rb[i] < m_end && (unsigned)row < (unsigned)p.Lin;
            vm |= v ? (1u << (8 + i)) : 0u;
            rx[set][i] = *reinterpret_cast<const u32x4*>(v ? xptr[i] : p.x);
            // next step: 32 rows on (plan() guarantees Lq >= 32: at most one clip boundary per step)
            const bool wrap = xq[i] + 32 >= p.Lq;
            xq[i] += wrap ? 32 - p.Lq : 32;
            xptr[i] += wrap ? x_step + x_wrap : x_step;
        }
        vmask[set] = vm;
        next_mb = mb + 32;
    };
    float bacc[NA][8];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) bacc[i][j] = 0.f;
    auto stash = [&](int set, int buf) {
        const unsigned vm = vmask[set];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (!((vm >> i) & 1u)) rg[set][i] = u32x4{0u, 0u, 0u, 0u};
            if ((32 * CA) % 256 == 0 || i + 1 < NA || tid + 256 * i < 32 * CA)
                *reinterpret_cast<u32x4*>(&Gs[buf][ra[i] * PA + ca[i] * 8]) = rg[set][i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (!((vm >> (8 + i)) & 1u)) rx[set][i] = u32x4{0u, 0u, 0u, 0u};
            if ((32 * CB) % 256 == 0 || i + 1 < NB || tid + 256 * i < 32 * CB)
                *reinterpret_cast<u32x4*>(&Xs[buf][rb[i] * PB + cb[i] * 8]) = rx[set][i];
        }
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const unsigned w[4] = {rg[set][i].x, rg[set][i].y, rg[set][i].z, rg[set][i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bacc[i][2 * j] += __uint_as_float(w[j] << 16);
                    bacc[i][2 * j + 1] += __uint_as_float(w[j] & 0xffff0000u);
                }
            }
        }
    };
    f32x4 acc[WA][WB];
#pragma unroll
    for (int a = 0; a < WA; ++a)
#pragma unroll
        for (int b = 0; b < WB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // transpose-read address of this lane inside a 16-column tile: row 8g + t/4, column 4*(t % 4)
    const int g = lane >> 4, t = lane & 15;
    const int tr_a = (8 * g + (t >> 2)) * PA + (t & 3) * 4 + wr * (TCO / 2);
    const int tr_b = (8 * g + (t >> 2)) * PB + (t & 3) * 4 + wc * (TK / 2);
    auto frag = [&](const bf16_t* img, int off, int pitch) {
        using lds_p = __attribute__((address_space(3))) s16x4*;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + off + 4 * pitch));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto mma = [&](int buf) {
        bf16x8 af[WA], bfr[WB];
#pragma unroll
        for (int a = 0; a < WA; ++a) af[a] = frag(Gs[buf], tr_a + a * 16, PA);
#pragma unroll
        for (int b = 0; b < WB; ++b) bfr[b] = frag(Xs[buf], tr_b + b * 16, PB);
#pragma unroll
        for (int a = 0; a < WA; ++a)
#pragma unroll
            for (int b = 0; b < WB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    };

    // every fetch is safe beyond m_end (all chunks invalid -> zeros), so the pipeline needs no tail cases: m_chunk is a
    // multiple of 32 * RING rows and the steps past m_end multiply zeros.  The loads of step s + RING are issued when the
    // registers of step s have gone to LDS: RING - 1 full steps of global round trip per block (a workgroup of the
    // <160, 160> shape owns a CU alone -- 344 VGPRs -- so nothing else hides the ~2 us a request takes under this load:
    // with two sets a step took 2 us, 112 us for the TCN's eight gradients)
#pragma unroll
    for (int r = 0; r < RING; ++r) fetch(r);
    for (int mb = m_beg; mb < m_end; mb += 32 * RING) {
#pragma unroll
        for (int r = 0; r < RING; ++r) {
            stash(r, r & 1);
            fetch(r);
            __syncthreads();
            mma(r & 1);
        }
    }
    // D[i][j]: i = output channel (A row) = (lane >> 4)*4 + q, j = x column (B column) = lane & 15
    float* dst = p.part + ((long long)split * p.ntiles + tile) * (TCO * TK);
#pragma unroll
    for (int a = 0; a < WA; ++a)
#pragma unroll
        for (int b = 0; b < WB; ++b) {
            const int kcol = wc * (TK / 2) + b * 16 + (lane & 15);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = wr * (TCO / 2) + a * 16 + (lane >> 4) * 4 + q;
                if (co0 + col < p.Cout) dst[col * TK + kcol] = acc[a][b][q];
            }
            __builtin_amdgcn_sched_barrier(0);                  // tile by tile: 100 store addresses at once cost 200 registers
        }
    if (do_bias) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (tid + 256 * i < 32 * CA) {
#pragma unroll
                for (int j = 0; j < 8; ++j) atomicAdd(&bsum[ca[i] * 8 + j], bacc[i][j]);
            }
        __syncthreads();
        for (int i = tid; i < TCO; i += 256)
            if (co0 + i < p.Cout) p.part_b[((long long)split * p.nco + cot) * TCO + i] = bsum[i];
    }
}

// dw (+ db) += sum over the splits of the stored tiles; blockIdx.y = job; 32 consecutive elements per block, the 8 thread
// rows share the splits
// (DIRECT: few splits -- one thread per element sums them all, 256 consecutive elements per block)
template <int TCO, int TK, bool DIRECT>
__global__ __launch_bounds__(256) void wgrad_tr_reduce_k(const TrJobs js) {
    const TrP& p = js.j[blockIdx.y];
    __shared__ float red[8][33];
    constexpr int EPB = DIRECT ? 256 : 32, NG = DIRECT ? 1 : 8;
    const long long total = (long long)p.ntiles * (TCO * TK);
    const int nb = p.nco * TCO;
    const int e = DIRECT ? threadIdx.x : (threadIdx.x & 31), grp = DIRECT ? 0 : (threadIdx.x >> 5);
    const long long i = (long long)blockIdx.x * EPB + e;
    if ((long long)blockIdx.x * EPB >= total + nb) return;      // whole block beyond this job
    const bool is_bias = i >= total;
    const int j = (int)(i - total);
    int tile = 0, col = 0, kl = 0;
    if (!is_bias) {
        tile = (int)(i / (TCO * TK));
        const int r = (int)(i - (long long)tile * (TCO * TK));
        col = r / TK;
        kl = r - col * TK;
    }
    const int cot = tile % p.nco, kt = tile / p.nco;
    const int co = cot * TCO + col;
    float sum = 0.f;
    if (!is_bias) {
        if (co < p.Cout) {
            const float* src = p.part + i;
#pragma unroll 4
            for (int sp = grp; sp < p.splits; sp += NG) sum += src[(long long)sp * total];
        }
    } else if (j < nb && p.db) {
        const int cb = j / TCO, ci = j - cb * TCO;
        if (cb * TCO + ci < p.Cout)
            for (int sp = grp; sp < p.splits; sp += NG) sum += p.part_b[(long long)sp * nb + j];
    }
    if (!DIRECT) {
        red[grp][e] = sum;
        __syncthreads();
        if (grp != 0) return;
#pragma unroll
        for (int k = 1; k < 8; ++k) sum += red[k][e];
    }
    if (is_bias) {
        if (p.db && j < nb) {
            const int cb = j / TCO, ci = j - cb * TCO;
            if (cb * TCO + ci < p.Cout) p.db[cb * TCO + ci] += sum;
        }
        return;
    }
    const int tap = kt / p.kct, k = (kt - tap * p.kct) * TK + kl;
    int t = tap, c = k;
    if (p.flat_cin > 0) {
        t = k / p.flat_cin;
        c = k - t * p.flat_cin;
    }
    if (co >= p.Cout || c >= p.Cin || t >= p.ks_out || k >= p.Cp) return;
    p.dw[(long long)co * p.d_co + (long long)t * p.d_t + (long long)c * p.d_c] += sum;
}

int plan(const s2ag_bf16_wgrad_args* g, TrP& p, int TCO, int TK, int blocks_for_job) {
    if (!g || !g->gy || !g->x || !g->dw || g->N <= 0 || g->Lq <= 0 || g->ks <= 0) return S2AG_E_BADARG;
    if ((g->Cvalid & 7) || (g->ldx & 7) || (g->ldg & 7) || g->Cvalid > g->Cp) return S2AG_E_BADARG;
    if (g->Lq < 32) return S2AG_E_UNSUPPORTED;                  // the loader steps (clip, frame) by 32 rows with one wrap
    if ((reinterpret_cast<uintptr_t>(g->x) | reinterpret_cast<uintptr_t>(g->gy)) & 15) return S2AG_E_BADARG;
    p.gy = static_cast<const bf16_t*>(g->gy); p.x = static_cast<const bf16_t*>(g->x); p.dw = g->dw; p.db = g->db;
    p.M = g->N * g->Lq; p.Lq = g->Lq; p.Lin = g->Lin; p.x_clip = g->x_clip; p.ldx = g->ldx; p.ldg = g->ldg;
    p.pos_mul = g->pos_mul; p.pos_off = g->pos_off; p.pos_tap = g->pos_tap;
    p.ks = g->ks; p.Cp = g->Cp; p.Cvalid = g->Cvalid; p.Cout = g->Cout; p.Cin = g->Cin;
    p.d_co = g->d_co; p.d_t = g->d_t; p.d_c = g->d_c; p.flat_cin = g->flat_cin;
    p.ks_out = g->flat_cin > 0 ? g->ks_out : g->ks;
    p.nco = cdiv(g->Cout, TCO);
    p.kct = cdiv(g->Cvalid, TK);
    p.ntiles = p.nco * g->ks * p.kct;
    int splits = cdiv(blocks_for_job, p.ntiles);
    const int max_splits = cdiv(p.M, 256);                      // at least 8 steps of 32 rows per block
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.m_chunk = cdiv(cdiv(p.M, splits), 128) * 128;             // multiple of 32 rows * the kernel's ring of 4
    p.splits = cdiv(p.M, p.m_chunk);
    return 0;
}

long long part_floats(const TrP& p, int TCO, int TK) {
    return (long long)p.splits * p.ntiles * TCO * TK + (long long)p.splits * p.nco * TCO;
}

bool big_tiles(const s2ag_bf16_wgrad_args* jobs, int n) {
    for (int k = 0; k < n; ++k)
        if (jobs[k].Cout <= 128 || jobs[k].flat_cin > 0 || jobs[k].Cvalid < 160) return false;
    return true;
}

int target_blocks() {
    static const int t = [] { const char* e = getenv("S2AG_BF16_WGRAD_TR_BLOCKS"); return e ? atoi(e) : 0; }();
    return t;
}
}  // namespace

extern "C" long long s2ag_bf16_conv_wgrad_tr_scratch_floats(const s2ag_bf16_wgrad_args* jobs, int njobs) {
    if (!jobs || njobs < 1 || njobs > S2AG_BF16_MAX_WGRAD_JOBS) return S2AG_E_BADARG;
    const bool big = big_tiles(jobs, njobs);
    const int TCO = big ? 160 : 64, TK = big ? 160 : 64;
    const int target = target_blocks() > 0 ? target_blocks() : (big ? 256 : 1024);
    long long tot = 0;
    for (int k = 0; k < njobs; ++k) {
        TrP p{};
        const int rc = plan(jobs + k, p, TCO, TK, cdiv(target, njobs));
        if (rc) return rc;
        tot += part_floats(p, TCO, TK);
    }
    return tot;
}

extern "C" int s2ag_bf16_conv_wgrad_tr(const s2ag_bf16_wgrad_args* jobs, int njobs, float* scratch, long long scratch_floats,
                                       void* stream) {
    if (!jobs || njobs < 1 || njobs > S2AG_BF16_MAX_WGRAD_JOBS || !scratch) return S2AG_E_BADARG;
    const bool big = big_tiles(jobs, njobs);
    const int TCO = big ? 160 : 64, TK = big ? 160 : 64;
    const int target = target_blocks() > 0 ? target_blocks() : (big ? 256 : 1024);
    TrJobs js{};
    static const int remap = [] { const char* e = getenv("S2AG_WGRAD_TR_XCD"); return e ? atoi(e) : 1; }();
    js.xcd_remap = remap;
    long long off = 0, max_red = 0;
    int mt = 0, ms = 0;
    for (int k = 0; k < njobs; ++k) {
        TrP& p = js.j[k];
        const int rc = plan(jobs + k, p, TCO, TK, cdiv(target, njobs));
        if (rc) return rc;
        p.part = scratch + off;
        p.part_b = p.part + (long long)p.splits * p.ntiles * TCO * TK;
        off += part_floats(p, TCO, TK);
        mt = p.ntiles > mt ? p.ntiles : mt;
        ms = p.splits > ms ? p.splits : ms;
        const long long red = (long long)p.ntiles * TCO * TK + p.nco * TCO;
        max_red = red > max_red ? red : max_red;
    }
    if (off > scratch_floats) return S2AG_E_BADARG;
    const bool direct = ms <= 16;
    const dim3 grid(mt, ms, njobs), rgrid(cdiv(max_red, direct ? 256 : 32), njobs);
    hipStream_t st = (hipStream_t)stream;
    if (big) {
        hipLaunchKernelGGL((wgrad_tr_k<160, 160>), grid, dim3(256), 0, st, js);
        if (direct) hipLaunchKernelGGL((wgrad_tr_reduce_k<160, 160, true>), rgrid, dim3(256), 0, st, js);
        else hipLaunchKernelGGL((wgrad_tr_reduce_k<160, 160, false>), rgrid, dim3(256), 0, st, js);
    } else {
        hipLaunchKernelGGL((wgrad_tr_k<64, 64>), grid, dim3(256), 0, st, js);
        if (direct) hipLaunchKernelGGL((wgrad_tr_reduce_k<64, 64, true>), rgrid, dim3(256), 0, st, js);
        else hipLaunchKernelGGL((wgrad_tr_reduce_k<64, 64, false>), rgrid, dim3(256), 0, st, js);
    }
    S2AG_LAUNCH_CHECK();
    return 0;
}
