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

extern "C" int s2ag_gru_coop_split_pieces(void);      // the step's product setting (gru_coop.hip)
namespace s2ag {                                       // csrc/wgrad_tr32p.hip: the opt-in pipelined form of wgrad_tr32_k<160, 160> (option WGRAD32_PIPE)
bool wgrad_tr32p_supported(const s2ag_bf16_wgrad_args* jobs, int njobs);
int wgrad_tr32p_launch(const void* jobs_struct, int nblk, int pieces, int ring, hipStream_t st);
}

namespace {
using namespace s2ag;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef unsigned short bf16_t;

#include "wgrad_tr_shared.h"

__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// WR x WC waves share the TCO x TK tile (a wave: TCO / WR output channels x TK / WC columns).  <160, 160, 2, 2> and
// <64, 64, 2, 2>: four waves, one per SIMD.  <320, 160, 4, 2> (the TCN's 300-channel layers): eight waves -- two per SIMD, so
// one wave's transpose reads and LDS stores run beside the other's 25 MFMAs -- on a tile that reads every gy element twice
// instead of four times: a step of the four-wave kernel was ~3 000 cycles for 400 cycles of MFMA per SIMD.
template <int TCO, int TK, int WR, int WC>
__global__ __launch_bounds__(64 * WR * WC) void wgrad_tr_k(const TrJobs js) {
    constexpr int NTH = 64 * WR * WC;
    // XCD-aware block order: the dispatcher places hardware block b on XCD b % 8, each XCD has its own L2, and the tiles of
    // one (layer, split) read the same operand rows (the k tiles share gy, the co tiles share x).  In hardware order
    // (tile fastest) the 8 tiles of a group land on 8 different XCDs and every L2 fetches its own copy through the
    // Infinity Cache (356 MB for 89 MB of operands: the TCN launch ran at that fabric's ~3.7 TB/s).  The bijective remap
    // gives every XCD a contiguous range of virtual ids, so a group's tiles share one L2.
    const int nwg = gridDim.x;                                   // compact 1-D grid: every block has work
    const int hw = blockIdx.x;
    const int xcd = hw & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int v = js.xcd_remap ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (hw >> 3) : hw;
    int job = 0;
    while (job + 1 < js.njobs && v >= js.start[job + 1]) ++job;
    const TrP& p = js.j[job];
    const int local = v - js.start[job];
    const int tile = local % p.ntiles, split = local / p.ntiles;
    constexpr int PA = TCO + 8, PB = TK + 8;                    // LDS row pitches (bf16): rows stay 16-byte aligned
    constexpr int CA = TCO / 8, CB = TK / 8;                    // 16-byte chunks per row
    constexpr int NA = (32 * CA + NTH - 1) / NTH, NB = (32 * CB + NTH - 1) / NTH;
    constexpr int WA = TCO / (16 * WR), WB = TK / (16 * WC);      // 16-wide tiles per wave along co / along k
    static_assert(TCO % (16 * WR) == 0 && TK % (16 * WC) == 0, "wave tiling");
    __shared__ __attribute__((aligned(16))) bf16_t Gs[2][32 * PA];
    __shared__ __attribute__((aligned(16))) bf16_t Xs[2][32 * PB];
    __shared__ float bsum[TCO];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int cot = tile % p.nco, kt = tile / p.nco;
    const int tap = kt / p.kct, c0 = (kt - tap * p.kct) * TK, co0 = cot * TCO;
    const bool do_bias = p.db != nullptr && kt == 0;
    for (int i = tid; i < TCO; i += NTH) bsum[i] = 0.f;

    const int m_beg = split * p.m_chunk;
    const int m_end = min(p.M, m_beg + p.m_chunk);
    int ra[NA], ca[NA], rb[NB], cb[NB];
    bool oka[NA], okb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int id = tid + NTH * i;
        ra[i] = id / CA;
        ca[i] = id - ra[i] * CA;
        oka[i] = id < 32 * CA && co0 + ca[i] * 8 < p.ldg;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int id = tid + NTH * i;
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
    const bf16_t* gy0 = static_cast<const bf16_t*>(p.gy);
    const bf16_t* x0 = static_cast<const bf16_t*>(p.x);
#pragma unroll
    for (int i = 0; i < NA; ++i) gptr[i] = gy0 + (long long)(m_beg + ra[i]) * p.ldg + co0 + ca[i] * 8;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int m = min(m_beg + rb[i], p.M - 1);
        const int n = m / p.Lq;
        xq[i] = m - n * p.Lq;
        xptr[i] = x0 + (long long)n * p.x_clip + (long long)(xq[i] * p.pos_mul + p.pos_off + tap * p.pos_tap) * p.ldx + c0 + cb[i] * 8;
    }
    const long long g_step = 32ll * p.ldg, x_step = 32ll * p.pos_mul * p.ldx;
    const long long x_wrap = p.x_clip - (long long)p.Lq * p.pos_mul * p.ldx;      // extra advance across a clip boundary
    const int row_off = p.pos_off + tap * p.pos_tap;
    int next_mb = m_beg;                                         // fetches are issued for consecutive steps
    constexpr int RING = WR * WC > 4 ? 2 : 4;                     // register sets = steps in flight (+ the one being stored)
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
            rg[set][i] = *reinterpret_cast<const u32x4*>(v ? gptr[i] : gy0);
            gptr[i] += g_step;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = xq[i] * p.pos_mul + row_off;
            const bool v = okb[i] && mb + rb[i] < m_end && (unsigned)row < (unsigned)p.Lin;
            vm |= v ? (1u << (8 + i)) : 0u;
            rx[set][i] = *reinterpret_cast<const u32x4*>(v ? xptr[i] : x0);
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
            if ((32 * CA) % NTH == 0 || i + 1 < NA || tid + NTH * i < 32 * CA)
                *reinterpret_cast<u32x4*>(&Gs[buf][ra[i] * PA + ca[i] * 8]) = rg[set][i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (!((vm >> (8 + i)) & 1u)) rx[set][i] = u32x4{0u, 0u, 0u, 0u};
            if ((32 * CB) % NTH == 0 || i + 1 < NB || tid + NTH * i < 32 * CB)
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
    const int tr_a = (8 * g + (t >> 2)) * PA + (t & 3) * 4 + wr * (TCO / WR);
    const int tr_b = (8 * g + (t >> 2)) * PB + (t & 3) * 4 + wc * (TK / WC);
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
            const int kcol = wc * (TK / WC) + b * 16 + (lane & 15);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = wr * (TCO / WR) + a * 16 + (lane >> 4) * 4 + q;
                if (co0 + col < p.Cout) dst[col * TK + kcol] = acc[a][b][q];
            }
            __builtin_amdgcn_sched_barrier(0);                  // tile by tile: 100 store addresses at once cost 200 registers
        }
    if (do_bias) {
        __syncthreads();
        S2AG_DET_WAVES_BEGIN          // (deterministic mode: the waves add their bias sums one after the other)
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (tid + NTH * i < 32 * CA) {
#pragma unroll
                for (int j = 0; j < 8; ++j) atomicAdd(&bsum[ca[i] * 8 + j], bacc[i][j]);
            }
        S2AG_DET_WAVES_END
        __syncthreads();
        for (int i = tid; i < TCO; i += NTH)
            if (co0 + i < p.Cout) p.part_b[((long long)split * p.nco + cot) * TCO + i] = bsum[i];
    }
}

// ---- fp32 operands (the GRU's weight gradients of the default, fp32 step) ----------------------------------------------
// dW_ih / dW_hh of nn.GRU (net/multimodal_context_net_v2.py:281,406,480) are 56 GFLOP per step on the f32 MFMA
// (wgrad_multi_k: 1.1 ms of kernel time at ~45 TFLOP/s, beside -- and slowing down -- the next layer's recurrence).
// Same kernel shape as above, but the operand rows arrive as fp32 and every value is split by the loader into two bf16
// pieces (hi = rn(v), lo = rn(v - hi): 16 mantissa bits, the precision of the step's other large products, bench.py
// `matrix_products`); LDS holds a hi and a lo image per operand, the transpose read serves both, and a tile pair costs
// three MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulation).  Row pitches / column offsets in floats, multiples of 4.
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

// NPC: pieces per operand (2: hi + lo, three products, the fp32 step; 1: hi only, one product: the bf16 step mode)
template <int TCO, int TK, int RING, int BPC, int WR = 2, int WC = 2, int NPC = 2>
__global__ __launch_bounds__(64 * WR * WC, BPC) void wgrad_tr32_k(const TrJobs js) {
    constexpr int NTH = 64 * WR * WC;
    const int nwg = gridDim.x;                                   // compact 1-D grid: every block has work
    const int hw = blockIdx.x;
    const int xcd = hw & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int v = js.xcd_remap ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (hw >> 3) : hw;
    int job = 0;
    while (job + 1 < js.njobs && v >= js.start[job + 1]) ++job;
    const TrP& p = js.j[job];
    const int local = v - js.start[job];
    const int tile = local % p.ntiles, split = local / p.ntiles;
    constexpr int PA = TCO + 8, PB = TK + 8;                    // LDS row pitches (bf16)
    constexpr int CA = TCO / 4, CB = TK / 4;                    // 16-byte (4-float) chunks per row
    constexpr int NA = (32 * CA + NTH - 1) / NTH, NB = (32 * CB + NTH - 1) / NTH;
    constexpr int WA = TCO / (16 * WR), WB = TK / (16 * WC);
    __shared__ __attribute__((aligned(16))) bf16_t Gs[2][NPC][32 * PA];     // [buffer][hi / lo]
    __shared__ __attribute__((aligned(16))) bf16_t Xs[2][NPC][32 * PB];
    __shared__ float bsum[TCO];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int cot = tile % p.nco, kt = tile / p.nco;
    const int tap = kt / p.kct, c0 = (kt - tap * p.kct) * TK, co0 = cot * TCO;
    const bool do_bias = p.db != nullptr && kt == 0;
    for (int i = tid; i < TCO; i += NTH) bsum[i] = 0.f;
    const int m_beg = split * p.m_chunk;
    const int m_end = min(p.M, m_beg + p.m_chunk);
    int ra[NA], ca[NA], rb[NB], cb[NB];
    bool oka[NA], okb[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int id = tid + NTH * i;
        ra[i] = id / CA;
        ca[i] = id - ra[i] * CA;
        oka[i] = id < 32 * CA && co0 + ca[i] * 4 < p.Cout;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int id = tid + NTH * i;
        rb[i] = id / CB;
        cb[i] = id - rb[i] * CB;
        okb[i] = id < 32 * CB && c0 + cb[i] * 4 < p.Cvalid;
    }
    const float* gy0 = static_cast<const float*>(p.gy);
    const float* x0 = static_cast<const float*>(p.x);
    const float* gptr[NA];
    const float* xptr[NB];
    int xq[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) gptr[i] = gy0 + (long long)(m_beg + ra[i]) * p.ldg + co0 + ca[i] * 4;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int m = min(m_beg + rb[i], p.M - 1);
        const int n = m / p.Lq;
        xq[i] = m - n * p.Lq;
        xptr[i] = x0 + (long long)n * p.x_clip + (long long)(xq[i] * p.pos_mul + p.pos_off + tap * p.pos_tap) * p.ldx + c0 + cb[i] * 4;
    }
    const long long g_step = 32ll * p.ldg, x_step = 32ll * p.pos_mul * p.ldx;
    const long long x_wrap = p.x_clip - (long long)p.Lq * p.pos_mul * p.ldx;
    const int row_off = p.pos_off + tap * p.pos_tap;
    int next_mb = m_beg;
    f32x4 rg[RING][NA], rx[RING][NB];
    unsigned vmask[RING];
    auto fetch = [&](int set) {
        const int mb = next_mb;
        unsigned vm = 0u;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bool ok = oka[i] && mb + ra[i] < m_end;
            vm |= ok ? (1u << i) : 0u;
            rg[set][i] = *reinterpret_cast<const f32x4*>(ok ? gptr[i] : gy0);
            gptr[i] += g_step;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int row = xq[i] * p.pos_mul + row_off;
            const bool ok = okb[i] && mb + rb[i] < m_end && (unsigned)row < (unsigned)p.Lin;
            vm |= ok ? (1u << (8 + i)) : 0u;
            rx[set][i] = *reinterpret_cast<const f32x4*>(ok ? xptr[i] : x0);
            const bool wrap = xq[i] + 32 >= p.Lq;
            xq[i] += wrap ? 32 - p.Lq : 32;
            xptr[i] += wrap ? x_step + x_wrap : x_step;
        }
        vmask[set] = vm;
        next_mb = mb + 32;
    };
    float bacc[NA][4];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) bacc[i][j] = 0.f;
    // 4 floats -> 4 hi + 4 lo bf16 (8 bytes each)
    auto split_store = [&](bf16_t* hi_img, bf16_t* lo_img, int off, f32x4 v) {
        const unsigned h01 = pk_bf16(v[0], v[1]), h23 = pk_bf16(v[2], v[3]);
        *reinterpret_cast<uint2*>(hi_img + off) = make_uint2(h01, h23);
        if (NPC == 2) {
            const unsigned l01 = pk_bf16(v[0] - __uint_as_float(h01 << 16), v[1] - __uint_as_float(h01 & 0xffff0000u));
            const unsigned l23 = pk_bf16(v[2] - __uint_as_float(h23 << 16), v[3] - __uint_as_float(h23 & 0xffff0000u));
            *reinterpret_cast<uint2*>(lo_img + off) = make_uint2(l01, l23);
        }
    };
    auto stash = [&](int set, int buf) {
        const unsigned vm = vmask[set];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            if (!((vm >> i) & 1u)) rg[set][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if ((32 * CA) % NTH == 0 || i + 1 < NA || tid + NTH * i < 32 * CA)
                split_store(Gs[buf][0], Gs[buf][NPC - 1], ra[i] * PA + ca[i] * 4, rg[set][i]);
            if (do_bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bacc[i][j] += rg[set][i][j];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (!((vm >> (8 + i)) & 1u)) rx[set][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if ((32 * CB) % NTH == 0 || i + 1 < NB || tid + NTH * i < 32 * CB)
                split_store(Xs[buf][0], Xs[buf][NPC - 1], rb[i] * PB + cb[i] * 4, rx[set][i]);
        }
    };
    f32x4 acc[WA][WB];
#pragma unroll
    for (int a = 0; a < WA; ++a)
#pragma unroll
        for (int b = 0; b < WB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, t = lane & 15;
    const int tr_a = (8 * g + (t >> 2)) * PA + (t & 3) * 4 + wr * (TCO / WR);
    const int tr_b = (8 * g + (t >> 2)) * PB + (t & 3) * 4 + wc * (TK / WC);
    auto frag = [&](const bf16_t* img, int off, int pitch) {
        using lds_p = __attribute__((address_space(3))) s16x4*;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(img + off + 4 * pitch));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto mma = [&](int buf) {
        bf16x8 bh[WB], bl[NPC == 2 ? WB : 1];
#pragma unroll
        for (int b = 0; b < WB; ++b) {
            bh[b] = frag(Xs[buf][0], tr_b + b * 16, PB);
            if (NPC == 2) bl[b * (NPC - 1)] = frag(Xs[buf][NPC - 1], tr_b + b * 16, PB);
        }
#pragma unroll
        for (int a = 0; a < WA; ++a) {
            const bf16x8 ah = frag(Gs[buf][0], tr_a + a * 16, PA);
            // the three piece products of a tile are five MFMAs apart: back to back they wait for each other's result
            if (NPC == 2) {
                const bf16x8 al = frag(Gs[buf][NPC - 1], tr_a + a * 16, PA);
#pragma unroll
                for (int b = 0; b < WB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[b], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < WB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[b * (NPC - 1)], acc[a][b], 0, 0, 0);
            }
#pragma unroll
            for (int b = 0; b < WB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[b], acc[a][b], 0, 0, 0);
        }
    };
    // m_chunk is a multiple of 192 rows (covers a ring of 2 or 3 register sets x 2 LDS buffers); steps past m_end multiply zeros
#pragma unroll
    for (int r = 0; r < RING; ++r) fetch(r);
    int nst = 0;
    const bool tr_on = js.trace != nullptr && blockIdx.x == 0 && tid == 0;
#define TR32_STAMP() do { if (tr_on && nst < 250) js.trace[nst++] = __builtin_amdgcn_s_memtime(); } while (0)
    constexpr int PERIOD = (RING % 2) ? 2 * RING : RING;         // steps until (register set, LDS buffer) repeats
    for (int mb = m_beg; mb < m_end; mb += 32 * PERIOD) {
#pragma unroll
        for (int r = 0; r < PERIOD; ++r) {
            TR32_STAMP();
            stash(r % RING, r & 1);
            TR32_STAMP();
            fetch(r % RING);
            __syncthreads();
            TR32_STAMP();
            mma(r & 1);
        }
    }
    TR32_STAMP();
#undef TR32_STAMP
    float* dst = p.part + ((long long)split * p.ntiles + tile) * (TCO * TK);
#pragma unroll
    for (int a = 0; a < WA; ++a)
#pragma unroll
        for (int b = 0; b < WB; ++b) {
            const int kcol = wc * (TK / WC) + b * 16 + (lane & 15);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = wr * (TCO / WR) + a * 16 + (lane >> 4) * 4 + q;
                if (co0 + col < p.Cout) dst[col * TK + kcol] = acc[a][b][q];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    if (do_bias) {
        __syncthreads();
        S2AG_DET_WAVES_BEGIN          // (deterministic mode: the waves add their bias sums one after the other)
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (tid + NTH * i < 32 * CA) {
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(&bsum[ca[i] * 4 + j], bacc[i][j]);
            }
        S2AG_DET_WAVES_END
        __syncthreads();
        for (int i = tid; i < TCO; i += NTH)
            if (co0 + i < p.Cout) p.part_b[((long long)split * p.nco + cot) * TCO + i] = bsum[i];
    }
}



bool big_tiles(const s2ag_bf16_wgrad_args* jobs, int n) {
    for (int k = 0; k < n; ++k)
        if (jobs[k].Cout <= 128 || jobs[k].flat_cin > 0 || jobs[k].Cvalid < 160) return false;
    return true;
}

// fp32 form: 160 x 160 tiles unless a weight is small or flat-window (the wave encoder's convs)
bool big_tiles32(const s2ag_bf16_wgrad_args* jobs, int n) {
    for (int k = 0; k < n; ++k)
        if (jobs[k].Cout <= 128 || jobs[k].flat_cin > 0) return false;
    return true;
}

unsigned long long* g_tr_trace = nullptr;

// Workgroups of the fp32 form (one per CU: 506 VGPRs).  96, not 256: the launch runs beside the next layer's cooperative
// recurrence, whose 160 workgroups each need a CU of their own -- a chip-filling weight-gradient launch that starts first
// makes them queue (gru_coop_bwd_k 224 us inside the step against 157 us alone); 96 + 160 = the chip.  Measured on the step:
// 15 210-15 310 (96) / 15 300 (128) / 14 900 (160) / 15 110-15 120 (256) clips/s.
int target_blocks32() {
    return 96;
}

int target_blocks() {
    return 0;
}

// 320 x 160 tiles on eight waves when every weight has more than 160 output channels (the TCN: 300)
bool wide_tiles(const s2ag_bf16_wgrad_args* jobs, int n) {
    if (!big_tiles(jobs, n)) return false;
    for (int k = 0; k < n; ++k)
        if (jobs[k].Cout <= 160 || jobs[k].Cout > 320) return false;
    return true;
}
}  // namespace

extern "C" long long s2ag_bf16_conv_wgrad_tr_scratch_floats(const s2ag_bf16_wgrad_args* jobs, int njobs) {
    if (!jobs || njobs < 1 || njobs > S2AG_BF16_MAX_WGRAD_JOBS) return S2AG_E_BADARG;
    const bool big = big_tiles(jobs, njobs), wide = wide_tiles(jobs, njobs);
    const int TCO = wide ? 320 : (big ? 160 : 64), TK = big ? 160 : 64;
    const int target = target_blocks() > 0 ? target_blocks() : (big ? 256 : 1024);
    long long tot = 0;
    const int rpb = rows_per_block(jobs, njobs, TCO, TK, target);
    for (int k = 0; k < njobs; ++k) {
        TrP p{};
        const int rc = plan(jobs + k, p, TCO, TK, rpb);
        if (rc) return rc;
        tot += part_floats(p, TCO, TK);
    }
    return tot;
}

extern "C" int s2ag_bf16_conv_wgrad_tr(const s2ag_bf16_wgrad_args* jobs, int njobs, float* scratch, long long scratch_floats,
                                       void* stream) {
    if (!jobs || njobs < 1 || njobs > S2AG_BF16_MAX_WGRAD_JOBS || !scratch) return S2AG_E_BADARG;
    const bool big = big_tiles(jobs, njobs), wide = wide_tiles(jobs, njobs);
    const int TCO = wide ? 320 : (big ? 160 : 64), TK = big ? 160 : 64;
    const int target = target_blocks() > 0 ? target_blocks() : (big ? 256 : 1024);
    TrJobs js{};
    js.xcd_remap = 1;
    long long off = 0, max_red = 0;
    int ms = 0, nblk = 0;
    const int rpb = rows_per_block(jobs, njobs, TCO, TK, target);
    js.njobs = njobs;
    for (int k = 0; k < njobs; ++k) {
        TrP& p = js.j[k];
        const int rc = plan(jobs + k, p, TCO, TK, rpb);
        if (rc) return rc;
        p.part = scratch + off;
        p.part_b = p.part + (long long)p.splits * p.ntiles * TCO * TK;
        off += part_floats(p, TCO, TK);
        js.start[k] = nblk;
        nblk += p.ntiles * p.splits;
        ms = p.splits > ms ? p.splits : ms;
        const long long red = (long long)p.ntiles * TCO * TK + p.nco * TCO;
        max_red = red > max_red ? red : max_red;
    }
    js.start[njobs] = nblk;
    if (off > scratch_floats) return S2AG_E_BADARG;
    const bool direct = ms <= 16;
    const dim3 grid(nblk), rgrid(cdiv(max_red, direct ? 256 : 32), njobs);
    hipStream_t st = (hipStream_t)stream;
    if (wide) {
        hipLaunchKernelGGL((wgrad_tr_k<320, 160, 4, 2>), grid, dim3(512), 0, st, js);
        if (direct) hipLaunchKernelGGL((wgrad_tr_reduce_k<320, 160, true>), rgrid, dim3(256), 0, st, js);
        else hipLaunchKernelGGL((wgrad_tr_reduce_k<320, 160, false>), rgrid, dim3(256), 0, st, js);
    } else if (big) {
        hipLaunchKernelGGL((wgrad_tr_k<160, 160, 2, 2>), grid, dim3(256), 0, st, js);
        if (direct) hipLaunchKernelGGL((wgrad_tr_reduce_k<160, 160, true>), rgrid, dim3(256), 0, st, js);
        else hipLaunchKernelGGL((wgrad_tr_reduce_k<160, 160, false>), rgrid, dim3(256), 0, st, js);
    } else {
        hipLaunchKernelGGL((wgrad_tr_k<64, 64, 2, 2>), grid, dim3(256), 0, st, js);
        if (direct) hipLaunchKernelGGL((wgrad_tr_reduce_k<64, 64, true>), rgrid, dim3(256), 0, st, js);
        else hipLaunchKernelGGL((wgrad_tr_reduce_k<64, 64, false>), rgrid, dim3(256), 0, st, js);
    }
    S2AG_LAUNCH_CHECK();
    return 0;
}

/* fp32 operands (row pitches in floats): the GRU weight gradients of the fp32 step.  Same job struct; Cvalid / ldx / ldg
 * multiples of 4; products from two bf16 pieces per operand (16 mantissa bits), fp32 accumulation. */
// (A 160 x 128 variant with two register sets and two workgroups per CU -- one's loader beside the other's MFMAs -- spilled
// 87 VGPRs at the 256-register cap and ran at half the speed: measured, removed.  So did the eight-wave 320 x 128 form that
// works for the bf16 kernel above: hi / lo fragments and fp32 register sets do not fit 256 registers, 35 spilled, 281 us against
// 158 us for the TCN's eight gradients.)
extern "C" long long s2ag_f32_wgrad_tr_scratch_floats_n(const s2ag_bf16_wgrad_args* jobs, int njobs, int blocks) {
    if (!jobs || njobs < 1 || njobs > S2AG_BF16_MAX_WGRAD_JOBS) return S2AG_E_BADARG;
    const bool big = big_tiles32(jobs, njobs);                    // small / flat-window weights (the wave encoder): 64 x 64 tiles
    const int TCO = big ? 160 : 64, TK = big ? 160 : 64;
    const int target = blocks > 0 ? blocks : (big ? target_blocks32() : 1024);
    long long tot = 0;
    const int rpb = rows_per_block(jobs, njobs, TCO, TK, target);
    for (int k = 0; k < njobs; ++k) {
        TrP p{};
        const int rc = plan(jobs + k, p, TCO, TK, rpb, 4, 192);
        if (rc) return rc;
        tot += part_floats(p, TCO, TK);
    }
    return tot;
}

extern "C" int s2ag_f32_wgrad_tr_n(const s2ag_bf16_wgrad_args* jobs, int njobs, float* scratch, long long scratch_floats,
                                   int blocks, void* stream) {
    if (!jobs || njobs < 1 || njobs > S2AG_BF16_MAX_WGRAD_JOBS || !scratch) return S2AG_E_BADARG;
    const bool big = big_tiles32(jobs, njobs);                    // small / flat-window weights (the wave encoder): 64 x 64 tiles
    const int TCO = big ? 160 : 64, TK = big ? 160 : 64;
    const int target = blocks > 0 ? blocks : (big ? target_blocks32() : 1024);
    TrJobs js{};
    js.xcd_remap = 1;
    js.trace = g_tr_trace;
    long long off = 0, max_red = 0;
    int ms = 0, nblk = 0;
    const int rpb = rows_per_block(jobs, njobs, TCO, TK, target);
    js.njobs = njobs;
    for (int k = 0; k < njobs; ++k) {
        TrP& p = js.j[k];
        const int rc = plan(jobs + k, p, TCO, TK, rpb, 4, 192);
        if (rc) return rc;
        if ((reinterpret_cast<uintptr_t>(jobs[k].x) | reinterpret_cast<uintptr_t>(jobs[k].gy)) & 15) return S2AG_E_BADARG;
        p.part = scratch + off;
        p.part_b = p.part + (long long)p.splits * p.ntiles * TCO * TK;
        off += part_floats(p, TCO, TK);
        js.start[k] = nblk;
        nblk += p.ntiles * p.splits;
        ms = p.splits > ms ? p.splits : ms;
        const long long red = (long long)p.ntiles * TCO * TK + p.nco * TCO;
        max_red = red > max_red ? red : max_red;
    }
    js.start[njobs] = nblk;
    if (off > scratch_floats) return S2AG_E_BADARG;
    const bool direct = ms <= 16;
    const dim3 grid(nblk), rgrid(cdiv(max_red, direct ? 256 : 32), njobs);
    hipStream_t st = (hipStream_t)stream;
    const bool one = s2ag_gru_coop_split_pieces() == 1;       // bf16 step mode: one piece per operand, one product
    if (big && s2ag::option(s2ag::OPT_WGRAD32_PIPE) && s2ag::wgrad_tr32p_supported(jobs, njobs)) {
        // opt-in variant (same tiles, same splits, same order of products: bit-identical dw): stash of step s + 1 beside the
        // MFMAs of step s, buffer loads with hardware bounds checks
        const int rc = s2ag::wgrad_tr32p_launch(&js, nblk, one ? 1 : 2, s2ag::option(s2ag::OPT_WGRAD32_PIPE) == 2 ? 2 : 3, st);
        if (rc) return rc;
        if (direct) hipLaunchKernelGGL((wgrad_tr_reduce_k<160, 160, true>), rgrid, dim3(256), 0, st, js);
        else hipLaunchKernelGGL((wgrad_tr_reduce_k<160, 160, false>), rgrid, dim3(256), 0, st, js);
    } else if (big && one) {
        hipLaunchKernelGGL((wgrad_tr32_k<160, 160, 3, 1, 2, 2, 1>), grid, dim3(256), 0, st, js);
        if (direct) hipLaunchKernelGGL((wgrad_tr_reduce_k<160, 160, true>), rgrid, dim3(256), 0, st, js);
        else hipLaunchKernelGGL((wgrad_tr_reduce_k<160, 160, false>), rgrid, dim3(256), 0, st, js);
    } else if (big) {
        hipLaunchKernelGGL((wgrad_tr32_k<160, 160, 3, 1>), grid, dim3(256), 0, st, js);
        if (direct) hipLaunchKernelGGL((wgrad_tr_reduce_k<160, 160, true>), rgrid, dim3(256), 0, st, js);
        else hipLaunchKernelGGL((wgrad_tr_reduce_k<160, 160, false>), rgrid, dim3(256), 0, st, js);
    } else if (one) {
        hipLaunchKernelGGL((wgrad_tr32_k<64, 64, 3, 2, 2, 2, 1>), grid, dim3(256), 0, st, js);
        if (direct) hipLaunchKernelGGL((wgrad_tr_reduce_k<64, 64, true>), rgrid, dim3(256), 0, st, js);
        else hipLaunchKernelGGL((wgrad_tr_reduce_k<64, 64, false>), rgrid, dim3(256), 0, st, js);
    } else {
        hipLaunchKernelGGL((wgrad_tr32_k<64, 64, 3, 2>), grid, dim3(256), 0, st, js);
        if (direct) hipLaunchKernelGGL((wgrad_tr_reduce_k<64, 64, true>), rgrid, dim3(256), 0, st, js);
        else hipLaunchKernelGGL((wgrad_tr_reduce_k<64, 64, false>), rgrid, dim3(256), 0, st, js);
    }
    S2AG_LAUNCH_CHECK();
    return 0;
}

// default workgroup count (96: beside a cooperative recurrence, see target_blocks32)
extern "C" long long s2ag_f32_wgrad_tr_scratch_floats(const s2ag_bf16_wgrad_args* jobs, int njobs) {
    return s2ag_f32_wgrad_tr_scratch_floats_n(jobs, njobs, 0);
}
extern "C" int s2ag_f32_wgrad_tr(const s2ag_bf16_wgrad_args* jobs, int njobs, float* scratch, long long scratch_floats,
                                 void* stream) {
    return s2ag_f32_wgrad_tr_n(jobs, njobs, scratch, scratch_floats, 0, stream);
}

extern "C" int s2ag_wgrad_tr_set_trace(void* buf) {
    g_tr_trace = static_cast<unsigned long long*>(buf);
    return 0;
}
S2AG_DET_HOOK(wgrad_tr)
