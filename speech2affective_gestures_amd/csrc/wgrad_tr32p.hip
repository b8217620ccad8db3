// OPT-IN VARIANT of wgrad_tr32_k<160, 160, 3, 1> (csrc/wgrad_tr.hip): the fp32-operand weight gradients of the GRUs and of the
// text TCN (net/multimodal_context_net_v2.py:281,406,480; net/tcn.py:19,25 -- the dW of loss.backward(), processor_v2.py:841,937).
// Config switch WGRAD32_PIPE (default OFF: the default kernel's binary does not move; tools/ab_variants.sh times one against
// the other).  A file of its own on purpose: a variant that is to be A/B-timed must not perturb the default translation unit.
//
// SAME MATH, SAME ORDER -- the results are bit-identical to the default kernel's (tests/test_gpu_zy_variants.py holds dw to
// torch.equal): the same 160 x 160 tile on four waves, the same hi / lo bf16 split of every operand value, the same three
// piece products per tile in the same order, the same 32-row steps, the same split of the contraction over workgroups, the
// same second launch that sums the splits.  What differs is the SCHEDULE of a step, because the default kernel's step is its
// instruction count (one wave per SIMD at ~500 VGPRs; profiles/r03_cfg3_roofline_table.md: 156 us at 7.5 % of HBM and 7.4 %
// MFMA-busy; per 32-row step and wave 403 vector-ALU instructions -- 94 v_cndmask, 47 AGPR<->VGPR copies, 16 v_readlane of
// spilled scalars -- then a barrier, then 75 MFMAs: split+LDS ~1 200 cycles, loads+barrier 1 300-1 900, MFMAs 1 560, serialised):
//
//   1. SOFTWARE PIPELINE ACROSS THE BARRIER.  The default loop is  stash(s) | fetch | barrier | mma(s) : the vector ALU (split
//      into bf16 pieces, LDS stores) and the matrix pipe never work at the same time, because the barrier sits between them
//      and there is no second wave on the SIMD to fill the gap.  Here step s + 1 is split and stored into the OTHER LDS image
//      while the MFMAs of step s run:   { mma(s) || stash(s + 1) } | fetch | barrier.   One barrier per step as before; the
//      stash code is placed between the MFMA groups of the five row tiles and `sched_group_barrier` puts two of its
//      instructions behind every MFMA, so the 75 MFMAs (1 200 cycles at 16 per v_mfma_f32_16x16x32_bf16) cover most of the
//      ~210 vector-ALU instructions that are left (profiles/r06_variants_isa.txt: 150 of 212 are issued between MFMAs).
//   2. BUFFER LOADS WITH HARDWARE BOUNDS CHECKS instead of 64-bit flat pointers with per-chunk select + zero-fill: a chunk that
//      must read as zeros (rows past the end of the matrix, pad channels, the causal pad rows of a dilated tap) gets a byte
//      offset beyond num_records and the load returns zeros -- no pointer select, no 4 x v_cndmask per chunk, 32-bit offsets
//      (half the address registers and half the adds).  The gy offsets advance by a constant: one v_add_u32 per chunk and step.
//   3. No diagnostic stamps in the loop (the default carries 3 conditional s_memtime stores per step).
//
// Static evidence (no GPU in r06): profiles/r06_variants_isa.txt -- registers, spills, instruction mix of the main loop, the
// spread of the MFMAs and an in-order issue model (2 953 -> 1 689 cycles per step with two register sets), default vs variant.
// Random job mixes bit-identical on the device model: profiles/r06_variants_fuzz.txt.
#include <stdlib.h>

#include "s2ag_common.h"

namespace {
using namespace s2ag;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
typedef unsigned short bf16_t;

#include "wgrad_tr_shared.h"

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

constexpr unsigned OOB = 0x80000000u;      // >= every num_records the host admits (wgrad_tr32p_supported): reads as zeros

// NPC: pieces per operand (2: hi + lo, three products, the fp32 step; 1: hi only, one product: the bf16 step mode)
// RING: register sets = steps of global loads in flight.  3 (option value 1): as the default kernel, ~460 registers, the sets
// in flight live in AGPRs (~40 copy instructions per step).  2 (option value 2): everything in 256 architectural VGPRs + the
// accumulators, two steps (~2 x 1 300 cycles) of latency cover instead of three -- which wins is a question for the clock.
template <int NPC, int RING>
__global__ __launch_bounds__(256, 1) void wgrad_tr32p_k(const TrJobs js) {
    constexpr int TCO = 160, TK = 160, NTH = 256, WR = 2, WC = 2;
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
    constexpr int NA = (32 * CA) / NTH, NB = (32 * CB) / NTH;   // 5 + 5 chunks per thread and step
    static_assert((32 * CA) % NTH == 0 && (32 * CB) % NTH == 0, "every thread owns whole chunks");
    constexpr int WA = TCO / (16 * WR), WB = TK / (16 * WC);
    static_assert(WA == NA && WA == NB, "one G chunk and one X chunk are stashed beside every row tile's MFMAs");
    __shared__ __attribute__((aligned(16))) bf16_t Gs[2][NPC][32 * PA];     // [image][hi / lo]
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

    // descriptors from wave-uniform values only (the job is chosen from blockIdx): base pointer + byte count
    const int n_clips = p.M / p.Lq;
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.gy), 0, (int)((long long)p.M * p.ldg * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.x), 0, (int)(((long long)(n_clips - 1) * p.x_clip + (long long)p.Lin * p.ldx) * 4), 0x00020000);

    unsigned goff[NA], xoff[NB];                                // byte offsets (OOB: this chunk always reads zeros)
    unsigned xq[NB];
    int lds_a[NA], lds_b[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int id = tid + NTH * i;
        const int ra = id / CA, ca = id - ra * CA;
        lds_a[i] = ra * PA + ca * 4;
        goff[i] = co0 + ca * 4 < p.Cout ? (unsigned)(((m_beg + ra) * p.ldg + co0 + ca * 4) * 4) : OOB;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int id = tid + NTH * i;
        const int rb = id / CB, cb = id - rb * CB;
        lds_b[i] = rb * PB + cb * 4;
        const int m = m_beg + rb;                                // (rows >= M: clip index >= n_clips, beyond num_records)
        const int n = m / p.Lq;
        xq[i] = (unsigned)(m - n * p.Lq);
        // a pad-channel chunk starts beyond 2^31 and stays there (as goff): no per-step column test
        xoff[i] = c0 + cb * 4 < p.Cvalid
            ? (unsigned)((n * (int)p.x_clip + ((int)xq[i] * p.pos_mul + p.pos_off + tap * p.pos_tap) * p.ldx + c0 + cb * 4) * 4) : OOB;
    }
    const unsigned g_step = 32u * p.ldg * 4u, x_step = 32u * p.pos_mul * p.ldx * 4u;
    // (+ the extra advance across a clip boundary)
    const unsigned x_step_wrap = x_step + (unsigned)((p.x_clip - (long long)p.Lq * p.pos_mul * p.ldx) * 4);
    const int row_off = p.pos_off + tap * p.pos_tap;
    const unsigned Lq = (unsigned)p.Lq, Lin = (unsigned)p.Lin, pos_mul = (unsigned)p.pos_mul;
    f32x4 rg[RING][NA], rx[RING][NB];
    auto fetch = [&](int set) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            rg[set][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_g, goff[i], 0, 0));
            goff[i] += g_step;                                   // (an OOB chunk stays beyond 2^31: the host bounds the total advance)
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const unsigned row = xq[i] * pos_mul + (unsigned)row_off;       // (a negative row wraps to >= Lin)
            rx[set][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, row < Lin ? xoff[i] : OOB, 0, 0));
            // next step: 32 rows on (plan() guarantees Lq >= 32: at most one clip boundary per step); frame index modulo Lq
            // without a select: min(q + 32, q + 32 - Lq) as unsigned (the second operand wraps to ~2^32 while q + 32 < Lq)
            const unsigned qn = xq[i] + 32u;
            xq[i] = min(qn, qn - Lq);
            xoff[i] += xq[i] < qn ? x_step_wrap : x_step;
        }
    };
    float bacc[NA][4];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) bacc[i][j] = 0.f;
    // 4 floats -> 4 hi + 4 lo bf16 (8 bytes each): exactly the default kernel's split
    auto split_store = [&](bf16_t* hi_img, bf16_t* lo_img, int off, f32x4 val) {
        const unsigned h01 = pk_bf16(val[0], val[1]), h23 = pk_bf16(val[2], val[3]);
        *reinterpret_cast<uint2*>(hi_img + off) = make_uint2(h01, h23);
        if (NPC == 2) {
            const unsigned l01 = pk_bf16(val[0] - __uint_as_float(h01 << 16), val[1] - __uint_as_float(h01 & 0xffff0000u));
            const unsigned l23 = pk_bf16(val[2] - __uint_as_float(h23 << 16), val[3] - __uint_as_float(h23 & 0xffff0000u));
            *reinterpret_cast<uint2*>(lo_img + off) = make_uint2(l01, l23);
        }
    };
    // one G chunk + one X chunk of register set `set` into LDS image `img` (five such pairs = the whole step)
    auto stash_pair = [&](int set, int img, int i, bool bias) {
        split_store(Gs[img][0], Gs[img][NPC - 1], lds_a[i], rg[set][i]);
        if (bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bacc[i][j] += rg[set][i][j];
        }
        split_store(Xs[img][0], Xs[img][NPC - 1], lds_b[i], rx[set][i]);
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
    // the MFMAs of the step in image `img`; beside the MFMAs of row tile a, chunk pair a of the NEXT step goes from register
    // set `set` into image img ^ 1 (nobody reads that image before the barrier that follows)
    auto mma_stash = [&](int img, int set, bool bias) {
        bf16x8 bh[WB], bl[NPC == 2 ? WB : 1];
#pragma unroll
        for (int b = 0; b < WB; ++b) {
            bh[b] = frag(Xs[img][0], tr_b + b * 16, PB);
            if (NPC == 2) bl[b * (NPC - 1)] = frag(Xs[img][NPC - 1], tr_b + b * 16, PB);
        }
#pragma unroll
        for (int a = 0; a < WA; ++a) {
            const bf16x8 ah = frag(Gs[img][0], tr_a + a * 16, PA);
            if (NPC == 2) {
                const bf16x8 al = frag(Gs[img][NPC - 1], tr_a + a * 16, PA);
#pragma unroll
                for (int b = 0; b < WB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[b], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < WB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[b * (NPC - 1)], acc[a][b], 0, 0, 0);
            }
#pragma unroll
            for (int b = 0; b < WB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[b], acc[a][b], 0, 0, 0);
            stash_pair(set, img ^ 1, a, bias);
            // the emitted order of this row tile: every MFMA followed by two vector-ALU instructions of the stash (an in-order
            // wave issues them while the 16-cycle MFMA occupies the matrix pipe), the LDS stores spread between them
#pragma unroll
            for (int k = 0; k < WB * (NPC == 2 ? 3 : 1); ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, NPC == 2 ? 2 : 4, 0);      // VALU
                if (k % 4 == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // DS write
            }
        }
    };

    // Prologue: RING steps in flight, step 0 into image 0.  m_chunk is a multiple of 192 rows = 6 steps (plan(): row_multiple),
    // so a split that is not the matrix's last consists of whole periods and its loads stay inside [m_beg, m_end); the rows
    // past M of the last split are beyond num_records.  The step stashed by the LAST iteration is never multiplied.
#pragma unroll
    for (int r = 0; r < RING; ++r) fetch(r);
#pragma unroll
    for (int i = 0; i < NA; ++i) stash_pair(0, 0, i, do_bias);
    fetch(0);
    __syncthreads();
    constexpr int PERIOD = 6;                                    // steps until (register set, LDS image) repeats; 6 steps = 192 rows
    static_assert(PERIOD % RING == 0 && PERIOD % 2 == 0, "ring of 2 or 3 register sets x 2 LDS images");
    for (int mb = m_beg; mb < m_end; mb += 32 * PERIOD) {
#pragma unroll
        for (int r = 0; r < PERIOD; ++r) {
            // bias sums: only rows of THIS split (the step stashed beside the last MFMAs belongs to the next one)
            const bool bias = do_bias && mb + 32 * (r + 1) < m_end;
            mma_stash(r & 1, (r + 1) % RING, bias);
            fetch((r + 1) % RING);                               // the set just emptied takes the loads of step s + 1 + RING
            __syncthreads();
        }
    }
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
        S2AG_DET_WAVES_BEGIN          // (deterministic mode: the waves add their bias sums one after the other)
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int ca = (tid + NTH * i) % CA;
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(&bsum[ca * 4 + j], bacc[i][j]);
        }
        S2AG_DET_WAVES_END
        __syncthreads();
        for (int i = tid; i < TCO; i += NTH)
            if (co0 + i < p.Cout) p.part_b[((long long)split * p.nco + cot) * TCO + i] = bsum[i];
    }
}
}  // namespace

namespace s2ag {
// 32-bit byte offsets with the top bit reserved for "reads as zeros": every operand of the launch must span < 2 GiB
bool wgrad_tr32p_supported(const s2ag_bf16_wgrad_args* jobs, int njobs) {
    for (int k = 0; k < njobs; ++k) {
        const s2ag_bf16_wgrad_args& g = jobs[k];
        const long long gb = (long long)g.N * g.Lq * g.ldg * 4;
        const long long xb = ((long long)(g.N - 1) * g.x_clip + (long long)g.Lin * g.ldx) * 4;
        // (+ the advance of an offset that runs past the end of its split: one period of rows)
        const long long slack = 192ll * 4 * (g.ldg > (long long)g.pos_mul * g.ldx ? g.ldg : (long long)g.pos_mul * g.ldx) + 4 * llabs(g.x_clip);
        if (gb <= 0 || xb <= 0 || gb + slack >= (1ll << 31) || xb + slack >= (1ll << 31)) return false;
        if (g.x_clip < (long long)g.Lq * g.pos_mul * g.ldx) return false;        // offsets must grow with the row index
        if (g.x_clip < (long long)g.Lin * g.ldx) return false;                   // ... and a row past the last clip lies past num_records
    }
    return true;
}

// `jobs_struct`: the TrJobs of s2ag_f32_wgrad_tr_n (same layout: both files include wgrad_tr_shared.h), passed by value to the kernel
int wgrad_tr32p_launch(const void* jobs_struct, int nblk, int pieces, int ring, hipStream_t st) {
    const TrJobs& js = *static_cast<const TrJobs*>(jobs_struct);
    if (pieces == 1 && ring == 2) hipLaunchKernelGGL((wgrad_tr32p_k<1, 2>), dim3(nblk), dim3(256), 0, st, js);
    else if (pieces == 1) hipLaunchKernelGGL((wgrad_tr32p_k<1, 3>), dim3(nblk), dim3(256), 0, st, js);
    else if (ring == 2) hipLaunchKernelGGL((wgrad_tr32p_k<2, 2>), dim3(nblk), dim3(256), 0, st, js);
    else hipLaunchKernelGGL((wgrad_tr32p_k<2, 3>), dim3(nblk), dim3(256), 0, st, js);
    return (int)hipGetLastError();
}
}  // namespace s2ag
S2AG_DET_HOOK(wgrad_tr32p)
