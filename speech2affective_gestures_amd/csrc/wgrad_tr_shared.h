// Shared between csrc/wgrad_tr.hip (the default weight-gradient kernels) and the opt-in variants that live in files of their
// own (csrc/wgrad_tr32p.hip): the job description of a launch, the planner that cuts the contraction into equally long
// blocks, and the second launch that sums the stored split tiles into dw / db.  Included INSIDE the anonymous namespace of the
// including file, after `using namespace s2ag;` (moved here verbatim from wgrad_tr.hip in r06 -- no default kernel changed:
// profiles/r06_isa_diff_since_298c878.txt).
#ifndef S2AG_WGRAD_TR_SHARED_H
#define S2AG_WGRAD_TR_SHARED_H

struct TrP {
    const void* gy;             // bf16 rows (wgrad_tr_k) or fp32 rows (wgrad_tr32_k)
    const void* x;
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
    int start[S2AG_BF16_MAX_WGRAD_JOBS + 1];   // first block of every job in the compact 1-D grid (tile fastest, then split)
    int njobs;
    int xcd_remap;
    unsigned long long* trace;                 // diagnostics (s2ag_wgrad_tr_set_trace): s_memtime stamps of block 0, thread 0
};

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

int plan(const s2ag_bf16_wgrad_args* g, TrP& p, int TCO, int TK, int rows_per_block, int elems16 = 8, int row_multiple = 128) {
    if (!g || !g->gy || !g->x || !g->dw || g->N <= 0 || g->Lq <= 0 || g->ks <= 0) return S2AG_E_BADARG;
    const int am = elems16 - 1;                                  // elements per 16 bytes - 1
    if ((g->Cvalid & am) || (g->ldx & am) || (g->ldg & am) || g->Cvalid > g->Cp) return S2AG_E_BADARG;
    if (g->Lq < 32) return S2AG_E_UNSUPPORTED;                  // the loader steps (clip, frame) by 32 rows with one wrap
    if ((reinterpret_cast<uintptr_t>(g->x) | reinterpret_cast<uintptr_t>(g->gy)) & 15) return S2AG_E_BADARG;
    p.gy = g->gy; p.x = g->x; p.dw = g->dw; p.db = g->db;
    p.M = g->N * g->Lq; p.Lq = g->Lq; p.Lin = g->Lin; p.x_clip = g->x_clip; p.ldx = g->ldx; p.ldg = g->ldg;
    p.pos_mul = g->pos_mul; p.pos_off = g->pos_off; p.pos_tap = g->pos_tap;
    p.ks = g->ks; p.Cp = g->Cp; p.Cvalid = g->Cvalid; p.Cout = g->Cout; p.Cin = g->Cin;
    p.d_co = g->d_co; p.d_t = g->d_t; p.d_c = g->d_c; p.flat_cin = g->flat_cin;
    p.ks_out = g->flat_cin > 0 ? g->ks_out : g->ks;
    p.nco = cdiv(g->Cout, TCO);
    p.kct = cdiv(g->Cvalid, TK);
    p.ntiles = p.nco * g->ks * p.kct;
    int splits = cdiv(p.M, rows_per_block > 256 ? rows_per_block : 256);     // at least 8 steps of 32 rows per block
    if (splits < 1) splits = 1;
    p.m_chunk = cdiv(cdiv(p.M, splits), row_multiple) * row_multiple;      // 32 rows * the kernel's ring
    p.splits = cdiv(p.M, p.m_chunk);
    return 0;
}

// rows of the contraction per block such that all jobs together make ~`target` equally long blocks
int rows_per_block(const s2ag_bf16_wgrad_args* jobs, int njobs, int TCO, int TK, int target) {
    long long tile_rows = 0;
    for (int k = 0; k < njobs; ++k)
        tile_rows += (long long)cdiv(jobs[k].Cout, TCO) * jobs[k].ks * cdiv(jobs[k].Cvalid, TK) * jobs[k].N * jobs[k].Lq;
    // rounded UP to the kernels' row granularity: a block count just above the number of CUs would cost a whole second
    // round (one workgroup per CU at these register counts: 288 blocks took twice the time of 256)
    long long r = (tile_rows + target - 1) / (target > 0 ? target : 1);
    r = (r + 191) / 192 * 192;
    if (r < 384) r = 384;
    for (int it = 0; it < 64; ++it) {                            // grow until the block count really is <= target
        long long blocks = 0;
        for (int k = 0; k < njobs; ++k)
            blocks += (long long)cdiv(jobs[k].Cout, TCO) * jobs[k].ks * cdiv(jobs[k].Cvalid, TK) *
                      cdiv((long long)jobs[k].N * jobs[k].Lq, r);
        if (blocks <= target) break;
        r += 192;
    }
    return (int)r;
}

long long part_floats(const TrP& p, int TCO, int TK) {
    return (long long)p.splits * p.ntiles * TCO * TK + (long long)p.splits * p.nco * TCO;
}

#endif
