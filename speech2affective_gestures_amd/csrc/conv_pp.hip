// Poly-phase data gradient of the STRIDED wave-encoder convs (fp32, f32 MFMA) -- replaces ConvolutionBackward (input
// grad) of net/multimodal_context_net_v2.py:18-27 (Conv1d(16,32,15,stride=6), (32,64,15,6), (64,32,15,6)), which the
// general implicit-GEMM kernel ran in "residue mode" at 341 / 135 / 28 us (B = 256).
//
//   dx[n, p, ci] = sum_{t, co} gy[n, (p - t) / S, co] * w[co, ci, t]     over the taps t = p (mod S), 0 <= (p - t)/S < Lout
//
// With p = S*q + r the taps of PHASE r are t = r + S*i, i = 0 .. ceil((KS - r) / S) - 1, and the source frame is q - i:
// every phase is a stride-1 conv of gy with 2-3 taps whose output rows lie S frames apart in dx.  The contraction is
// short (K = taps * Cout <= 192) and the output narrow (Cin = 16 .. 64), so the 64 x 64 x 32 block tile of the general
// kernels wastes most of its MFMAs here and the work is really bound by moving gy in and dx out once:
//   * a wave keeps the B operands -- w[.., ci-tile, r + S*i] for ITS (phase, 16-column tile) pairs -- in registers for
//     the whole launch (<= 144 VGPRs), read once straight from the leaf's layout (reference (Cout, Cin, KS) or the
//     cached tap-major copy);
//   * a sub-tile = 16 consecutive q: its 16 + 2 source frames go through a wave-private LDS image (16-byte global loads,
//     next sub-tile in flight in registers, no block barrier anywhere), the A fragment of (tap i, 4 channels) is read
//     ONCE and serves every phase -- gy[q - i] does not depend on r;
//   * the S phases of a sub-tile interleave into 16 S consecutive frames of dx: the 16 x 16 accumulator tiles go to a
//     double-buffered LDS image of the workgroup's round and leave as 16-byte stores of whole rows, one barrier per
//     round (written straight from the accumulators -- 64-byte segments 1.5 KB apart -- the kernel ran at 1.8 TB/s).
// Wave teams: Cin = 16 -> one wave does all S phases of its own sub-tiles; Cin = 32 -> four waves share a sub-tile
// (column tile = wave & 1, phases of one parity); Cin = 64 -> wave = column tile, all phases.
#include "s2ag_common.h"

namespace {
using namespace s2ag;
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct PpP {
    const float* gy;
    const float* w;
    float* dx;
    int N, Lin, Lout, ldg, ldx, wtm, accumulate;
    int Q;        // ceil(Lin / S): q positions per clip
    int chunks;   // q chunks per clip
    int QC;       // q per chunk (multiple of 16)
};

template <int COUT, int CIN, int KS, int S>
__global__ __launch_bounds__(256) void conv_dgrad_pp_k(const PpP p) {
    constexpr int NCT = CIN / 16;                    // 16-column tiles of dx
    constexpr int WT = NCT == 1 ? 1 : 4;             // waves per team (one team works on one sub-tile)
    constexpr int PSTEP = NCT == 2 ? 2 : 1;          // phase step of a wave
    constexpr int NPH = S / PSTEP;                   // phases per wave
    constexpr int NTAP = (KS + S - 1) / S;           // taps per phase (max)
    constexpr int KC = COUT / 4;                     // 4-channel chunks of the contraction per tap
    constexpr int ROWS = 16 + NTAP - 1;              // source frames under a sub-tile
    constexpr int PITCH = COUT + 4;                  // LDS row pitch: 16 rows x 4 k land on 64 distinct banks
    constexpr int NLD = (ROWS * COUT / 4 + 63) / 64; // float4 loads per lane and sub-tile
    static_assert(S % PSTEP == 0 && COUT % 4 == 0 && CIN % 16 == 0, "shape");
    constexpr int TEAMS = 4 / WT;
    constexpr int OROWS = TEAMS * 16 * S;            // dx frames a workgroup produces per round
    constexpr int OPITCH = CIN + 4;                  // pitch of the output image (the 4 row groups of a tile: 2 banks sets)
    constexpr int C4 = CIN / 4;
    __shared__ float lds[4][ROWS * PITCH];
    __shared__ float oimg[2][OROWS * OPITCH];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int team = wave / WT, tw = wave % WT;
    const int ct = NCT == 1 ? 0 : tw % NCT;
    const int pstart = NCT == 2 ? tw / NCT : 0;
    const int lr = lane & 15, lk = lane >> 4;

    // ---- B operands of this wave: b[j][i][c] = w[co = 4c + lk][ci = 16 ct + lr][t = r_j + S i], r_j = pstart + j PSTEP
    float b[NPH][NTAP][KC];
#pragma unroll
    for (int j = 0; j < NPH; ++j) {
        const int r = pstart + j * PSTEP;
#pragma unroll
        for (int i = 0; i < NTAP; ++i) {
            const int t = r + S * i;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
                const int co = 4 * c + lk, ci = 16 * ct + lr;
                float v = 0.f;
                if (t < KS) v = p.wtm ? p.w[((long long)co * KS + t) * CIN + ci] : p.w[((long long)co * CIN + ci) * KS + t];
                b[j][i][c] = v;
            }
        }
    }

    const int n = blockIdx.x / p.chunks;
    const int q_lo = (blockIdx.x - n * p.chunks) * p.QC;
    int q_hi = q_lo + p.QC;
    if (q_hi > p.Q) q_hi = p.Q;
    const float* gyc = p.gy + (long long)n * p.Lout * p.ldg;
    float* dxc = p.dx + (long long)n * p.Lin * p.ldx;
    float* img = lds[wave];

    // source frames q0 - (NTAP - 1) .. q0 + 15 of a sub-tile, zero outside [0, Lout)
    float4 st[NLD];
    auto fetch = [&](int q0) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = (u * 64 + lane) * 4;                 // float index inside the (ROWS, COUT) block
            const int row = e / COUT, col = e - row * COUT;
            const int l = q0 - (NTAP - 1) + row;
            st[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < ROWS && (unsigned)l < (unsigned)p.Lout)
                st[u] = *reinterpret_cast<const float4*>(gyc + (long long)l * p.ldg + col);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = (u * 64 + lane) * 4;
            const int row = e / COUT, col = e - row * COUT;
            if (row < ROWS) *reinterpret_cast<float4*>(img + row * PITCH + col) = st[u];
        }
    };

    // a round: every team one sub-tile (the round's frames of dx are contiguous: S * 16 * TEAMS rows from S * qr)
    int buf = 0;
    if (q_lo + team * 16 < q_hi) fetch(q_lo + team * 16);
    for (int qr = q_lo; qr < q_hi; qr += TEAMS * 16, buf ^= 1) {
        const int q0 = qr + team * 16;
        float* out = oimg[buf];
        if (q0 < q_hi) {
            __builtin_amdgcn_wave_barrier();
            stash();
            __builtin_amdgcn_wave_barrier();
            const int qn = q0 + TEAMS * 16;
            if (qn < q_hi) fetch(qn);
            // A fragments: a[i][c] = gy[q0 + lr - i][4c + lk]  (image row lr + NTAP - 1 - i)
            float a[NTAP][KC];
#pragma unroll
            for (int i = 0; i < NTAP; ++i)
#pragma unroll
                for (int c = 0; c < KC; ++c) a[i][c] = img[(lr + NTAP - 1 - i) * PITCH + 4 * c + lk];
            // the phases' accumulator chains interleaved (independent MFMAs back to back)
            f32x4 acc[NPH];
#pragma unroll
            for (int j = 0; j < NPH; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NTAP; ++i)
#pragma unroll
                for (int c = 0; c < KC; ++c)
#pragma unroll
                    for (int j = 0; j < NPH; ++j)
                        if (pstart + j * PSTEP + S * i < KS)       // wave-uniform: the last phases have one tap less
                            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][c], b[j][i][c], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NPH; ++j) {
                const int r = pstart + j * PSTEP;
                float* o = out + ((team * 16 + lk * 4) * S + r) * OPITCH + 16 * ct + lr;
#pragma unroll
                for (int v = 0; v < 4; ++v) o[v * S * OPITCH] = acc[j][v];
            }
        }
        __syncthreads();
        // the round's rows of dx: frames S*qr .. , whole rows as 16-byte stores (the other image is being filled meanwhile)
        int rows = (q_hi - qr) * S;
        if (rows > OROWS) rows = OROWS;
        const int pos0 = S * qr;
        if (pos0 + rows > p.Lin) rows = p.Lin - pos0;
        for (int e = threadIdx.x; e < rows * C4; e += 256) {
            const int row = e / C4, c4 = e - row * C4;
            float4 v = *reinterpret_cast<const float4*>(out + row * OPITCH + 4 * c4);
            float4* d = reinterpret_cast<float4*>(dxc + (long long)(pos0 + row) * p.ldx + 4 * c4);
            if (p.accumulate) {
                const float4 u = *d;
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            *d = v;
        }
    }
}

template <int COUT, int CIN, int KS, int S>
int launch_pp(const PpP& p0, hipStream_t stream) {
    PpP p = p0;
    p.Q = cdiv(p.Lin, S);
    // two workgroups per CU (one round of the chip: measured 256 / 512 / 768 / 1024 workgroups = 67 / 61 / 75 / 71 us at the
    // conv2 shape); a chunk is a multiple of the 4 x 16 (or 16, in team mode) q a workgroup covers per round
    const int round = (CIN == 16 ? 4 : 1) * 16;
    int per_clip = cdiv(512, p.N);
    if (per_clip < 1) per_clip = 1;
    p.QC = cdiv(cdiv(p.Q, per_clip), round) * round;
    p.chunks = cdiv(p.Q, p.QC);
    hipLaunchKernelGGL((conv_dgrad_pp_k<COUT, CIN, KS, S>), dim3(p.N * p.chunks), dim3(256), 0, stream, p);
    return 1;
}
// ---------------------------------------------------------------------------------------------------------------------
// Forward of the same layers in the same style (Conv1d(16, 32, 15, stride=6) of the wave encoder; the general straight-
// line kernel spent half of its 64-wide column tile on nothing at Cout = 32: 154 us at B = 256).  Without padding and with
// channels-last rows of exactly Cin floats the window of output frame l is ONE run of K = KS * Cin contiguous floats that
// starts S * Cin floats after the previous one ("flat window"):
//     y[n, l, co] = act( bias[co] + sum_k x[n, S Cin l + k] * W[co, k] ),    W = the tap-major weight (Cout, KS * Cin)
// A wave keeps W for all its column tiles in registers (K / 4 x Cout / 16 <= 120 VGPRs), stages the (15 S + KS) Cin floats
// under 16 output frames in a wave-private LDS image (a 4-float pad after every S Cin floats puts the 16 rows x 4 k of a
// fragment read on 64 different banks), and emits the fp64 column sums of the BatchNorm that follows (one partial row per
// workgroup).
struct FwP {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    double* stats;
    int N, Lin, Lout, ldy, wtm, act;
    float slope;
    int chunks, LC;     // output-frame chunks per clip, frames per chunk (multiple of 64)
};

template <int CIN, int COUT, int KS, int S>
__global__ __launch_bounds__(256) void conv_fwd_fw_k(const FwP p) {
    constexpr int K = KS * CIN, KC = K / 4, NCT = COUT / 16;
    constexpr int RS = S * CIN;                          // floats between the windows of consecutive frames
    constexpr int SPAN = 15 * RS + K;                    // floats under a sub-tile of 16 frames
    constexpr int IMG = SPAN + 4 * (SPAN / RS) + 4;
    constexpr int NLD = (SPAN / 4 + 63) / 64;
    static_assert(RS % 4 == 0 && K % 4 == 0 && KC * NCT <= 128, "shape");
    __shared__ float lds[4][IMG];
    __shared__ double red[4][2][COUT];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lk = lane >> 4;

    // b[ct][c] = W[co = 16 ct + lr][k = 4 c + lk]
    float b[NCT][KC];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const int co = 16 * ct + lr, k = 4 * c + lk;
            const int t = k / CIN, ci = k - t * CIN;
            b[ct][c] = p.wtm ? p.w[(long long)co * K + k] : p.w[((long long)co * CIN + ci) * KS + t];
        }
    float bias[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) bias[ct] = p.bias ? p.bias[16 * ct + lr] : 0.f;

    const int n = blockIdx.x / p.chunks;
    const int l_lo = (blockIdx.x - n * p.chunks) * p.LC;
    int l_hi = l_lo + p.LC;
    if (l_hi > p.Lout) l_hi = p.Lout;
    const float* xc = p.x + (long long)n * p.Lin * CIN;
    const long long xlen = (long long)p.Lin * CIN;
    float* yc = p.y + (long long)n * p.Lout * p.ldy;
    float* img = lds[wave];

    float4 st[NLD];
    auto fetch = [&](int l0) {
        const long long base = (long long)l0 * RS;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = (u * 64 + lane) * 4;
            st[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < SPAN && base + e + 3 < xlen) st[u] = *reinterpret_cast<const float4*>(xc + base + e);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = (u * 64 + lane) * 4;
            if (e < SPAN) *reinterpret_cast<float4*>(img + e + 4 * (e / RS)) = st[u];
        }
    };

    double s1[NCT], s2[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) s1[ct] = s2[ct] = 0.0;

    int l0 = l_lo + wave * 16;
    if (l0 < l_hi) fetch(l0);
    for (; l0 < l_hi; l0 += 64) {
        __builtin_amdgcn_wave_barrier();
        stash();
        __builtin_amdgcn_wave_barrier();
        if (l0 + 64 < l_hi) fetch(l0 + 64);
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* arow = img + (RS + 4) * lr + lk;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            const float a = arow[4 * c + 4 * ((4 * c) / RS)];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[ct][c], acc[ct], 0, 0, 0);
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int l = l0 + 4 * lk + v;
                if (l < l_hi) {
                    const float r = apply_act(acc[ct][v] + bias[ct], p.act, p.slope);
                    yc[(long long)l * p.ldy + 16 * ct + lr] = r;
                    s1[ct] += (double)r;
                    s2[ct] += (double)r * (double)r;
                }
            }
    }
    if (p.stats) {          // (2, gridDim.x, COUT): one partial row per workgroup
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            double a1 = s1[ct], a2 = s2[ct];
            a1 += __shfl_xor(a1, 16, 64);
            a1 += __shfl_xor(a1, 32, 64);
            a2 += __shfl_xor(a2, 16, 64);
            a2 += __shfl_xor(a2, 32, 64);
            if (lane < 16) {
                red[wave][0][16 * ct + lane] = a1;
                red[wave][1][16 * ct + lane] = a2;
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * COUT) {
            const int which = threadIdx.x / COUT, c = threadIdx.x - which * COUT;
            const double v = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
            p.stats[((size_t)which * gridDim.x + blockIdx.x) * COUT + c] = v;
        }
    }
}
}  // namespace

// rows of BatchNorm partials written (= workgroups) if launched, 0 if the shape is outside this kernel
int s2ag_conv_fwd_fw(const float* x, const float* w, const float* bias, float* y, int N, int Lin, int Lout, int Cin,
                     int Cout, int ks, int stride, int pad, int dil, int ldx, int ldy, int wtm, int act, float slope,
                     float drop_p, double* stats, hipStream_t stream) {
    if (ks != 15 || stride != 6 || pad != 0 || dil != 1 || Cin != 16 || Cout != 32 || ldx != Cin || drop_p > 0.f ||
        (act != S2AG_ACT_NONE && act != S2AG_ACT_LEAKY) || ((uintptr_t)x & 15) != 0 || Lout < 64)
        return 0;
    FwP p{};
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.stats = stats; p.N = N; p.Lin = Lin; p.Lout = Lout; p.ldy = ldy; p.wtm = wtm;
    p.act = act; p.slope = slope;
    int per_clip = cdiv(512, N);
    if (per_clip < 1) per_clip = 1;
    p.LC = cdiv(cdiv(Lout, per_clip), 64) * 64;
    p.chunks = cdiv(Lout, p.LC);
    hipLaunchKernelGGL((conv_fwd_fw_k<16, 32, 15, 6>), dim3(N * p.chunks), dim3(256), 0, stream, p);
    return N * p.chunks;
}

namespace {
}  // namespace

// 1 = launched; 0 = shape outside this kernel (the caller falls back to the general one)
int s2ag_conv_dgrad_pp(const float* gy, const float* w, float* dx, int N, int Lin, int Lout, int Cin, int Cout, int ks,
                       int stride, int pad, int dil, int ldg, int ldx, int wtm, int accumulate, hipStream_t stream) {
    if (ks != 15 || stride != 6 || pad != 0 || dil != 1 || ldg % 4 != 0 || ((uintptr_t)gy & 15) != 0 || ldx % 4 != 0 ||
        ((uintptr_t)dx & 15) != 0)
        return 0;
    if ((long long)N * cdiv(cdiv(Lin, 6), 16) > 0x7fffffffLL) return 0;
    PpP p{};
    p.gy = gy; p.w = w; p.dx = dx; p.N = N; p.Lin = Lin; p.Lout = Lout; p.ldg = ldg; p.ldx = ldx; p.wtm = wtm;
    p.accumulate = accumulate;
    if (Cout == 32 && Cin == 16) return launch_pp<32, 16, 15, 6>(p, stream);
    if (Cout == 64 && Cin == 32) return launch_pp<64, 32, 15, 6>(p, stream);
    if (Cout == 32 && Cin == 64) return launch_pp<32, 64, 15, 6>(p, stream);
    return 0;
}
