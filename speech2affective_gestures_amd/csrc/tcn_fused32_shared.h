// Shared between csrc/tcn_fused32.hip (the default clip-resident TemporalConvNet of the fp32 step) and its opt-in variant in a
// file of its own (csrc/tcn32p.hip: two clips per workgroup): tile constants, the packed-weight layout and the launch
// parameters.  Included INSIDE the anonymous namespace of the including file, after its vector typedefs (moved here verbatim
// from tcn_fused32.hip in r06 -- no default kernel changed: profiles/r06_isa_diff_since_298c878.txt).
#ifndef S2AG_TCN_FUSED32_SHARED_H
#define S2AG_TCN_FUSED32_SHARED_H

constexpr int CP = 320;                 // padded channels of an LDS row
constexpr int NCT = CP / 16;
constexpr int KT_TAP = CP / 32;
constexpr int NKT = 2 * KT_TAP;
constexpr int PITCH = 324;              // LDS row pitch in floats (1 296 B)
constexpr int MT = 3;                   // 16-row tiles: up to 48 frames
constexpr int CT_W = NCT / 4;
constexpr long long FRAG = (long long)NCT * NKT * 64 * 8;      // bf16 elements of one plane of one conv

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}

struct T32P {
    const float* x;                             // (clips*T, C) fp32
    float* h1[S2AG_TCN_MAX_BLOCKS];
    float* h2[S2AG_TCN_MAX_BLOCKS];
    float* y[S2AG_TCN_MAX_BLOCKS];
    const bf16_t* wfrag;                        // [conv][forward / data gradient][plane hi / lo][ct][kt][64][8]
    // backward (tcn32_bwd_k)
    const float* gy;                            // (clips*T, C) gradient w.r.t. the last block's output
    float* gx;
    float* gp1[S2AG_TCN_MAX_BLOCKS];            // gradients w.r.t. the convs' pre-activations (weight-gradient operands)
    float* gp2[S2AG_TCN_MAX_BLOCKS];
    const float* bias[2 * S2AG_TCN_MAX_BLOCKS];
    int dil[S2AG_TCN_MAX_BLOCKS];
    int n_blocks, n_clips, T, C;
    float drop_p, inv_keep;
    const unsigned long long* rng;
    unsigned site[2 * S2AG_TCN_MAX_BLOCKS];
    u32x4* keep;                                // one u32x4 per thread, workgroup and conv (tcn32_keep_k)
    int keep_total, keep_off;                   // clips of the keep layout [conv][clip][256]; first clip of this pass in it
    int save_clips;                             // clips < save_clips leave h1 / h2 / y of every block, the others only the last y
};

#endif
