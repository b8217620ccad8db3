// How fast do 16x16x32 bf16 MFMAs issue on gfx950 when consecutive instructions use (a) ONE accumulator, (b) a new
// accumulator every CH instructions (CH = 1, 2, 3, 6, 12) out of 16?   build: hipcc --offload-arch=gfx950 -O3 -o p this.hip
// Measured (MI355X, one wave): CH = 1: 27.0 cycles per MFMA; CH = 2 / 3 / 6 / 12: 16.5 / 16.3 / 16.2 / 16.1 -- chains of two
// or more on one accumulator already run at the pipe's 16-cycle rate, so the three-product chains of gemm_sp_k are not what
// holds it at ~37 cycles per MFMA (operand delivery from LDS with one wave per SIMD and workgroup is).
#include <hip/hip_runtime.h>
#include <stdio.h>
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int CH>
__global__ __launch_bounds__(64) void probe(float* out, long long* cyc, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    f32x4 acc[16];
    for (int t = 0; t < 16; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int CH>
void run(float* out, long long* cyc, int iters) {
    hipLaunchKernelGGL(probe<CH>, dim3(1), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(probe<CH>, dim3(1), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("chain %2d per accumulator (16 accumulators round robin): %.1f cycle-counter ticks per MFMA\n", CH,
           (double)c / ((double)iters * 16 * CH));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256); hipMalloc(&cyc, 8);
    run<1>(out, cyc, 2000); run<2>(out, cyc, 2000); run<3>(out, cyc, 2000); run<6>(out, cyc, 1000); run<12>(out, cyc, 500);
    return 0;
}
