// Probe of ds_read_b64_tr_b16 (gfx950): which elements does lane l receive?  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/probe/tr16_probe.hip -o /tmp/tr16_probe && /tmp/tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
constexpr int PITCH = 72;
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short img[64 * PITCH];
    for (int i = threadIdx.x; i < 64 * PITCH; i += 64) img[i] = (short)((i / PITCH) * 100 + (i % PITCH));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, t = l & 15;
    const short* a = &img[(8 * g + t / 4) * PITCH + (t % 4) * 4];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d;
    hipMalloc(&d, 256 * sizeof(short));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j)
            if (h[l * 4 + j] != (8 * (l >> 4) + j) * 100 + (l & 15)) ++bad;
    printf("mismatches vs hypothesis (lane l gets img[8*(l>>4) + j][l & 15]): %d\n", bad);
    for (int l = 0; l < 64; l += 1) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
