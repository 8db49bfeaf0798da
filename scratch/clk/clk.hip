// Measures the sustained shader clock under an fp32-MFMA load (clock64 = shader cycles, wall_clock64 = 100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ void __launch_bounds__(256) burn(float *out, long long *clk, int iters, int nacc) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    long long c1 = clock64(), w1 = wall_clock64();
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
int main(int argc, char **argv) {
    int nb = 256 * (argc > 1 ? atoi(argv[1]) : 1);     // blocks of 4 waves per CU
    float *out; long long *clk;
    hipMalloc(&out, nb * 256 * 4); hipMalloc(&clk, nb * 16);
    for (int rep = 0; rep < 2; ++rep) {
        int iters = 100000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(burn, dim3(nb), dim3(256), 0, 0, out, clk, iters, 8);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[4]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        double flops = (double)nb * 4 * iters * 8 * 2048.0;
        printf("rep %d: %.3f ms  %.1f TFLOP/s  shader cycles %lld wall(100MHz) %lld -> %.3f GHz; mfma per wave %lld -> %.2f cycles/MFMA/wave\n", rep, ms,
               flops / ms / 1e9, h[0], h[1], (double)h[0] / h[1] * 0.1, (long long)iters * 8, (double)h[0] / (iters * 8.0));
    }
    return 0;
}
