// Sustained fp32-MFMA issue rate and shader clock under load.  Every wave runs `iters` iterations of 8 independent, in-place
// v_mfma_f32_16x16x4_f32 (mode 0; 3: all-zero operands; 4: random mantissas), the same with one 1 KB store per 8 MFMAs (2), or
// 32 packed VALU fmas (1), and records s_memtime (shader clock) and s_memrealtime (100 MHz constant) around the loop:
// MHz = d(shader) / d(real) * 100.  Scratch experiment, not part of the library.
#include <hip/hip_runtime.h>
#include <stdint.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int MODE>
__device__ __forceinline__ void body(int iters, float *sink, unsigned long long *out) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    if (MODE == 3) { a = 0.0f; b = 0.0f; }
    if (MODE == 4) {
        a = __builtin_bit_cast(float, 0x3f800000u + (0x9e3779b9u * threadIdx.x) % 0x7fffffu);
        b = __builtin_bit_cast(float, 0x3f800000u + (0x85ebca6bu * (threadIdx.x + 7)) % 0x7fffffu);
    }
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{a + i, b};
    float *dst = sink + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 5 || MODE == 6) {
            // co-issue: MODE 5 = the waves of the upper half of the block stream VALU fmas while the lower half streams MFMAs (do VALU
            // instructions of one wave take matrix-pipe time from its SIMD partner?); MODE 6 = every wave: 8 MFMAs + 16 VALU fmas
            const bool valu_wave = MODE == 5 && (threadIdx.x >> 6) >= (blockDim.x >> 7);
            if (valu_wave || MODE == 6) {
#pragma unroll
                for (int u = 0; u < (MODE == 6 ? 2 : 4); ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i][0]) : "v"(b), "v"(a));
            }
            if (!valu_wave) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            }
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_elementwise_fma(v[i], f32x2{b, b}, f32x2{a, a});
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if constexpr (MODE == 2) {
                asm volatile("s_nop 7\n\ts_nop 7\n\tglobal_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(acc[0]) : "memory");
                dst += (size_t)gridDim.x * blockDim.x * 4 * ((it & 15) == 15 ? -15 : 1);
            }
        }
    }
    const unsigned long long c1 = clock64(), r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i][0] + v[i][1];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}
#define PROBE(M) extern "C" __global__ void __launch_bounds__(512) probe##M(int iters, float *sink, unsigned long long *out) { body<M>(iters, sink, out); }
PROBE(0) PROBE(1) PROBE(2) PROBE(3) PROBE(4) PROBE(5) PROBE(6)
