// v_mfma_f32_4x4x1_16b_f32 probe: lane layout of A / B / D and issue cost alone and mixed with v_mfma_f32_16x16x4_f32.
#include <hip/hip_runtime.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;

extern "C" __global__ void layout(float *out) {
    const int l = threadIdx.x;
    const float a = (float)(l + 1), b = (float)(1000 * (l + 1));
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
}

template <int MODE> __device__ void body(int iters, float *sink, long long *out) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 1e-4f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {                 // 8 x 16x16x4
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        } else if constexpr (MODE == 1) {          // 8 x 4x4x1 (independent accumulators)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
        } else if constexpr (MODE == 2) {          // 6 x 16x16x4 + 2 x 4x4x1
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[7], 0, 0, 0);
        } else if constexpr (MODE == 3) {          // 4 dependent 4x4x1 on ONE accumulator + 4 on another
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[1], 0, 0, 0); }
        } else {                                   // 8 dependent 4x4x1 on one accumulator
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[0], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
extern "C" __global__ void __launch_bounds__(256) t0(int iters, float *sink, long long *out) { body<0>(iters, sink, out); }
extern "C" __global__ void __launch_bounds__(256) t1(int iters, float *sink, long long *out) { body<1>(iters, sink, out); }
extern "C" __global__ void __launch_bounds__(256) t2(int iters, float *sink, long long *out) { body<2>(iters, sink, out); }
extern "C" __global__ void __launch_bounds__(256) t3(int iters, float *sink, long long *out) { body<3>(iters, sink, out); }
extern "C" __global__ void __launch_bounds__(256) t4(int iters, float *sink, long long *out) { body<4>(iters, sink, out); }
