// fp32 products from bf16 matrix instructions (scratch probe, not product code).
//   a = a1 + a2 + a3 exactly (three bf16 pieces of the 24-bit mantissa); a*b ~ a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1): six
//   v_mfma_f32_16x16x32_bf16 per 32 k instead of eight v_mfma_f32_16x16x4_f32.  Questions: accuracy vs fp32 MFMA, issue cost, clock.
#include <hip/hip_runtime.h>
#include <stdint.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
union Frag { bf16x8 v; uint32_t u[4]; };

__device__ __forceinline__ void split3(float a, uint32_t &p1, uint32_t &p2, uint32_t &p3) {
    const uint32_t b1 = __float_as_uint(a) & 0xffff0000u;
    const float r1 = a - __uint_as_float(b1);
    const uint32_t b2 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(b2);
    p1 = b1; p2 = b2; p3 = __float_as_uint(r2);
}
__device__ __forceinline__ uint32_t pack_hi(uint32_t x0, uint32_t x1) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }

// A [16][K] row-major, B [K][16] row-major, K % 32 == 0.  out[0] = fp32 MFMA, out[1] = 6-term, out[2] = 3-term; each [16][16].
extern "C" __global__ void accuracy(const float *A, const float *B, int K, float *out) {
    const int l = threadIdx.x, j = l & 15, g = l >> 4;
    f32x4 d32 = {0, 0, 0, 0}, d6 = {0, 0, 0, 0}, d3 = {0, 0, 0, 0};
    for (int k = 0; k < K; k += 4) d32 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[j * K + k + g], B[(k + g) * 16 + j], d32, 0, 0, 0);
    for (int k = 0; k < K; k += 32) {
        Frag a[3], b[3];
        for (int e = 0; e < 8; e += 2) {
            uint32_t x[2][3], y[2][3];
            for (int h = 0; h < 2; ++h) {
                split3(A[j * K + k + 8 * g + e + h], x[h][0], x[h][1], x[h][2]);
                split3(B[(k + 8 * g + e + h) * 16 + j], y[h][0], y[h][1], y[h][2]);
            }
            for (int p = 0; p < 3; ++p) { a[p].u[e / 2] = pack_hi(x[0][p], x[1][p]); b[p].u[e / 2] = pack_hi(y[0][p], y[1][p]); }
        }
        // small terms first
        d6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[2].v, d6, 0, 0, 0);
        d6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, b[1].v, d6, 0, 0, 0);
        d6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2].v, b[0].v, d6, 0, 0, 0);
        d6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[1].v, d6, 0, 0, 0);
        d6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, b[0].v, d6, 0, 0, 0);
        d6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[0].v, d6, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[1].v, d3, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1].v, b[0].v, d3, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0].v, b[0].v, d3, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) {
        out[(4 * g + r) * 16 + j] = d32[r];
        out[256 + (4 * g + r) * 16 + j] = d6[r];
        out[512 + (4 * g + r) * 16 + j] = d3[r];
    }
}

// issue cost: MODE 0 = 8 independent 16x16x4 f32, 1 = 8 independent 16x16x32 bf16, 2 = 8 dependent bf16, 3 = the same 8 bf16 with 8 VALU
// (split-like: and / sub) interleaved.  out[block] = shader cycles, out[4096 + block] = 100 MHz ticks.
template <int MODE> __device__ void body(int iters, float *sink, long long *out) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 1e-4f;
    Frag fa, fb;
    for (int i = 0; i < 4; ++i) { fa.u[i] = 0x3f803f80u + threadIdx.x; fb.u[i] = 0x3f003f00u + i; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb.v, acc[i], 0, 0, 0);
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb.v, acc[0], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb.v, acc[i], 0, 0, 0);
                const uint32_t m = __float_as_uint(v[i]) & 0xffff0000u;
                v[i] = (v[i] - __uint_as_float(m)) * 1.0001f + b;
            }
        }
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) { atomicMax((unsigned long long *)&out[blockIdx.x], (unsigned long long)(t1 - t0)); atomicMax((unsigned long long *)&out[4096 + blockIdx.x], (unsigned long long)(w1 - w0)); }
}
extern "C" __global__ void __launch_bounds__(512) r0(int iters, float *sink, long long *out) { body<0>(iters, sink, out); }
extern "C" __global__ void __launch_bounds__(512) r1(int iters, float *sink, long long *out) { body<1>(iters, sink, out); }
extern "C" __global__ void __launch_bounds__(512) r2(int iters, float *sink, long long *out) { body<2>(iters, sink, out); }
extern "C" __global__ void __launch_bounds__(512) r3(int iters, float *sink, long long *out) { body<3>(iters, sink, out); }

// VALU beside fp32 MFMAs: NV independent-of-the-MFMA VALU instructions (and / sub / fma, the dropout-generator mix) per v_mfma_f32_16x16x4_f32
template <int NV, bool IMUL> __device__ void body_v(int iters, float *sink, long long *out) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 1e-4f;
    float v[8]; uint32_t h[8];
    for (int i = 0; i < 8; ++i) { v[i] = a + i; h[i] = threadIdx.x * 2654435761u + i; }
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                if constexpr (IMUL) { h[(i + n) & 7] = (h[(i + n) & 7] ^ (h[(i + n) & 7] >> 15)) * 0x2c1b3c6du; }
                else { v[(i + n) & 7] = fmaf(v[(i + n) & 7], 1.0001f, b); }
            }
        }
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i] + (float)h[i];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) { atomicMax((unsigned long long *)&out[blockIdx.x], (unsigned long long)(t1 - t0)); atomicMax((unsigned long long *)&out[4096 + blockIdx.x], (unsigned long long)(w1 - w0)); }
}
#define VK(name, NV, IM) extern "C" __global__ void __launch_bounds__(1024) name(int iters, float *sink, long long *out) { body_v<NV, IM>(iters, sink, out); }
VK(v0, 0, false) VK(v2, 2, false) VK(v4, 4, false) VK(v6, 6, false) VK(v8, 8, false) VK(v12, 12, false)
VK(m1, 1, true) VK(m2, 2, true) VK(m4, 4, true)

// which VALU classes cost matrix-pipe time beside v_mfma_f32_16x16x4_f32?  NV instructions of ONE class per MFMA
//   CLS 0 v_fma_f32 | 1 v_xor_b32 | 2 v_lshrrev_b32 | 3 v_mul_lo_u32 | 4 v_cmp + v_cndmask (2 instr) | 5 v_max_f32 | 6 v_mul_f32 | 7 v_add_u32
template <int NV, int CLS> __device__ void body_c(int iters, float *sink, long long *out) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 1e-4f;
    float v[8]; uint32_t h[8];
    for (int i = 0; i < 8; ++i) { v[i] = a + i; h[i] = threadIdx.x * 2654435761u + i; }
    const uint32_t c1 = 0x9e3779b1u + threadIdx.x;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int q = (i + n) & 7;
                if constexpr (CLS == 0) v[q] = fmaf(v[q], 1.0001f, b);
                else if constexpr (CLS == 1) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(h[q]) : "v"(c1));
                else if constexpr (CLS == 2) asm volatile("v_lshrrev_b32 %0, 1, %0\n\tv_or_b32 %0, 0x40000000, %0" : "+v"(h[q]));
                else if constexpr (CLS == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(h[q]) : "v"(c1));
                else if constexpr (CLS == 4) asm volatile("v_cmp_ge_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(h[q]) : "v"(c1) : "vcc");
                else if constexpr (CLS == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[q]) : "v"(b));
                else if constexpr (CLS == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[q]) : "v"(b));
                else asm volatile("v_add_u32 %0, %0, %1" : "+v"(h[q]) : "v"(c1));
            }
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i] + (float)h[i];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)&out[blockIdx.x], (unsigned long long)(t1 - t0));
}
#define CK(name, NV, CL) extern "C" __global__ void __launch_bounds__(1024) name(int iters, float *sink, long long *out) { body_c<NV, CL>(iters, sink, out); }
CK(c0, 4, 0) CK(c1, 4, 1) CK(c2, 2, 2) CK(c3, 4, 3) CK(c4, 2, 4) CK(c5, 4, 5) CK(c6, 4, 6) CK(c7, 4, 7)
