// r4 probes behind scorer_x6.hip (scratch, not product code):
//  co<NV,CLS>: NV vector-ALU instructions of one class per v_mfma_f32_16x16x32_bf16 — do they hide in the matrix instruction's 16 cycles?
//  trread:     what ds_read_b64_tr_b16 delivers (lane, element) -> LDS halfword index, for a per-lane address table given by the host
#include <hip/hip_runtime.h>
#include <stdint.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
union Frag { bf16x8 v; uint32_t u[4]; };

template <int NV, int CLS> __device__ void body(int iters, float *sink, long long *out) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    Frag fa, fb;
    for (int i = 0; i < 4; ++i) { fa.u[i] = 0x3f803f80u + threadIdx.x; fb.u[i] = 0x3f003f00u + i; }
    float b = 0.5f + threadIdx.x * 1e-4f;
    float v[8]; uint32_t h[8];
    for (int i = 0; i < 8; ++i) { v[i] = 1.0f + threadIdx.x * 1e-3f + i; h[i] = threadIdx.x * 2654435761u + i; }
    const uint32_t c1 = 0x9e3779b1u + threadIdx.x;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb.v, acc[i], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                const int q = (i + n) & 7;
                if constexpr (CLS == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[q]) : "v"(b));
                else if constexpr (CLS == 1) asm volatile("v_and_b32 %0, %1, %0" : "+v"(h[q]) : "v"(c1));
                else if constexpr (CLS == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(h[q]) : "v"(c1), "v"(c1));
                else if constexpr (CLS == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(h[q]) : "v"(c1));
                else if constexpr (CLS == 4) asm volatile("v_cmp_ge_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(h[q]) : "v"(c1) : "vcc");
                else if constexpr (CLS == 5) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[q]) : "v"(b));
                else asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(h[q]));
            }
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i] + (float)h[i];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)&out[blockIdx.x], (unsigned long long)(t1 - t0));
}
#define CK(name, NV, CLS) extern "C" __global__ void __launch_bounds__(512) name(int iters, float *sink, long long *out) { body<NV, CLS>(iters, sink, out); }
CK(f0, 0, 0) CK(f1, 1, 0) CK(f2, 2, 0) CK(f3, 3, 0) CK(f4, 4, 0) CK(f6, 6, 0)
CK(a2, 2, 1) CK(a3, 3, 1) CK(p2, 2, 2) CK(p3, 3, 2) CK(m1, 1, 3) CK(m2, 2, 3) CK(s1, 1, 4) CK(s2, 2, 4) CK(u3, 3, 5) CK(h3, 3, 6)


// dependent-accumulator patterns of v_mfma_f32_16x16x32_bf16: NA accumulators visited round-robin (NA = 1: every MFMA depends on the one in
// front of it; 2: on the one two back, ...), 8 MFMAs per iteration
template <int NA> __device__ void body_dep(int iters, float *sink, long long *out) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    Frag fa, fb;
    for (int i = 0; i < 4; ++i) { fa.u[i] = 0x3f803f80u + threadIdx.x; fb.u[i] = 0x3f003f00u + i; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i % NA] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa.v, fb.v, acc[i % NA], 0, 0, 0);
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)&out[blockIdx.x], (unsigned long long)(t1 - t0));
}
#define DK(name, NA) extern "C" __global__ void __launch_bounds__(512) name(int iters, float *sink, long long *out) { body_dep<NA>(iters, sink, out); }
DK(d1, 1) DK(d2, 2) DK(d4, 4) DK(d8, 8)

// LDS halfword i holds the value i (as uint16).  Lane l reads 8 bytes at byte address addr[l] with ds_read_b64_tr_b16; out[l][0..3] = the
// four halfwords it received.
extern "C" __global__ void trread(const uint32_t *addr, uint32_t *out) {
    __shared__ uint16_t lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)lds + addr[threadIdx.x];
    uint64_t r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = (uint32_t)(r & 0xffff);
    out[threadIdx.x * 4 + 1] = (uint32_t)((r >> 16) & 0xffff);
    out[threadIdx.x * 4 + 2] = (uint32_t)((r >> 32) & 0xffff);
    out[threadIdx.x * 4 + 3] = (uint32_t)((r >> 48) & 0xffff);
}
// time: NR tr reads (independent) per iteration from a [rows][stride] image, per-lane address table; conflicts show as cycles per read
extern "C" __global__ void __launch_bounds__(512) trtime(const uint32_t *addr, int iters, long long *out, uint32_t *sink) {
    extern __shared__ uint16_t ldsd[];
    for (int i = threadIdx.x; i < 32768; i += 512) ldsd[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)ldsd + addr[threadIdx.x & 63];
    uint64_t acc = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        uint64_t r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r[k]) : "v"(a), "n"(0) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += r[k];
    }
    const long long t1 = clock64();
    if (acc == 0x123456789ull) sink[0] = 1;
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)&out[blockIdx.x], (unsigned long long)(t1 - t0));
}
