// Issue cost of the VALU instruction classes the ring kernels are made of (scratch experiment, not part of the library): every wave runs
// `iters` iterations of 32 independent instructions of ONE class (8 accumulators x 4), or of a mix, between two s_memtime reads.
// SIMD-cycles per instruction = cycles / (instructions per wave x waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdint.h>
using f32x2 = __attribute__((ext_vector_type(2))) float;

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define FMA(i)   asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pb), "v"(pc));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
#define EXP(i)   asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define RCP(i)   asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define LOG(i)   asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
#define FRACT(i) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
#define BFI(i)   asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(m), "v"(b));
#define MAXF(i)  asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define DPP(i)   asm volatile("v_mov_b32_dpp %0, %0 wave_rol:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define ADD(i)   asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
// mixes: one transcendental + K plain fmas on other registers (does the transcendental unit run beside the fma pipe?)
#define EXP_FMA1(i) asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %2, %3, %1" : "+v"(a[i]), "+v"(d[i]) : "v"(b), "v"(c));
#define EXP_FMA3(i) asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %2, %3, %1\n v_fma_f32 %4, %2, %3, %4\n v_fma_f32 %5, %2, %3, %5" : "+v"(a[i]), "+v"(d[i]), "+v"(e[i]), "+v"(f[i]) : "v"(b), "v"(c));
#define EXP_PK2(i)  asm volatile("v_exp_f32 %0, %0\n v_pk_fma_f32 %1, %2, %3, %1\n v_pk_fma_f32 %4, %2, %3, %4" : "+v"(a[i]), "+v"(p[i]), "+v"(q[i]) : "v"(pb), "v"(pc));

template <int MODE>
__global__ void __launch_bounds__(1024) probe(int iters, float *sink, unsigned long long *out) {
    float a[8], d[8], e[8], f[8]; f32x2 p[8], q[8];
    const float b = 1.0f + threadIdx.x * 1e-7f, c = 1e-9f; const f32x2 pb = {b, b}, pc = {c, c}; const unsigned m = 0x7fffffffu;
    for (int i = 0; i < 8; ++i) { a[i] = 1.0f + i; d[i] = e[i] = f[i] = 0.5f * i; p[i] = f32x2{1.0f * i, b}; q[i] = p[i]; }
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define X4(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)
        if constexpr (MODE == 0) { X4(FMA) }
        if constexpr (MODE == 1) { X4(PKFMA) }
        if constexpr (MODE == 2) { X4(PKADD) }
        if constexpr (MODE == 3) { X4(PKMUL) }
        if constexpr (MODE == 4) { X4(EXP) }
        if constexpr (MODE == 5) { X4(RCP) }
        if constexpr (MODE == 6) { X4(LOG) }
        if constexpr (MODE == 7) { X4(FRACT) }
        if constexpr (MODE == 8) { X4(BFI) }
        if constexpr (MODE == 9) { X4(MAXF) }
        if constexpr (MODE == 10) { X4(DPP) }
        if constexpr (MODE == 11) { X4(ADD) }
        if constexpr (MODE == 12) { X4(EXP_FMA1) }
        if constexpr (MODE == 13) { X4(EXP_FMA3) }
        if constexpr (MODE == 14) { X4(EXP_PK2) }
    }
    const unsigned long long c1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + d[i] + e[i] + f[i] + p[i].x + p[i].y + q[i].x + q[i].y;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = c1 - c0;
}
#define INST(M) template __global__ void probe<M>(int, float *, unsigned long long *);
INST(0) INST(1) INST(2) INST(3) INST(4) INST(5) INST(6) INST(7) INST(8) INST(9) INST(10) INST(11) INST(12) INST(13) INST(14)
extern "C" int run_probe(int mode, int grid, int block, int iters, float *sink, unsigned long long *out, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (mode) {
#define C_(M) case M: probe<M><<<grid, block, 0, s>>>(iters, sink, out); break;
        C_(0) C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11) C_(12) C_(13) C_(14)
        default: return -1;
    }
    return (int)hipGetLastError();
}
