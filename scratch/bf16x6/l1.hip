// PROTOTYPE (scratch, not product code): layer 1 of the pointsf scorer, h = relu(X W^T + b), with fp32 products formed from six bf16 matrix
// instructions (see probe.hip).  Question: what does the forward cost when the matrix pipe is 2.5x cheaper but operands must be split (VALU) and
// the weight fragments are 48 bytes per lane per six MFMAs (LDS read rate)?
//   X  [R][F] fp32 (F % 8 == 0, F <= 160), h [R][112] fp32, Wp [3][112][KP] bf16 planes prepared by the host (KP = 160, zero padded), b [112].
//   8 waves x 32-row tiles; LDS: the three weight planes, row stride KP + 8 bf16 (336 B: float4 reads of 8 consecutive rows cover the banks).
#include <hip/hip_runtime.h>
#include <stdint.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
constexpr int KP = 160, LDW = KP + 8, MT = 7, RT = 2, NS = KP / 32;

__device__ __forceinline__ void split3(float a, uint32_t &p1, uint32_t &p2, uint32_t &p3) {
    const uint32_t b1 = __float_as_uint(a) & 0xffff0000u;
    const float r1 = a - __uint_as_float(b1);
    const uint32_t b2 = __float_as_uint(r1) & 0xffff0000u;
    p1 = b1; p2 = b2; p3 = __float_as_uint(r1 - __uint_as_float(b2));
}
__device__ __forceinline__ uint32_t pack_hi(uint32_t x0, uint32_t x1) { return __builtin_amdgcn_perm(x1, x0, 0x07060302u); }
__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { union { u32x4 u; bf16x8 b; } c; c.u = v; return c.b; }

template <int MODE>   // 0: full; 1: no split (planes = raw bits, wrong numbers: VALU-free timing); 2: MFMAs only on plane 0 (one term)
__device__ void body(const float *__restrict__ X, const uint16_t *__restrict__ Wp, const float *__restrict__ bias, int R, int F, float *__restrict__ H) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];        // [3][112][LDW]
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int i = tid; i < 3 * 112 * (KP / 8); i += nthr) {                  // 16-byte pieces
        const int p = i / (112 * (KP / 8)), rem = i - p * 112 * (KP / 8), r = rem / (KP / 8), c = rem - r * (KP / 8);
        *reinterpret_cast<u32x4 *>(smem + ((size_t)p * 112 + r) * LDW + 8 * c) = *reinterpret_cast<const u32x4 *>(Wp + ((size_t)p * 112 + r) * KP + 8 * c);
    }
    __syncthreads();
    const int lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6, nw = nthr >> 6;
    const int ntiles = (R + 16 * RT - 1) / (16 * RT);
    for (int tile = blockIdx.x * nw + wave; tile < ntiles; tile += gridDim.x * nw) {
        const float *xr[RT];
        int row[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) { row[rt] = tile * 16 * RT + 16 * rt + j; xr[rt] = X + (size_t)(row[rt] < R ? row[rt] : R - 1) * F; }
        f32x4 acc[MT][RT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias + 16 * mt + 4 * g);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[mt][rt] = b4;
        }
        f32x4 xn[RT][2];
        auto load = [&](int s, f32x4 (&x)[RT][2]) {
            const int k0 = 32 * s + 8 * g;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                x[rt][0] = *reinterpret_cast<const f32x4 *>(xr[rt] + (k0 < F ? k0 : 0));
                x[rt][1] = *reinterpret_cast<const f32x4 *>(xr[rt] + (k0 + 4 < F ? k0 + 4 : 0));
            }
        };
        load(0, xn);
#pragma unroll 1
        for (int s = 0; s < NS; ++s) {
            f32x4 xc[RT][2];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) { xc[rt][0] = xn[rt][0]; xc[rt][1] = xn[rt][1]; }
            if (s + 1 < NS) load(s + 1, xn);
            // B operand planes of this slice
            u32x4 bq[RT][3];
            const int k0 = 32 * s + 8 * g;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (k0 + e < F) ? xc[rt][e >> 2][e & 3] : 0.0f;
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) { bq[rt][0][q] = __float_as_uint(v[2 * q]); bq[rt][1][q] = __float_as_uint(v[2 * q + 1]); bq[rt][2][q] = __float_as_uint(v[2 * q]) ^ 1u; }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t a1, a2, a3, c1, c2, c3;
                        split3(v[2 * q], a1, a2, a3);
                        split3(v[2 * q + 1], c1, c2, c3);
                        bq[rt][0][q] = pack_hi(a1, c1); bq[rt][1][q] = pack_hi(a2, c2); bq[rt][2][q] = pack_hi(a3, c3);
                    }
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                u32x4 aq[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) aq[p] = *reinterpret_cast<const u32x4 *>(smem + ((size_t)p * 112 + 16 * mt + j) * LDW + 32 * s + 8 * g);
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    f32x4 c = acc[mt][rt];
                    if constexpr (MODE != 2) {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(aq[0]), as_bf(bq[rt][2]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(aq[1]), as_bf(bq[rt][1]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(aq[2]), as_bf(bq[rt][0]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(aq[0]), as_bf(bq[rt][1]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(aq[1]), as_bf(bq[rt][0]), c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(aq[0]), as_bf(bq[rt][0]), c, 0, 0, 0);
                    acc[mt][rt] = c;
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 h = acc[mt][rt];
#pragma unroll
                for (int c = 0; c < 4; ++c) h[c] = fmaxf(h[c], 0.0f);
                if (row[rt] < R) *reinterpret_cast<f32x4 *>(H + (size_t)row[rt] * 112 + 16 * mt + 4 * g) = h;
            }
    }
}
extern "C" __global__ void __launch_bounds__(512) l1_full(const float *X, const uint16_t *Wp, const float *b, int R, int F, float *H) { body<0>(X, Wp, b, R, F, H); }
extern "C" __global__ void __launch_bounds__(512) l1_nosplit(const float *X, const uint16_t *Wp, const float *b, int R, int F, float *H) { body<1>(X, Wp, b, R, F, H); }
extern "C" __global__ void __launch_bounds__(512) l1_oneterm(const float *X, const uint16_t *Wp, const float *b, int R, int F, float *H) { body<2>(X, Wp, b, R, F, H); }
