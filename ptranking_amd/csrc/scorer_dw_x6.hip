// bf16x6 row contraction dW_1 = dZ_1^T . dropout(X) for WIDE inputs (F > 192: config 4 of BASELINE.json, F = 700), the layer-wise backward's
// largest kernel: 73 GFLOP at 524288 x 700, which the fp32-MFMA kernel (scorer.hip mlp_bwd_dw_lds_kernel<6>) runs at ~77 TFLOP/s (0.95 ms).
//
// Reference: the autograd graph of ptranking/ltr_adhoc/pointwise/point_ranker.py / ptranking/base/ranker.py:117-178 (d Linear.weight of the first
// layer); dropout site 0 recomputed from the key (ptr_dropout.h), as in the kernel this one replaces.
//
// Arithmetic: each fp32 operand is split into three bf16 planes (round to nearest) and a product is six v_mfma_f32_16x16x32_bf16 with fp32
// accumulation (scorer_x6.hip has the error analysis: not above the fp32-MFMA path's).  One workgroup of 8 waves walks its row chunk in slabs of
// 32 rows = one MFMA contraction step:
//   * the X slab [32 x 384 columns of this pass] and the dZ slab [32 x 112] are loaded ONE SLAB AHEAD into registers (16-byte loads, in flight
//     under the MFMAs of the current slab), then split and written as [row][feature] bf16 plane images in LDS (row strides 800 / 224 B: = 32 mod 256,
//     the conflict-free strides of ds_read_b64_tr_b16 measured in scratch/x6probe);
//   * both MFMA operands are "documents down the contraction": ds_read_b64_tr_b16 delivers the 16-feature x 32-document fragment of a tile out of
//     the row-major image (same reads as scorer_bwd_x6.hip);
//   * wave w owns the column tiles w, w + 8, w + 16 of the pass and all 7 output tiles: 21 accumulator tiles, 126 MFMAs per slab.
// HBM traffic is the floor here, not the matrix pipe: X is read once (1.47 GB at 524288 x 700), dZ once per pass of 384 columns — 1.94 GB in
// ~630 us = 3.1 TB/s, 80 % of what a read-only stream reaches on this part (3.9 TB/s, scratch/x6probe/bw.py).  Two variants measured and dropped
// (round 4): the two wave groups skewed by half an iteration so that one converts while the other multiplies (half images, one barrier per
// phase: 1.47 vs 1.27 ms for the whole backward — a lone wave per SIMD hides neither its LDS reads nor its load waits), and term-major MFMA
// order (dependent matrix instructions three issues apart: 1.32 vs 1.27 ms).
// Interface, partial layout (ws[block][flat parameter layout]) and bias gradient are those of mlp_bwd_dw_lds_kernel: the reduction is unchanged.
#include "ptr_device.h"
#include "ptr_dropout.h"
#include "ptr_mlp.h"

namespace ptr {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using i16x4 = __attribute__((ext_vector_type(4))) short;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
union DFrag { bf16x8 v; u32x4 q; uint32_t u[4]; };
using lds_u32x2_d = __attribute__((address_space(3))) u32x2;
using lds_i16x4_d = __attribute__((address_space(3))) i16x4;

constexpr int kD6S = 32;                         // rows per slab
// column tiles per pass: 24 (384 columns, eight waves x three tiles: r4 / r5) or — r6, where it costs no extra pass — 16 (sixteen waves: waves 0-7 / 8-15 share the column tiles
// W, W + 8 and split the seven dZ tiles even / odd; 2 x 4 accumulators per wave, <= 128 registers, four waves per SIMD: the staging pass is issued by twice the
// waves and a wave's MFMA phase is 2 x 4 instead of 3 x 7 tile products — linear_bw_x6.hip's 16-wave form, measured there)
__host__ __device__ constexpr int d6_xrs(int CT) { return CT == 24 ? 800 : 544; }      // X image: 768 / 512 B per row, padded by 32 (= 32 mod 256)
constexpr int kD6ZRS = 224, kD6ZPL = kD6S * kD6ZRS;      // dZ image: 112 bf16 per row
__host__ __device__ constexpr int d6_lds(int CT) { return 3 * kD6S * d6_xrs(CT) + 3 * kD6ZPL + kD6S * kAL * 4; }      // + bias scratch: 32 x 112 floats
static_assert(d6_lds(24) <= 160 * 1024 && d6_lds(16) <= 160 * 1024, "LDS budget");

__device__ __forceinline__ uint32_t d6_cvt_pk(float x0, float x1) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{x0, x1}, bf16x2)); }
__device__ __forceinline__ void d6_split2(float x0, float x1, uint32_t &p1, uint32_t &p2, uint32_t &p3) {
    p1 = d6_cvt_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xffff0000u);
    p2 = d6_cvt_pk(r0, r1);
    const float s0 = r0 - __uint_as_float(p2 << 16), s1 = r1 - __uint_as_float(p2 & 0xffff0000u);
    p3 = d6_cvt_pk(s0, s1);
}
__device__ __forceinline__ uint32_t d6_lds_addr(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p; }
__device__ __forceinline__ void d6_write4(uint32_t addr, int plane_bytes, const f32x4 v) {
    uint32_t a[3], b[3];
    d6_split2(v[0], v[1], a[0], a[1], a[2]);
    d6_split2(v[2], v[3], b[0], b[1], b[2]);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<lds_u32x2_d *>((uintptr_t)(addr + (uint32_t)(p * plane_bytes))) = u32x2{a[p], b[p]};
}

template <bool SITE0, int CT, int NW>
__global__ void __launch_bounds__(NW * 64, 1)
mlp_bwd_dw_x6_kernel(const float *__restrict__ A, int lda, const float *__restrict__ dZ, int K, int nt_base, MlpArgs a, float *__restrict__ ws,
                     size_t np_stride, size_t w_off, size_t b_off) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_d6[];
    constexpr int NT = NW * 64, NWT = CT / 8, HALVES = NW / 8, MTL = (kMT + HALVES - 1) / HALVES;
    constexpr int kD6XRS = d6_xrs(CT), kD6XPL = kD6S * kD6XRS;
    constexpr int kD6_X = 0, kD6_Z = 3 * kD6XPL, kD6_B = kD6_Z + 3 * kD6ZPL;
    constexpr int XW4 = CT * 4;                            // float4 per X-slice row (96 / 64)
    constexpr int SX = (kD6S * XW4 + NT - 1) / NT;         // X load slots per thread (6 / 2)
    constexpr int ZW4 = kAL / 4;                           // float4 per dZ row (28)
    constexpr int SZ = (kD6S * ZW4 + NT - 1) / NT;         // dZ load slots per thread (2 / 1, the last one partly used)
    const int R = a.R;
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4, Wall = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = Wall & 7, half = Wall >> 3;               // column tiles W, W + 8, ..; dZ tiles half, half + HALVES, ..
    const int chunk = ((R + gridDim.x - 1) / gridDim.x + kD6S - 1) / kD6S * kD6S;
    const int r_begin = blockIdx.x * chunk, r_end = min(R, r_begin + chunk);
    const uint32_t thr = drop_thr(a.p_drop);
    const float scale = (SITE0 && a.p_drop > 0.0f) ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    const int col0 = 16 * nt_base;
    const uint32_t lds0 = d6_lds_addr(smem_d6);

    // slot geometry is RECOMPUTED at each use from an opaque copy of the thread index: hoisted out of the slab loop it costs ~24 registers this
    // kernel does not have (21 accumulator tiles + 9 + 3 fragments + 8 prefetch registers)
    auto xgeo = [&](int s, int t_, int &row, int &col, bool &ok) __attribute__((always_inline)) {
        const int idx = s * NT + t_;
        row = idx / XW4; col = col0 + 4 * (idx % XW4); ok = col < K && idx < kD6S * XW4;      // K % 4 == 0 on this path: a float4 is inside the row or beyond it
    };
    auto zgeo = [&](int s, int t_, int &row, int &col, bool &ok) __attribute__((always_inline)) {
        const int idx = s * NT + t_;
        ok = idx < kD6S * ZW4; row = ok ? idx / ZW4 : 0; col = ok ? 4 * (idx % ZW4) : 0;
    };
    f32x4 rx[SX], rz[SZ], zsum[SZ];
#pragma unroll
    for (int s = 0; s < SZ; ++s) zsum[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    // raw loads from clamped (always valid) addresses; masks and the recomputed input dropout are applied at the split
    auto gload = [&](int r0) __attribute__((always_inline)) {
        int t_ = tid;
        asm volatile("" : "+v"(t_));
#pragma unroll
        for (int s = 0; s < SX; ++s) {
            int row, col; bool ok;
            xgeo(s, t_, row, col, ok);
            const int r = r0 + row;
            const int rc = r < r_end ? r : r_end - 1;
            rx[s] = *reinterpret_cast<const f32x4 *>(A + (size_t)rc * lda + (ok ? col : 0));
        }
#pragma unroll
        for (int s = 0; s < SZ; ++s) {
            int row, col; bool ok;
            zgeo(s, t_, row, col, ok);
            const int r = r0 + row;
            const int rc = r < r_end ? r : r_end - 1;
            rz[s] = *reinterpret_cast<const f32x4 *>(dZ + (size_t)rc * kAL + col);
        }
    };
    auto stage = [&](int r0) __attribute__((always_inline)) {                   // registers -> split -> plane images
        int t_ = tid;
        asm volatile("" : "+v"(t_));
#pragma unroll
        for (int s = 0; s < SX; ++s) {
            int row, col; bool ok;
            xgeo(s, t_, row, col, ok);
            const int r = r0 + row;
            const bool rok = r < r_end;
            f32x4 v = rx[s];
            if constexpr (SITE0) {
                uint32_t w0, w1;
                drop_bits(a.seed_lo, a.seed_hi, 0, rok ? r : r_end - 1, col >> 2, w0, w1);
                v = drop4(v, w0, w1, thr, scale);
            }
            const float okf = (rok & ok) ? 1.0f : 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] *= okf;
            d6_write4(lds0 + (uint32_t)(kD6_X + row * kD6XRS + 2 * (col - col0)), kD6XPL, v);
        }
#pragma unroll
        for (int s = 0; s < SZ; ++s) {
            int row, col; bool ok;
            zgeo(s, t_, row, col, ok);
            if (ok) {
                const float okf = (r0 + row < r_end) ? 1.0f : 0.0f;
                f32x4 v = rz[s];
#pragma unroll
                for (int c = 0; c < 4; ++c) { v[c] *= okf; zsum[s][c] += v[c]; }
                d6_write4(lds0 + (uint32_t)(kD6_Z + row * kD6ZRS + 2 * col), kD6ZPL, v);
            }
        }
    };
    // the fragment of tile t (16 features x 32 rows) of an image: lane (j, g) -> feature 16 t + j, contraction slots = rows {4g..4g+3, 16+4g..16+4g+3}
    auto read_tr = [&](DFrag (&f)[3], uint32_t img_lane, int plane_bytes, int row_bytes, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_i16x4_d *>((uintptr_t)(img_lane + (uint32_t)(p * plane_bytes + 32 * t))));
            const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_i16x4_d *>((uintptr_t)(img_lane + (uint32_t)(p * plane_bytes + 32 * t + 16 * row_bytes))));
            const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
            f[p].u[0] = l2[0]; f[p].u[1] = l2[1]; f[p].u[2] = h2[0]; f[p].u[3] = h2[1];
        }
    };
    const uint32_t tr_x = lds0 + (uint32_t)(kD6_X + (4 * g + (j >> 2)) * kD6XRS + 8 * (j & 3));
    const uint32_t tr_z = lds0 + (uint32_t)(kD6_Z + (4 * g + (j >> 2)) * kD6ZRS + 8 * (j & 3));

    f32x4 acc[NWT][MTL];
#pragma unroll
    for (int n = 0; n < NWT; ++n)
#pragma unroll
        for (int ml = 0; ml < MTL; ++ml) acc[n][ml] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (r_begin < r_end) gload(r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += kD6S) {
        stage(r0);
        __syncthreads();
        if (r0 + kD6S < r_end) gload(r0 + kD6S);           // next slab in flight under this slab's MFMAs
        DFrag xb[NWT][3];
#pragma unroll
        for (int n = 0; n < NWT; ++n) read_tr(xb[n], tr_x, kD6XPL, kD6XRS, W + 8 * n);
#pragma unroll
        for (int ml = 0; ml < MTL; ++ml) {
            const int mt = ml * HALVES + half;
            if (mt >= kMT) break;                          // wave-uniform
            DFrag za[3];
            read_tr(za, tr_z, kD6ZPL, kD6ZRS, mt);
#pragma unroll
            for (int n = 0; n < NWT; ++n) {
                f32x4 c = acc[n][ml];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(za[0].v, xb[n][2].v, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(za[1].v, xb[n][1].v, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(za[2].v, xb[n][0].v, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(za[0].v, xb[n][1].v, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(za[1].v, xb[n][0].v, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(za[0].v, xb[n][0].v, c, 0, 0, 0);
                acc[n][ml] = c;
            }
        }
        __syncthreads();
    }
    // partials: dW[o][k], o = 16 mt + 4 g + c (rows of the D tile), k = this wave's column (lane j of the D tile)
    float *out = ws + (size_t)blockIdx.x * np_stride + w_off;
#pragma unroll
    for (int n = 0; n < NWT; ++n) {
        const int k = col0 + 16 * (W + 8 * n) + j;
#pragma unroll
        for (int ml = 0; ml < MTL; ++ml)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int mt = ml * HALVES + half;
                const int o = 16 * mt + 4 * g + c;
                if (k < K && o < kH && mt < kMT) out[(size_t)o * K + k] = acc[n][ml][c];
            }
    }
    if (nt_base == 0) {                                    // d bias = column sums of dZ: per-thread sums over the slabs, then over the 32 slab rows
        float *bs = reinterpret_cast<float *>(smem_d6 + kD6_B);
#pragma unroll
        for (int s = 0; s < SZ; ++s) {
            int row, col; bool ok;
            zgeo(s, tid, row, col, ok);
            if (ok) *reinterpret_cast<f32x4 *>(bs + row * kAL + col) = zsum[s];
        }
        __syncthreads();
        if (tid < kH) {
            float v = 0.0f;
#pragma unroll 8
            for (int r = 0; r < kD6S; ++r) v += bs[r * kAL + tid];
            ws[(size_t)blockIdx.x * np_stride + b_off + tid] = v;
        }
    }
}

// all passes over the ntk column tiles of the layer.  The sixteen-wave form when it needs no more passes than the eight-wave / 24-tile form (every pass re-reads and
// re-splits dZ): F = 700 = 44 tiles is 2 passes of 24 against 3 of 16 — config 4 measured 1.858 ms against 1.881 with the extra pass.  PTR_DW_X6_FORM=16 / 24 forces a form.
int launch_dw_x6(const float *A, int lda, const float *dZ, int K, int ntk, const MlpArgs &a, float *ws, size_t np_stride, size_t w_off, size_t b_off,
                 int nblk, hipStream_t st, const char *who) {
    const char *fe = getenv("PTR_DW_X6_FORM");
    const int forced = fe ? atoi(fe) : 0;
    const bool f24 = forced == 24 || (forced != 16 && (ntk + 15) / 16 > (ntk + 23) / 24);
    auto go = [&](auto kern, int ct, int threads) -> int {
        if (int e = allow_lds(kern, d6_lds(ct))) return e;
        for (int base = 0; base < ntk; base += ct) {
            hipLaunchKernelGGL(kern, dim3(nblk), dim3(threads), d6_lds(ct), st, A, lda, dZ, K, base, a, ws, np_stride, w_off, b_off);
            if (int e = check_hip(hipGetLastError(), who)) return e;
        }
        return 0;
    };
    return f24 ? go(mlp_bwd_dw_x6_kernel<true, 24, 8>, 24, 512) : go(mlp_bwd_dw_x6_kernel<true, 16, 16>, 16, 1024);
}

// PTR_DW_X6: "0" never, "1" (default) the first layer's dW of wide inputs (more than 12 column tiles, i.e. F > 192) from 32768 rows on, "2" always
bool dw_x6_supported(int R, int K, int lda, const void *A) {
    const char *e = getenv("PTR_DW_X6");                   // read per call: tests and A/B runs switch it inside one process
    const int mode = e ? atoi(e) : 1;
    if (mode <= 0) return false;
    if (K % 4 != 0 || lda % 4 != 0 || (reinterpret_cast<uintptr_t>(A) & 15) != 0) return false;
    if ((K + 15) / 16 <= 12) return false;
    return mode >= 2 || R >= 32768;
}

}  // namespace ptr
