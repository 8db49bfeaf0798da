// bf16x6 weight gradient of a Linear layer with one NARROW side (at most 144 columns): dW[N][K] = dY^T X, db = column sums of dY — the
// listsf projections (136 -> 408, 136 -> 136, 136 -> 128, 128 -> 256) and the default pointsf's 100 x 100 layers, which the fp32-MFMA
// kernel (linear.hip linear_bwd_w_kernel) runs at 40-50 % of its matrix peak and with the narrow operand re-read once per block of output tiles.
//
// Reference: the autograd backward of torch.nn.Linear inside ptranking/base/utils.py:288-356 (`get_stacked_FFNet`) and
// ptranking/base/list_ranker.py:176-254 (the encoder's projections).
//
// Same machine as scorer_dw_x6.hip (fp32 products from six v_mfma_f32_16x16x32_bf16 on exactly split operands, [row][column] bf16 plane images
// in LDS, both MFMA operands through ds_read_b64_tr_b16, slabs of 32 rows prefetched one slab ahead): the NARROW operand (<= 16 MT columns,
// MT = 7 or 9) is held whole, the WIDE operand streams in passes of 384 columns; a wave owns 3 wide tiles x MT narrow tiles.  Either side of
// the product can be the narrow one:
//   narrow = dY (N <= 144): tile rows are out-features, lanes in-features: partial[n * K + k]; db from per-thread column sums of the dY slab;
//   narrow = X  (K <= 140): tile rows are in-features: partial[n * K + k] written with lanes along n; the X image carries a column of ONES
//                           behind its last feature, so row K of the product IS db (the fused scorer backward's trick).
// Partials: ws[chunk][N * K + N], the layout of linear_bwd_w_kernel — reduce_chunks_kernel is unchanged.
#include "ptr_device.h"

namespace ptr {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using i16x4 = __attribute__((ext_vector_type(4))) short;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
union LFrag { bf16x8 v; u32x4 q; uint32_t u[4]; };
using lds_u32x2_l = __attribute__((address_space(3))) u32x2;
using lds_i16x4_l = __attribute__((address_space(3))) i16x4;

constexpr int kL6S = 32;                         // rows per slab
constexpr int kL6CT = 24;                        // wide tiles per pass (384 columns); r6: CT = 8 (128 columns) for the products whose wide side is that small
// wide image: 768 B per row padded to 800, 512 to 544, or 256 to 288 (all = 32 mod 256: conflict-free transpose reads)
__host__ __device__ constexpr int l6_wrs(int CT) { return CT == 24 ? 800 : CT == 16 ? 544 : 288; }      // 768 / 512 / 256 B of tiles + 32
__host__ __device__ constexpr int l6_nrs(int MT) { return MT == 9 ? 288 : 224; }             // narrow image row stride (bytes): 144 / 112 bf16, both = +-32 mod 256
// CT = 8: the column-sum scratch of the epilogue lies over the wide image (dead by then) — 49 / 55 KB per workgroup, two workgroups per CU
__host__ __device__ constexpr int l6_lds(int MT, int CT) { return 3 * kL6S * l6_wrs(CT) + 3 * kL6S * l6_nrs(MT) + (CT == 24 ? kL6S * 16 * MT * 4 : 0); }

__device__ __forceinline__ uint32_t l6_cvt_pk(float x0, float x1) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{x0, x1}, bf16x2)); }
__device__ __forceinline__ void l6_split2(float x0, float x1, uint32_t &p1, uint32_t &p2, uint32_t &p3) {      // round-to-nearest split, scorer_x6.hip
    p1 = l6_cvt_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xffff0000u);
    p2 = l6_cvt_pk(r0, r1);
    const float s0 = r0 - __uint_as_float(p2 << 16), s1 = r1 - __uint_as_float(p2 & 0xffff0000u);
    p3 = l6_cvt_pk(s0, s1);
}
__device__ __forceinline__ uint32_t l6_lds_addr(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p; }
__device__ __forceinline__ void l6_write4(uint32_t addr, int plane_bytes, const f32x4 v) {
    uint32_t a[3], b[3];
    l6_split2(v[0], v[1], a[0], a[1], a[2]);
    l6_split2(v[2], v[3], b[0], b[1], b[2]);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<lds_u32x2_l *>((uintptr_t)(addr + (uint32_t)(p * plane_bytes))) = u32x2{a[p], b[p]};
}

// Zn: the narrow operand [R][ldn], NN columns (+ a column of ones at index NN when `ones`); Aw: the wide operand [R][ldw], KW columns, this launch
// covers its tiles tile0 .. tile0 + 23.  Element (o, k) of the product -> part[o * s_n + k * s_w] for o < NN, the ones row -> part[bias_off + k];
// colsum != 0 (narrow = dY, first pass): the column sums of the narrow slab -> part[bias_off + o].
// r6, CT = 8: ONE wide tile per wave, <= 128 registers, two workgroups per CU.  The slab loop is stage (VALU split) -> barrier -> MFMAs -> barrier with the next
// slab's loads in flight behind the first barrier; with one wide tile per wave the MFMA phase is 42-54 instructions and a slab took 3.2 us of which the matrix
// pipe worked 0.3: a second workgroup per CU runs its stage while the first multiplies (default pointsf's 100 x 100 layers: 51 -> 3x us per call).
// r6, CT = 16 with NW = 16 waves: the waves 0-7 / 8-15 split the narrow tiles (even / odd), a wave holds 2 wide x 5 narrow accumulators (<= 128 registers, four
// waves per SIMD): the staging pass is issued by twice the waves and the MFMA phase is half as long per wave — for products with 9..16 wide tiles per pass.
template <int MT, int CT, int NW>
__global__ void __launch_bounds__(NW * 64, CT == 8 ? 2 : 1)
linear_bw_x6_kernel(const float *__restrict__ Zn, int ldn, int NN, int ones, const float *__restrict__ Aw, int ldw, int KW, int tile0, int ntp, int R,
                    float *__restrict__ ws, size_t ws_stride, size_t s_n, size_t s_w, size_t bias_off, int colsum) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_l6[];
    constexpr int NRS = l6_nrs(MT), NPL = kL6S * NRS;
    constexpr int kL6WRS = l6_wrs(CT), kL6WPL = kL6S * kL6WRS, NWT = CT / 8;      // NWT: wide tiles per wave (W, W + 8, ..)
    constexpr int WW4 = CT * 4;                            // float4 per wide-slice row (96 / 32)
    constexpr int NT = NW * 64, HALVES = NW / 8, MTL = (MT + HALVES - 1) / HALVES;      // HALVES: groups of 8 waves sharing the wide tiles, each with every HALVES-th narrow tile
    constexpr int SW = (kL6S * WW4 + NT - 1) / NT;         // wide load slots per thread (6 / 2)
    constexpr int ZW4 = 4 * MT;                            // float4 per narrow row (28 / 36)
    constexpr int SZ = (kL6S * ZW4 + NT - 1) / NT;         // narrow load slots per thread (2 / 3)
    constexpr int kW_ = 0, kZ_ = 3 * kL6WPL, kB_ = CT == 24 ? kZ_ + 3 * NPL : 0;      // CT = 8 / 16: the column-sum scratch lies over the (dead) wide image
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4, Wall = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = Wall & 7, half = Wall >> 3;               // wide tiles W, W + 8, ..; narrow tiles half, half + HALVES, ..
    const int chunk = ((R + gridDim.x - 1) / gridDim.x + kL6S - 1) / kL6S * kL6S;
    const int r_begin = blockIdx.x * chunk, r_end = min(R, r_begin + chunk);
    const int col0 = 16 * tile0, colE = min(KW, col0 + 16 * ntp);        // this pass: wide columns [col0, colE), ntp <= 24 tiles
    const uint32_t lds0 = l6_lds_addr(smem_l6);

    // slot geometry recomputed at each use from an opaque thread index (hoisted it costs the registers the accumulators need)
    auto wgeo = [&](int s, int t_, int &row, int &col, bool &ok) __attribute__((always_inline)) {
        const int idx = s * NT + t_;
        row = idx / WW4; col = col0 + 4 * (idx % WW4); ok = col < colE && idx < kL6S * WW4;
    };
    auto zgeo = [&](int s, int t_, int &row, int &col, bool &in, bool &ok) __attribute__((always_inline)) {
        const int idx = s * NT + t_;
        in = idx < kL6S * ZW4; row = in ? idx / ZW4 : 0; col = in ? 4 * (idx % ZW4) : 0; ok = col < NN;
    };
    f32x4 rw[SW], rz[SZ], zsum[SZ];
#pragma unroll
    for (int s = 0; s < SZ; ++s) zsum[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto gload = [&](int r0) __attribute__((always_inline)) {
        int t_ = tid;
        asm volatile("" : "+v"(t_));
#pragma unroll
        for (int s = 0; s < SW; ++s) {
            int row, col; bool ok;
            wgeo(s, t_, row, col, ok);
            const int r = r0 + row;
            const int rc = r < r_end ? r : r_end - 1;
            rw[s] = ok ? *reinterpret_cast<const f32x4 *>(Aw + (size_t)rc * ldw + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s = 0; s < SZ; ++s) {
            int row, col; bool in, ok;
            zgeo(s, t_, row, col, in, ok);
            const int r = r0 + row;
            const int rc = r < r_end ? r : r_end - 1;
            rz[s] = *reinterpret_cast<const f32x4 *>(Zn + (size_t)rc * ldn + (ok ? col : 0));
        }
    };
    auto stage = [&](int r0) __attribute__((always_inline)) {
        int t_ = tid;
        asm volatile("" : "+v"(t_));
#pragma unroll
        for (int s = 0; s < SW; ++s) {
            int row, col; bool ok;
            wgeo(s, t_, row, col, ok);
            if (ok) {                                      // columns past the pass are neither loaded nor multiplied
                const float okf = r0 + row < r_end ? 1.0f : 0.0f;
                f32x4 v = rw[s];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] *= okf;
                l6_write4(lds0 + (uint32_t)(kW_ + row * kL6WRS + 2 * (col - col0)), kL6WPL, v);
            }
        }
#pragma unroll
        for (int s = 0; s < SZ; ++s) {
            int row, col; bool in, ok;
            zgeo(s, t_, row, col, in, ok);
            if (in) {
                const bool rok = r0 + row < r_end;
                const float okf = (rok && ok) ? 1.0f : 0.0f;
                f32x4 v = rz[s];
#pragma unroll
                for (int c = 0; c < 4; ++c) { v[c] *= okf; zsum[s][c] += v[c]; }
                if (ones && col == NN && rok) v[0] = 1.0f;            // NN % 4 == 0: the ones column opens a float4 of its own
                l6_write4(lds0 + (uint32_t)(kZ_ + row * NRS + 2 * col), NPL, v);
            }
        }
    };
    auto read_tr = [&](LFrag (&f)[3], uint32_t img_lane, int plane_bytes, int row_bytes, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_i16x4_l *>((uintptr_t)(img_lane + (uint32_t)(p * plane_bytes + 32 * t))));
            const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_i16x4_l *>((uintptr_t)(img_lane + (uint32_t)(p * plane_bytes + 32 * t + 16 * row_bytes))));
            const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
            f[p].u[0] = l2[0]; f[p].u[1] = l2[1]; f[p].u[2] = h2[0]; f[p].u[3] = h2[1];
        }
    };
    const uint32_t tr_w = lds0 + (uint32_t)(kW_ + (4 * g + (j >> 2)) * kL6WRS + 8 * (j & 3));
    const uint32_t tr_z = lds0 + (uint32_t)(kZ_ + (4 * g + (j >> 2)) * NRS + 8 * (j & 3));

    f32x4 acc[NWT][MTL];
#pragma unroll
    for (int n = 0; n < NWT; ++n)
#pragma unroll
        for (int ml = 0; ml < MTL; ++ml) acc[n][ml] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (r_begin < r_end) gload(r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += kL6S) {
        stage(r0);
        __syncthreads();
        if (r0 + kL6S < r_end) gload(r0 + kL6S);
        LFrag xb[NWT][3];
#pragma unroll
        for (int n = 0; n < NWT; ++n)
            if (W + 8 * n < ntp) read_tr(xb[n], tr_w, kL6WPL, kL6WRS, W + 8 * n);        // wave-uniform: a wave multiplies only the tiles the pass holds
#pragma unroll
        for (int ml = 0; ml < MTL; ++ml) {
            const int mt = ml * HALVES + half;
            if (W >= ntp || mt >= MT) break;
            LFrag za[3];
            read_tr(za, tr_z, NPL, NRS, mt);
#pragma unroll
            for (int n = 0; n < NWT; ++n) {
                if (W + 8 * n >= ntp) continue;
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
    float *part = ws + (size_t)blockIdx.x * ws_stride;
#pragma unroll
    for (int n = 0; n < NWT; ++n) {
        const int k = col0 + 16 * (W + 8 * n) + j;
#pragma unroll
        for (int ml = 0; ml < MTL; ++ml)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int mt = ml * HALVES + half;
                const int o = 16 * mt + 4 * g + c;
                if (k < colE && W + 8 * n < ntp && mt < MT) {
                    if (o < NN) part[(size_t)o * s_n + (size_t)k * s_w] = acc[n][ml][c];
                    else if (ones && o == NN) part[bias_off + k] = acc[n][ml][c];
                }
            }
    }
    if (colsum) {                                          // narrow = dY: db = its column sums (per-thread sums over the slabs, then over the 32 slab rows)
        float *bs = reinterpret_cast<float *>(smem_l6 + kB_);
#pragma unroll
        for (int s = 0; s < SZ; ++s) {
            int row, col; bool in, ok;
            zgeo(s, tid, row, col, in, ok);
            if (in) *reinterpret_cast<f32x4 *>(bs + row * (16 * MT) + col) = zsum[s];
        }
        __syncthreads();
        if (tid < NN) {
            float v = 0.0f;
#pragma unroll 8
            for (int r = 0; r < kL6S; ++r) v += bs[r * (16 * MT) + tid];
            part[bias_off + tid] = v;
        }
    }
}

// PTR_LIN_BW_X6: "0" never, "1" (default) from 32768 rows on, "2" always (tests).  Applicable when one side has at most 144 columns (140 for
// X, which needs a free column for the ones), every leading dimension and column count is a multiple of 4 and the pointers are 16-byte aligned.
int lin_bw_x6_plan(int R, int K, int N, int ldx, int ldy, const void *X, const void *dY) {       // 0 = not served, 1 = narrow dY, 2 = narrow X
    const char *e = getenv("PTR_LIN_BW_X6");
    const int mode = e ? atoi(e) : 1;
    if (mode <= 0 || (mode == 1 && R < 32768)) return 0;
    if ((K | N | ldx | ldy) & 3) return 0;
    if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(dY)) & 15) return 0;
    const bool a_ok = N <= 144, b_ok = K <= 140;
    if (!a_ok && !b_ok) return 0;
    if (a_ok && b_ok) {
        // both narrow: r6 — the orientation whose WIDE side fits 128 columns runs the one-tile-per-wave form (two workgroups per CU); otherwise hold
        // the SMALLER one's companion as the wide stream (one pass either way up to 384 columns)
        if (K <= 128 && N <= K) return 1;
        if (N <= 128 && K < N) return 2;
        if (K <= 128) return 1;
        if (N <= 128) return 2;
        return N <= K ? 1 : 2;
    }
    return a_ok ? 1 : 2;
}
static int l6_num_cus() {
    static const int ncu = [] {                          // queried once: hipGetDeviceProperties is not a per-call cost
        int dev = 0;
        hipDeviceProp_t pr;
        return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }();
    return ncu;
}
// the one-tile-per-wave form serves a product whose wide side (plan 1: X, plan 2: dY) has at most 128 columns
static bool l6_small(int plan, int K, int N) { return (plan == 1 ? K : N) <= 128; }
// row chunks (= partials): one per CU, two for the one-tile-per-wave form (two workgroups per CU); at least 256 rows each.  plan 0: the largest count any plan uses
int lin_bw_x6_chunks(int R, int K, int N, int plan) {
    const int per_cu = (plan == 0 || l6_small(plan, K, N)) ? 2 : 1;
    const int maxc = (R + 255) / 256, want = per_cu * l6_num_cus();
    return maxc < want ? (maxc < 1 ? 1 : maxc) : want;
}

int launch_lin_bw_x6(int plan, const float *X, int ldx, const float *dY, int ldy, int R, int K, int N, float *ws, int chunks, hipStream_t st, const char *who) {
    const size_t n = (size_t)N * K + N;
    const float *Zn = plan == 1 ? dY : X, *Aw = plan == 1 ? X : dY;
    const int ldn = plan == 1 ? ldy : ldx, ldw = plan == 1 ? ldx : ldy, NN = plan == 1 ? N : K, KW = plan == 1 ? K : N;
    const int ones = plan == 2 ? 1 : 0;
    const size_t s_n = plan == 1 ? (size_t)K : 1, s_w = plan == 1 ? 1 : (size_t)K;
    const int mt = (NN + ones) <= 112 ? 7 : 9;
    const int T = (KW + 15) / 16;
    if (l6_small(plan, K, N)) {                           // one pass, one wide tile per wave
        auto go8 = [&](auto kern, int MT) -> int {
            const size_t lds = (size_t)l6_lds(MT, 8);
            if (int e = allow_lds(kern, lds)) return e;
            hipLaunchKernelGGL(kern, dim3(chunks), dim3(512), lds, st, Zn, ldn, NN, ones, Aw, ldw, KW, 0, T, R, ws, n, s_n, s_w, (size_t)N * K, plan == 1 ? 1 : 0);
            return check_hip(hipGetLastError(), who);
        };
        return mt == 7 ? go8(linear_bw_x6_kernel<7, 8, 8>, 7) : go8(linear_bw_x6_kernel<9, 8, 8>, 9);
    }
    // 9+ wide tiles: passes of <= 16 tiles on the 16-wave form (PTR_LIN_BW_FORM=24: the 8-wave / 24-tile form, A/B measurements)
    const char *fe = getenv("PTR_LIN_BW_FORM");
    const int ct = (fe && atoi(fe) == 24) ? 24 : 16;
    auto go = [&](auto kern, int MT, int CT, int threads) -> int {
        const size_t lds = (size_t)l6_lds(MT, CT);
        if (int e = allow_lds(kern, lds)) return e;
        // balanced passes: ceil(T / CT) of them, each ceil(T / passes) tiles wide (26 tiles: 13 + 13, not 24 + 2)
        const int passes = (T + CT - 1) / CT, tpp = (T + passes - 1) / passes;
        for (int t0 = 0; t0 < T; t0 += tpp) {
            hipLaunchKernelGGL(kern, dim3(chunks), dim3(threads), lds, st, Zn, ldn, NN, ones, Aw, ldw, KW, t0, T - t0 < tpp ? T - t0 : tpp, R, ws, n, s_n, s_w,
                               (size_t)N * K, (plan == 1 && t0 == 0) ? 1 : 0);
            if (int e = check_hip(hipGetLastError(), who)) return e;
        }
        return 0;
    };
    if (ct == 16) return mt == 7 ? go(linear_bw_x6_kernel<7, 16, 16>, 7, 16, 1024) : go(linear_bw_x6_kernel<9, 16, 16>, 9, 16, 1024);
    return mt == 7 ? go(linear_bw_x6_kernel<7, kL6CT, 8>, 7, kL6CT, 512) : go(linear_bw_x6_kernel<9, kL6CT, 8>, 9, kL6CT, 512);
}

}  // namespace ptr
