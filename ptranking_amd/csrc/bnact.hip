// Batch normalisation + activation + dropout of the stacked feed-forward scoring functions, as hand-written streaming kernels around
// the linear-layer kernels of linear.hip.  One hidden layer of the reference's get_stacked_FFNet (ptranking/base/utils.py:296-315)
//     Dropout -> Linear -> [LTRBatchNorm] -> AF            (default pointsf: 5 x [.. -> BN(affine) -> GELU], then Linear -> BN -> Sigmoid,
//                                                           ptranking/ltr_adhoc/eval/parameter.py:145-146)
// becomes   z = linear(a_in)  ->  (mean, rstd) = column statistics of z over ALL B*L documents (LTRBatchNorm = BatchNorm1d without
// running statistics, utils.py:201-223: batch statistics in training AND evaluation, biased variance, eps 1e-5)  ->
// a_out = dropout_next(AF(gamma * (z - mean) * rstd + beta)).  Only z is stored; backward recomputes the normalised value, the
// activation derivative and the dropout mask (counter-based generator, ptr_dropout.h):
//     dy   = da_out * keep/(1-p) * AF'(y)
//     dz   = gamma * rstd * (dy - mean_rows(dy) - xhat * mean_rows(dy * xhat)),   dgamma = sum dy * xhat,   dbeta = sum dy
// i.e. two passes over z with one grid-wide reduction in between (ptr_colsum2 + ptr_bnact_backward), exactly BatchNorm's backward.
// All reductions are two-stage with a fixed order (deterministic).  Activations: the reference's get_AF (utils.py:100-143) minus the
// random (RReLU) and broken (PRelU / SWISH / softmax) entries.
#include "ptr_device.h"
#include "ptr_dropout.h"

namespace ptr {

// erf without libm's branches (r6): the element-wise batch-norm / GELU kernels evaluate it (and an exp) per element — colsum2_kernel<1> ran at 2.8 TB/s of its
// 104 MB on the library erff + expf.  |a| <= 0.875: a + a (c0 - 1 + s P(s)), s = a^2 (odd, degree 11); above: 1 - 2^q(t) with t = min(|a|, 4) and q a degree-7
// polynomial on v_exp_f32, sign restored — both evaluated, one select.  Least-squares fits on Chebyshev nodes (erfc-weighted above); maximum absolute error against
// float64 over [-6, 6] with fp32 evaluation: 6.8e-8 (one ulp of the results near 1), GELU / GELU' 4.5e-7 / 1.4e-7 over [-8, 8] — the rounding of the fp32 formula
// itself (the reference's torch GELU is `0.5 x (1 + erf(x / sqrt 2))` in fp32 too: ptranking/base/utils.py:201-212 -> nn.GELU).
__device__ __forceinline__ float erf_fast(float a) {
    const float t = fminf(fabsf(a), 4.0f), s = a * a;
    float r = -6.254293257e-04f;
    r = fmaf(r, s, 5.042351317e-03f);
    r = fmaf(r, s, -2.679867297e-02f);
    r = fmaf(r, s, 1.128267050e-01f);
    r = fmaf(r, s, -3.761257529e-01f);
    r = fmaf(r, s, 1.283791661e-01f);
    const float small = fmaf(r, a, a);
    float q = -3.014734466e-05f;
    q = fmaf(q, t, 6.231664447e-04f);
    q = fmaf(q, t, -5.998506676e-03f);
    q = fmaf(q, t, 3.618580848e-02f);
    q = fmaf(q, t, -1.561265737e-01f);
    q = fmaf(q, t, -9.138113260e-01f);
    q = fmaf(q, t, -1.629501581e+00f);
    q = fmaf(q, t, 2.424932900e-04f);
    const float large = __builtin_copysignf(1.0f - __builtin_amdgcn_exp2f(q), a);
    return t > 0.875f ? large : small;
}
// exp(-y^2 / 2) / sqrt(2 pi) on v_exp_f32
__device__ __forceinline__ float gauss_pdf_fast(float y) { return 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.7213475204444817f * y * y); }

__device__ __forceinline__ float af_fwd(int af, float y) {
    switch (af) {
        case PTR_AF_RELU: return fmaxf(y, 0.0f);
        case PTR_AF_LEAKY: return y > 0.0f ? y : 0.01f * y;
        case PTR_AF_ELU: return y > 0.0f ? y : expm1f(y);                       // ELU / CELU with alpha = 1
        case PTR_AF_SELU: return 1.0507009873554805f * (y > 0.0f ? y : 1.6732632423543772f * expm1f(y));
        case PTR_AF_GELU: return 0.5f * y * (1.0f + erf_fast(y * 0.7071067811865476f));
        case PTR_AF_SIGMOID: return 1.0f / (1.0f + expf(-y));
        case PTR_AF_TANH: return tanhf(y);
        default: return y;
    }
}
__device__ __forceinline__ float af_bwd(int af, float y) {
    switch (af) {
        case PTR_AF_RELU: return y > 0.0f ? 1.0f : 0.0f;
        case PTR_AF_LEAKY: return y > 0.0f ? 1.0f : 0.01f;
        case PTR_AF_ELU: return y > 0.0f ? 1.0f : expf(y);
        case PTR_AF_SELU: return 1.0507009873554805f * (y > 0.0f ? 1.0f : 1.6732632423543772f * expf(y));
        case PTR_AF_GELU: return 0.5f * (1.0f + erf_fast(y * 0.7071067811865476f)) + y * gauss_pdf_fast(y);
        case PTR_AF_SIGMOID: { const float s = 1.0f / (1.0f + expf(-y)); return s * (1.0f - s); }
        case PTR_AF_TANH: { const float t = tanhf(y); return 1.0f - t * t; }
        default: return 1.0f;
    }
}

#ifndef PTR_BN_BWD_ROWS
#define PTR_BN_BWD_ROWS 2
#endif

struct BnActArgs {
    int group;                   // rows per statistics group: 0 = one group (LTRBatchNorm, the whole batch), L = per query (LTRBatchNorm2)
    int R, N, ld;                // z / a / da are [R][N] with leading dimension ld
    int af;
    int has_bn;
    float p_drop;
    uint32_t seed_lo, seed_hi;
    int site;
    const int32_t *lens;         // padded query batches: row r is a real document iff (r % L) < lens[r / L]; NULL: all rows
    int L;                       // rows per query (only read when lens != NULL)
};

// real (non-padding) rows of [r0, r1)
__device__ __forceinline__ int real_rows(const BnActArgs &a, int r0, int r1) {
    if (!a.lens) return max(r1 - r0, 0);
    int cnt = 0;
    for (int q = r0 / a.L; q * a.L < r1; ++q) {
        const int n = min(max(a.lens[q], 0), a.L);
        cnt += max(0, min(r1, q * a.L + n) - max(r0, q * a.L));
    }
    return cnt;
}
__device__ __forceinline__ bool row_is_real(const BnActArgs &a, int r) {
    if (!a.lens) return true;
    const int q = r / a.L;
    return r - q * a.L < a.lens[q];
}

// keep / (1 - p) factor of element (row, col) of dropout site `site` (1 when p == 0)
__device__ __forceinline__ float drop_factor(const BnActArgs &a, int row, int col, uint32_t thr, float inv_keep) {
    if (a.p_drop <= 0.0f) return 1.0f;
    return drop_keep1(a.seed_lo, a.seed_hi, a.site, row, col, thr) ? inv_keep : 0.0f;
}

// keep / (1 - p) factors of columns [4*cg, 4*cg+3] of `row` (one hash for the four)
__device__ __forceinline__ f32x4 drop_factor4(const BnActArgs &a, int row, int cg, uint32_t thr, float inv_keep) {
    if (a.p_drop <= 0.0f) return f32x4{1.0f, 1.0f, 1.0f, 1.0f};
    uint32_t w0, w1;
    drop_bits(a.seed_lo, a.seed_hi, a.site, row, cg, w0, w1);
    return f32x4{(w0 & 0xFFFFu) >= thr ? inv_keep : 0.0f, (w0 >> 16) >= thr ? inv_keep : 0.0f, (w1 & 0xFFFFu) >= thr ? inv_keep : 0.0f,
                 (w1 >> 16) >= thr ? inv_keep : 0.0f};
}

// Column statistics of a chunk of rows -> partial[chunk][2][N], ONE pass over the data, W = 4 (float4 per thread) or 1 columns per thread.
//   MODE 0: (mean_b, M2_b = sum (z - mean_b)^2) of the chunk's rows: sums of d = z - k and d^2 around the pivot k = the chunk's first
//           row (the column's own scale: E[d^2] - E[d]^2 then cancels at most a few bits, unlike E[z^2] - mean^2), combined over the
//           chunks by the parallel-variance formula in the finishing kernel
//   MODE 1: (sum dy, sum dy * xhat), dy = da * dropout * AF'(y), y = gamma * xhat + beta        (batch-norm backward)
// Threads: column group c_in (W columns) x row lane rl; a row lane walks the chunk's rows rl, rl + rsub, ... four rows in flight.
template <int MODE, int W>
__global__ void __launch_bounds__(256)
colsum2_kernel(const float *__restrict__ z, const float *__restrict__ da, const float *__restrict__ mean, const float *__restrict__ rstd,
               const float *__restrict__ gamma, const float *__restrict__ beta, BnActArgs a, float *__restrict__ partial,
               float *__restrict__ counts /* [gridDim.x] real rows per chunk, or NULL */) {
    using vec = float __attribute__((ext_vector_type(W)));
    __shared__ float red[2 * W][256];
    const int N = a.N, R = a.R, NG = N / W;                  // W == 4 only when N % 4 == 0
    const int tid = threadIdx.x;
    const int groups_per_pass = NG < 256 ? NG : 256;
    const int rsub = 256 / groups_per_pass;                  // row lanes per pass
    const int c_in = tid % groups_per_pass, rl = tid / groups_per_pass;
    const int chunk = a.group > 0 ? a.group : (R + gridDim.x - 1) / gridDim.x;     // grouped: one workgroup per group (query)
    const int r_begin = blockIdx.x * chunk, r_end = min(R, r_begin + chunk);
    const size_t so = a.group > 0 ? (size_t)blockIdx.x * N : 0;                  // offset of this group's statistics
    const uint32_t thr = drop_thr(a.p_drop);
    const float inv_keep = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    auto ldv = [&](const float *ptr_) -> vec {
        if constexpr (W == 4) return *reinterpret_cast<const vec *>(ptr_);
        else return vec{*ptr_};
    };
    for (int g0 = 0; g0 < NG; g0 += groups_per_pass) {
        const int cg = g0 + c_in, c = cg * W;
        const bool on = cg < NG && rl < rsub && r_begin < r_end;
        vec s1, s2, kv, mu, rs, ga, be;
        for (int e = 0; e < W; ++e) { s1[e] = 0.f; s2[e] = 0.f; kv[e] = 0.f; mu[e] = 0.f; rs[e] = 1.f; ga[e] = 1.f; be[e] = 0.f; }
        if (on) {
            if constexpr (MODE == 0) kv = ldv(z + (size_t)r_begin * a.ld + c);
            if (MODE == 1 && a.has_bn) {
                mu = ldv(mean + so + c); rs = ldv(rstd + so + c);
                if (gamma) ga = ldv(gamma + c);
                if (beta) be = ldv(beta + c);
            }
            auto acc = [&](int r, vec zv, vec dv) {
                const float real = row_is_real(a, r) ? 1.0f : 0.0f;      // padded rows: loaded (finite) and multiplied away, no branch
                if constexpr (MODE == 0) {
                    const vec d = (zv - kv) * real;
                    s1 += d;
                    for (int e = 0; e < W; ++e) s2[e] = fmaf(d[e], d[e], s2[e]);
                } else {
                    dv = dv * real;
                    vec keep;
                    if constexpr (W == 4) keep = drop_factor4(a, r, cg, thr, inv_keep);
                    else keep[0] = drop_factor(a, r, c, thr, inv_keep);
                    for (int e = 0; e < W; ++e) {
                        const float xh = a.has_bn ? (zv[e] - mu[e]) * rs[e] : zv[e];
                        const float y = a.has_bn ? fmaf(ga[e], xh, be[e]) : zv[e];
                        const float dy = dv[e] * keep[e] * af_bwd(a.af, y);
                        s1[e] += dy;
                        s2[e] = fmaf(dy, xh, s2[e]);
                    }
                }
            };
            int r = r_begin + rl;
            // independent rows in flight: four for the statistics; TWO for the backward sums (r6) — with four, AF'(y) and the dropout hash of four float4 pairs need
            // 130 registers = three waves per SIMD, and a wave issues one instruction per ~5 cycles whatever its instruction-level parallelism (scratch/valu_rate):
            // 66 registers = seven waves per SIMD hide the loads AND issue faster
            constexpr int UR = MODE == 1 ? PTR_BN_BWD_ROWS : 4;
            for (; r + (UR - 1) * rsub < r_end; r += UR * rsub) {
                vec zv[UR], dv[UR];
#pragma unroll
                for (int u = 0; u < UR; ++u) {
                    zv[u] = ldv(z + (size_t)(r + u * rsub) * a.ld + c);
                    if constexpr (MODE == 1) dv[u] = ldv(da + (size_t)(r + u * rsub) * a.ld + c); else dv[u] = zv[u];
                }
#pragma unroll
                for (int u = 0; u < UR; ++u) acc(r + u * rsub, zv[u], dv[u]);
            }
            for (; r < r_end; r += rsub) {
                const vec zv = ldv(z + (size_t)r * a.ld + c);
                vec dv = zv;
                if constexpr (MODE == 1) dv = ldv(da + (size_t)r * a.ld + c);
                acc(r, zv, dv);
            }
        }
        for (int e = 0; e < W; ++e) { red[e][tid] = s1[e]; red[W + e][tid] = s2[e]; }
        __syncthreads();
        if (rl == 0 && cg < NG) {                                 // fixed-order sum over the row lanes
            vec t1, t2;
            for (int e = 0; e < W; ++e) { t1[e] = 0.f; t2[e] = 0.f; }
            for (int k = 0; k < rsub; ++k)
                for (int e = 0; e < W; ++e) { t1[e] += red[e][k * groups_per_pass + c_in]; t2[e] += red[W + e][k * groups_per_pass + c_in]; }
            if constexpr (MODE == 0) {
                const float nrows = (float)max(real_rows(a, r_begin, r_end), 1);
                for (int e = 0; e < W; ++e) {
                    const float m1 = t1[e] / nrows;
                    t2[e] = fmaxf(t2[e] - t1[e] * m1, 0.0f);      // M2 around the chunk mean
                    t1[e] = kv[e] + m1;                           // chunk mean
                }
            }
            for (int e = 0; e < W; ++e) {
                partial[((size_t)blockIdx.x * 2 + 0) * N + c + e] = t1[e];
                partial[((size_t)blockIdx.x * 2 + 1) * N + c + e] = t2[e];
            }
        }
        __syncthreads();
    }
    if (counts && tid == 0) counts[blockIdx.x] = (float)real_rows(a, r_begin, r_end);
}

// Fixed-order combination of the chunk partials by 8 columns x 32 partial lanes per workgroup (lane bl sums the partials bl, bl + 32, ...
// four at a time, the 32 lane sums are added in lane order).
//   FINISH 0: out1 = sum partial[.][0], out2 = sum partial[.][1]
//   FINISH 1: partials are (chunk mean, chunk M2) over `chunk` rows each (the last one shorter): out1 = mean, out2 = 1 / sqrt(var + eps),
//             var = (sum M2_b + n_b (mean_b - mean)^2) / R   (biased, as BatchNorm normalises)
// counts != NULL (padded batches): chunk b holds counts[b] real rows; their total replaces R and is also written to total_out[0]
template <int FINISH>
__global__ void __launch_bounds__(256)
colsum2_reduce_kernel(const float *__restrict__ partial, int nblk, int N, int R, int chunk, float eps, float *__restrict__ out1,
                      float *__restrict__ out2, const float *__restrict__ counts, float *__restrict__ total_out) {
    constexpr int CL = 8, BL = 32;
    __shared__ float red[2][BL][CL + 1];
    __shared__ float cnt_red[256];
    const int cl = threadIdx.x & (CL - 1), bl = threadIdx.x / CL;
    const int c = blockIdx.x * CL + cl;
    const bool on = c < N;
    float Rf = (float)R;
    if (counts) {                                            // fixed-order total of the chunk counts (integers: exact in fp32 up to 2^24)
        float t = 0.0f;
        for (int b = threadIdx.x; b < nblk; b += 256) t += counts[b];
        cnt_red[threadIdx.x] = t;
        __syncthreads();
        for (int s_ = 128; s_ > 0; s_ >>= 1) {
            if ((int)threadIdx.x < s_) cnt_red[threadIdx.x] += cnt_red[threadIdx.x + s_];
            __syncthreads();
        }
        Rf = fmaxf(cnt_red[0], 1.0f);
        if (total_out && blockIdx.x == 0 && threadIdx.x == 0) total_out[0] = Rf;
    }
    auto rows_of = [&](int b) { return counts ? counts[b] : (float)(min(R, (b + 1) * chunk) - b * chunk); };
    auto p0 = [&](int b) { return b < nblk ? partial[((size_t)b * 2 + 0) * N + c] : 0.0f; };
    auto p1 = [&](int b) { return b < nblk ? partial[((size_t)b * 2 + 1) * N + c] : 0.0f; };
    float s1 = 0.0f, s2 = 0.0f;
    if (on)
        for (int b = bl; b < nblk; b += 4 * BL) {
            const float a0 = p0(b), a1 = p0(b + BL), a2 = p0(b + 2 * BL), a3 = p0(b + 3 * BL);
            if (FINISH == 0) {
                const float b0 = p1(b), b1 = p1(b + BL), b2 = p1(b + 2 * BL), b3 = p1(b + 3 * BL);
                s1 += (a0 + a1) + (a2 + a3);
                s2 += (b0 + b1) + (b2 + b3);
            } else {
                s1 += (rows_of(b) * a0 + (b + BL < nblk ? rows_of(b + BL) * a1 : 0.0f)) +
                      ((b + 2 * BL < nblk ? rows_of(b + 2 * BL) * a2 : 0.0f) + (b + 3 * BL < nblk ? rows_of(b + 3 * BL) * a3 : 0.0f));
            }
        }
    red[0][bl][cl] = s1; red[1][bl][cl] = s2;
    __syncthreads();
    float t1 = 0.0f, t2 = 0.0f;
    for (int k = 0; k < BL; ++k) { t1 += red[0][k][cl]; t2 += red[1][k][cl]; }      // every lane: the same fixed order
    if (FINISH == 0) {
        if (on && bl == 0) { out1[c] = t1; out2[c] = t2; }
        return;
    }
    const float mean_tot = t1 / Rf;
    float m2 = 0.0f;
    if (on)
        for (int b = bl; b < nblk; b += 4 * BL) {
            float part[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int bb = b + u * BL;
                const float mb = p0(bb), qb = p1(bb);
                const float d = mb - mean_tot;
                part[u] = bb < nblk ? qb + rows_of(bb) * d * d : 0.0f;
            }
            m2 += (part[0] + part[1]) + (part[2] + part[3]);
        }
    __syncthreads();
    red[0][bl][cl] = m2;
    __syncthreads();
    if (on && bl == 0) {
        float v = 0.0f;
        for (int k = 0; k < BL; ++k) v += red[0][k][cl];
        out1[c] = mean_tot;
        out2[c] = 1.0f / sqrtf(v / Rf + eps);
    }
}

// grouped statistics: the per-group partial IS the group's (mean, M2): mean = partial[g][0], rstd = 1 / sqrt(M2 / L + eps)
__global__ void __launch_bounds__(256)
group_finish_kernel(const float *__restrict__ partial, size_t G, int N, int L, float eps, float *__restrict__ mean, float *__restrict__ rstd,
                    const int32_t *__restrict__ lens) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= G * N) return;
    const size_t gidx = i / N;
    const int c = (int)(i - gidx * N);
    const int n = lens ? max(min(max(lens[gidx], 0), L), 1) : L;          // a query's own documents (an empty query: mean 0, var 0)
    mean[i] = partial[(gidx * 2 + 0) * N + c];
    rstd[i] = 1.0f / sqrtf(partial[(gidx * 2 + 1) * N + c] / (float)n + eps);
}

// a_out = dropout(AF(BN(z))), W columns per thread
template <int W>
__global__ void __launch_bounds__(256)
bnact_fwd_kernel(const float *__restrict__ z, const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ gamma,
                 const float *__restrict__ beta, BnActArgs a, float *__restrict__ out) {
    using vec = float __attribute__((ext_vector_type(W)));
    const int NG = a.N / W;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)a.R * NG) return;
    const int r = (int)(i / NG), cg = (int)(i - (size_t)r * NG), c = cg * W;
    auto ldv = [&](const float *ptr_) -> vec {
        if constexpr (W == 4) return *reinterpret_cast<const vec *>(ptr_);
        else return vec{*ptr_};
    };
    vec y = ldv(z + (size_t)r * a.ld + c);
    const size_t so = a.group > 0 ? (size_t)(r / a.group) * a.N : 0;
    if (a.has_bn) {
        const vec mu = ldv(mean + so + c), rs = ldv(rstd + so + c);
        vec ga, be;
        for (int e = 0; e < W; ++e) { ga[e] = 1.0f; be[e] = 0.0f; }
        if (gamma) ga = ldv(gamma + c);
        if (beta) be = ldv(beta + c);
        for (int e = 0; e < W; ++e) y[e] = fmaf(ga[e], (y[e] - mu[e]) * rs[e], be[e]);
    }
    const uint32_t thr = drop_thr(a.p_drop);
    const float inv_keep = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    vec keep;
    if constexpr (W == 4) keep = drop_factor4(a, r, cg, thr, inv_keep);
    else keep[0] = drop_factor(a, r, c, thr, inv_keep);
    vec o;
    for (int e = 0; e < W; ++e) o[e] = af_fwd(a.af, y[e]) * keep[e];
    if constexpr (W == 4) *reinterpret_cast<vec *>(out + (size_t)r * a.ld + c) = o;
    else out[(size_t)r * a.ld + c] = o[0];
}

// dz = gamma * rstd * (dy - sum_dy / R - xhat * sum_dyx / R)        (no BN: dz = dy), W columns per thread
template <int W>
__global__ void __launch_bounds__(256)
bnact_bwd_kernel(const float *__restrict__ z, const float *__restrict__ da, const float *__restrict__ mean, const float *__restrict__ rstd,
                 const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ sum_dy,
                 const float *__restrict__ sum_dyx, BnActArgs a, float *__restrict__ dz, const float *__restrict__ total_real) {
    using vec = float __attribute__((ext_vector_type(W)));
    const int NG = a.N / W;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)a.R * NG) return;
    const int r = (int)(i / NG), cg = (int)(i - (size_t)r * NG), c = cg * W;
    auto ldv = [&](const float *ptr_) -> vec {
        if constexpr (W == 4) return *reinterpret_cast<const vec *>(ptr_);
        else return vec{*ptr_};
    };
    const vec zv = ldv(z + (size_t)r * a.ld + c), dv = ldv(da + (size_t)r * a.ld + c);
    const uint32_t thr = drop_thr(a.p_drop);
    const float inv_keep = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    vec keep;
    if constexpr (W == 4) keep = drop_factor4(a, r, cg, thr, inv_keep);
    else keep[0] = drop_factor(a, r, c, thr, inv_keep);
    vec o;
    if (a.has_bn) {
        // grouped: sum_dy / sum_dyx are the per-group partials [G][2][N] of colsum2_kernel<1> (sum_dyx = sum_dy + N), mean over L rows
        const size_t so = a.group > 0 ? (size_t)(r / a.group) * a.N : 0, ss = a.group > 0 ? 2 * so : 0;
        const vec mu = ldv(mean + so + c), rs = ldv(rstd + so + c), sdy = ldv(sum_dy + ss + c), sdyx = ldv(sum_dyx + ss + c);
        vec ga, be;
        for (int e = 0; e < W; ++e) { ga[e] = 1.0f; be[e] = 0.0f; }
        if (gamma) ga = ldv(gamma + c);
        if (beta) be = ldv(beta + c);
        float cnt = (float)(a.group > 0 ? a.group : a.R);
        if (a.lens) cnt = a.group > 0 ? (float)max(min(max(a.lens[r / a.group], 0), a.L), 1) : total_real[0];
        const float invR = 1.0f / cnt;
        const float real = row_is_real(a, r) ? 1.0f : 0.0f;      // a padded row takes no part in the statistics: its dz is 0
        for (int e = 0; e < W; ++e) {
            const float xh = (zv[e] - mu[e]) * rs[e];
            const float y = fmaf(ga[e], xh, be[e]);
            const float dy = dv[e] * keep[e] * af_bwd(a.af, y);
            o[e] = real * ((ga[e] * rs[e]) * (dy - sdy[e] * invR - xh * (sdyx[e] * invR)));
        }
    } else {
        for (int e = 0; e < W; ++e) o[e] = dv[e] * keep[e] * af_bwd(a.af, zv[e]);
    }
    if constexpr (W == 4) *reinterpret_cast<vec *>(dz + (size_t)r * a.ld + c) = o;
    else dz[(size_t)r * a.ld + c] = o[0];
}

// float4 path: column count and leading dimension multiples of 4, every (non-null) pointer 16-byte aligned
template <class... P> static bool vec4_ok(int N, int ld, P... ptrs) {
    bool ok = (N % 4 == 0) && (ld % 4 == 0);
    ((ok = ok && ((reinterpret_cast<uintptr_t>(ptrs) & 15) == 0)), ...);
    return ok;
}

static int bn_blocks(int R) {
    int b = (R + 255) / 256;
    return b < 1 ? 1 : (b > 512 ? 512 : b);
}
// chunks of the BACKWARD sums (colsum2_kernel<1>; PTR_BN_BWD_BLOCKS overrides the cap for measurements).  r6 measured 512 / 2048 chunks with two / four rows in flight
// at 131 072 x 100: 1.267 / 1.277 / 1.272 / 1.324 ms per default-pointsf step — more chunks cost the fixed-order reduction more than they give the sums
static int bn_blocks_bwd_cap() {
    static int cap = 0;
    if (!cap) { const char *e = getenv("PTR_BN_BWD_BLOCKS"); cap = e ? atoi(e) : 0; if (cap < 1 || cap > 4096) cap = 512; }
    return cap;
}
static int bn_blocks_bwd(int R) {
    int b = (R + 63) / 64;
    const int cap = bn_blocks_bwd_cap();
    return b < 1 ? 1 : (b > cap ? cap : b);
}

static int check_bnact(const char *who, int R, int N, int ld, int af, float p) {
    if (R < 0 || N <= 0 || ld < N) { set_error("%s: bad shape R=%d N=%d ld=%d", who, R, N, ld); return PTR_ERR_INVALID_ARG; }
    if (af < PTR_AF_NONE || af > PTR_AF_TANH) { set_error("%s: unknown activation %d", who, af); return PTR_ERR_INVALID_ARG; }
    if (!(p >= 0.0f && p < 1.0f)) { set_error("%s: dropout p=%g out of [0,1)", who, (double)p); return PTR_ERR_INVALID_ARG; }
    return 0;
}

}  // namespace ptr

// group_rows: 0 = statistics over all R rows (LTRBatchNorm); L > 0 = per group of L consecutive rows (per query, LTRBatchNorm2; R % L == 0)
// layout: [blocks][2][N] partials | [2][N] totals (backward) | [blocks] real rows per chunk | [4] total real rows (padded batches)
extern "C" size_t ptr_bn_ws_floats(int R, int N, int group_rows) {
    const size_t blocks = group_rows > 0 ? (size_t)(R / group_rows) : (size_t)std::max(ptr::bn_blocks(R), ptr::bn_blocks_bwd(R));
    return blocks * 2 * (size_t)N + 2 * (size_t)N + blocks + 4;
}

namespace ptr {
static int check_lens(const char *who, const int32_t *lens, int rows_per_query, int R, int group_rows) {
    if (!lens) return 0;
    if (rows_per_query <= 0 || R % rows_per_query) { set_error("%s: lens given but R=%d is not a multiple of rows_per_query=%d", who, R, rows_per_query); return PTR_ERR_INVALID_ARG; }
    if (group_rows > 0 && group_rows != rows_per_query) { set_error("%s: per-query statistics need group_rows == rows_per_query (%d vs %d)", who, group_rows, rows_per_query); return PTR_ERR_INVALID_ARG; }
    return 0;
}
}  // namespace ptr

// mean / rstd ([N], or [R / group_rows][N]) of the columns of z (biased variance, rstd = 1 / sqrt(var + eps), two-pass)
// lens / rows_per_query (nullable / ignored): padded query batches — only the rows (r % rows_per_query) < lens[r / rows_per_query] enter
// the statistics (the reference never pads: data_utils.py:683-742 batches equal-length lists; a padded batch must score and train
// like the per-length batches it replaces)
extern "C" int ptr_bn_stats(const float *z, int ld, int R, int N, int group_rows, const int32_t *lens, int rows_per_query, float eps, float *ws,
                            float *mean, float *rstd, void *stream) {
    using namespace ptr;
    const char *who = "ptr_bn_stats";
    if (int rc = check_bnact(who, R, N, ld, 0, 0.0f)) return rc;
    if (R == 0 || !z || !ws || !mean || !rstd) { set_error("%s: NULL pointer / empty batch", who); return PTR_ERR_INVALID_ARG; }
    if (group_rows < 0 || (group_rows > 0 && R % group_rows)) { set_error("%s: R=%d is not a multiple of group_rows=%d", who, R, group_rows); return PTR_ERR_INVALID_ARG; }
    if (int rc = check_lens(who, lens, rows_per_query, R, group_rows)) return rc;
    hipStream_t st = as_stream(stream);
    BnActArgs a{group_rows, R, N, ld, 0, 0, 0.0f, 0, 0, 0, lens, lens ? rows_per_query : 0};
    const bool v4 = vec4_ok(N, ld, z, ws, mean, rstd);
    const int nb = group_rows > 0 ? R / group_rows : bn_blocks(R);
    float *counts = (lens && group_rows == 0) ? ws + (size_t)nb * 2 * N + 2 * (size_t)N : nullptr;
    if (v4) hipLaunchKernelGGL((colsum2_kernel<0, 4>), dim3(nb), dim3(256), 0, st, z, nullptr, nullptr, nullptr, nullptr, nullptr, a, ws, counts);
    else hipLaunchKernelGGL((colsum2_kernel<0, 1>), dim3(nb), dim3(256), 0, st, z, nullptr, nullptr, nullptr, nullptr, nullptr, a, ws, counts);
    if (group_rows > 0) {
        const unsigned fin = (unsigned)(((size_t)nb * N + 255) / 256);
        hipLaunchKernelGGL(group_finish_kernel, dim3(fin), dim3(256), 0, st, ws, (size_t)nb, N, group_rows, eps, mean, rstd, lens);
    } else {
        hipLaunchKernelGGL(colsum2_reduce_kernel<1>, dim3((N + 7) / 8), dim3(256), 0, st, ws, nb, N, R, (R + nb - 1) / nb, eps, mean, rstd, counts,
                           (float *)nullptr);
    }
    return check_hip(hipGetLastError(), who);
}

// out = dropout(AF(gamma * (z - mean) * rstd + beta))   (mean == NULL: no batch norm; gamma / beta NULL: no affine)
extern "C" int ptr_bnact_forward(const float *z, int ld, int R, int N, int group_rows, const int32_t *lens, int rows_per_query, const float *mean,
                                 const float *rstd, const float *gamma,
                                 const float *beta, int af, float p_drop, uint64_t seed, int site, float *out, void *stream) {
    using namespace ptr;
    const char *who = "ptr_bnact_forward";
    if (int rc = check_bnact(who, R, N, ld, af, p_drop)) return rc;
    if (R == 0) return 0;
    if (!z || !out || (mean && !rstd)) { set_error("%s: NULL pointer", who); return PTR_ERR_INVALID_ARG; }
    if (group_rows < 0 || (group_rows > 0 && R % group_rows)) { set_error("%s: R=%d is not a multiple of group_rows=%d", who, R, group_rows); return PTR_ERR_INVALID_ARG; }
    (void)lens; (void)rows_per_query;      // padded rows are normalised like the others (finite, never read back into a statistic or a gradient)
    BnActArgs a{mean ? group_rows : 0, R, N, ld, af, mean ? 1 : 0, p_drop, (uint32_t)seed, (uint32_t)(seed >> 32), site, nullptr, 0};
    if (vec4_ok(N, ld, z, out, mean, rstd, gamma, beta)) {
        const size_t n = (size_t)R * (N / 4);
        hipLaunchKernelGGL(bnact_fwd_kernel<4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), z, mean, rstd, gamma, beta, a, out);
    } else {
        const size_t n = (size_t)R * N;
        hipLaunchKernelGGL(bnact_fwd_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), z, mean, rstd, gamma, beta, a, out);
    }
    return check_hip(hipGetLastError(), who);
}

// da -> dz (and, with batch norm, dgamma[N] = sum dy * xhat, dbeta[N] = sum dy over ALL rows; either may be NULL).
// ws: ptr_bn_ws_floats(R, N, group_rows) + 2 * N floats.
extern "C" int ptr_bnact_backward(const float *z, const float *da, int ld, int R, int N, int group_rows, const int32_t *lens, int rows_per_query,
                                  const float *mean, const float *rstd, const float *gamma, const float *beta, int af, float p_drop, uint64_t seed, int site, float *ws,
                                  float *dz, float *dgamma, float *dbeta, void *stream) {
    using namespace ptr;
    const char *who = "ptr_bnact_backward";
    if (int rc = check_bnact(who, R, N, ld, af, p_drop)) return rc;
    if (R == 0) return 0;
    if (!z || !da || !dz || (mean && (!rstd || !ws))) { set_error("%s: NULL pointer", who); return PTR_ERR_INVALID_ARG; }
    if (group_rows < 0 || (group_rows > 0 && R % group_rows)) { set_error("%s: R=%d is not a multiple of group_rows=%d", who, R, group_rows); return PTR_ERR_INVALID_ARG; }
    if (!mean) lens = nullptr;             // without batch norm a padded row's dz only depends on its own da (0 from the loss kernels)
    if (int rc = check_lens(who, lens, rows_per_query, R, mean ? group_rows : 0)) return rc;
    hipStream_t st = as_stream(stream);
    BnActArgs a{mean ? group_rows : 0, R, N, ld, af, mean ? 1 : 0, p_drop, (uint32_t)seed, (uint32_t)(seed >> 32), site, lens, lens ? rows_per_query : 0};
    const float *sum_dy = nullptr, *sum_dyx = nullptr;
    const bool v4 = vec4_ok(N, ld, z, da, dz, ws, mean, rstd, gamma, beta, dgamma, dbeta);
    float *total_real = nullptr;
    if (mean) {
        const int nb = a.group > 0 ? R / a.group : bn_blocks_bwd(R);
        // totals over all rows: straight into dbeta / dgamma when the caller wants them
        float *tot_dy = dbeta ? dbeta : ws + (size_t)nb * 2 * N, *tot_dyx = dgamma ? dgamma : ws + (size_t)nb * 2 * N + N;
        float *counts = (lens && a.group == 0) ? ws + (size_t)nb * 2 * N + 2 * (size_t)N : nullptr;
        total_real = counts ? counts + nb : nullptr;
        if (v4) hipLaunchKernelGGL((colsum2_kernel<1, 4>), dim3(nb), dim3(256), 0, st, z, da, mean, rstd, gamma, beta, a, ws, counts);
        else hipLaunchKernelGGL((colsum2_kernel<1, 1>), dim3(nb), dim3(256), 0, st, z, da, mean, rstd, gamma, beta, a, ws, counts);
        if (a.group == 0 || dbeta || dgamma)
            hipLaunchKernelGGL(colsum2_reduce_kernel<0>, dim3((N + 7) / 8), dim3(256), 0, st, ws, nb, N, R, 0, 0.0f, tot_dy, tot_dyx, counts, total_real);
        sum_dy = a.group > 0 ? ws : tot_dy;                 // grouped: the per-group partials themselves
        sum_dyx = a.group > 0 ? ws + N : tot_dyx;
    }
    if (v4) {
        const size_t n = (size_t)R * (N / 4);
        hipLaunchKernelGGL(bnact_bwd_kernel<4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, z, da, mean, rstd, gamma, beta, sum_dy, sum_dyx, a, dz,
                           total_real);
    } else {
        const size_t n = (size_t)R * N;
        hipLaunchKernelGGL(bnact_bwd_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, z, da, mean, rstd, gamma, beta, sum_dy, sum_dyx, a, dz,
                           total_real);
    }
    return check_hip(hipGetLastError(), who);
}
