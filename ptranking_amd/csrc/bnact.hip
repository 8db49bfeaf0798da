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

__device__ __forceinline__ float af_fwd(int af, float y) {
    switch (af) {
        case PTR_AF_RELU: return fmaxf(y, 0.0f);
        case PTR_AF_LEAKY: return y > 0.0f ? y : 0.01f * y;
        case PTR_AF_ELU: return y > 0.0f ? y : expm1f(y);                       // ELU / CELU with alpha = 1
        case PTR_AF_SELU: return 1.0507009873554805f * (y > 0.0f ? y : 1.6732632423543772f * expm1f(y));
        case PTR_AF_GELU: return 0.5f * y * (1.0f + erff(y * 0.7071067811865476f));
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
        case PTR_AF_GELU: return 0.5f * (1.0f + erff(y * 0.7071067811865476f)) + y * expf(-0.5f * y * y) * 0.3989422804014327f;
        case PTR_AF_SIGMOID: { const float s = 1.0f / (1.0f + expf(-y)); return s * (1.0f - s); }
        case PTR_AF_TANH: { const float t = tanhf(y); return 1.0f - t * t; }
        default: return 1.0f;
    }
}

struct BnActArgs {
    int group;                   // rows per statistics group: 0 = one group (LTRBatchNorm, the whole batch), L = per query (LTRBatchNorm2)
    int R, N, ld;                // z / a / da are [R][N] with leading dimension ld
    int af;
    int has_bn;
    float p_drop;
    uint32_t seed_lo, seed_hi;
    int site;
};

// keep / (1 - p) factor of element (row, col) of dropout site `site` (1 when p == 0)
__device__ __forceinline__ float drop_factor(const BnActArgs &a, int row, int col, uint32_t thr, float inv_keep) {
    if (a.p_drop <= 0.0f) return 1.0f;
    return drop_keep1(a.seed_lo, a.seed_hi, a.site, row, col, thr) ? inv_keep : 0.0f;
}

// Column sums of two per-element quantities over a chunk of rows -> partial[chunk][2][N].
//   MODE 0: (z, -)                            forward statistics, pass 1 (mean)
//   MODE 2: ((z - mean)^2, -)                 forward statistics, pass 2 (two-pass variance: no E[z^2] - mean^2 cancellation)
//   MODE 1: (dy, dy * xhat)                   backward statistics, dy = da * dropout * AF'(y), y = gamma * xhat + beta
template <int MODE>
__global__ void __launch_bounds__(256)
colsum2_kernel(const float *__restrict__ z, const float *__restrict__ da, const float *__restrict__ mean, const float *__restrict__ rstd,
               const float *__restrict__ gamma, const float *__restrict__ beta, BnActArgs a, float *__restrict__ partial) {
    __shared__ float red[2][256];
    const int N = a.N, R = a.R;
    const int tid = threadIdx.x;
    const int cols_per_pass = N < 256 ? N : 256;
    const int rsub = 256 / cols_per_pass;                    // row lanes per pass
    const int c_in = tid % cols_per_pass, rl = tid / cols_per_pass;
    const int chunk = a.group > 0 ? a.group : (R + gridDim.x - 1) / gridDim.x;     // grouped: one workgroup per group (query)
    const int r_begin = blockIdx.x * chunk, r_end = min(R, r_begin + chunk);
    const size_t so = a.group > 0 ? (size_t)blockIdx.x * N : 0;                  // offset of this group's statistics
    const uint32_t thr = drop_thr(a.p_drop);
    const float inv_keep = a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
    for (int c0 = 0; c0 < N; c0 += cols_per_pass) {
        const int c = c0 + c_in;
        float s1 = 0.0f, s2 = 0.0f;
        if (c < N && rl < rsub) {
            float mu = 0.f, rs = 1.f, ga = 1.f, be = 0.f;
            if (MODE == 1 && a.has_bn) { mu = mean[so + c]; rs = rstd[so + c]; ga = gamma ? gamma[c] : 1.0f; be = beta ? beta[c] : 0.0f; }
            if (MODE == 2) mu = mean[so + c];
            for (int r = r_begin + rl; r < r_end; r += rsub) {
                const float zv = z[(size_t)r * a.ld + c];
                if (MODE == 0) {
                    s1 += zv;
                } else if (MODE == 2) {
                    const float d = zv - mu;
                    s1 = fmaf(d, d, s1);
                } else {
                    const float xh = a.has_bn ? (zv - mu) * rs : zv;
                    const float y = a.has_bn ? fmaf(ga, xh, be) : zv;
                    const float dy = da[(size_t)r * a.ld + c] * drop_factor(a, r, c, thr, inv_keep) * af_bwd(a.af, y);
                    s1 += dy;
                    s2 = fmaf(dy, xh, s2);
                }
            }
        }
        red[0][tid] = s1;
        red[1][tid] = s2;
        __syncthreads();
        if (rl == 0 && c < N) {                               // fixed-order sum over the row lanes
            float t1 = 0.0f, t2 = 0.0f;
            for (int k = 0; k < rsub; ++k) { t1 += red[0][k * cols_per_pass + c_in]; t2 += red[1][k * cols_per_pass + c_in]; }
            partial[((size_t)blockIdx.x * 2 + 0) * N + c] = t1;
            partial[((size_t)blockIdx.x * 2 + 1) * N + c] = t2;
        }
        __syncthreads();
    }
}

// out1 / out2 = fixed-order sums of the partials; FINISH 1: out1 = sum / R (mean); FINISH 2: out1 = 1 / sqrt(sum / R + eps) (rstd)
template <int FINISH>
__global__ void __launch_bounds__(256)
colsum2_reduce_kernel(const float *__restrict__ partial, int nblk, int N, int R, float eps, float *__restrict__ out1, float *__restrict__ out2) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    float s1 = 0.0f, s2 = 0.0f;
    for (int b = 0; b < nblk; ++b) { s1 += partial[((size_t)b * 2 + 0) * N + c]; s2 += partial[((size_t)b * 2 + 1) * N + c]; }
    if (FINISH == 1) {
        out1[c] = s1 / (float)R;
    } else if (FINISH == 2) {
        out1[c] = 1.0f / sqrtf(s1 / (float)R + eps);
    } else {
        out1[c] = s1;
        out2[c] = s2;
    }
}

// grouped statistics: the per-group partial IS the group's sum.  FINISH 1: mean = sum / L; FINISH 2: rstd = 1 / sqrt(sum / L + eps)
template <int FINISH>
__global__ void __launch_bounds__(256)
group_finish_kernel(const float *__restrict__ partial, size_t G, int N, int L, float eps, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= G * N) return;
    const size_t gidx = i / N;
    const int c = (int)(i - gidx * N);
    const float sv = partial[(gidx * 2 + 0) * N + c];
    out[i] = FINISH == 1 ? sv / (float)L : 1.0f / sqrtf(sv / (float)L + eps);
}

// a_out = dropout(AF(BN(z)))
__global__ void __launch_bounds__(256)
bnact_fwd_kernel(const float *__restrict__ z, const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ gamma,
                 const float *__restrict__ beta, BnActArgs a, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)a.R * a.N) return;
    const int r = (int)(i / a.N), c = (int)(i - (size_t)r * a.N);
    float y = z[(size_t)r * a.ld + c];
    const size_t so = a.group > 0 ? (size_t)(r / a.group) * a.N : 0;
    if (a.has_bn) y = fmaf(gamma ? gamma[c] : 1.0f, (y - mean[so + c]) * rstd[so + c], beta ? beta[c] : 0.0f);
    const float h = af_fwd(a.af, y);
    out[(size_t)r * a.ld + c] = h * drop_factor(a, r, c, drop_thr(a.p_drop), a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f);
}

// dz = gamma * rstd * (dy - sum_dy / R - xhat * sum_dyx / R)        (no BN: dz = dy)
__global__ void __launch_bounds__(256)
bnact_bwd_kernel(const float *__restrict__ z, const float *__restrict__ da, const float *__restrict__ mean, const float *__restrict__ rstd,
                 const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ sum_dy,
                 const float *__restrict__ sum_dyx, BnActArgs a, float *__restrict__ dz) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)a.R * a.N) return;
    const int r = (int)(i / a.N), c = (int)(i - (size_t)r * a.N);
    const float zv = z[(size_t)r * a.ld + c];
    const float thr_keep = drop_factor(a, r, c, drop_thr(a.p_drop), a.p_drop > 0.0f ? 1.0f / (1.0f - a.p_drop) : 1.0f);
    if (a.has_bn) {
        // grouped: sum_dy / sum_dyx are the per-group partials [G][2][N] of colsum2_kernel<1> (sum_dyx = sum_dy + N), mean over L rows
        const size_t so = a.group > 0 ? (size_t)(r / a.group) * a.N : 0, ss = a.group > 0 ? 2 * so : 0;
        const float ga = gamma ? gamma[c] : 1.0f, rs = rstd[so + c];
        const float xh = (zv - mean[so + c]) * rs;
        const float y = fmaf(ga, xh, beta ? beta[c] : 0.0f);
        const float dy = da[(size_t)r * a.ld + c] * thr_keep * af_bwd(a.af, y);
        const float invR = 1.0f / (float)(a.group > 0 ? a.group : a.R);
        dz[(size_t)r * a.ld + c] = (ga * rs) * (dy - sum_dy[ss + c] * invR - xh * (sum_dyx[ss + c] * invR));
    } else {
        dz[(size_t)r * a.ld + c] = da[(size_t)r * a.ld + c] * thr_keep * af_bwd(a.af, zv);
    }
}

static int bn_blocks(int R) {
    int b = (R + 255) / 256;
    return b < 1 ? 1 : (b > 512 ? 512 : b);
}

static int check_bnact(const char *who, int R, int N, int ld, int af, float p) {
    if (R < 0 || N <= 0 || ld < N) { set_error("%s: bad shape R=%d N=%d ld=%d", who, R, N, ld); return PTR_ERR_INVALID_ARG; }
    if (af < PTR_AF_NONE || af > PTR_AF_TANH) { set_error("%s: unknown activation %d", who, af); return PTR_ERR_INVALID_ARG; }
    if (!(p >= 0.0f && p < 1.0f)) { set_error("%s: dropout p=%g out of [0,1)", who, (double)p); return PTR_ERR_INVALID_ARG; }
    return 0;
}

}  // namespace ptr

// group_rows: 0 = statistics over all R rows (LTRBatchNorm); L > 0 = per group of L consecutive rows (per query, LTRBatchNorm2; R % L == 0)
extern "C" size_t ptr_bn_ws_floats(int R, int N, int group_rows) {
    const size_t blocks = group_rows > 0 ? (size_t)(R / group_rows) : (size_t)ptr::bn_blocks(R);
    return blocks * 2 * (size_t)N;
}

// mean / rstd ([N], or [R / group_rows][N]) of the columns of z (biased variance, rstd = 1 / sqrt(var + eps), two-pass)
extern "C" int ptr_bn_stats(const float *z, int ld, int R, int N, int group_rows, float eps, float *ws, float *mean, float *rstd, void *stream) {
    using namespace ptr;
    const char *who = "ptr_bn_stats";
    if (int rc = check_bnact(who, R, N, ld, 0, 0.0f)) return rc;
    if (R == 0 || !z || !ws || !mean || !rstd) { set_error("%s: NULL pointer / empty batch", who); return PTR_ERR_INVALID_ARG; }
    if (group_rows < 0 || (group_rows > 0 && R % group_rows)) { set_error("%s: R=%d is not a multiple of group_rows=%d", who, R, group_rows); return PTR_ERR_INVALID_ARG; }
    hipStream_t st = as_stream(stream);
    BnActArgs a{group_rows, R, N, ld, 0, 0, 0.0f, 0, 0, 0};
    if (group_rows > 0) {
        const int G = R / group_rows;
        const unsigned fin = (unsigned)(((size_t)G * N + 255) / 256);
        hipLaunchKernelGGL(colsum2_kernel<0>, dim3(G), dim3(256), 0, st, z, nullptr, nullptr, nullptr, nullptr, nullptr, a, ws);
        hipLaunchKernelGGL(group_finish_kernel<1>, dim3(fin), dim3(256), 0, st, ws, (size_t)G, N, group_rows, eps, mean);
        hipLaunchKernelGGL(colsum2_kernel<2>, dim3(G), dim3(256), 0, st, z, nullptr, mean, nullptr, nullptr, nullptr, a, ws);
        hipLaunchKernelGGL(group_finish_kernel<2>, dim3(fin), dim3(256), 0, st, ws, (size_t)G, N, group_rows, eps, rstd);
        return check_hip(hipGetLastError(), who);
    }
    const int nb = bn_blocks(R);
    hipLaunchKernelGGL(colsum2_kernel<0>, dim3(nb), dim3(256), 0, st, z, nullptr, nullptr, nullptr, nullptr, nullptr, a, ws);
    hipLaunchKernelGGL(colsum2_reduce_kernel<1>, dim3((N + 255) / 256), dim3(256), 0, st, ws, nb, N, R, eps, mean, nullptr);
    hipLaunchKernelGGL(colsum2_kernel<2>, dim3(nb), dim3(256), 0, st, z, nullptr, mean, nullptr, nullptr, nullptr, a, ws);
    hipLaunchKernelGGL(colsum2_reduce_kernel<2>, dim3((N + 255) / 256), dim3(256), 0, st, ws, nb, N, R, eps, rstd, nullptr);
    return check_hip(hipGetLastError(), who);
}

// out = dropout(AF(gamma * (z - mean) * rstd + beta))   (mean == NULL: no batch norm; gamma / beta NULL: no affine)
extern "C" int ptr_bnact_forward(const float *z, int ld, int R, int N, int group_rows, const float *mean, const float *rstd, const float *gamma,
                                 const float *beta, int af, float p_drop, uint64_t seed, int site, float *out, void *stream) {
    using namespace ptr;
    const char *who = "ptr_bnact_forward";
    if (int rc = check_bnact(who, R, N, ld, af, p_drop)) return rc;
    if (R == 0) return 0;
    if (!z || !out || (mean && !rstd)) { set_error("%s: NULL pointer", who); return PTR_ERR_INVALID_ARG; }
    if (group_rows < 0 || (group_rows > 0 && R % group_rows)) { set_error("%s: R=%d is not a multiple of group_rows=%d", who, R, group_rows); return PTR_ERR_INVALID_ARG; }
    BnActArgs a{mean ? group_rows : 0, R, N, ld, af, mean ? 1 : 0, p_drop, (uint32_t)seed, (uint32_t)(seed >> 32), site};
    const size_t n = (size_t)R * N;
    hipLaunchKernelGGL(bnact_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), z, mean, rstd, gamma, beta, a, out);
    return check_hip(hipGetLastError(), who);
}

// da -> dz (and, with batch norm, dgamma[N] = sum dy * xhat, dbeta[N] = sum dy over ALL rows; either may be NULL).
// ws: ptr_bn_ws_floats(R, N, group_rows) + 2 * N floats.
extern "C" int ptr_bnact_backward(const float *z, const float *da, int ld, int R, int N, int group_rows, const float *mean, const float *rstd,
                                  const float *gamma, const float *beta, int af, float p_drop, uint64_t seed, int site, float *ws,
                                  float *dz, float *dgamma, float *dbeta, void *stream) {
    using namespace ptr;
    const char *who = "ptr_bnact_backward";
    if (int rc = check_bnact(who, R, N, ld, af, p_drop)) return rc;
    if (R == 0) return 0;
    if (!z || !da || !dz || (mean && (!rstd || !ws))) { set_error("%s: NULL pointer", who); return PTR_ERR_INVALID_ARG; }
    if (group_rows < 0 || (group_rows > 0 && R % group_rows)) { set_error("%s: R=%d is not a multiple of group_rows=%d", who, R, group_rows); return PTR_ERR_INVALID_ARG; }
    hipStream_t st = as_stream(stream);
    BnActArgs a{mean ? group_rows : 0, R, N, ld, af, mean ? 1 : 0, p_drop, (uint32_t)seed, (uint32_t)(seed >> 32), site};
    const float *sum_dy = nullptr, *sum_dyx = nullptr;
    if (mean) {
        const int nb = a.group > 0 ? R / a.group : bn_blocks(R);
        float *tot_dy = ws + (size_t)nb * 2 * N, *tot_dyx = tot_dy + N;
        hipLaunchKernelGGL(colsum2_kernel<1>, dim3(nb), dim3(256), 0, st, z, da, mean, rstd, gamma, beta, a, ws);
        hipLaunchKernelGGL(colsum2_reduce_kernel<0>, dim3((N + 255) / 256), dim3(256), 0, st, ws, nb, N, R, 0.0f, tot_dy, tot_dyx);
        if (dbeta) { if (int e = check_hip(hipMemcpyAsync(dbeta, tot_dy, N * sizeof(float), hipMemcpyDeviceToDevice, st), who)) return e; }
        if (dgamma) { if (int e = check_hip(hipMemcpyAsync(dgamma, tot_dyx, N * sizeof(float), hipMemcpyDeviceToDevice, st), who)) return e; }
        sum_dy = a.group > 0 ? ws : tot_dy;                 // grouped: the per-group partials themselves
        sum_dyx = a.group > 0 ? ws + N : tot_dyx;
    }
    const size_t n = (size_t)R * N;
    hipLaunchKernelGGL(bnact_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, z, da, mean, rstd, gamma, beta, sum_dy, sum_dyx, a, dz);
    return check_hip(hipGetLastError(), who);
}
