// Listwise softmax-over-list kernels: ListNet, ListMLE, and the device tie-shuffle that feeds ListMLE.
// These are the HBM-bound members of the path: O(L) work per query, one wavefront per query, the query tile staged once
// in LDS (one coalesced HBM read of scores/labels[/perm], one coalesced write of the gradient).
//
// Reference: ptranking/ltr_adhoc/listwise/listnet.py:39; ptranking/ltr_adhoc/listwise/listmle.py:82,92-97;
//            ptranking/ltr_adhoc/util/sampling_utils.py:13-28 (arg_shuffle_ties).
#include "ptr_device.h"

namespace ptr {

// queries per workgroup given the LDS bytes one query needs (one wave per query, at most 4 waves)
static inline int waves_per_block(size_t bytes_per_query) {
    const size_t budget = 150 * 1024;
    int w = (int)(budget / (bytes_per_query ? bytes_per_query : 1));
    return w >= 4 ? 4 : (w >= 2 ? 2 : 1);
}

// ------------------------------------------------------------------------------------------------ ListNet
// loss_q = -sum_i softmax(labels)_i * log_softmax(preds)_i ; grad_i = softmax(preds)_i * sum_j softmax(labels)_j - softmax(labels)_i
// STListNet (ptranking/ltr_adhoc/listwise/st_listnet.py:41-49): the same loss on (preds + gumbel) / temperature with
// gumbel = -log(-log(u + 1e-20) + 1e-20), u = `unif` (the reference draws it with torch.rand); unif == nullptr, inv_temp == 1
// is plain ListNet.  d loss / d preds picks up the 1/temperature factor.
__global__ void __launch_bounds__(kBlock)
listnet_kernel(const float *__restrict__ preds, const float *__restrict__ labels, const float *__restrict__ unif,
               const int32_t *__restrict__ lens, int B, int L, int Lp, float inv_temp, float *__restrict__ loss_q,
               float *__restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int q = blockIdx.x * wpb + wv;
    if (q >= B) return;                                   // no workgroup barriers below: waves are independent
    const int n = query_len(lens, q, L);
    float *s = smem + (size_t)wv * 2 * Lp, *y = s + Lp;
    const float *ps = preds + (size_t)q * L, *py = labels + (size_t)q * L;

    float ms = -INFINITY, my = -INFINITY;
    for (int i = lane; i < n; i += 64) {
        float a = ps[i];
        const float b = py[i];
        if (unif) {
            const float u = unif[(size_t)q * L + i];
            a = (a + -logf(-logf(u + 1e-20f) + 1e-20f)) * inv_temp;      // st_listnet.py:41-45 ((x + g) / T as a multiply by 1/T)
        }
        s[i] = a; y[i] = b;
        ms = fmaxf(ms, a); my = fmaxf(my, b);
    }
    ms = wave_max(ms); my = wave_max(my);
    float zs = 0.0f, zy = 0.0f;
    for (int i = lane; i < n; i += 64) {                  // each lane re-reads only what it wrote: no fence needed
        const float ea = expf(s[i] - ms), eb = expf(y[i] - my);
        y[i] = eb;
        zs += ea; zy += eb;
    }
    zs = wave_sum(zs); zy = wave_sum(zy);
    const float lzs = logf(zs);
    float loss = 0.0f, sumpy = 0.0f;
    for (int i = lane; i < n; i += 64) {
        const float pyv = y[i] / zy;
        const float lsm = (s[i] - ms) - lzs;              // log_softmax
        loss -= pyv * lsm;
        sumpy += pyv;
        y[i] = pyv; s[i] = lsm;
    }
    loss = wave_sum(loss); sumpy = wave_sum(sumpy);
    float *g = grad + (size_t)q * L;
    for (int i = lane; i < L; i += 64) g[i] = i < n ? (expf(s[i]) * sumpy - y[i]) * inv_temp : 0.0f;   // log_softmax backward
    if (lane == 0) loss_q[q] = loss;
}

// ------------------------------------------------------------------------------------------------ ListMLE
// u = s[perm]; m = max u; T_k = sum_{j>=k} exp(u_j - m); loss = sum_k (log T_k + m - u_k);
// d loss / d u_k = exp(u_k - m) * sum_{i<=k} 1/T_i - 1, scattered back through perm.
__global__ void __launch_bounds__(kBlock)
listmle_kernel(const float *__restrict__ preds, const int64_t *__restrict__ perm, const int32_t *__restrict__ lens, int B, int L,
               int Lp, float *__restrict__ loss_q, float *__restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int q = blockIdx.x * wpb + wv;
    if (q >= B) return;
    const int n = query_len(lens, q, L);
    float *s = smem + (size_t)wv * 4 * Lp;                // scores by original index; later: gradient by original index
    float *T = s + Lp;                                    // tail sums by position
    float *E = T + Lp;                                    // exp(u - m) by position
    int *pi = reinterpret_cast<int *>(E + Lp);            // permutation, narrowed to int32
    const float *ps = preds + (size_t)q * L;
    const int64_t *pp = perm + (size_t)q * L;

    float m = -INFINITY;
    for (int i = lane; i < n; i += 64) {
        const float a = ps[i];
        s[i] = a;
        m = fmaxf(m, a);
        long long k = pp[i];
        pi[i] = (k < 0 || k >= n) ? i : (int)k;           // malformed input cannot index out of the tile
    }
    m = wave_max(m);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // s[] / pi[] written by other lanes are read below
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    const int nchunk = (n + 63) >> 6;
    float carry = 0.0f, loss = 0.0f;
    for (int c = nchunk - 1; c >= 0; --c) {               // flip-cumsum-flip (listmle.py:95), chunk by chunk from the tail
        const int k = c * 64 + lane;
        const bool in = k < n;
        const float u = in ? s[pi[k]] : 0.0f;
        const float e = in ? expf(u - m) : 0.0f;
        const float Tk = wave_incl_suffix_sum(e, lane) + carry;
        carry = __shfl(Tk, 0, 64);
        if (in) { T[k] = Tk; E[k] = e; loss += (logf(Tk) + m) - u; }
    }
    loss = wave_sum(loss);
    // every lane now re-reads only its own T[k]/E[k]; the gradient is scattered into s[] which nobody reads any more
    __builtin_amdgcn_wave_barrier();
    float pc = 0.0f;
    for (int c = 0; c < nchunk; ++c) {
        const int k = c * 64 + lane;
        const bool in = k < n;
        const float inv = in ? 1.0f / T[k] : 0.0f;
        const float P = wave_incl_sum(inv, lane) + pc;
        pc = __shfl(P, 63, 64);
        if (in) s[pi[k]] = E[k] * P - 1.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float *g = grad + (size_t)q * L;
    for (int i = lane; i < L; i += 64) g[i] = i < n ? s[i] : 0.0f;
    if (lane == 0) loss_q[q] = loss;
}

// MDPRank (ptranking/ltr_adhoc/listwise/mdprank.py:24-78): a policy-gradient ListMLE.  pi is a ranking SAMPLED from the
// Plackett-Luce model (the host draws it), u = scores in sampled order, and the first top_k positions are weighted with the
// discounted long-term return of the episode:
//   reward_t = (2^{l_pi(t)} - 1) / log2(2 + t)  (t < top_k),   G_t = gamma^{t+1} * sum_{t' = t}^{top_k - 1} reward_t'
//   loss = sum_{t < top_k} G_t * ( log sum_{j >= t} exp(u_j) - u_t )
// Gradient: dL/du_j = e^{u_j - m} * sum_{i <= min(j, top_k-1)} G_i / T_i  -  (j < top_k ? G_j : 0).
// One wavefront per query, same scans as listmle_kernel plus a suffix scan of the rewards.
// LDS per query: s[Lp] | T[Lp] | E[Lp] | W[Lp] | pi[Lp]
__global__ void __launch_bounds__(kBlock)
mdprank_kernel(const float *__restrict__ preds, const float *__restrict__ labels, const int64_t *__restrict__ perm,
               const int32_t *__restrict__ lens, int B, int L, int Lp, int top_k, float gamma, float *__restrict__ loss_q,
               float *__restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int q = blockIdx.x * wpb + wv;
    if (q >= B) return;
    const int n = query_len(lens, q, L);
    float *s = smem + (size_t)wv * 5 * Lp;
    float *T = s + Lp, *E = T + Lp, *W = E + Lp;
    int *pi = reinterpret_cast<int *>(W + Lp);
    const float *ps = preds + (size_t)q * L, *ys = labels + (size_t)q * L;
    const int64_t *pp = perm + (size_t)q * L;
    const int top = (top_k <= 0 || top_k > n) ? n : top_k;       // top_k=None -> the whole list (mdprank.py:46)

    float m = -INFINITY;
    for (int i = lane; i < n; i += 64) {
        const float a = ps[i];
        s[i] = a;
        m = fmaxf(m, a);
        long long k = pp[i];
        pi[i] = (k < 0 || k >= n) ? i : (int)k;
    }
    m = wave_max(m);                                             // mdprank.py:65
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    const int nchunk = (n + 63) >> 6;
    float carry = 0.0f, rcarry = 0.0f, loss = 0.0f;
    for (int c = nchunk - 1; c >= 0; --c) {                      // both tail sums, chunk by chunk from the tail
        const int k = c * 64 + lane;
        const bool in = k < n;
        const int src = in ? pi[k] : 0;
        const float u = in ? s[src] : 0.0f;
        const float e = in ? expf(u - m) : 0.0f;
        const float r = (in && k < top) ? gain_of(ys[src]) / log2f(2.0f + (float)k) : 0.0f;     // mdprank.py:53-56
        const float Tk = wave_incl_suffix_sum(e, lane) + carry;
        const float Rk = wave_incl_suffix_sum(r, lane) + rcarry;                                // :59
        carry = __shfl(Tk, 0, 64);
        rcarry = __shfl(Rk, 0, 64);
        if (in) {
            const float w = k < top ? Rk * (gamma == 1.0f ? 1.0f : powf(gamma, (float)(k + 1))) : 0.0f;   // :61-63
            T[k] = Tk; E[k] = e; W[k] = w;
            loss += w * ((logf(Tk) + m) - u);                                                   // :68-70
        }
    }
    loss = wave_sum(loss);
    __builtin_amdgcn_wave_barrier();
    float pc = 0.0f;
    for (int c = 0; c < nchunk; ++c) {
        const int k = c * 64 + lane;
        const bool in = k < n;
        const float term = in ? W[k] / T[k] : 0.0f;
        const float P = wave_incl_sum(term, lane) + pc;
        pc = __shfl(P, 63, 64);
        if (in) s[pi[k]] = E[k] * P - W[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float *g = grad + (size_t)q * L;
    for (int i = lane; i < L; i += 64) g[i] = i < n ? s[i] : 0.0f;
    if (lane == 0) loss_q[q] = loss;
}

// ------------------------------------------------------------------------------------------------ RankMSE / RankCosine
// RankMSE (ptranking/ltr_adhoc/pointwise/rank_mse.py:13-22): mean over queries of sum_i (s_i - y_i)^2.  loss_q holds the
// per-query sums; the caller's reduction applies 1/B, the gradient 2 (s - y) / B is written here.
__global__ void __launch_bounds__(kBlock)
rankmse_kernel(const float *__restrict__ preds, const float *__restrict__ labels, const int32_t *__restrict__ lens, int B, int L,
               float inv_b, float *__restrict__ loss_q, float *__restrict__ grad) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int q = blockIdx.x * wpb + wv;
    if (q >= B) return;
    const int n = query_len(lens, q, L);
    float acc = 0.0f;
    for (int i = lane; i < L; i += 64) {
        float gi = 0.0f;
        if (i < n) {
            const float d = preds[(size_t)q * L + i] - labels[(size_t)q * L + i];
            acc = fmaf(d, d, acc);
            gi = 2.0f * d * inv_b;
        }
        grad[(size_t)q * L + i] = gi;
    }
    acc = wave_sum(acc);
    if (lane == 0) loss_q[q] = acc;
}

// RankCosine (ptranking/ltr_adhoc/listwise/rank_cosine.py:15,32): sum over queries of (1 - cos(s, y)) / 0.5 with
// nn.CosineSimilarity(dim=1, eps=1e-8) = <s,y> / (max(|s|, eps) * max(|y|, eps)).
__global__ void __launch_bounds__(kBlock)
rankcosine_kernel(const float *__restrict__ preds, const float *__restrict__ labels, const int32_t *__restrict__ lens, int B, int L,
                  float *__restrict__ loss_q, float *__restrict__ grad) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
    const int q = blockIdx.x * wpb + wv;
    if (q >= B) return;
    const int n = query_len(lens, q, L);
    const float *ps = preds + (size_t)q * L, *py = labels + (size_t)q * L;
    float sy = 0.0f, ss = 0.0f, yy = 0.0f;
    for (int i = lane; i < n; i += 64) {
        const float a = ps[i], b = py[i];
        sy = fmaf(a, b, sy); ss = fmaf(a, a, ss); yy = fmaf(b, b, yy);
    }
    sy = wave_sum(sy); ss = wave_sum(ss); yy = wave_sum(yy);
    const float eps = 1e-8f;
    const float ns = sqrtf(ss), ny = sqrtf(yy);
    const float ds = fmaxf(ns, eps), dy = fmaxf(ny, eps);
    const float c = sy / (ds * dy);
    // d cos / d s_i = y_i / (ds dy) - [ns > eps] * cos * s_i / ns^2
    const float k1 = 1.0f / (ds * dy), k2 = ns > eps ? c / ss : 0.0f;
    float *g = grad + (size_t)q * L;
    for (int i = lane; i < L; i += 64) g[i] = i < n ? -2.0f * (py[i] * k1 - k2 * ps[i]) : 0.0f;
    if (lane == 0) loss_q[q] = (1.0f - c) / 0.5f;
}

// ------------------------------------------------------------------------------------------------ tie shuffle
__device__ __forceinline__ uint32_t mix64(uint64_t x) {      // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (uint32_t)(x >> 16);
}

// perm = argsort by (label descending, random key ascending, index ascending): a uniformly random order inside every
// group of equal labels — the distribution arg_shuffle_ties draws from (sampling_utils.py:13-28).
// Integer grades in [0, 63] (MultiLabel) take the fast path: label and an 18-bit random field are packed into ONE float key
// (exact below 2^24) and ranked by the shared counting sort; field collisions fall back to index order inside count_ranks'
// tie pass.  Anything else ranks with the exact three-way comparison.
template <int G, int DPT>
__global__ void __launch_bounds__(kBlock)
shuffle_ties_kernel(const float *__restrict__ labels, const int32_t *__restrict__ lens, int B, int L, int Lp, uint64_t seed,
                    int64_t *__restrict__ perm) {
    constexpr int QPB = kBlock / G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, grp = tid / G, t = tid % G;
    const int q = blockIdx.x * QPB + grp;
    const bool valid = q < B;
    const int n = valid ? query_len(lens, q, L) : 0;
    float *keys = smem + (size_t)grp * 3 * Lp;                 // packed keys (fast path) / labels (general path)
    uint32_t *rnd = reinterpret_cast<uint32_t *>(keys + Lp);
    int *out = reinterpret_cast<int *>(keys + 2 * Lp);
    float y[DPT], key[DPT];
    uint32_t r[DPT];
    bool small_int = true;
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        const bool in = i < n;
        y[m] = in ? labels[(size_t)q * L + i] : 0.0f;
        r[m] = mix64(seed ^ ((uint64_t)q * 0x100000001B3ull + (uint64_t)i) * 0xD6E8FEB86659FD93ull);
        small_int &= !in || (y[m] >= 0.0f && y[m] < 64.0f && y[m] == floorf(y[m]));
        key[m] = in ? y[m] * 262144.0f + (float)(262143u - (r[m] >> 14)) : -INFINITY;
    }
    // one decision per workgroup keeps the barriers below uniform
    int *all_small = reinterpret_cast<int *>(smem + (size_t)QPB * 3 * Lp);   // carved from the dynamic region (no static LDS in
    if (tid == 0) *all_small = 1;                                            // front of it: keeps the float4 tiles 16-byte aligned)
    __syncthreads();
    if (!small_int) *all_small = 0;
    __syncthreads();
    const bool fast = *all_small != 0;
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        if (i < Lp) { keys[i] = fast ? key[m] : (i < n ? y[m] : -INFINITY); rnd[i] = r[m]; }
    }
    __syncthreads();
    int rk[DPT];
    if (fast) {
        // integer keys below 2^24; field collisions (equal keys) recount exactly inside either form
        if constexpr (G == kWave) count_ranks_wave<DPT>(keys, reinterpret_cast<float *>(out), n, Lp, t, key, rk);
        else count_ranks_fast<G, DPT>(keys, out, n, t, key, rk);
    } else {
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int i = t + m * G;
            int c = 0;
            if (i < n)
                for (int j = 0; j < n; ++j) {
                    const float yj = keys[j];
                    const uint32_t rj = rnd[j];
                    c += (yj > y[m] || (yj == y[m] && (rj < r[m] || (rj == r[m] && j < i)))) ? 1 : 0;
                }
            rk[m] = c;
        }
    }
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        if (i < n) out[rk[m]] = i;
    }
    __syncthreads();
    if (valid) {
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int i = t + m * G;
            if (i < L) perm[(size_t)q * L + i] = i < n ? (int64_t)out[i] : (int64_t)i;
        }
    }
}

}  // namespace ptr

extern "C" int ptr_listnet_fwd_bwd(const float *preds, const float *labels, const int32_t *lens, int B, int L, float *loss_out,
                                   float *loss_q, float *grad, void *stream) {
    using namespace ptr;
    const char *who = "ptr_listnet_fwd_bwd";
    if (int rc = check_batch(preds, labels, B, L, who)) return rc;
    if (B > 0 && (!loss_q || !grad)) { set_error("%s: NULL output pointer", who); return PTR_ERR_INVALID_ARG; }
    if (B > 0) {
        const int Lp = round_up(L, 4);
        const size_t per_q = 2 * (size_t)Lp * sizeof(float);
        const int wpb = waves_per_block(per_q);
        if (int e = allow_lds(listnet_kernel, wpb * per_q)) return e;
        hipLaunchKernelGGL(listnet_kernel, dim3((B + wpb - 1) / wpb), dim3(wpb * kWave), wpb * per_q, as_stream(stream), preds, labels,
                           (const float *)nullptr, lens, B, L, Lp, 1.0f, loss_q, grad);
        if (int rc = check_hip(hipGetLastError(), who)) return rc;
    }
    return loss_out ? ptr_sum_f32(loss_q, B, 1.0f, loss_out, stream) : 0;
}

extern "C" int ptr_stlistnet_fwd_bwd(const float *preds, const float *labels, const float *unif, const int32_t *lens, int B, int L,
                                     float temperature, float *loss_out, float *loss_q, float *grad, void *stream) {
    using namespace ptr;
    const char *who = "ptr_stlistnet_fwd_bwd";
    if (int rc = check_batch(preds, labels, B, L, who)) return rc;
    if (B > 0 && (!loss_q || !grad || !unif)) { set_error("%s: NULL pointer", who); return PTR_ERR_INVALID_ARG; }
    if (!(temperature > 0.0f)) { set_error("%s: temperature must be > 0 (got %g)", who, (double)temperature); return PTR_ERR_INVALID_ARG; }
    if (B > 0) {
        const int Lp = round_up(L, 4);
        const size_t per_q = 2 * (size_t)Lp * sizeof(float);
        const int wpb = waves_per_block(per_q);
        if (int e = allow_lds(listnet_kernel, wpb * per_q)) return e;
        hipLaunchKernelGGL(listnet_kernel, dim3((B + wpb - 1) / wpb), dim3(wpb * kWave), wpb * per_q, as_stream(stream), preds, labels,
                           unif, lens, B, L, Lp, 1.0f / temperature, loss_q, grad);
        if (int rc = check_hip(hipGetLastError(), who)) return rc;
    }
    return loss_out ? ptr_sum_f32(loss_q, B, 1.0f, loss_out, stream) : 0;
}

extern "C" int ptr_rankmse_fwd_bwd(const float *preds, const float *labels, const int32_t *lens, int B, int L, float *loss_out,
                                   float *loss_q, float *grad, void *stream) {
    using namespace ptr;
    const char *who = "ptr_rankmse_fwd_bwd";
    if (int rc = check_batch(preds, labels, B, L, who)) return rc;
    if (B > 0 && (!loss_q || !grad)) { set_error("%s: NULL output pointer", who); return PTR_ERR_INVALID_ARG; }
    if (B > 0) {
        hipLaunchKernelGGL(rankmse_kernel, dim3((B + 3) / 4), dim3(kBlock), 0, as_stream(stream), preds, labels, lens, B, L, 1.0f / (float)B,
                           loss_q, grad);
        if (int rc = check_hip(hipGetLastError(), who)) return rc;
    }
    return loss_out ? ptr_sum_f32(loss_q, B, B > 0 ? 1.0f / (float)B : 0.0f, loss_out, stream) : 0;
}

extern "C" int ptr_rankcosine_fwd_bwd(const float *preds, const float *labels, const int32_t *lens, int B, int L, float *loss_out,
                                      float *loss_q, float *grad, void *stream) {
    using namespace ptr;
    const char *who = "ptr_rankcosine_fwd_bwd";
    if (int rc = check_batch(preds, labels, B, L, who)) return rc;
    if (B > 0 && (!loss_q || !grad)) { set_error("%s: NULL output pointer", who); return PTR_ERR_INVALID_ARG; }
    if (B > 0) {
        hipLaunchKernelGGL(rankcosine_kernel, dim3((B + 3) / 4), dim3(kBlock), 0, as_stream(stream), preds, labels, lens, B, L, loss_q, grad);
        if (int rc = check_hip(hipGetLastError(), who)) return rc;
    }
    return loss_out ? ptr_sum_f32(loss_q, B, 1.0f, loss_out, stream) : 0;
}

extern "C" int ptr_listmle_fwd_bwd(const float *preds, const int64_t *perm, const int32_t *lens, int B, int L, float *loss_out,
                                   float *loss_q, float *grad, void *stream) {
    using namespace ptr;
    const char *who = "ptr_listmle_fwd_bwd";
    if (int rc = check_batch(preds, perm, B, L, who)) return rc;
    if (B > 0 && (!loss_q || !grad)) { set_error("%s: NULL output pointer", who); return PTR_ERR_INVALID_ARG; }
    if (B > 0) {
        const int Lp = round_up(L, 4);
        const size_t per_q = 4 * (size_t)Lp * sizeof(float);
        const int wpb = waves_per_block(per_q);
        if (int e = allow_lds(listmle_kernel, wpb * per_q)) return e;
        hipLaunchKernelGGL(listmle_kernel, dim3((B + wpb - 1) / wpb), dim3(wpb * kWave), wpb * per_q, as_stream(stream), preds, perm,
                           lens, B, L, Lp, loss_q, grad);
        if (int rc = check_hip(hipGetLastError(), who)) return rc;
    }
    return loss_out ? ptr_sum_f32(loss_q, B, 1.0f, loss_out, stream) : 0;
}

extern "C" int ptr_mdprank_fwd_bwd(const float *preds, const float *labels, const int64_t *perm, const int32_t *lens, int B, int L,
                                   int top_k, float gamma, float *loss_out, float *loss_q, float *grad, void *stream) {
    using namespace ptr;
    const char *who = "ptr_mdprank_fwd_bwd";
    if (int rc = check_batch(preds, perm, B, L, who)) return rc;
    if (B > 0 && (!labels || !loss_q || !grad)) { set_error("%s: NULL pointer", who); return PTR_ERR_INVALID_ARG; }
    if (!(gamma > 0.0f)) { set_error("%s: gamma must be > 0 (got %g)", who, (double)gamma); return PTR_ERR_INVALID_ARG; }
    if (B > 0) {
        const int Lp = round_up(L, 4);
        const size_t per_q = 5 * (size_t)Lp * sizeof(float);
        const int wpb = waves_per_block(per_q);
        if (int e = allow_lds(mdprank_kernel, wpb * per_q)) return e;
        hipLaunchKernelGGL(mdprank_kernel, dim3((B + wpb - 1) / wpb), dim3(wpb * kWave), wpb * per_q, as_stream(stream), preds, labels,
                           perm, lens, B, L, Lp, top_k, gamma, loss_q, grad);
        if (int rc = check_hip(hipGetLastError(), who)) return rc;
    }
    return loss_out ? ptr_sum_f32(loss_q, B, 1.0f, loss_out, stream) : 0;
}

extern "C" int ptr_shuffle_ties_order(const float *labels, const int32_t *lens, int B, int L, uint64_t seed, int64_t *perm,
                                      void *stream) {
    using namespace ptr;
    const char *who = "ptr_shuffle_ties_order";
    if (int rc = check_batch(labels, perm, B, L, who)) return rc;
    if (B == 0) return 0;
    return dispatch_wave_tiling(L, [&]<int G, int DPT>() -> int {
        constexpr int QPB = kBlock / G;
        const int Lp = G == kWave ? kWave * DPT : round_up(L, 4);      // one wavefront per query: 64*DPT padded keys through the register sort
        auto kern = shuffle_ties_kernel<G, DPT>;
        const size_t lds = ((size_t)QPB * 3 * Lp + 4) * sizeof(float);
        if (int e = allow_lds(kern, lds)) return e;
        hipLaunchKernelGGL(kern, dim3((B + QPB - 1) / QPB), dim3(kBlock), lds, as_stream(stream), labels, lens, B, L, Lp, seed, perm);
        return check_hip(hipGetLastError(), who);
    });
}
