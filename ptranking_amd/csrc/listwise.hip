// Listwise softmax-over-list kernels: ListNet, ListMLE, and the device tie-shuffle that feeds ListMLE.
// These are the HBM-bound members of the path: O(L) work per query, one wavefront per query, the query tile staged once
// in LDS (one coalesced HBM read of scores/labels[/perm], one coalesced write of the gradient).
//
// Reference: ptranking/ltr_adhoc/listwise/listnet.py:39; ptranking/ltr_adhoc/listwise/listmle.py:82,92-97;
//            ptranking/ltr_adhoc/util/sampling_utils.py:13-28 (arg_shuffle_ties).
#include "ptr_device.h"
#include "ptr_dropout.h"          // f32x4

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

// r5: the same ListNet out of REGISTERS for 16-byte aligned rows (L % 4 == 0) up to 1024 documents.  G = 16 / 32 / 64 lanes own one query
// (4 / 2 / 1 queries per wavefront), a lane owns the float4 chunks t, t + G, ...: one global_load_dwordx4 per operand and chunk, one
// global_store_dwordx4 of the gradient — no LDS round trip (the LDS kernel staged values only the writing lane ever re-read), no scalar
// 4-byte accesses.  Arithmetic on the transcendental pipe where the 1e-5 bar allows (the ring kernel's policy): e^x = v_exp_f32(x log2 e)
// on arguments <= 0, ln by v_log_f32, 1/z once per query by an IEEE division and a multiply per element; softmax(preds) in the backward
// is e^{s - m} / z from the forward's own exponentials (the reference re-evaluates exp(log_softmax), listnet.py:39 through autograd).
// Group reductions are xor butterflies inside the G lanes (DPP / ds_swizzle / ds_bpermute, lane_xor<>): every lane ends with the
// bit-identical sum, fixed order.
template <int G, class Op> __device__ __forceinline__ float group_butterfly(float v, int lane, Op op) {
    v = op(v, lane_xor<1>(v, lane));
    v = op(v, lane_xor<2>(v, lane));
    v = op(v, lane_xor<4>(v, lane));
    v = op(v, lane_xor<8>(v, lane));
    if constexpr (G >= 32) v = op(v, lane_xor<16>(v, lane));
    if constexpr (G >= 64) v = op(v, lane_xor<32>(v, lane));
    return v;
}
template <int G, int V>
__global__ void __launch_bounds__(kBlock)
listnet_vec_kernel(const float *__restrict__ preds, const float *__restrict__ labels, const int32_t *__restrict__ lens, int B, int L,
                   float *__restrict__ loss_q, float *__restrict__ grad) {
    constexpr int QW = kWave / G;                                   // queries per wavefront
    const int lane = threadIdx.x & 63, t = lane & (G - 1);
    const int q = (blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * QW + lane / G;
    const bool valid = q < B;
    const int qc = valid ? q : B - 1;                               // lanes past the batch recompute the last query and store nothing
    const int n = query_len(lens, qc, L);
    const f32x4 *ps = reinterpret_cast<const f32x4 *>(preds + (size_t)qc * L), *py = reinterpret_cast<const f32x4 *>(labels + (size_t)qc * L);
    const int L4 = L >> 2;
    f32x4 a[V], b[V];
#pragma unroll
    for (int m = 0; m < V; ++m) {
        const int c = t + m * G;
        const int cc = c < L4 ? c : 0;                              // chunks past the row: re-read chunk 0, masked below
        a[m] = ps[cc]; b[m] = py[cc];
    }
    const auto fmax2 = [](float x, float y) { return fmaxf(x, y); };
    const auto fadd2 = [](float x, float y) { return x + y; };
    float ms = -INFINITY, my = -INFINITY;
#pragma unroll
    for (int m = 0; m < V; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool in = 4 * (t + m * G) + e < n;
            a[m][e] = in ? a[m][e] : -INFINITY;                     // e^{-inf} = 0: padding leaves every sum
            b[m][e] = in ? b[m][e] : -INFINITY;
            ms = fmaxf(ms, a[m][e]); my = fmaxf(my, b[m][e]);
        }
    ms = group_butterfly<G>(ms, lane, fmax2); my = group_butterfly<G>(my, lane, fmax2);
    constexpr float kLog2e = 1.4426950408889634f;
    float zs = 0.0f, zy = 0.0f;
#pragma unroll
    for (int m = 0; m < V; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[m][e] -= ms;                                          // s - max (kept: log_softmax = (s - max) - ln z); -inf for padding (n > 0)
            const float ea = __builtin_amdgcn_exp2f(a[m][e] * kLog2e), eb = __builtin_amdgcn_exp2f((b[m][e] - my) * kLog2e);
            b[m][e] = eb;
            zs += ea; zy += eb;
        }
    zs = group_butterfly<G>(zs, lane, fadd2); zy = group_butterfly<G>(zy, lane, fadd2);
    const float lzs = fast_ln(zs), rzs = 1.0f / zs, rzy = 1.0f / zy;
    float loss = 0.0f, sumpy = 0.0f;
#pragma unroll
    for (int m = 0; m < V; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool in = 4 * (t + m * G) + e < n;
            const float pyv = b[m][e] * rzy;                        // softmax(labels)
            const float lsm = a[m][e] - lzs;                        // log_softmax(preds)
            loss -= in ? pyv * lsm : 0.0f;                          // (0 * -inf on padding otherwise)
            sumpy += pyv;
            b[m][e] = pyv;
        }
    loss = group_butterfly<G>(loss, lane, fadd2); sumpy = group_butterfly<G>(sumpy, lane, fadd2);
    f32x4 *g = reinterpret_cast<f32x4 *>(grad + (size_t)qc * L);
#pragma unroll
    for (int m = 0; m < V; ++m) {
        const int c = t + m * G;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sm = __builtin_amdgcn_exp2f(a[m][e] * kLog2e) * rzs;       // softmax(preds); 0 on padding
            o[e] = 4 * c + e < n ? sm * sumpy - b[m][e] : 0.0f;
        }
        if (valid && c < L4) g[c] = o;
    }
    if (valid && t == 0) loss_q[q] = n > 0 ? loss : 0.0f;
}

// true when the register kernel serves the batch; launches it
static int launch_listnet_vec(const float *preds, const float *labels, const int32_t *lens, int B, int L, float *loss_q, float *grad,
                              hipStream_t st, bool *served) {
    *served = false;
    static const bool off = [] { const char *e = getenv("PTR_LISTNET_VEC"); return e && atoi(e) == 0; }();
    if (off || L % 4 != 0 || L > 1024) return 0;
    if ((reinterpret_cast<uintptr_t>(preds) | reinterpret_cast<uintptr_t>(labels) | reinterpret_cast<uintptr_t>(grad)) & 15) return 0;
    *served = true;
    auto go = [&]<int G, int V>() -> int {
        constexpr int QPB = (kBlock / kWave) * (kWave / G);
        hipLaunchKernelGGL((listnet_vec_kernel<G, V>), dim3((B + QPB - 1) / QPB), dim3(kBlock), 0, st, preds, labels, lens, B, L, loss_q, grad);
        return check_hip(hipGetLastError(), "ptr_listnet_fwd_bwd");
    };
    if (L <= 64) return go.template operator()<16, 1>();
    if (L <= 128) return go.template operator()<32, 1>();
    if (L <= 256) return go.template operator()<64, 1>();
    if (L <= 512) return go.template operator()<64, 2>();
    return go.template operator()<64, 4>();
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

// r5: ListMLE with 16-byte accesses and the scans out of registers (L % 4 == 0, L <= 1024; one wavefront per query).  Lane t owns the 4 V
// CONSECUTIVE positions from 4 V t on: scores by one float4 per 4 positions, the int64 permutation by two dwordx4, the gradient written as
// float4.  The only LDS traffic left is what the permutation needs: the scores are parked once (ds_write_b128) for the gather u_k = s[pi_k],
// and the gradient is scattered through the same tile.  Tail sums T_k: sequential inside the lane's positions, then ONE cross-lane suffix
// scan of the lane totals shifted by a lane (an exclusive scan — inclusive minus own would cancel on the small tails); 1 / T_k prefix sums
// likewise (DPP).  exp / log on the transcendental pipe while the list's score range is below 24 (argument error <= 24 * 2^-24: 1.4e-6
// relative), libm otherwise (EXACT: denormal tail sums, ranges up to the fp32 exponent) — decided per wavefront.
using i64x2 = __attribute__((ext_vector_type(2))) long long;
template <int V, bool EXACT>
__device__ __forceinline__ void listmle_vec_body(const f32x4 (&a)[V], const int (&pi)[4 * V], int n, int lane, float m, float *S,
                                                 float &loss_out, f32x4 (&gout)[V]) {
    constexpr int E = 4 * V;
    constexpr float kLog2e = 1.4426950408889634f;
    const int p0 = E * lane;
    float u[E], e[E], T[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const bool in = p0 + i < n;
        u[i] = in ? S[pi[i]] : 0.0f;
        e[i] = in ? (EXACT ? expf(u[i] - m) : __builtin_amdgcn_exp2f((u[i] - m) * kLog2e)) : 0.0f;
    }
    float run = 0.0f;
#pragma unroll
    for (int i = E - 1; i >= 0; --i) { run += e[i]; T[i] = run; }
    const float later = dpp_wave_shl1(wave_incl_suffix_sum(run, lane));          // sum over the lanes behind this one
    float loss = 0.0f, inv[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const bool in = p0 + i < n;
        T[i] += later;
        const float lg = EXACT ? logf(T[i]) : fast_ln(T[i]);
        loss += in ? (lg + m) - u[i] : 0.0f;
        if constexpr (EXACT) inv[i] = in ? 1.0f / T[i] : 0.0f;
        else { const float r = __builtin_amdgcn_rcpf(T[i]); inv[i] = in ? fmaf(r, fmaf(-T[i], r, 1.0f), r) : 0.0f; }
    }
    loss_out = wave_sum_dpp(loss);
    run = 0.0f;
#pragma unroll
    for (int i = 0; i < E; ++i) { run += inv[i]; inv[i] = run; }
    const float before = dpp_wave_shr1(wave_incl_sum(run, lane));
    wave_lds_sync();                                            // every gather of S is done: the tile becomes the gradient
#pragma unroll
    for (int i = 0; i < E; ++i)
        if (p0 + i < n) S[pi[i]] = e[i] * (inv[i] + before) - 1.0f;
    wave_lds_sync();
#pragma unroll
    for (int mm = 0; mm < V; ++mm) {
        const f32x4 g4 = *reinterpret_cast<const f32x4 *>(S + p0 + 4 * mm);
#pragma unroll
        for (int c = 0; c < 4; ++c) gout[mm][c] = p0 + 4 * mm + c < n ? g4[c] : 0.0f;
    }
    (void)a;
}
template <int V>
__global__ void __launch_bounds__(kBlock)
listmle_vec_kernel(const float *__restrict__ preds, const int64_t *__restrict__ perm, const int32_t *__restrict__ lens, int B, int L,
                   float *__restrict__ loss_q, float *__restrict__ grad) {
    constexpr int E = 4 * V, LT = kWave * E;                        // positions per lane / per tile
    __shared__ __attribute__((aligned(16))) float tiles[(kBlock / kWave) * LT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = blockIdx.x * (kBlock / kWave) + wv;
    if (q >= B) return;                                             // waves are independent: no workgroup barrier below
    const int n = query_len(lens, q, L);
    float *S = tiles + wv * LT;
    const int p0 = E * lane;
    const f32x4 *ps = reinterpret_cast<const f32x4 *>(preds + (size_t)q * L);
    const i64x2 *pp = reinterpret_cast<const i64x2 *>(perm + (size_t)q * L);
    f32x4 a[V];
    i64x2 k2[2 * V];
#pragma unroll
    for (int mm = 0; mm < V; ++mm) {
        const int c = p0 + 4 * mm < L ? (p0 >> 2) + mm : 0;        // chunks past the row re-read chunk 0 (masked by n <= L)
        a[mm] = ps[c]; k2[2 * mm] = pp[2 * c]; k2[2 * mm + 1] = pp[2 * c + 1];
    }
    float mx = -INFINITY, mn = INFINITY;
    int pi[E];
#pragma unroll
    for (int mm = 0; mm < V; ++mm) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = p0 + 4 * mm + c;
            const bool in = i < n;
            mx = fmaxf(mx, in ? a[mm][c] : -INFINITY); mn = fminf(mn, in ? a[mm][c] : INFINITY);
            const long long k = k2[2 * mm + (c >> 1)][c & 1];
            pi[4 * mm + c] = (k < 0 || k >= n) ? (in ? i : 0) : (int)k;      // malformed input cannot index out of the tile
        }
        *reinterpret_cast<f32x4 *>(S + p0 + 4 * mm) = a[mm];
    }
    mx = wave_max(mx); mn = -wave_max(-mn);
    wave_lds_sync();
    float loss;
    f32x4 g[V];
    if (mx - mn < 24.0f) listmle_vec_body<V, false>(a, pi, n, lane, mx, S, loss, g);
    else listmle_vec_body<V, true>(a, pi, n, lane, mx, S, loss, g);
    f32x4 *go = reinterpret_cast<f32x4 *>(grad + (size_t)q * L);
#pragma unroll
    for (int mm = 0; mm < V; ++mm)
        if (p0 + 4 * mm < L) go[(p0 >> 2) + mm] = g[mm];
    if (lane == 0) loss_q[q] = loss;
}
static int launch_listmle_vec(const float *preds, const int64_t *perm, const int32_t *lens, int B, int L, float *loss_q, float *grad,
                              hipStream_t st, bool *served) {
    *served = false;
    static const bool off = [] { const char *e = getenv("PTR_LISTMLE_VEC"); return e && atoi(e) == 0; }();
    if (off || L % 4 != 0 || L > 1024) return 0;
    if ((reinterpret_cast<uintptr_t>(preds) | reinterpret_cast<uintptr_t>(perm) | reinterpret_cast<uintptr_t>(grad)) & 15) return 0;
    *served = true;
    auto go = [&]<int V>() -> int {
        hipLaunchKernelGGL((listmle_vec_kernel<V>), dim3((B + 3) / 4), dim3(kBlock), 0, st, preds, perm, lens, B, L, loss_q, grad);
        return check_hip(hipGetLastError(), "ptr_listmle_fwd_bwd");
    };
    if (L <= 256) return go.template operator()<1>();
    if (L <= 512) return go.template operator()<2>();
    return go.template operator()<4>();
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

// r5: the random field of a document from 32-bit arithmetic — one lowbias32 finaliser per document on a per-query key (the splitmix64
// finaliser above costs two 64-bit multiplies per document: six quarter-rate 32-bit multiplies; the shuffle kernel was issue-bound on them)
__device__ __forceinline__ uint32_t tie_query_key(uint64_t seed, uint32_t q) {
    return lowbias32((uint32_t)seed ^ lowbias32((uint32_t)(seed >> 32) + q * 0x9E3779B1u));
}
__device__ __forceinline__ uint32_t tie_hash(uint32_t qkey, uint32_t i) { return lowbias32(qkey + i * 0x85EBCA77u); }

// perm = argsort by (label descending, random key ascending, index ascending): a uniformly random order inside every
// group of equal labels — the distribution arg_shuffle_ties draws from (sampling_utils.py:13-28).
//
// r5, lists of up to 1024 documents (one wavefront per query): integer grades in [0, 63] (MultiLabel) are packed with the random field and
// the document index into ONE 32-bit key   label : LB | (2^FB - 1 - field) : FB | (N - 1 - index) : IB   (N = 64 DPT padded positions,
// IB = log2 N; LB = 3 when every grade of the list is below 8 — the reference's 0..4 — else 6; FB = 32 - LB - IB: 21 bits at 256 documents,
// 19 at 1024), the keys are sorted in registers by the bitonic network and the permutation is READ OFF the sorted keys' index bits: no rank
// search, no LDS, 16-byte loads and stores (lane t owns positions 4t .. 4t+3).  r6: two documents of one grade that draw the
// SAME field would be ordered by index — a small systematic departure from arg_shuffle_ties' uniform order (6-8 pairs per query in a tie
// group of 1000 documents at the 16 bits r5 left there).  Such keys end up ADJACENT in the sorted order, so one neighbour compare per position
// finds them, and a list that has any draws its fields again (up to four times; then the exact (label, 32-bit hash, index) comparison below,
// which other labels — or another wave's query in the same launch — take anyway).
typedef long i64x2_t __attribute__((ext_vector_type(2)));
template <int DPT>
__device__ __forceinline__ bool shuffle_ties_wave(const float *__restrict__ labels, int q, int n, int L, uint32_t qkey, int t,
                                                  int64_t *__restrict__ perm, bool aligned) {
    constexpr int N = kWave * DPT;
    constexpr int IB = DPT == 1 ? 6 : DPT == 2 ? 7 : DPT == 4 ? 8 : DPT == 8 ? 9 : 10;
    const float *row = labels + (size_t)q * L;
    const bool vec = DPT % 4 == 0 && (L & 3) == 0 && aligned;
    float y[DPT];
    if (vec) {
#pragma unroll
        for (int r4 = 0; r4 < DPT; r4 += 4) {
            const int i = t * DPT + r4;
            const f32x4 v = i < L ? *reinterpret_cast<const f32x4 *>(row + i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) y[r4 + c] = v[c];
        }
    } else {
#pragma unroll
        for (int r = 0; r < DPT; ++r) y[r] = t * DPT + r < n ? row[t * DPT + r] : 0.0f;
    }
    bool small_int = true, below8 = true;
#pragma unroll
    for (int r = 0; r < DPT; ++r) {
        const bool in = t * DPT + r < n;
        small_int &= !in || (y[r] >= 0.0f && y[r] < 64.0f && y[r] == floorf(y[r]));
        below8 &= !in || y[r] < 8.0f;
    }
    if (!__all(small_int)) return false;
    const int FB = __all(below8) ? 29 - IB : 26 - IB;         // wave-uniform: the shifts below take it from a scalar register
    const uint32_t fmask = (1u << FB) - 1u;
    uint32_t key[DPT];
    // Fields are drawn until no two documents of one grade share one (rejection sampling: the fields are i.i.d., so the order given
    // "all distinct" is still uniform over every tie group's permutations); 0.3 % of 1024-document lists with a 500-document tie group
    // exhaust the four draws and take the exact comparison
    bool done = false;
#pragma unroll 1
    for (uint32_t attempt = 0; attempt < 4u && !done; ++attempt) {
        const uint32_t akey = qkey + attempt * 0xC2B2AE3Du;
#pragma unroll
        for (int r = 0; r < DPT; ++r) {
            const int i = t * DPT + r;
            const uint32_t field = tie_hash(akey, (uint32_t)i) >> (32 - FB);
            key[r] = i < n ? ((uint32_t)y[r] << (FB + IB)) | ((fmask - field) << IB) | (uint32_t)(N - 1 - i) : 0u;
        }
        wave_sort_desc<DPT>(key, t);                           // padded positions (key 0) sort behind every document
        // two documents with one (grade, field) are neighbours now: position p against p + 1 (the next lane's first key for a lane's last one)
        const uint32_t next_first = (uint32_t)__shfl_down((int)key[0], 1, 64);
        bool collide = false;
#pragma unroll
        for (int r = 0; r < DPT; ++r) {
            const uint32_t nx = r + 1 < DPT ? key[r + 1] : next_first;
            collide |= t * DPT + r + 1 < n && ((key[r] ^ nx) >> IB) == 0u;
        }
        done = !__any(collide);
    }
    if (!done) return false;
    int64_t *orow = perm + (size_t)q * L;
    if (vec) {
#pragma unroll
        for (int r2 = 0; r2 < DPT; r2 += 2) {
            const int p = t * DPT + r2;
            i64x2_t o;
            o[0] = p < n ? (long)(N - 1 - (int)(key[r2] & (N - 1))) : (long)p;
            o[1] = p + 1 < n ? (long)(N - 1 - (int)(key[r2 + 1] & (N - 1))) : (long)(p + 1);
            if (p < L) *reinterpret_cast<i64x2_t *>(orow + p) = o;
        }
    } else {
#pragma unroll
        for (int r = 0; r < DPT; ++r) {
            const int p = t * DPT + r;
            if (p < L) orow[p] = p < n ? (int64_t)(N - 1 - (int)(key[r] & (N - 1))) : (int64_t)p;
        }
    }
    return true;
}

// Longer lists (four wavefronts per query) and the general path: label and an 18-bit random field are packed into ONE float key
// (exact below 2^24) and ranked by the shared counting sort; field collisions fall back to index order inside count_ranks'
// tie pass.  Anything else ranks with the exact three-way comparison.
template <int G, int DPT>
__global__ void __launch_bounds__(kBlock)
shuffle_ties_kernel(const float *__restrict__ labels, const int32_t *__restrict__ lens, int B, int L, int Lp, uint64_t seed,
                    int64_t *__restrict__ perm, int aligned) {
    constexpr int QPB = kBlock / G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, grp = tid / G, t = tid % G;
    const int q = G == kWave ? __builtin_amdgcn_readfirstlane(blockIdx.x * QPB + grp) : blockIdx.x * QPB + grp;   // one wavefront per query: scalar
    const bool valid = q < B;
    const int n = valid ? query_len(lens, q, L) : 0;
    float *keys = smem + (size_t)grp * 3 * Lp;                 // packed keys (fast path) / labels (general path)
    uint32_t *rnd = reinterpret_cast<uint32_t *>(keys + Lp);
    int *out = reinterpret_cast<int *>(keys + 2 * Lp);
    const uint32_t qkey = tie_query_key(seed, (uint32_t)q);
    if constexpr (G == kWave) {
        if (!valid) return;                                    // the waves of a block are independent: no workgroup barrier on this path
        if (shuffle_ties_wave<DPT>(labels, q, n, L, qkey, t, perm, aligned != 0)) return;
    }
    float y[DPT], key[DPT];
    uint32_t r[DPT];
    bool small_int = true;
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        const bool in = i < n;
        y[m] = in ? labels[(size_t)q * L + i] : 0.0f;
        r[m] = tie_hash(qkey, (uint32_t)i);
        small_int &= !in || (y[m] >= 0.0f && y[m] < 64.0f && y[m] == floorf(y[m]));
        key[m] = in ? y[m] * 262144.0f + (float)(262143u - (r[m] >> 14)) : -INFINITY;
    }
    bool fast;
    if constexpr (G == kWave) {
        fast = false;                             // the packed-key path above declined: labels that are not small integer grades
    } else {
        // one decision per workgroup keeps the barriers below uniform
        int *all_small = reinterpret_cast<int *>(smem + (size_t)QPB * 3 * Lp);   // carved from the dynamic region (no static LDS in
        if (tid == 0) *all_small = 1;                                            // front of it: keeps the float4 tiles 16-byte aligned)
        __syncthreads();
        if (!small_int) *all_small = 0;
        __syncthreads();
        fast = *all_small != 0;
    }
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        if (i < Lp) { keys[i] = fast ? key[m] : (i < n ? y[m] : -INFINITY); rnd[i] = r[m]; }
    }
    if constexpr (G == kWave) wave_lds_sync(); else __syncthreads();
    int rk[DPT];
    if (fast) {
        // integer keys below 2^24; field collisions (equal keys) recount exactly inside either form
        if constexpr (G == kWave) {
            count_ranks_wave<DPT>(keys, reinterpret_cast<float *>(out), n, Lp, t, key, rk);
            // `out` served as the FLOAT scratch of the sorted keys; the int stores below must not move above another lane's last float
            // loads of the binary search (type-based alias analysis would allow it): same fence as sort_desc_kernel
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else count_ranks_fast<G, DPT>(keys, out, n, t, key, rk);
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
    if constexpr (G == kWave) wave_lds_sync(); else __syncthreads();
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
        bool served = false;
        if (int rc = launch_listnet_vec(preds, labels, lens, B, L, loss_q, grad, as_stream(stream), &served)) return rc;
        if (!served) {                                        // unaligned rows, L % 4 != 0, L > 1024: the LDS kernel
            const int Lp = round_up(L, 4);
            const size_t per_q = 2 * (size_t)Lp * sizeof(float);
            const int wpb = waves_per_block(per_q);
            if (int e = allow_lds(listnet_kernel, wpb * per_q)) return e;
            hipLaunchKernelGGL(listnet_kernel, dim3((B + wpb - 1) / wpb), dim3(wpb * kWave), wpb * per_q, as_stream(stream), preds, labels,
                               (const float *)nullptr, lens, B, L, Lp, 1.0f, loss_q, grad);
            if (int rc = check_hip(hipGetLastError(), who)) return rc;
        }
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
        bool served = false;
        if (int rc = launch_listmle_vec(preds, perm, lens, B, L, loss_q, grad, as_stream(stream), &served)) return rc;
        if (!served) {                                        // unaligned rows, L % 4 != 0, L > 1024: the LDS kernel
            const int Lp = round_up(L, 4);
            const size_t per_q = 4 * (size_t)Lp * sizeof(float);
            const int wpb = waves_per_block(per_q);
            if (int e = allow_lds(listmle_kernel, wpb * per_q)) return e;
            hipLaunchKernelGGL(listmle_kernel, dim3((B + wpb - 1) / wpb), dim3(wpb * kWave), wpb * per_q, as_stream(stream), preds, perm,
                               lens, B, L, Lp, loss_q, grad);
            if (int rc = check_hip(hipGetLastError(), who)) return rc;
        }
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
        hipLaunchKernelGGL(kern, dim3((B + QPB - 1) / QPB), dim3(kBlock), lds, as_stream(stream), labels, lens, B, L, Lp, seed, perm,
                           (int)(((reinterpret_cast<uintptr_t>(labels) | reinterpret_cast<uintptr_t>(perm)) & 15) == 0));
        return check_hip(hipGetLastError(), who);
    });
}
