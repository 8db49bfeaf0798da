// Fused ApproxNDCG kernels.
//
// Reference: ptranking/ltr_adhoc/listwise/approxNDCG.py:19-27 (approximate ranks), :45-62 (loss), :83-109 (ranker),
//            Robust_Sigmoid ptranking/base/utils.py:57-95.
//
//   pi_hat_i = 0.5 + sum_j rs(alpha * (s_j - s_i))          (j == i contributes 0.5)
//   DCG_b    = sum_i (2^l_i - 1) / log2(pi_hat_i + 1)       in ideal (label-descending) order
//   loss     = -(sum_b DCG_b) * (sum_a 1/IDCG_a)            <- the reference's [B]/[B,1] broadcast (SURVEY.md §7 vi)
//
// Kernel 1 (per query, O(L^2) sigmoids, two circulant half-matrix passes):
//   pass 1: every unordered pair {a,b} is visited once; e = exp(-alpha*|s_a - s_b|) gives BOTH indicators
//           (1/(1+e) for the higher score's partner, e/(1+e) for the other) — Robust_Sigmoid's two branches;
//   pass 2: gradient; d(pi_hat_a)/d(s_b) = alpha*y_ab*(1-y_ab) with y_ab the stored-forward value, exactly the tensor the
//           reference saves for backward (base/utils.py:78-79).
//   Outputs per query: DCG_b, 1/IDCG_b and the gradient for scale 1 (or already times 1/IDCG_b when un-coupled).
// Kernel 2 (one workgroup): S = sum 1/IDCG, loss, scale.   Kernel 3: grad *= S (coupled mode only).
#include "ptr_device.h"

namespace ptr {

// LDS per group (floats): S_id[Lp] | Y_id[Lp] | acc[NW][Lp] | red[4]
__host__ __device__ constexpr size_t approx_group_floats(int Lp, int NW) { return (size_t)Lp * (2 + NW) + 4; }

// Both Robust_Sigmoid values of an unordered pair from one exponential.
// delta = s_b - s_a.  ya = rs(alpha*delta) (contribution of b to pi_hat_a), yb = rs(-alpha*delta).
__device__ __forceinline__ void robust_pair(float delta, float alpha, float &ya, float &yb) {
    const float x = alpha * fabsf(delta);
    const float e = __expf(-x);
    const float dd = 1.0f + e;
    float r = __builtin_amdgcn_rcpf(dd);
    r = fmaf(r, fmaf(-dd, r, 1.0f), r);          // 1/(1+e)      (base/utils.py:71)
    const float sm = e * r;                       // e/(1+e)      (base/utils.py:73-74)
    const bool pos = delta > 0.0f, neg = delta < 0.0f;
    ya = pos ? r : (neg ? sm : 0.5f);
    yb = pos ? sm : (neg ? r : 0.5f);
}

// SoftRank (ptranking/ltr_adhoc/listwise/softrank.py:47-69) is the same two-pass scheme with a Gaussian rank indicator:
//   E[rank_i] = 1 + sum_{j != i} 0.5*erfc((s_i - s_j)/den),  den = sqrt(2*(2*delta^2))               (:50-56)
//   loss      = -sum_q sum_{i < top_k} (2^l_i - 1) / (log2(E[rank_i] + 1) * IDCG_q)                  (:58-69; no batch coupling)
// Both indicators of an unordered pair from one erfc: the small one directly, the other as its complement.
// `alpha` carries 1/den.  delta = s_b - s_a; ya = 0.5*erfc((s_a - s_b)/den) is b's contribution to E[rank_a].
__device__ __forceinline__ void soft_pair(float delta, float inv_den, float &ya, float &yb) {
    const float sm = 0.5f * erfcf(fabsf(delta) * inv_den);
    const float lg = 1.0f - sm;
    const bool pos = delta > 0.0f;
    ya = pos ? lg : sm;
    yb = pos ? sm : lg;
}

template <int G, int DPT, bool SOFT>
__global__ void __launch_bounds__(kBlock)
approxndcg_kernel(const float *__restrict__ preds, const float *__restrict__ labels, const int32_t *__restrict__ lens, int B,
                  int L, int Lp, float alpha, int presort, int couple_batch, int top_k, float *__restrict__ dcg_q,
                  float *__restrict__ inv_idcg_q, float *__restrict__ grad) {
    constexpr int QPB = kBlock / G, NW = G / kWave;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, grp = tid / G, t = tid % G, wv = t >> 6;
    const int q = blockIdx.x * QPB + grp;
    const bool valid = q < B;
    const int n = valid ? query_len(lens, q, L) : 0;

    float *base = smem + (size_t)grp * approx_group_floats(Lp, NW);
    float *S_id = base, *Y_id = base + Lp, *acc = base + 2 * (size_t)Lp, *red = acc + (size_t)NW * Lp;

    float si[DPT], li[DPT];
    int ipos[DPT];
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        const bool in = i < n;
        si[m] = in ? preds[(size_t)q * L + i] : -INFINITY;
        li[m] = in ? labels[(size_t)q * L + i] : 0.0f;
        if (i < Lp) {
#pragma unroll
            for (int w = 0; w < NW; ++w) acc[(size_t)w * Lp + i] = 0.0f;
        }
    }
    stage_ideal_order<G, DPT>(S_id, Y_id, n, Lp, t, presort != 0, si, li, ipos);

    // thread t owns ideal positions a = t + m*G
    float sa[DPT], gna[DPT], pia[DPT];
    float part = 0.0f;
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int a = t + m * G;
        sa[m] = a < n ? S_id[a] : 0.0f;
        gna[m] = a < n ? gain_of(Y_id[a]) : 0.0f;
        pia[m] = 0.0f;
        if (a < n) part += gna[m] / log2f((float)a + 2.0f);
    }
    const float idcg = group_sum<G>(part, red, t);
    float *aw = acc + (size_t)wv * Lp;
    const int half = (n - 1) >> 1;

    // ---- pass 1: approximate rank positions
    for (int d = 1; d <= half; ++d) {
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int a = t + m * G;
            if (a < n) {
                int b = a + d; if (b >= n) b -= n;
                float ya, yb;
                if constexpr (SOFT) soft_pair(S_id[b] - sa[m], alpha, ya, yb); else robust_pair(S_id[b] - sa[m], alpha, ya, yb);
                pia[m] += ya;
                aw[b] += yb;                                  // per-wave row, distinct b per lane (no atomics needed)
            }
        }
    }
    if (n > 0 && (n & 1) == 0) {
        const int d = n >> 1;
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int a = t + m * G;
            if (a < d) {
                float ya, yb;
                if constexpr (SOFT) soft_pair(S_id[a + d] - sa[m], alpha, ya, yb); else robust_pair(S_id[a + d] - sa[m], alpha, ya, yb);
                pia[m] += ya;
                aw[a + d] += yb;
            }
        }
    }
    __syncthreads();

    // ---- per-position DCG term and dLoss/d(pi_hat) (for scale 1)
    const float ln2 = 0.6931471805599453f;
    float ca[DPT];
    float dpart = 0.0f;
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int a = t + m * G;
        ca[m] = 0.0f;
        if (a < n && a < top_k) {                            // softrank.py:66-68 (top_k = n for ApproxNDCG)
            float pi = pia[m];
#pragma unroll
            for (int w = 0; w < NW; ++w) pi += acc[(size_t)w * Lp + a];
            pi += 1.0f;                                       // 0.5 (diagonal term) + 0.5 (approxNDCG.py:25); softrank.py:56
            const float lg = log2f(pi + 1.0f);
            dpart += gna[m] / lg;                             // approxNDCG.py:58
            ca[m] = gna[m] / (ln2 * (1.0f + pi) * lg * lg);   // d(-g/log2(1+pi))/d(pi)
        }
    }
    const float dcg = group_sum<G>(dpart, red, t);            // (G == 256: contains the barriers that fence `acc` reuse)
    if constexpr (G == kWave) __syncthreads();
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int a = t + m * G;
        if (a < Lp) {
            Y_id[a] = ca[m];                                  // c by ideal position (labels no longer needed)
#pragma unroll
            for (int w = 0; w < NW; ++w) acc[(size_t)w * Lp + a] = 0.0f;
        }
    }
    __syncthreads();

    // ---- pass 2: gradient.  Entry (a,b): pi_hat_a depends on s_b with +d_ab and on s_a with -d_ab, d_ab = (alpha*y_ab)*(1-y_ab).
    float ga[DPT];
#pragma unroll
    for (int m = 0; m < DPT; ++m) ga[m] = 0.0f;
    auto gpair = [&](int m, int a, int b) {
        float dab, dba;
        if constexpr (SOFT) {                                 // d(0.5*erfc(x/den))/dx = -exp(-(x/den)^2)/(sqrt(pi)*den), symmetric
            const float x = (S_id[b] - sa[m]) * alpha;
            dab = dba = 0.5641895835477563f * alpha * __expf(-x * x);
        } else {
            float ya, yb;
            robust_pair(S_id[b] - sa[m], alpha, ya, yb);
            dab = (alpha * ya) * (1.0f - ya);                 // base/utils.py:78
            dba = (alpha * yb) * (1.0f - yb);
        }
        const float cb = Y_id[b];
        const float flow = cb * dba - ca[m] * dab;            // d loss / d s_a from this pair
        ga[m] += flow;
        aw[b] -= flow;
    };
    for (int d = 1; d <= half; ++d) {
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int a = t + m * G;
            if (a < n) { int b = a + d; if (b >= n) b -= n; gpair(m, a, b); }
        }
    }
    if (n > 0 && (n & 1) == 0) {
        const int d = n >> 1;
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int a = t + m * G;
            if (a < d) gpair(m, a, a + d);
        }
    }
    __syncthreads();

    const float inv_idcg = 1.0f / idcg;
    const float scale = couple_batch ? 1.0f : inv_idcg;
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int a = t + m * G;
        if (a < n) {
            float tot = ga[m];
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += acc[(size_t)w * Lp + a];
            S_id[a] = tot * scale;                            // own index only
        }
    }
    __syncthreads();
    if (valid) {
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int i = t + m * G;
            if (i < L) grad[(size_t)q * L + i] = i < n ? S_id[ipos[m]] : 0.0f;
        }
        if (t == 0) {
            if constexpr (SOFT) { dcg_q[q] = -(dcg * inv_idcg); }         // per-query loss
            else { dcg_q[q] = dcg; inv_idcg_q[q] = inv_idcg; }
        }
    }
}

// =====================================================================================================================
// ApproxNDCG "ring" kernel (list lengths up to 512): ONE wavefront per query, both O(L^2) passes out of registers — the scheme of
// lambdarank_ring_kernel (pairwise.hip).  Lane a owns documents a, a+64, ... in INPUT order (nothing in the loss depends on the ideal
// ORDER, only the IDCG does: a value-only sort of the labels, or the labels as they are under `presort`); every own record {s} / {s, c}
// stays put, every slot has two travelling copies {s, acc} / {s, c, acc} (the records 1..16 and 17..32 lanes ahead, one packed
// instruction stream for both), rotated one lane per step with v_mov_b32_dpp wave_rol:1.  The partner's share of a pair accumulates in the
// travelling record instead of an LDS read-modify-write, and the partner's score / coefficient arrive by rotation instead of LDS reads:
// per pair and pass ~10 / ~15 VALU slots against ~23 / ~30 of approxndcg_kernel.  Padding slots carry s = -1e30, c = 0: e = 0, y in
// {0, 1}, y(1-y) = 0 — every pair with a padding record contributes exactly 0 to real documents (scores are assumed far above -1e30).
template <int DPT, int PASS>
__device__ __forceinline__ void approx_ring(const float (&s)[DPT], const float (&c)[DPT], float c2, float alpha, int lane, float (&out)[DPT]) {
    f32x2 so2[DPT], co2[DPT], acc2[DPT];                          // own records {v, v}; own accumulators of the two copies
    f32x2 Ts[DPT], Tc[DPT], Ta[DPT];                              // travelling {copy A, copy B}
    const int ahead16 = (lane + 16) & 63;
#pragma unroll
    for (int k = 0; k < DPT; ++k) {
        so2[k] = f32x2{s[k], s[k]};
        Ts[k] = f32x2{s[k], __shfl(s[k], ahead16, 64)};
        acc2[k] = f32x2{0.f, 0.f}; Ta[k] = f32x2{0.f, 0.f};
        if constexpr (PASS == 2) { co2[k] = f32x2{c[k], c[k]}; Tc[k] = f32x2{c[k], __shfl(c[k], ahead16, 64)}; }
    }
    const f32x2 c22 = {c2, c2}, one2 = {1.0f, 1.0f}, al2 = {alpha, alpha};
    auto pair2 = [&](int k, int t, f32x2 mask, bool use_mask) __attribute__((always_inline)) {
        // Robust_Sigmoid (base/utils.py:57-95) of +-alpha*delta from ONE exponential, as robust_pair() above: r = 1/(1+e), sm = e/(1+e)
        const f32x2 dl = pk_sub(Ts[t], so2[k]);                   // delta = s_b - s_a
        const f32x2 x = dl * c22;                                 // alpha*log2(e) folded
        const f32x2 e = {__builtin_amdgcn_exp2f(-fabsf(x.x)), __builtin_amdgcn_exp2f(-fabsf(x.y))};
        const f32x2 dd = one2 + e;      // compiler-emitted: an inline-asm reader right behind v_exp_f32 would miss the trans-use wait state
        f32x2 r = {__builtin_amdgcn_rcpf(dd.x), __builtin_amdgcn_rcpf(dd.y)};
        r = __builtin_elementwise_fma(r, __builtin_elementwise_fma(-dd, r, one2), r);
        const f32x2 sm = e * r;
        f32x2 ya, yb;                                             // delta == 0: e = 1, r = sm = 0.5 on its own
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool pos = dl[h] > 0.0f;
            ya[h] = pos ? r[h] : sm[h];
            yb[h] = pos ? sm[h] : r[h];
        }
        if constexpr (PASS == 1) {
            if (use_mask) { ya = ya * mask; yb = yb * mask; }
            acc2[k] = pk_add(acc2[k], ya);                        // b's contribution to pi_hat_a
            Ta[t] = pk_add(Ta[t], yb);                            // a's contribution to pi_hat_b
        } else {
            const f32x2 dab = (ya * al2) * pk_sub(one2, ya), dba = (yb * al2) * pk_sub(one2, yb);      // base/utils.py:78
            f32x2 flow = __builtin_elementwise_fma(-co2[k], dab, Tc[t] * dba);                         // d loss / d s_a from this pair
            if (use_mask) flow = flow * mask;
            acc2[k] = pk_add(acc2[k], flow);
            Ta[t] = pk_sub(Ta[t], flow);
        }
    };
    auto rotate = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < DPT; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Ts[t][h] = dpp_rol1(Ts[t][h]); Ta[t][h] = dpp_rol1(Ta[t][h]);
                if constexpr (PASS == 2) Tc[t][h] = dpp_rol1(Tc[t][h]);
            }
    };
    // offset 0: pairs inside a lane (travelling slot t > own slot k), copy A only
#pragma unroll
    for (int k = 0; k < DPT; ++k)
#pragma unroll
        for (int t = k + 1; t < DPT; ++t) pair2(k, t, f32x2{1.0f, 0.0f}, true);
    // steps 1..15: lane offsets r (copy A) and r + 16 (copy B)
    for (int r = 1; r < 16; ++r) {
        rotate();
#pragma unroll
        for (int k = 0; k < DPT; ++k)
#pragma unroll
            for (int t = 0; t < DPT; ++t) pair2(k, t, one2, false);
    }
    // step 16: offset 16 (A) and the half step 32 (B), where lanes a and a+32 see each other from both ends — the lower half keeps them
    {
        rotate();
        const float lm = lane < 32 ? 1.0f : 0.0f;
#pragma unroll
        for (int k = 0; k < DPT; ++k)
#pragma unroll
            for (int t = 0; t < DPT; ++t) pair2(k, t, f32x2{1.0f, lm}, true);
    }
    // the travelling accumulators sit 16 (copy A) / 32 (copy B) lanes behind their owners
    const int behind16 = (lane - 16) & 63;
#pragma unroll
    for (int k = 0; k < DPT; ++k) out[k] = (acc2[k].x + acc2[k].y) + (__shfl(Ta[k].x, behind16, 64) + __shfl(Ta[k].y, lane ^ 32, 64));
}

constexpr int approx_ring_sort_width(int dpt) { int e = 1; while (e < dpt) e *= 2; return e; }

template <int DPT>
__global__ void __launch_bounds__(kBlock)
approxndcg_ring_kernel(const float *__restrict__ preds, const float *__restrict__ labels, const int32_t *__restrict__ lens, int B, int L,
                       float alpha, int presort, int couple_batch, float *__restrict__ dcg_q, float *__restrict__ inv_idcg_q,
                       float *__restrict__ grad) {
    constexpr int E = approx_ring_sort_width(DPT), NS = 64 * E;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int q = blockIdx.x * (kBlock / kWave) + wv;
    const bool valid = q < B;
    const int n = __builtin_amdgcn_readfirstlane(valid ? query_len(lens, q, L) : 0);
    float *lab = smem + (size_t)wv * NS;                          // label staging of the ideal sort (presort == 0 only)

    float si[DPT], gi[DPT], li[DPT];
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = lane + 64 * m;
        const bool in = i < n;
        si[m] = in ? preds[(size_t)q * L + i] : -1e30f;
        li[m] = in ? labels[(size_t)q * L + i] : -INFINITY;
        gi[m] = in ? gain_of(li[m]) : 0.0f;
    }
    // IDCG (approxNDCG.py:52-53): gains of the labels in descending order over 1/log2(position + 2)
    float part = 0.0f;
    if (presort) {
#pragma unroll
        for (int m = 0; m < DPT; ++m) part = fmaf(gi[m], inv_log2_pos(lane + 64 * m), part);
    } else {
#pragma unroll
        for (int m = 0; m < E; ++m) lab[lane + 64 * m] = m < DPT ? li[m < DPT ? m : 0] : -INFINITY;
        wave_lds_sync();
        float v[E];
#pragma unroll
        for (int r = 0; r < E; ++r) v[r] = lab[lane * E + r];
        wave_sort_desc<E>(v, lane);                               // equal labels are interchangeable: no tie handling
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const int p = lane * E + r;
            part = fmaf(p < n ? gain_of(v[r]) : 0.0f, inv_log2_pos(p), part);
        }
    }
    const float idcg = wave_sum_dpp(part);

    const float c2 = alpha * 1.4426950408889634f;
    float pia[DPT], ca[DPT], tot[DPT];
    approx_ring<DPT, 1>(si, si, c2, alpha, lane, pia);
    const float ln2 = 0.6931471805599453f;
    float dpart = 0.0f;
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const bool in = lane + 64 * m < n;
        const float pi = pia[m] + 1.0f;                           // 0.5 (diagonal term) + 0.5 (approxNDCG.py:25)
        const float lg = log2f(pi + 1.0f);
        dpart += in ? gi[m] / lg : 0.0f;                          // approxNDCG.py:58
        ca[m] = in ? gi[m] / (ln2 * (1.0f + pi) * lg * lg) : 0.0f;   // d(-g/log2(1+pi))/d(pi)
    }
    const float dcg = wave_sum_dpp(dpart);
    approx_ring<DPT, 2>(si, ca, c2, alpha, lane, tot);
    const float inv_idcg = 1.0f / idcg;
    const float scale = couple_batch ? 1.0f : inv_idcg;
    if (valid) {
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int i = lane + 64 * m;
            if (i < L) grad[(size_t)q * L + i] = i < n ? tot[m] * scale : 0.0f;
        }
        if (lane == 0) { dcg_q[q] = dcg; inv_idcg_q[q] = inv_idcg; }
    }
}

static int approx_ring_enabled() {                   // PTR_APPROX_RING=0 selects the LDS kernel (A/B measurements, tests); read per call
    const char *e = getenv("PTR_APPROX_RING");
    return !(e && e[0] == '0');
}

// One workgroup: S = sum 1/IDCG, loss, the factor kernel 3 applies.  out_scale[0] = applied factor, out_scale[1] = local S.
__global__ void __launch_bounds__(kBlock)
approx_finish_kernel(const float *__restrict__ dcg_q, const float *__restrict__ inv_q, int B, int couple_batch,
                     float scale_override, float *__restrict__ loss_out, float *__restrict__ scale_ws) {
    __shared__ float red[4];
    float sd = 0.0f, ss = 0.0f, sn = 0.0f;
    for (int i = threadIdx.x; i < B; i += kBlock) { const float d = dcg_q[i], v = inv_q[i]; sd += d; ss += v; sn += d * v; }
    const float D = group_sum<kBlock>(sd, red, threadIdx.x);
    const float S = group_sum<kBlock>(ss, red, threadIdx.x);
    const float N = group_sum<kBlock>(sn, red, threadIdx.x);
    if (threadIdx.x == 0) {
        const float f = couple_batch ? (scale_override > 0.0f ? scale_override : S) : 1.0f;
        loss_out[0] = couple_batch ? -(D * f) : -N;
        scale_ws[0] = f;
        scale_ws[1] = S;
    }
}

__global__ void __launch_bounds__(kBlock) scale_inplace_kernel(float *__restrict__ x, size_t n, const float *__restrict__ f) {
    const float s = f[0];
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) x[i] *= s;
}

}  // namespace ptr

extern "C" int ptr_approxndcg_fwd_bwd(const float *preds, const float *labels, const int32_t *lens, int B, int L, float alpha,
                                      int presort, int couple_batch, float grad_scale_override, float *loss_out, float *dcg_q,
                                      float *inv_idcg_q, float *scale_out, float *grad, void *stream) {
    using namespace ptr;
    const char *who = "ptr_approxndcg_fwd_bwd";
    if (int rc = check_batch(preds, labels, B, L, who)) return rc;
    if (!loss_out || !scale_out || (B > 0 && (!dcg_q || !inv_idcg_q || !grad))) { set_error("%s: NULL output pointer", who); return PTR_ERR_INVALID_ARG; }
    if (!(alpha > 0.0f)) { set_error("%s: alpha must be > 0 (got %g)", who, (double)alpha); return PTR_ERR_INVALID_ARG; }
    hipStream_t st = as_stream(stream);
    if (B > 0 && L <= 512 && approx_ring_enabled()) {
        auto go = [&](auto kern, int dpt) -> int {
            constexpr int QPB = kBlock / kWave;
            const size_t lds = (size_t)QPB * 64 * approx_ring_sort_width(dpt) * sizeof(float);
            hipLaunchKernelGGL(kern, dim3((B + QPB - 1) / QPB), dim3(kBlock), lds, st, preds, labels, lens, B, L, alpha, presort, couple_batch,
                               dcg_q, inv_idcg_q, grad);
            return check_hip(hipGetLastError(), who);
        };
        const int rc = L <= 64 ? go(approxndcg_ring_kernel<1>, 1) : L <= 128 ? go(approxndcg_ring_kernel<2>, 2)
                     : L <= 192 ? go(approxndcg_ring_kernel<3>, 3) : L <= 256 ? go(approxndcg_ring_kernel<4>, 4)
                     : L <= 384 ? go(approxndcg_ring_kernel<6>, 6) : go(approxndcg_ring_kernel<8>, 8);
        if (rc) return rc;
    } else if (B > 0) {
        const int Lp = round_up(L, 4);
        int rc = dispatch_tiling(L, [&]<int G, int DPT>() -> int {
            constexpr int QPB = kBlock / G, NW = G / kWave;
            auto kern = approxndcg_kernel<G, DPT, false>;
            const size_t lds = QPB * approx_group_floats(Lp, NW) * sizeof(float);
            if (int e = allow_lds(kern, lds)) return e;
            hipLaunchKernelGGL(kern, dim3((B + QPB - 1) / QPB), dim3(kBlock), lds, st, preds, labels, lens, B, L, Lp, alpha, presort,
                               couple_batch, PTR_MAX_LIST_LEN, dcg_q, inv_idcg_q, grad);
            return check_hip(hipGetLastError(), who);
        });
        if (rc) return rc;
    }
    hipLaunchKernelGGL(approx_finish_kernel, dim3(1), dim3(kBlock), 0, st, dcg_q, inv_idcg_q, B, couple_batch, grad_scale_override,
                       loss_out, scale_out);
    if (int rc = check_hip(hipGetLastError(), who)) return rc;
    if (couple_batch && B > 0 && grad_scale_override != 1.0f) {
        const size_t nel = (size_t)B * L;
        const int blocks = (int)((nel + kBlock * 4 - 1) / (kBlock * 4));
        hipLaunchKernelGGL(scale_inplace_kernel, dim3(blocks < 2048 ? (blocks < 1 ? 1 : blocks) : 2048), dim3(kBlock), 0, st, grad, nel,
                           scale_out);
        if (int rc = check_hip(hipGetLastError(), who)) return rc;
    }
    return 0;
}

extern "C" int ptr_softrank_fwd_bwd(const float *preds, const float *labels, const int32_t *lens, int B, int L, float delta,
                                    int top_k, float *loss_out, float *loss_q, float *grad, void *stream) {
    using namespace ptr;
    const char *who = "ptr_softrank_fwd_bwd";
    if (int rc = check_batch(preds, labels, B, L, who)) return rc;
    if (B > 0 && (!loss_q || !grad)) { set_error("%s: NULL output pointer", who); return PTR_ERR_INVALID_ARG; }
    if (!(delta > 0.0f)) { set_error("%s: delta must be > 0 (got %g)", who, (double)delta); return PTR_ERR_INVALID_ARG; }
    hipStream_t st = as_stream(stream);
    if (B > 0) {
        const float var = 2.0f * (delta * delta);             // softrank.py:52, in fp32 like the reference's tensor arithmetic
        const float inv_den = 1.0f / sqrtf(2.0f * var);       // :54
        const int Lp = round_up(L, 4);
        int rc = dispatch_tiling(L, [&]<int G, int DPT>() -> int {
            constexpr int QPB = kBlock / G, NW = G / kWave;
            auto kern = approxndcg_kernel<G, DPT, true>;
            const size_t lds = QPB * approx_group_floats(Lp, NW) * sizeof(float);
            if (int e = allow_lds(kern, lds)) return e;
            hipLaunchKernelGGL(kern, dim3((B + QPB - 1) / QPB), dim3(kBlock), lds, st, preds, labels, lens, B, L, Lp, inv_den, 1, 0,
                               top_k > 0 ? top_k : PTR_MAX_LIST_LEN, loss_q, (float *)nullptr, grad);
            return check_hip(hipGetLastError(), who);
        });
        if (rc) return rc;
    }
    return loss_out ? ptr_sum_f32(loss_q, B, 1.0f, loss_out, stream) : 0;
}
