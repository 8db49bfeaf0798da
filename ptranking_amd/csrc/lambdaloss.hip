// Fused LambdaLoss kernel (NDCG_Loss1 / NDCG_Loss2 / NDCG_Loss2++): ideal-order staging -> rank by score -> normalised gains ->
// the k x k block of (winner, loser) pairs -> gradient scattered back through both permutations.
//
// Reference: ptranking/ltr_adhoc/listwise/lambdaloss.py:36-58 (power weights), :83-132 (loss), epsilon = 1e-8
// (ptranking/ltr_global.py:10).  Bug-compatible details reproduced on purpose (SURVEY.md §7 iii):
//   * the discount table is inverted twice: inv[r] = (1/log2(r+2))^-1, delta_d = |inv[d-1] - inv[d]|,
//     rho_ij = |inv[i] - inv[j]| (not the paper's |1/D - 1/D|);
//   * the mask is `label_i > label_j` AND both predicted ranks < k (a full-matrix mask: every unordered pair with
//     different grades contributes exactly one term, seen from its winner);
//   * clamp(min=1e-8) before and after the power, log2 (not ln), score differences clamped to +-1e8, NaN -> 0.
#include <utility>
#include "ptr_device.h"
#include "ptr_dropout.h"          // f32x4

namespace ptr {

// LDS per group (floats): pk float4[Lp] | S_id[Lp] | Y_id[Lp] | extra[max(NW-2,0)][Lp] | red[4].  Once the per-rank tile pk
// is built, S_id / Y_id are dead and become partner-gradient accumulator rows 0 / 1 (rows 2.. live in `extra`).
__host__ __device__ constexpr size_t lambdaloss_group_floats(int Lp, int NW) {
    return (size_t)Lp * (4 + 2 + (NW > 2 ? NW - 2 : 0)) + 4;
}

template <int G, int DPT>
__global__ void __launch_bounds__(kBlock)
lambdaloss_kernel(const float *__restrict__ preds, const float *__restrict__ labels, const int32_t *__restrict__ lens, int B,
                  int L, int Lp, int k, float sigma, float mu, int loss_type, int presort, float *__restrict__ loss_q,
                  float *__restrict__ grad) {
    constexpr int QPB = kBlock / G, NW = G / kWave;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, grp = tid / G, t = tid % G, wv = t >> 6;
    const int q = blockIdx.x * QPB + grp;
    const bool valid = q < B;
    const int n = valid ? query_len(lens, q, L) : 0;

    float *base = smem + (size_t)grp * lambdaloss_group_floats(Lp, NW);
    float4 *pk = reinterpret_cast<float4 *>(base);   // by predicted rank: {score, normalised gain, inv[rank], label}
    float *S_id = base + 4 * (size_t)Lp;              // scores by ideal position; later: accumulator row 0 / grad by rank
    float *Y_id = S_id + Lp;                          // labels by ideal position; later: accumulator row 1 / grad by ideal pos
    float *extra = Y_id + Lp;                         // accumulator rows 2..NW-1
    float *red = extra + (size_t)(NW > 2 ? NW - 2 : 0) * Lp;
    auto acc_row = [&](int w) -> float * { return w == 0 ? S_id : (w == 1 ? Y_id : extra + (size_t)(w - 2) * Lp); };

    float si[DPT], li[DPT];
    int ipos[DPT];
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        const bool in = i < n;
        si[m] = in ? preds[(size_t)q * L + i] : -INFINITY;
        li[m] = in ? labels[(size_t)q * L + i] : 0.0f;
    }
    stage_ideal_order<G, DPT>(S_id, Y_id, n, Lp, t, presort != 0, si, li, ipos);

    // from here on thread t owns IDEAL POSITIONS p = t + m*G
    float sp[DPT], yp[DPT];
    int rk[DPT];
    float part = 0.0f;
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int p = t + m * G;
        sp[m] = p < Lp ? S_id[p] : -INFINITY;
        yp[m] = p < Lp ? Y_id[p] : 0.0f;
        if (p < n) part += gain_of(yp[m]) / log2f((float)p + 2.0f);     // adhoc_metric.py:205-217 on the ideal ranking
    }
    // lambdaloss.py:89 (pk, filled below, is the scratch of either form: 4*Lp floats >= the 64*DPT sorted keys of the one-wave form)
    if constexpr (G == kWave) count_ranks_wave<DPT>(S_id, reinterpret_cast<float *>(pk), n, Lp, t, sp, rk);
    else count_ranks_fast<G, DPT>(S_id, reinterpret_cast<int *>(pk), n, t, sp, rk);
    const float idcg = group_sum<G>(part, red, t);
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int p = t + m * G;
        if (p < n) {
            float *dst = reinterpret_cast<float *>(pk + rk[m]);
            dst[0] = sp[m];
            dst[1] = gain_of(yp[m]) / idcg;                               // lambdaloss.py:106
            dst[3] = yp[m];
            const float disc = 1.0f / log2f((float)p + 2.0f);             // lambdaloss.py:94
            reinterpret_cast<float *>(pk + p)[2] = 1.0f / disc;           // torch.pow(discounts, -1.) (:41,49)
        }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int p = t + m * G;
        if (p < Lp) {
#pragma unroll
            for (int w = 0; w < NW; ++w) acc_row(w)[p] = 0.0f;
        }
    }
    __syncthreads();

    const int kk = k < n ? (k < 0 ? 0 : k) : n;
    const float eps = 1e-8f, inv_ln2 = 1.4426950408889634f, log2_eps = -26.575424759098897f;   // log2(1e-8)
    float4 me[DPT];
    float ga[DPT];
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int a = t + m * G;
        me[m] = a < kk ? pk[a] : make_float4(0.f, 0.f, 0.f, 0.f);
        ga[m] = 0.0f;
    }
    float lacc = 0.0f;
    float *gw = acc_row(wv);

    auto pair = [&](int m, int a, int d, float delta_fwd, float delta_wrap) {
        int b = a + d;
        const bool fwd = b < kk;
        if (!fwd) b -= kk;
        const float4 o = pk[b];
        if (loss_type == PTR_LAMBDALOSS_NDCG_LOSS1) {
            // NDCG_Loss1 (lambdaloss.py:33-34,108-109,130): no label mask; BOTH ordered entries (a,b) and (b,a) of the k x k
            // block count, entry (i,j) carrying the column's weight w_j = G_j / D_j = G_j * inv[j].
            float df = fminf(fmaxf(me[m].x - o.x, -1e8f), 1e8f);
            if (df != df) df = 0.0f;
            const float x = sigma * df;
            const float wb = o.y * o.z, wa = me[m].y * me[m].z;
            const float pab = 1.0f / (1.0f + expf(-x)), pba = 1.0f / (1.0f + expf(x));
            const float lab = pab >= eps ? -log1pf(expf(-x)) * inv_ln2 : log2_eps;
            const float lba = pba >= eps ? -log1pf(expf(x)) * inv_ln2 : log2_eps;
            const float zab = wb * lab, zba = wa * lba;
            const bool okab = zab >= log2_eps, okba = zba >= log2_eps;
            lacc -= (okab ? zab : log2_eps) + (okba ? zba : log2_eps);
            float g = 0.0f;                                               // d loss / d s_a; s_b gets -g
            if (pab >= eps && okab) g -= (wb * sigma * (1.0f - pab)) * inv_ln2;
            if (pba >= eps && okba) g += (wa * sigma * (1.0f - pba)) * inv_ln2;
            ga[m] += g;
            gw[b] -= g;
            return;
        }
        const bool a_wins = me[m].w > o.w, b_wins = o.w > me[m].w;        // lambdaloss.py:127-128
        if (!(a_wins || b_wins)) return;
        const float delta = fwd ? delta_fwd : delta_wrap;                  // |rank distance| = d or kk - d
        const float absG = fabsf(me[m].y - o.y);
        float w = delta * absG;                                            // NDCG_Loss2, lambdaloss.py:44
        if (loss_type == PTR_LAMBDALOSS_NDCG_LOSS2PP) w = (fabsf(me[m].z - o.z) + mu * delta) * absG;   // :57
        float df = a_wins ? me[m].x - o.x : o.x - me[m].x;                 // s_winner - s_loser
        df = fminf(fmaxf(df, -1e8f), 1e8f);
        if (df != df) df = 0.0f;                                           // lambdaloss.py:115-116
        const float x = sigma * df;
        const float p0 = 1.0f / (1.0f + expf(-x));
        // log2(clamp(p, eps)): evaluated as -log1p(e^-x)/ln2 so that p close to 1 keeps full relative accuracy (the
        // hardware log has an absolute error floor near 1 that would bias a sum over ~1e5 pairs).
        const float lp = p0 >= eps ? -log1pf(expf(-x)) * inv_ln2 : log2_eps;
        // log2(clamp(p^w, eps)) = max(w*log2(p), log2(eps)); the reference's fp32 rounding of p^w is unbiased noise
        const float z = w * lp;
        const bool wp_ok = z >= log2_eps;
        lacc -= wp_ok ? z : log2_eps;                                      // lambdaloss.py:118-119,132
        float g = 0.0f;
        if (p0 >= eps && wp_ok) g = -(w * sigma * (1.0f - p0)) * inv_ln2;  // d/ds_winner; loser gets -g
        const float ga_ = a_wins ? g : -g;
        ga[m] += ga_;
        gw[b] -= ga_;                                                      // per-wave row, distinct b per lane: race-free
    };

    if (loss_type == PTR_LAMBDALOSS_NDCG_LOSS1) {
        // diagonal entries: sigmoid(0)^w_j = 2^-w_j, constant in the scores
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int a = t + m * G;
            if (a < kk) lacc += fminf(me[m].y * me[m].z, -log2_eps);
        }
    }
    const int half = (kk - 1) >> 1;
    for (int d = 1; d <= half; ++d) {
        const float dfw = fabsf(reinterpret_cast<const float *>(pk + (d - 1))[2] - reinterpret_cast<const float *>(pk + d)[2]);
        const int dw = kk - d;
        const float dwr = fabsf(reinterpret_cast<const float *>(pk + (dw - 1))[2] - reinterpret_cast<const float *>(pk + dw)[2]);
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int a = t + m * G;
            if (a < kk) pair(m, a, d, dfw, dwr);
        }
    }
    if (kk > 0 && (kk & 1) == 0) {
        const int d = kk >> 1;
        const float dfw = fabsf(reinterpret_cast<const float *>(pk + (d - 1))[2] - reinterpret_cast<const float *>(pk + d)[2]);
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int a = t + m * G;
            if (a < d) pair(m, a, d, dfw, dfw);
        }
    }
    __syncthreads();

    // gradient by predicted rank -> by ideal position -> by original index
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int a = t + m * G;
        if (a < n) {
            float tot = ga[m];
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += acc_row(w)[a];
            S_id[a] = tot;                               // own index only: safe to overwrite row 0 in place
        }
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int p = t + m * G;
        if (p < n) Y_id[p] = S_id[rk[m]];
    }
    __syncthreads();
    const float loss = group_sum<G>(lacc, red, t);
    if (valid) {
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int i = t + m * G;
            if (i < L) grad[(size_t)q * L + i] = i < n ? Y_id[ipos[m]] : 0.0f;
        }
        if (t == 0) loss_q[q] = loss;
    }
}

// r5: NDCG_Loss2 / Loss2++ with a SMALL cut-off (kk = min(k, n) <= 11: the default k = 5 of BASELINE config 5) on presorted labels.  Only
// the kk best-scored documents enter the loss (lambdaloss.py:127-128: both predicted ranks < k), so no sort: kk rounds of a wavefront
// arg-max (score descending, document index ascending on ties — the order count_ranks gives) pick them, their kk (kk - 1) / 2 pairs are
// evaluated ONE PAIR PER LANE with the arithmetic of lambdaloss_kernel's pair body, and the gradient row is zeros plus kk patched
// entries.  One wavefront per query, everything in registers, float4 loads / stores; the IDCG is the only full-length pass.
//
// Persistent wavefronts (a grid-stride walk over the queries): what depends on the lane alone — the discount of its positions, the
// double-inverted discount inv[lane], the (a, b) pair of the lane and that pair's position weights — is computed once per wavefront
// (the pair table again only when kk changes, i.e. for lists shorter than k).  Up to 256 documents (V = 1) each lane keeps its four
// documents ordered, so a selection round is one DPP max ladder, a ballot and a pop in the winning lane (the lowest lane among equal
// heads holds the lowest index); the pair body runs on the transcendental pipe while every pair of the query has |sigma ds| <= 80 (the
// RankNet kernel's policy: pairwise.hip), the library functions otherwise.
template <int V>
__global__ void __launch_bounds__(kBlock)
lambdaloss_topk_kernel(const float *__restrict__ preds, const float *__restrict__ labels, const int32_t *__restrict__ lens, int B, int L,
                       int k, float sigma, float mu, int loss_type, float *__restrict__ loss_q, float *__restrict__ grad) {
    constexpr int E = 4 * V;
    constexpr int kDead = 0x7fffffff;
    const int lane = threadIdx.x & 63;
    const int wave0 = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6));
    const int nwaves = gridDim.x * (kBlock / kWave), L4 = L >> 2;
    const float eps = 1e-8f, inv_ln2 = 1.4426950408889634f, log2_eps = -26.575424759098897f;
    float disc[E];
#pragma unroll
    for (int m = 0; m < V; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) disc[4 * m + e] = inv_log2_pos(4 * (lane + 64 * m) + e);
    const float rinv = 1.0f / (1.0f / log2f((float)lane + 2.0f));                          // inv[rank], rank = lane (lambdaloss.py:41,49,94)
    int kk_tab = -1, aa = 0, b = 0;                                                        // the lane's pair (aa < b < kk) for kk == kk_tab
    bool has = false;
    float delta = 0.0f, dpos = 0.0f;
    int plo[11];                                                                           // record lane r: byte address (ds_bpermute) of the lane
#pragma unroll                                                                             // that holds pair (r, o); lane 63 (no pair: zero) otherwise
    for (int o = 0; o < 11; ++o) plo[o] = 63 << 2;
    for (int q = wave0; q < B; q += nwaves) {                                              // waves are independent
        const int n = query_len(lens, q, L);
        const f32x4 *ps = reinterpret_cast<const f32x4 *>(preds + (size_t)q * L), *py = reinterpret_cast<const f32x4 *>(labels + (size_t)q * L);
        float s[E], y[E];
        float part = 0.0f;
        const bool full = n == 256 * V;
#pragma unroll
        for (int m = 0; m < V; ++m) {
            const int c = lane + 64 * m;
            const f32x4 a4 = ps[c < L4 ? c : 0], b4 = py[c < L4 ? c : 0];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * c + e;
                const bool in = full || i < n;                                             // full lists (n == L == 256 V): no masks
                s[4 * m + e] = in ? a4[e] : -INFINITY;
                y[4 * m + e] = in ? b4[e] : 0.0f;
                part += in ? (__builtin_amdgcn_exp2f(b4[e]) - 1.0f) * disc[4 * m + e] : 0.0f;   // IDCG of the (presorted = ideal) label order,
                // adhoc_metric.py:205-217 (2^l - 1 on v_exp_f32: exact for the integer grades, 1 ulp otherwise)
            }
        }
        const float idcg = wave_sum_dpp(part);
        const int kk = k < n ? (k < 0 ? 0 : k) : n;
        if (kk != kk_tab) {                                                                // uniform: kk depends on the query alone
            kk_tab = kk;
            const int npairs = kk * (kk - 1) / 2;
            int a = 0, rem = lane;
            for (int it = 0; it < 11; ++it) { const int row = kk - 1 - a; if (rem >= row && row > 0) { rem -= row; ++a; } }
            has = lane < npairs;
            b = has ? a + 1 + rem : 0;
            aa = has ? a : 0;
            const float ia = __shfl(rinv, aa, 64), ib = __shfl(rinv, b, 64);
            const int dist = b - aa;
            const float id0 = __shfl(rinv, dist > 0 ? dist - 1 : 0, 64), id1 = __shfl(rinv, dist, 64);
            delta = fabsf(id0 - id1);                                                      // :44
            dpos = fabsf(ia - ib);                                                         // :57
#pragma unroll
            for (int o = 0; o < 11; ++o) {                                                 // pair (a, b) sits in lane a (kk - 1) - a (a - 1) / 2 + b - a - 1
                const int a_ = lane < o ? lane : o, b_ = lane < o ? o : lane;
                const int pl = a_ * (kk - 1) - ((a_ * (a_ - 1)) >> 1) + (b_ - a_ - 1);
                plo[o] = (o != lane && lane < kk && o < kk ? pl : 63) << 2;
            }
        }
        // ---- the kk best documents, in rank order: record r lives in lane r
        float rs = 0.0f, ry = 0.0f;
        int ri = -1;
        if constexpr (V == 1) {
            // a lane's four documents stay where they are; a document that cannot win any more (padding, a NaN score — the reference sorts
            // those last —, taken in an earlier round) is a quiet NaN, which v_max ignores and no comparison matches
            float hs[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) hs[e] = (full || 4 * lane + e < n) ? __builtin_canonicalizef(s[e]) : __builtin_nanf("");
            float hm;                                                                      // the lane's best score and its register
            int he;
            auto head = [&]() __attribute__((always_inline)) {
                float m3;
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m3) : "v"(hs[0]), "v"(hs[1]), "v"(hs[2]));
                asm("v_max_f32 %0, %1, %2" : "=v"(hm) : "v"(m3), "v"(hs[3]));
                he = hs[0] == hm ? 0 : (hs[1] == hm ? 1 : (hs[2] == hm ? 2 : 3));         // equal scores: the lowest index
            };
            head();
            int rsb = 0, ryb = 0;                                                          // records as bits: written lane by lane (v_writelane)
            auto round = [&]<int R>() __attribute__((always_inline)) -> bool {
                const float gmax = wave_max_dpp(hm);                                       // DPP ladder: no LDS crossbar round trips in the selection
                const uint64_t own = __builtin_amdgcn_ballot_w64(hm == gmax);
                if (own == 0) return false;                                                // fewer than kk rankable documents (all NaN)
                const int wl = (int)__builtin_ctzll(own);                                  // the lowest lane holds the lowest index
                const int we = __builtin_amdgcn_readlane(he, wl);
                const int wi = 4 * wl + we;
                const float ysel = we == 0 ? y[0] : (we == 1 ? y[1] : (we == 2 ? y[2] : y[3]));   // scalar conditions
                const int wy = __builtin_amdgcn_readlane(__builtin_bit_cast(int, ysel), wl);
                const int gb = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, gmax));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(rsb) : "s"(gb), "n"(R));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(ryb) : "s"(wy), "n"(R));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(ri) : "s"(wi), "n"(R));
                const bool mine = lane == wl;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (we == e) hs[e] = mine ? __builtin_nanf("") : hs[e];                // scalar condition: one select
                head();
                return true;
            };
            [&]<int... R>(std::integer_sequence<int, R...>) {
                bool go = true;
                ((go = go && R < kk && round.template operator()<R>()), ...);
            }(std::make_integer_sequence<int, 11>{});
            rs = __builtin_bit_cast(float, rsb); ry = __builtin_bit_cast(float, ryb);
        } else {
            for (int r = 0; r < kk; ++r) {
                float best = -INFINITY, blab = 0.0f;
                int bidx = kDead;
#pragma unroll
                for (int m = 0; m < V; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * (lane + 64 * m) + e;
                        const bool live = i < n && s[4 * m + e] == s[4 * m + e];
                        const bool better = live && (bidx == kDead || s[4 * m + e] > best);
                        best = better ? s[4 * m + e] : best; blab = better ? y[4 * m + e] : blab; bidx = better ? i : bidx;
                    }
                const float gmax = wave_max_dpp(bidx == kDead ? -INFINITY : best);
                const int cand = wave_min_i32_dpp((bidx != kDead && best == gmax) ? bidx : kDead);
                if (cand == kDead) break;
                const uint64_t own = __builtin_amdgcn_ballot_w64(bidx == cand);
                const int wl = (int)__builtin_ctzll(own);
                const float wlab = __shfl(blab, wl, 64);
                if (lane == r) { rs = gmax; ry = wlab; ri = cand; }
#pragma unroll
                for (int m = 0; m < V; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (4 * (lane + 64 * m) + e == cand) s[4 * m + e] = __builtin_nanf("");   // taken: never live again
            }
        }
        // ---- pairs (a < b < kk), one per lane; arithmetic of lambdaloss_kernel's pair body
        const float rG = ri >= 0 ? (__builtin_amdgcn_exp2f(ry) - 1.0f) / idcg : 0.0f;      // lambdaloss.py:106
        const float sa = __shfl(rs, aa, 64), sb = __shfl(rs, b, 64), ya = __shfl(ry, aa, 64), yb = __shfl(ry, b, 64);
        const float Ga = __shfl(rG, aa, 64), Gb = __shfl(rG, b, 64);
        float lacc = 0.0f, g_a = 0.0f;
        const bool a_wins = ya > yb, b_wins = yb > ya;                                     // lambdaloss.py:127-128
        const bool act = has && (a_wins || b_wins);
        const float absG = fabsf(Ga - Gb);
        float w = delta * absG;                                                            // NDCG_Loss2, :44
        if (loss_type == PTR_LAMBDALOSS_NDCG_LOSS2PP) w = (dpos + mu * delta) * absG;      // :57
        float df = a_wins ? sa - sb : sb - sa;
        df = fminf(fmaxf(df, -1e8f), 1e8f);
        if (df != df) df = 0.0f;                                                           // :115-116
        const float x = sigma * df;
        float p0, lp;
        if (__any(act && !(fabsf(x) <= 80.0f))) {
            p0 = 1.0f / (1.0f + expf(-x));
            lp = p0 >= eps ? -log1pf(expf(-x)) * inv_ln2 : log2_eps;
        } else {
            // e = exp(-|x|) is a normal float; the larger probability is 1 / (1 + e) by v_rcp + one Newton step, log2 p = min(x, 0) log2(e) -
            // log2(1 + e) with one v_log_f32 (1 ulp: 1e-7 absolute near p = 1)
            const float e = __expf(-fabsf(x));
            const float dd = 1.0f + e;
            float pb = __builtin_amdgcn_rcpf(dd);
            pb = fmaf(pb, fmaf(-dd, pb, 1.0f), pb);
            p0 = x >= 0.0f ? pb : e * pb;
            lp = p0 >= eps ? fminf(x, 0.0f) * inv_ln2 - __builtin_amdgcn_logf(dd) : log2_eps;
        }
        if (act) {
            const float z = w * lp;
            const bool wp_ok = z >= log2_eps;
            lacc = -(wp_ok ? z : log2_eps);                                                // :118-119,132
            float g = 0.0f;
            if (p0 >= eps && wp_ok) g = -(w * sigma * (1.0f - p0)) * inv_ln2;
            g_a = a_wins ? g : -g;                                                         // d loss / d s_a; s_b gets the negative
        }
        const float loss = wave_sum_dpp(lacc);
        // ---- gradient of record r = lane: its kk - 1 pairs, gathered from the pair lanes (pair (a, b) sits in lane a (kk - 1) - a (a - 1) / 2
        //      + b - a - 1)
        float gr = 0.0f;
#pragma unroll
        for (int o = 0; o < 11; ++o) {
            if (o < kk) {                                                                  // uniform
                const float gv = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(plo[o], __builtin_bit_cast(int, g_a)));
                gr += lane < o ? gv : -gv;                                                 // lanes without this pair read lane 63's zero
            }
        }
        // ---- gradient row: zeros + the kk records' entries
        f32x4 o4[V];
#pragma unroll
        for (int m = 0; m < V; ++m) o4[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < kk; ++r) {
            const int idx = __builtin_amdgcn_readlane(ri, r);
            const float gq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gr), r));
            if (idx < 0) break;                                                            // records are filled in order
            const bool mine = lane == ((idx >> 2) & 63);
#pragma unroll
            for (int m = 0; m < V; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if ((idx >> 8) == m && (idx & 3) == e) o4[m][e] = mine ? gq : o4[m][e];   // scalar condition: one select per record
        }
        f32x4 *go = reinterpret_cast<f32x4 *>(grad + (size_t)q * L);
#pragma unroll
        for (int m = 0; m < V; ++m) {
            const int c = lane + 64 * m;
            if (c < L4) go[c] = o4[m];
        }
        if (lane == 0) loss_q[q] = loss;
    }
}

}  // namespace ptr

extern "C" int ptr_lambdaloss_fwd_bwd(const float *preds, const float *labels, const int32_t *lens, int B, int L, int k,
                                      float sigma, float mu, int loss_type, int presort, float *loss_out, float *loss_q,
                                      float *grad, void *stream) {
    using namespace ptr;
    const char *who = "ptr_lambdaloss_fwd_bwd";
    if (int rc = check_batch(preds, labels, B, L, who)) return rc;
    if (B > 0 && (!loss_q || !grad)) { set_error("%s: NULL output pointer", who); return PTR_ERR_INVALID_ARG; }
    if (loss_type != PTR_LAMBDALOSS_NDCG_LOSS1 && loss_type != PTR_LAMBDALOSS_NDCG_LOSS2 && loss_type != PTR_LAMBDALOSS_NDCG_LOSS2PP) {
        set_error("%s: loss_type %d not supported (0 = NDCG_Loss1, 1 = NDCG_Loss2, 2 = NDCG_Loss2++)", who, loss_type);
        return PTR_ERR_INVALID_ARG;
    }
    hipStream_t st = as_stream(stream);
    static const bool topk_off = [] { const char *e = getenv("PTR_LAMBDALOSS_TOPK"); return e && atoi(e) == 0; }();
    const bool topk = !topk_off && presort && k <= 11 && loss_type != PTR_LAMBDALOSS_NDCG_LOSS1 && L % 4 == 0 && L <= 1024 &&
                      ((reinterpret_cast<uintptr_t>(preds) | reinterpret_cast<uintptr_t>(labels) | reinterpret_cast<uintptr_t>(grad)) & 15) == 0;
    if (B > 0 && topk) {
        // small cut-off on presorted labels: wavefront arg-max selection instead of a sort, one pair per lane (lambdaloss_topk_kernel)
        auto go = [&](auto kern) -> int {
            const int blocks = persistent_grid(kern, kBlock, 0, (B + 3) / 4);   // persistent wavefronts: the resident set walks the queries
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(kBlock), 0, st, preds, labels, lens, B, L, k, sigma, mu, loss_type, loss_q, grad);
            return check_hip(hipGetLastError(), who);
        };
        if (int rc = L <= 256 ? go(lambdaloss_topk_kernel<1>) : (L <= 512 ? go(lambdaloss_topk_kernel<2>) : go(lambdaloss_topk_kernel<4>))) return rc;
    } else if (B > 0) {
        // one wavefront per query up to 256 documents (ranks from a register bitonic sort, one accumulator row), four beyond
        int rc = dispatch_wave256_tiling(L, [&]<int G, int DPT>() -> int {
            constexpr int QPB = kBlock / G, NW = G / kWave;
            const int Lp = G == kWave ? kWave * DPT : round_up(L, 4);
            auto kern = lambdaloss_kernel<G, DPT>;
            const size_t lds = QPB * lambdaloss_group_floats(Lp, NW) * sizeof(float);
            if (int e = allow_lds(kern, lds)) return e;
            hipLaunchKernelGGL(kern, dim3((B + QPB - 1) / QPB), dim3(kBlock), lds, st, preds, labels, lens, B, L, Lp, k, sigma, mu,
                               loss_type, presort, loss_q, grad);
            return check_hip(hipGetLastError(), who);
        });
        if (rc) return rc;
    }
    return loss_out ? ptr_sum_f32(loss_q, B, 1.0f, loss_out, stream) : 0;
}
