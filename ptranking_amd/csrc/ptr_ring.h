// The LambdaRank ring kernel (list lengths up to 512, sigma > 0) — shared by its two translation units (their code is large: every list-length class
// is a family of straight-line pair loops):
//   pairwise_ring.hip  lambdarank_ring_kernel<1 / 2 / 4> (lists of up to 256 documents: the north-star kernel)
//   pairwise.hip       lambdarank_ring_kernel<8> (r6: 257..512 documents, one wave per SIMD: 362 registers) next to the LDS kernel that serves longer lists
// (r6 experiment, ptranking_amd/build.py EXTRA_FLAGS: LLVM's max-ilp scheduling strategy removes the loop's s_nop wait states but costs a wave of occupancy — slower.)
// Reference: ptranking/ltr_adhoc/listwise/lambdarank.py:39-56, ptranking/ltr_adhoc/util/lambda_utils.py:5-23, ptranking/metric/metric_utils.py:19-45.
#pragma once
#include <stdlib.h>

#include "ptr_device.h"

namespace ptr {

// =====================================================================================================================
// LambdaRank "ring" kernel (list lengths up to 256, sigma > 0): ONE wavefront per query, the pair loop runs entirely out of
// registers with wavefront shuffles — no LDS access, no address arithmetic, no branch inside the O(L^2) loop.
//
// Lane a owns documents a, a+64, ... (DPT = ceil(L/64) per lane, ring size RS = 64*DPT; slots n..RS-1 hold padding records)
// in INPUT order: the pair body needs every record's D = 1/log2(rank+2), not a sorted arrangement, so the ranks are counted
// (one packed fma-with-clamp per two compares, see below) and nothing is scattered; loads and gradient stores are coalesced.
// Every lane keeps its own records {s, G, D} fixed and owns DPT travelling records {s, G, D, g}; one ring step moves the
// travelling records to the neighbouring lane (v_mov_b32_dpp wave_rol:1) and pairs every own slot with every travelling slot:
// after r steps lane a meets the records of lane a+r, i.e. the circulant schedule of the LDS kernel above, with the partner's
// gradient share accumulated in the travelling g instead of an LDS read-modify-write.  After 32 steps the travelling g of a
// record sits 32 lanes away from its owner and is added back with one shuffle.
//
// Per pair: 16.75 VALU-issue slots incl. 3 transcendentals (the LDS kernel: ~42 + 3 + 3 LDS).  What makes the body that short:
//   * D = 1/log2(rank+2) is strictly decreasing in the rank, so the sign of dD = D_own - D_T tells which of the two is ranked
//     first: no position arithmetic and no wrap-around bookkeeping.  prod = (G_own-G_T)*dD > 0  <=>  the first-ranked document
//     has the larger gain (target 1); wsg = sigma*|dG|*dD is the pair weight carrying the orientation sign, so the own
//     gradient share is wsg*(p - t) and the partner's its negative, for either orientation;
//   * s_first - s_second = |s_own - s_T| (rank order = score order), so x = sigma*|ds| needs no select;
//   * padding needs no mask: s = -1e30 (=> p = 1, q = 0, gradient factor 0), G = -1 (=> target 1, log p = 0) make every pair
//     with a padding record contribute exactly 0 to loss and gradients (scores are assumed to lie far above -1e30);
//   * the loss is accumulated as sum |wsg| * max(log2(.), -100/ln2) and scaled by ln2/sigma once per query.
// Arithmetic per pair is otherwise the reference's (see the header): p = fl(1/(1+e^-x)), q = fl(1-p), BCE's -100 clamp, and a
// gradient that is exactly 0 once p rounds to 1.
// (f32x2 / pk_fma_clamp: ptr_device.h)
// (f32x2 / pk_fma_clamp / pk_sub / pk_add / wave_sum_dpp / inv_log2_pos / wave_lds_sync / dpp_rol1: ptr_device.h)
// Pair loop of the ring kernel for DPT >= 2 documents per lane.  Own records stay put; every slot t has TWO travelling copies packed
// in one register pair, {copy A, copy B} = the records of the lanes r and r + 16 ahead: 16 ring steps (v_mov_b32_dpp wave_rol:1 on
// both copies) cover the lane offsets 1..16 (A) and 17..32 (B), i.e. the circulant half ring, and one packed instruction stream per
// step handles the WHOLE block (own slot k, travelling slot t) at two offsets — so a block is either evaluated or skipped as a unit.
// Z > 0: the last Z slots hold equal-label documents only; the blocks among them (k >= DPT - Z and t >= DPT - Z) are not evaluated
// (compile-time: each Z is its own straight-line loop, selected by a wave-uniform switch).
// Orientation-free pair form: with dDn = D_T - D_own (< 0: own is ranked first) and un = sigma*(G_own-G_T)*dDn,
//   target 1 <=> un < 0;   own gradient += -un * m,  partner's += un * m,   m = sign(dDn) * (t1 ? q : fract(p))
// (fract(p) = p for p in [0.5, 1) and 0 at p == 1: the factor that vanishes once p(1-p) underflows).
// tot[k] = dLoss/ds of document lane + 64k; lacc = sum |un| * max(log2(.), clamp) (scaled by -ln2/sigma by the caller).
template <int DPT, int Z>
__device__ __forceinline__ void ring_pairs(const float (&si)[DPT], const float (&Gi)[DPT], const float (&Di)[DPT], float sigma, float c2,
                                           float kClamp, int lane, float (&tot)[DPT], float &lacc) {
    constexpr int ZB = DPT - Z;                                   // slots ZB..DPT-1 are mutually weight-free
    auto active = [](int k, int t) constexpr { return !(k >= ZB && t >= ZB); };
    f32x2 so2[DPT], go2[DPT], Do2[DPT], ga2[DPT];                 // own records, broadcast pairs {v, v}
    f32x2 Ts[DPT], Tg[DPT], Td[DPT], Ta[DPT];                     // travelling {copy A, copy B} of slot t
    const int ahead16 = (lane + 16) & 63;
#pragma unroll
    for (int k = 0; k < DPT; ++k) {
        const float s_ = si[k], g_ = Gi[k], d_ = Di[k] * sigma;   // un = sigma*dG*dD: sigma rides on D (signs unchanged)
        so2[k] = f32x2{s_, s_}; go2[k] = f32x2{g_, g_}; Do2[k] = f32x2{d_, d_};
        ga2[k] = f32x2{0.f, 0.f};
        Ts[k] = f32x2{s_, __shfl(s_, ahead16, 64)}; Tg[k] = f32x2{g_, __shfl(g_, ahead16, 64)}; Td[k] = f32x2{d_, __shfl(d_, ahead16, 64)};
        Ta[k] = f32x2{0.f, 0.f};
    }
    const f32x2 c22 = {c2, c2}, one2 = {1.0f, 1.0f}, half2 = {0.5f, 0.5f};
    auto pair2 = [&](int k, int t, f32x2 mask, bool use_mask) {
        const f32x2 x = (so2[k] - Ts[t]) * c22;
        const f32x2 e = {__builtin_amdgcn_exp2f(-fabsf(x.x)), __builtin_amdgcn_exp2f(-fabsf(x.y))};
        const f32x2 dd = one2 + e;
        f32x2 p = {__builtin_amdgcn_rcpf(dd.x), __builtin_amdgcn_rcpf(dd.y)};
        p = __builtin_elementwise_fma(p, __builtin_elementwise_fma(-dd, p, one2), p);
        const f32x2 dDn = Td[t] - Do2[k];                        // D carries sigma
        f32x2 un = (go2[k] - Tg[t]) * dDn;
        if (use_mask) un = un * mask;
        // a = probability of the target's outcome (t1 ? p : 1-p), r = 1 - a the gradient factor, without compare/select:
        // with h = p - 1/2 (exact for p in [0.5, 1]) and c = copysign(h, un):  a = 1/2 - c,  r = 1/2 + c  (both exact:
        // un < 0 <=> target 1 gives a = p, r = 1-p; otherwise a = 1-p, r = p).  fract() sends r = 1 (p has rounded to 1 on a
        // target-0 pair) to 0, where the reference's p(1-p) factor vanishes.
        const f32x2 hh = pk_sub(p, half2);
        f32x2 c;
#pragma unroll
        for (int h = 0; h < 2; ++h) c[h] = __builtin_copysignf(hh[h], un[h]);
        const f32x2 a = pk_sub(half2, c), r = pk_add(half2, c);
        f32x2 m;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            lacc = fmaf(fabsf(un[h]), fmaxf(__builtin_amdgcn_logf(a[h]), kClamp), lacc);
            m[h] = __builtin_copysignf(__builtin_amdgcn_fractf(r[h]), dDn[h]);      // one v_bfi_b32
        }
        ga2[k] = __builtin_elementwise_fma(-un, m, ga2[k]);
        Ta[t] = __builtin_elementwise_fma(un, m, Ta[t]);
    };
    auto rotate = [&]() {
#pragma unroll
        for (int t = 0; t < DPT; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Ts[t][h] = dpp_rol1(Ts[t][h]); Tg[t][h] = dpp_rol1(Tg[t][h]); Td[t][h] = dpp_rol1(Td[t][h]); Ta[t][h] = dpp_rol1(Ta[t][h]);
            }
    };
    // offset 0: pairs inside a lane (travelling slot t > own slot k), copy A only (copy B at offset 16 is visited at step 16)
#pragma unroll
    for (int k = 0; k < DPT; ++k)
#pragma unroll
        for (int t = k + 1; t < DPT; ++t)
            if (active(k, t)) pair2(k, t, f32x2{1.0f, 0.0f}, true);
    // steps 1..15: offsets r (copy A) and r + 16 (copy B)
    for (int r = 1; r < 16; ++r) {
        rotate();
#pragma unroll
        for (int k = 0; k < DPT; ++k) {
#pragma unroll
            for (int t = 0; t < DPT; ++t)
                if (active(k, t)) pair2(k, t, one2, false);
        }
    }
    // step 16: offset 16 (A) and the half step 32 (B), where lanes a and a+32 see each other from both ends — the lower half keeps them
    {
        rotate();
        const float lm = lane < 32 ? 1.0f : 0.0f;
#pragma unroll
        for (int k = 0; k < DPT; ++k)
#pragma unroll
            for (int t = 0; t < DPT; ++t)
                if (active(k, t)) pair2(k, t, f32x2{1.0f, lm}, true);
    }
    // the travelling accumulators sit 16 (copy A) / 32 (copy B) lanes behind their owners
    const int behind16 = (lane - 16) & 63;
#pragma unroll
    for (int k = 0; k < DPT; ++k) tot[k] = (ga2[k].x + ga2[k].y) + (__shfl(Ta[k].x, behind16, 64) + __shfl(Ta[k].y, lane ^ 32, 64));
}

// The waves of a block are independent (one query each, wave-local LDS hand-overs), so the block size is a launch-time choice:
// up to 16 waves per workgroup.  256 workgroups of 16 waves spread evenly over the 256 CUs; 1024 workgroups of 4 do not (the
// dispatcher fills some CUs deeper than others: 33.7 us vs 29.9 us for 4096 queries of 128 documents).
// DPT = 4 (list lengths 129..256): 8 waves at most — the two-copy pair loop with its Z variants needs ~150 registers, and under the
// 128-register cap of a 16-wave block it spilled 25 of them to scratch (PMC: 348 MB of HBM traffic per 65 536 queries against 202 MB
// algorithmic); 3 waves per SIMD without spills run as fast as 4 with them.
constexpr int kRingBlock = 1024;
#ifndef PTR_RING4_WAVES
#define PTR_RING4_WAVES 8          // waves per workgroup of the DPT = 4 kernel (experiment: 16 = the 128-register cap)
#endif
template <int DPT> constexpr int ring_block() { return DPT >= 8 ? 256 : DPT >= 4 ? 64 * PTR_RING4_WAVES : kRingBlock; }      // DPT = 8: one wave per SIMD — 128 record registers + the temporaries of 64 blocks per step need more than the 256 of an 8-wave block (163 spills there)
template <int DPT>
__global__ void __launch_bounds__(ring_block<DPT>())
lambdarank_ring_kernel(const float *__restrict__ preds, const float *__restrict__ labels, const int32_t *__restrict__ lens,
                       int B, int L, float sigma, float *__restrict__ loss_q, float *__restrict__ grad) {
    constexpr int RS = 64 * DPT;
    const int QPB = blockDim.x >> 6;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int q = blockIdx.x * QPB + wv;
    const bool valid = q < B;
    const int n = __builtin_amdgcn_readfirstlane(valid ? query_len(lens, q, L) : 0);      // wave-uniform: scalar loop control
    float *keys = smem + (size_t)wv * (2 * RS);            // raw scores, broadcast-read by the rank count
    int *mark = reinterpret_cast<int *>(keys + RS);        // tie detection

    // ---- coalesced load: lane owns documents i = lane + 64 m, and keeps them (records stay in INPUT order: the pair body only
    // needs every record's D = 1/log2(rank+2), not a sorted arrangement)
    float si[DPT], li[DPT];
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = lane + 64 * m;
        const bool in = i < n;
        si[m] = in ? preds[(size_t)q * L + i] : -INFINITY;
        li[m] = in ? labels[(size_t)q * L + i] : 0.0f;
        keys[i] = si[m];
    }
    wave_lds_sync();
    // rank = #{j : s_j > s_i}, ONE VALU slot per compare: t = clamp(BIG*s_j - BIG*s_i, 0, 1) is exactly 1 for s_j > s_i and 0
    // otherwise (v_pk_fma_f32 with the clamp modifier: two compares per instruction; the fma is exact up to its final rounding, so
    // the sign is right and 0 means equal), summed in fp32 (exact up to 2^24).  BIG = 2^100: t is fractional only for
    // 0 < s_j - s_i < 2^-100, and BIG*s overflows only for |s| >= 2^28 (inf - inf = NaN clamps to 0) — either way the sums are
    // not all integers or two documents share a rank; both are detected below and the wave recounts with compares
    // (count_ranks), as it does for ties (equal scores; rank = original index order).
    int rk[DPT];
    {
        const float big = 0x1p100f;
        const f32x2 big2 = {big, big};
        f32x2 nsb[DPT], cnt[DPT];
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const float v = lane + 64 * m < n ? -si[m] * big : 0.0f;
            nsb[m] = f32x2{v, v}; cnt[m] = f32x2{0.0f, 0.0f};
        }
        const float4 *k4 = reinterpret_cast<const float4 *>(keys);
        // keys[n..RS) = -inf contribute 0: whole groups of 8 keys, the next group's reads issued ahead of this one's arithmetic
        const int n8 = (n + 7) >> 3;
        float4 va = k4[0], vb = k4[1];
        for (int j8 = 0; j8 < n8; ++j8) {
            const float4 ua = va, ub = vb;
            const int nx = min(j8 + 1, RS / 8 - 1);
            va = k4[2 * nx]; vb = k4[2 * nx + 1];
            const f32x2 u0 = {ua.x, ua.y}, u1 = {ua.z, ua.w}, u2 = {ub.x, ub.y}, u3 = {ub.z, ub.w};
#pragma unroll
            for (int m = 0; m < DPT; ++m) {
                cnt[m] += pk_fma_clamp(u0, big2, nsb[m]);
                cnt[m] += pk_fma_clamp(u1, big2, nsb[m]);
                cnt[m] += pk_fma_clamp(u2, big2, nsb[m]);
                cnt[m] += pk_fma_clamp(u3, big2, nsb[m]);
            }
        }
        bool redo = false;
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int i = lane + 64 * m;
            const float c = cnt[m].x + cnt[m].y;
            rk[m] = (int)c;
            redo |= i < n && ((float)rk[m] != c || rk[m] >= n);
            if (i < n && rk[m] < n) mark[rk[m]] = i;
        }
        wave_lds_sync();
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int i = lane + 64 * m;
            redo |= i < n && rk[m] < n && mark[rk[m]] != i;
        }
        if (__any(redo)) count_ranks<kWave, DPT>(keys, n, lane, si, rk);
    }
    // own records {s, G, D}: G = gain / IDCG (labels arrive in ideal order: DCG of the input order is the IDCG), D by rank.
    // 1/log2(.) = v_log_f32 + v_rcp_f32 with one Newton step (<= 1 ulp; the parity bar is 1e-5).
    float Di[DPT], Gi[DPT];
    float part = 0.0f;
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = lane + 64 * m;
        const bool in = i < n;
        Gi[m] = in ? gain_of(li[m]) : 0.0f;
        part = fmaf(Gi[m], inv_log2_pos(i), part);
        Di[m] = inv_log2_pos(in ? rk[m] : i);              // padding keeps positions n..RS-1
    }
    const float ridcg = 1.0f / wave_sum_dpp(part);
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const bool in = lane + 64 * m < n;
        Gi[m] = in ? Gi[m] * ridcg : -1.0f;
        si[m] = in ? si[m] : -1e30f;
    }

    // ---- own records by position p = DPT*lane + slot; travelling copies
    const float c2 = sigma * 1.4426950408889634f;          // x*log2(e) folded into sigma
    const float kClamp = -100.0f * 1.4426950408889634f;    // BCE's -100 clamp in the log2 domain
    float ga[DPT], Tacc[DPT];
    float lacc = 0.0f;
    if constexpr (DPT == 1) {
        // one document per lane: scalar pair body, the travelling record moves one lane per step
        const float so = si[0], go = Gi[0], Do = Di[0];
        float Ts = so, Tg = go, Td = Do;
        ga[0] = 0.0f; Tacc[0] = 0.0f;
        for (int step = 1; step <= 32; ++step) {
            Ts = dpp_rol1(Ts); Tg = dpp_rol1(Tg); Td = dpp_rol1(Td); Tacc[0] = dpp_rol1(Tacc[0]);
            const float e = __builtin_amdgcn_exp2f(-fabsf((so - Ts) * c2));
            const float dd = 1.0f + e;
            float p = __builtin_amdgcn_rcpf(dd);
            p = fmaf(p, fmaf(-dd, p, 1.0f), p);
            const float qv = 1.0f - p;
            const float dDn = Td - Do;
            float un = ((go - Tg) * dDn) * sigma;
            if (step == 32) un = dDn < 0.0f ? un : 0.0f;     // half step: every pair is seen from both ends, keep the first-ranked one's
            const bool t1 = un < 0.0f;
            lacc = fmaf(fabsf(un), fmaxf(__builtin_amdgcn_logf(t1 ? p : qv), kClamp), lacc);
            const float m = __builtin_copysignf(t1 ? qv : __builtin_amdgcn_fractf(p), dDn);
            ga[0] = fmaf(-un, m, ga[0]);
            Tacc[0] = fmaf(un, m, Tacc[0]);
        }
    } else {
        // DPT >= 2: see ring_pairs() — two pairs per instruction with packed fp32, whole (own slot, travelling slot) blocks of
        // equal-label documents skipped.  Slot k holds documents 64k..64k+63 of the label-sorted list (lambdarank.py:36), so the tail
        // of the list — the run of grade-0 documents, 51 % of MSLR-WEB30K — fills whole slots: Z = number of trailing slots whose real
        // documents all carry one and the same label (empty slots count).  Every pair inside those Z slots has |G_i - G_j| = 0
        // exactly (metric_utils.py:43) and contributes exactly 0 to loss and gradients; purity is CHECKED here (wave-uniform
        // compares), not inferred from sortedness.
        int Z = 0;
        {
            float zlab = 0.0f;
            bool have = false, open = true;
#pragma unroll
            for (int k = DPT - 1; k >= 0; --k) {
                const bool in = lane + 64 * k < n;
                const float first = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, li[k])));   // document 64k
                const bool empty = 64 * k >= n;
                const bool pure = __all(!in || li[k] == first);
                const bool ok = open && (empty || (pure && (!have || first == zlab)));
                if (ok && !empty) { zlab = first; have = true; }
                Z += ok ? 1 : 0;
                open = ok;
            }
            Z = __builtin_amdgcn_readfirstlane(Z);
        }
        float tot[DPT];
#pragma unroll
        for (int k = 0; k < DPT; ++k) tot[k] = 0.0f;
        auto run = [&](auto z_) { ring_pairs<DPT, decltype(z_)::value>(si, Gi, Di, sigma, c2, kClamp, lane, tot, lacc); };
        // all labels equal: every weight is 0 — unless no document is relevant at all (IDCG = 0: G = 0 * inf), where the reference's
        // loss and gradients are NaN (edge behaviours)
        auto degenerate = [&]() {
            const float bad = ridcg - ridcg;                 // 0 for a finite 1 / IDCG, NaN otherwise
            lacc = bad;
#pragma unroll
            for (int k = 0; k < DPT; ++k) tot[k] = bad;
        };
        if constexpr (DPT == 2) {
            switch (Z) {
                case 0: run(std::integral_constant<int, 0>{}); break;
                case 1: run(std::integral_constant<int, 1>{}); break;
                default: degenerate(); break;                // every label equal: loss and gradients are exactly 0
            }
        } else if constexpr (DPT == 8) {
            // r6: lists of 257..512 documents.  Each Z is its own straight-line loop of up to 64 blocks per ring step (~2 K instructions): four
            // variants (Z rounded down to an even count) keep the code at what the instruction cache holds; an odd trailing pure slot is evaluated
            switch (Z) {
                case 0: case 1: run(std::integral_constant<int, 0>{}); break;
                case 2: case 3: run(std::integral_constant<int, 2>{}); break;
                case 4: case 5: run(std::integral_constant<int, 4>{}); break;
                case 6: case 7: run(std::integral_constant<int, 6>{}); break;
                default: degenerate(); break;
            }
        } else {
            switch (Z) {
                case 0: run(std::integral_constant<int, 0>{}); break;
                case 1: run(std::integral_constant<int, 1>{}); break;
                case 2: run(std::integral_constant<int, 2>{}); break;
                case 3: run(std::integral_constant<int, 3>{}); break;
                default: degenerate(); break;
            }
        }
        const float loss = wave_sum_dpp(lacc) * (-0.6931471805599453f / sigma);
#pragma unroll
        for (int k = 0; k < DPT; ++k) {
            const int i = lane + 64 * k;
            if (valid && i < L) grad[(size_t)q * L + i] = i < n ? tot[k] : 0.0f;
        }
        if (valid && lane == 0) loss_q[q] = loss;
        return;
    }
    // the travelling accumulators are half a ring (32 lanes) away from their owners; gradients leave in input order, coalesced
    const float loss = wave_sum_dpp(lacc) * (-0.6931471805599453f / sigma);
#pragma unroll
    for (int k = 0; k < DPT; ++k) {
        const float tot = ga[k] + __shfl_xor(Tacc[k], 32, 64);
        const int i = lane + 64 * k;
        if (valid && i < L) grad[(size_t)q * L + i] = i < n ? tot : 0.0f;
    }
    if (valid && lane == 0) loss_q[q] = loss;
}

// enqueue lambdarank_ring_kernel<DPT> with QPB queries (waves) per workgroup; defined in the translation unit that instantiates DPT
int launch_lambdarank_ring_small(int dpt, int QPB, const float *preds, const float *labels, const int32_t *lens, int B, int L, float sigma, float *loss_q,
                                 float *grad, hipStream_t st);

}  // namespace ptr
