// Evaluation kernels: descending sort with indices, and the fused Evaluator prologue + nDCG / nERR / AP / P at cut-offs.
//
// Reference: ptranking/base/ranker.py:46-60 and :220-243 (predict -> .cpu() -> torch.sort -> gather -> ideal sort -> metric),
//            ptranking/metric/adhoc/adhoc_metric.py:36-62 (P@ks), :91-123 (AP@ks), :127-193 (nERR@ks), :219-260 (nDCG@ks).
// The reference moves every batch of predictions to the host and sorts there; here the whole chain runs on the device
// and only [B, len(ks)] numbers per metric ever need to leave it.
//
// Per query: stage in ideal order -> rank documents by score (counting sort, (score desc, index asc)) -> labels by
// predicted rank in LDS -> wave 0 walks the ranking in 64-position chunks with prefix scans (sum / product), writing
// a result whenever position+1 equals a requested cut-off.  Cut-offs beyond the list are zero-filled at the END of
// the row, like the reference's padded_*_at_ks.
#include "ptr_device.h"

namespace ptr {

struct Cutoffs { int nk; int k[PTR_MAX_CUTOFFS]; };

// ------------------------------------------------------------------------------------------------ sort_desc
template <int G, int DPT>
__global__ void __launch_bounds__(kBlock)
sort_desc_kernel(const float *__restrict__ preds, const int32_t *__restrict__ lens, int B, int L, int Lp, float *__restrict__ vals,
                 int64_t *__restrict__ idx, int aligned) {
    constexpr int QPB = kBlock / G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, grp = tid / G, t = tid % G;
    const int q = G == kWave ? __builtin_amdgcn_readfirstlane(blockIdx.x * QPB + grp) : blockIdx.x * QPB + grp;   // one wavefront per query: scalar
    const bool valid = q < B;
    const int n = valid ? query_len(lens, q, L) : 0;
    float *keys = smem + (size_t)grp * 3 * Lp, *sv = keys + Lp;
    int *si_ = reinterpret_cast<int *>(sv + Lp);
    if constexpr (G == kWave) {
        // r5, one wavefront per query: blocked layout (lane t owns documents / positions t*DPT ..), 16-byte loads and stores
        if (!valid) return;
        float own[DPT], v[DPT];
        int rk[DPT];
        load_blocked<DPT>(preds + (size_t)q * L, n, L, t, -INFINITY, own, aligned != 0);
        float *vrow = vals + (size_t)q * L;
        int64_t *irow = idx + (size_t)q * L;
        typedef long i64x2_t __attribute__((ext_vector_type(2)));
        if constexpr (DPT >= 2) {
            // 65 .. 1024 documents: packed (score, index) keys, the permutation read off the sorted keys (no rank search)
            float sc[DPT];
            int id[DPT];
            if (sort_scores_packed<DPT, true>(keys, n, t, own, sc, id)) {
                if (DPT % 4 == 0 && (L & 3) == 0 && aligned) {
#pragma unroll
                    for (int r = 0; r < DPT; r += 4) {
                        const int p = t * DPT + r;
                        if (p < L) {
                            *reinterpret_cast<float4 *>(vrow + p) = float4{p < n ? sc[r] : 0.0f, p + 1 < n ? sc[r + 1] : 0.0f, p + 2 < n ? sc[r + 2] : 0.0f, p + 3 < n ? sc[r + 3] : 0.0f};
                            *reinterpret_cast<i64x2_t *>(irow + p) = i64x2_t{p < n ? (long)id[r] : (long)p, p + 1 < n ? (long)id[r + 1] : (long)(p + 1)};
                            *reinterpret_cast<i64x2_t *>(irow + p + 2) = i64x2_t{p + 2 < n ? (long)id[r + 2] : (long)(p + 2), p + 3 < n ? (long)id[r + 3] : (long)(p + 3)};
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < DPT; ++r) {
                        const int p = t * DPT + r;
                        if (p < L) { vrow[p] = p < n ? sc[r] : 0.0f; irow[p] = p < n ? (int64_t)id[r] : (int64_t)p; }
                    }
                }
                return;
            }
            wave_lds_sync();
        }
        rank_blocked_wave<DPT>(keys, sv, n, t, own, rk, v);
#pragma unroll
        for (int r = 0; r < DPT; ++r) {
            const int i = t * DPT + r;
            if (i < n) { sv[rk[r]] = own[r]; si_[rk[r]] = i; }
        }
        wave_lds_sync();
        if (DPT % 4 == 0 && (L & 3) == 0 && aligned) {
#pragma unroll
            for (int r = 0; r < DPT; r += 4) {
                const int p = t * DPT + r;
                if (p < L) {
                    const float4 sv4 = *reinterpret_cast<const float4 *>(sv + p);
                    const int4 si4 = *reinterpret_cast<const int4 *>(si_ + p);
                    *reinterpret_cast<float4 *>(vrow + p) = float4{p < n ? sv4.x : 0.0f, p + 1 < n ? sv4.y : 0.0f, p + 2 < n ? sv4.z : 0.0f, p + 3 < n ? sv4.w : 0.0f};
                    *reinterpret_cast<i64x2_t *>(irow + p) = i64x2_t{p < n ? (long)si4.x : (long)p, p + 1 < n ? (long)si4.y : (long)(p + 1)};
                    *reinterpret_cast<i64x2_t *>(irow + p + 2) = i64x2_t{p + 2 < n ? (long)si4.z : (long)(p + 2), p + 3 < n ? (long)si4.w : (long)(p + 3)};
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < DPT; ++r) {
                const int p = t * DPT + r;
                if (p < L) {
                    vrow[p] = p < n ? sv[p] : 0.0f;
                    irow[p] = p < n ? (int64_t)si_[p] : (int64_t)p;
                }
            }
        }
    } else {
    float own[DPT];
    int rk[DPT];
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        own[m] = i < n ? preds[(size_t)q * L + i] : -INFINITY;
        if (i < Lp) keys[i] = own[m];
    }
    __syncthreads();
    count_ranks_fast<G, DPT>(keys, si_, n, t, own, rk);      // si_ doubles as the permutation-check scratch before it is filled
    __syncthreads();
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        if (i < n) { sv[rk[m]] = own[m]; si_[rk[m]] = i; }
    }
    __syncthreads();
    if (valid) {
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int r = t + m * G;
            if (r < L) {
                vals[(size_t)q * L + r] = r < n ? sv[r] : 0.0f;
                idx[(size_t)q * L + r] = r < n ? (int64_t)si_[r] : (int64_t)r;
            }
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------ batch max (nERR normaliser)
// torch.max(batch_ideal_rankings) (adhoc_metric.py:174-175).  Grid-wide: block maxima are merged with an integer atomicMax
// on an order-preserving encoding of the float (max is exact and order-independent, so the result is deterministic).
__device__ __forceinline__ int float_to_ordered(float f) { const int b = __float_as_int(f); return b >= 0 ? b : b ^ 0x7FFFFFFF; }
__device__ __forceinline__ float ordered_to_float(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

__global__ void __launch_bounds__(kBlock)
batch_max_kernel(const float *__restrict__ labels, const int32_t *__restrict__ lens, int B, int L, int *__restrict__ out_key) {
    __shared__ float red[4];
    float mx = -INFINITY;
    // flat, coalesced walk over the B x L label matrix (the per-query form left 3/4 of a workgroup idle at L = 64); the (query, position)
    // pair of an element advances incrementally with the grid stride
    const size_t total = (size_t)B * L, stride = (size_t)gridDim.x * kBlock;
    const int dq = (int)(stride / L), di = (int)(stride % L);
    size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;
    int q = (int)(e / L), i = (int)(e % L);
    for (; e < total; e += stride) {
        if (!lens || i < query_len(lens, q, L)) mx = fmaxf(mx, labels[e]);
        q += dq; i += di;
        if (i >= L) { i -= L; ++q; }
    }
    mx = group_max<kBlock>(mx, red, threadIdx.x);
    if (threadIdx.x == 0) atomicMax(out_key, float_to_ordered(mx));
}

// batch_max_kernel() without a length vector: a plain 16-byte-load stream over the B x L labels (r5: the scalar per-element walk reached
// 1.8 TB/s).  total4 = B L / 4.
__global__ void __launch_bounds__(kBlock)
batch_max_vec_kernel(const float4 *__restrict__ labels4, size_t total4, int *__restrict__ out_key) {
    __shared__ float red[4];
    float mx = -INFINITY;
    const size_t stride = (size_t)gridDim.x * kBlock;
    size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x;
    for (; e + 7 * stride < total4; e += 8 * stride) {             // eight loads in flight
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = labels4[e + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) mx = fmaxf(mx, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
    }
    for (; e < total4; e += stride) {
        const float4 a = labels4[e];
        mx = fmaxf(mx, fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
    }
    mx = group_max<kBlock>(mx, red, threadIdx.x);
    if (threadIdx.x == 0) atomicMax(out_key, float_to_ordered(mx));
}

// ------------------------------------------------------------------------------------------------ metrics
// 2^l - 1 on the transcendental pipe alone (v_exp_f32: exact for the integer grades, 1 ulp otherwise; labels are far from its denormal range)
__device__ __forceinline__ float gain_fast(float label) { return __builtin_amdgcn_exp2f(label) - 1.0f; }
// LDS per group (floats): S_id[Lp] | Y_id[Lp] | Y_sys[Lp]
// WHICH: compile-time set of requested metrics (bit 0 nDCG, 1 nERR, 2 AP, 3 P) for the sets the Evaluator asks for (one metric, or all
// four); -1 = decided at run time from the output pointers.  The walk's nine prefix scans shrink to the ones the set needs (r5).
template <int G, int DPT, int WHICH>
__global__ void __launch_bounds__(kBlock)
metrics_kernel(const float *__restrict__ preds, const float *__restrict__ labels, const int32_t *__restrict__ lens, int B, int L,
               int Lp, Cutoffs ck, int presort, int linear_gain, float max_label_host, const float *__restrict__ max_label_dev,
               float *__restrict__ o_ndcg, float *__restrict__ o_nerr, float *__restrict__ o_ap, float *__restrict__ o_p, int aligned) {
    constexpr int QPB = kBlock / G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, grp = tid / G, t = tid % G;
    const int q = G == kWave ? __builtin_amdgcn_readfirstlane(blockIdx.x * QPB + grp) : blockIdx.x * QPB + grp;   // one wavefront per query: scalar
    // (persistent wavefronts walking the queries with a grid stride were measured, r5: 88 registers instead of 52, 69 us against 66)
    float *S_id = smem + (size_t)grp * 3 * Lp, *Y_id = S_id + Lp, *Y_sys = Y_id + Lp;
    const bool valid = q < B;
    if (G == kWave && !valid) return;                          // the waves of a block are independent on this path
    const int n = valid ? query_len(lens, q, L) : 0;

    if constexpr (G == kWave) {
        // r5, one wavefront per query (lists of up to 1024 documents): the documents sit in the BLOCKED layout (lane t owns documents
        // t*DPT ..: 16-byte loads, no LDS staging ahead of the sorts), the waves of a block are independent (no workgroup barrier)
        float si[DPT], li[DPT], v[DPT];
        int rk[DPT];
        load_blocked<DPT>(preds + (size_t)q * L, n, L, t, -INFINITY, si, aligned != 0);
        load_blocked<DPT>(labels + (size_t)q * L, n, L, t, 0.0f, li, aligned != 0);
        // NOTE: the reference sorts the predictions in ORIGINAL order (ranker.py:50) — ties are broken by original index —
        // and sorts the labels separately for the ideal ranking (ranker.py:53-56).
        bool packed = false;
        if constexpr (DPT >= 2) {
            // 65 .. 1024 documents: packed (score, index) keys; the labels by predicted rank are gathered through the sorted keys' index bits
            int id[DPT];
            lds_store_blocked<DPT>(Y_id, t, li);                        // labels by document (Y_id is staged below)
            packed = sort_scores_packed<DPT, false>(S_id, n, t, si, v, id);
            if (packed) {
                float ls[DPT];
#pragma unroll
                for (int r = 0; r < DPT; ++r) ls[r] = Y_id[id[r]];      // torch.gather(labels, idx), ranker.py:52 (positions >= n: never walked)
                lds_store_blocked<DPT>(Y_sys, t, ls);
            }
            wave_lds_sync();
        }
        if (!packed) {
            rank_blocked_wave<DPT>(S_id, Y_sys, n, t, si, rk, v);      // bitonic sort + binary search (Y_sys: scratch of the exact recount)
#pragma unroll
            for (int r = 0; r < DPT; ++r) { if (t * DPT + r < n) Y_sys[rk[r]] = li[r]; }
        }
        if (!presort) {
            // only the ideal LABELS are walked below, and equal labels are interchangeable: a value-only sort, no tie handling
#pragma unroll
            for (int r = 0; r < DPT; ++r) v[r] = t * DPT + r < n ? li[r] : -INFINITY;
            wave_sort_desc<DPT>(v, t);
#pragma unroll
            for (int r = 0; r < DPT; ++r) li[r] = t * DPT + r < n ? v[r] : 0.0f;
        }
        lds_store_blocked<DPT>(Y_id, t, li);
        wave_lds_sync();
    } else {
    float si[DPT], li[DPT];
    int ipos[DPT];
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        const bool in = i < n;
        si[m] = in ? preds[(size_t)q * L + i] : -INFINITY;
        li[m] = in ? labels[(size_t)q * L + i] : 0.0f;
    }
    // rank by score first, on the raw tile (see the note above)
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        if (i < Lp) S_id[i] = si[m];
    }
    __syncthreads();
    int rk[DPT];
    // ranks by packed fma-clamp counting (one VALU slot per compare; ties / overflow fall back to the exact compares) — the O(L^2) count
    // is what the kernel's time is made of: 0.54 -> 0.2 ms for 65 536 x 256.  Y_id (not staged yet) is the permutation-check scratch.
    count_ranks_fast<G, DPT>(S_id, reinterpret_cast<int *>(Y_id), n, t, si, rk);
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        if (i < n) Y_sys[rk[m]] = li[m];                    // torch.gather(labels, idx), ranker.py:52
    }
    __syncthreads();
    stage_ideal_order<G, DPT>(S_id, Y_id, n, Lp, t, presort != 0, si, li, ipos);   // ends with a barrier
    }

    // ---- wave 0 of the group walks the two rankings
    if (valid && t < kWave) {
        const int lane = t;
        const float max_label = max_label_dev ? ordered_to_float(reinterpret_cast<const int *>(max_label_dev)[0]) : max_label_host;
        // adhoc_metric.py:133: 1 / 2^max_label.  Integer grades (every MultiLabel set): 2^-max_label from v_exp_f32, exact, like the reciprocal
        // of a power of two; anything else (or a grade beyond the normal range) keeps the library exp2 and the IEEE division
        const float ml = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, max_label)));   // uniform
        float rpow_max;
        if (ml == floorf(ml) && fabsf(ml) <= 64.0f) rpow_max = __builtin_amdgcn_exp2f(-ml);
        else rpow_max = 1.0f / exp2f(ml);
        const bool w_ndcg = WHICH < 0 ? o_ndcg != nullptr : (WHICH & 1) != 0, w_nerr = WHICH < 0 ? o_nerr != nullptr : (WHICH & 2) != 0;
        const bool w_ap = WHICH < 0 ? o_ap != nullptr : (WHICH & 4) != 0, w_p = WHICH < 0 ? o_p != nullptr : (WHICH & 8) != 0;
        float *r_ndcg = w_ndcg ? o_ndcg + (size_t)q * ck.nk : nullptr;
        float *r_nerr = w_nerr ? o_nerr + (size_t)q * ck.nk : nullptr;
        float *r_ap = w_ap ? o_ap + (size_t)q * ck.nk : nullptr;
        float *r_p = w_p ? o_p + (size_t)q * ck.nk : nullptr;
        // lane s of the wave holds the cut-off of output slot s (the cut-offs that fit the list, in order; zero padding goes last,
        // adhoc_metric.py:255-258): the walk's results are gathered from the lanes at rank k-1 and stored as ONE row per metric
        int used = 0, kslot = 0;
        for (int c = 0; c < ck.nk; ++c) {
            const bool use = ck.k[c] >= 1 && ck.k[c] <= n;
            if (use && lane == used) kslot = ck.k[c];
            used += use ? 1 : 0;
        }
        if (lane < ck.nk && lane >= used) {
            if (r_ndcg) r_ndcg[lane] = 0.0f;
            if (r_nerr) r_nerr[lane] = 0.0f;
            if (r_ap) r_ap[lane] = 0.0f;
            if (r_p) r_p[lane] = 0.0f;
        }
        int kmax = 0;
        for (int c = 0; c < ck.nk; ++c) if (ck.k[c] <= n && ck.k[c] > kmax) kmax = ck.k[c];
        float c_sdcg = 0.f, c_idcg = 0.f, c_rel = 0.f, c_prec = 0.f, c_ideal = 0.f, c_serr = 0.f, c_ierr = 0.f, c_sun = 1.f, c_iun = 1.f;
        const int nchunk = (kmax + 63) >> 6;
        for (int ch = 0; ch < nchunk; ++ch) {
            // ranks in [kmax, n) of the last chunk are NOT masked (inclusive prefix scans flow from low lanes to high lanes only, no cut-off
            // reads those lanes, the chunk carries of the last chunk are never used); ranks >= n ARE: Y_sys[n ..] is never written on the
            // counting paths, and whatever bits lie there must not enter gain_fast and the scans (two v_cndmask per chunk)
            const int r = ch * 64 + lane;
            const bool in = true;
            const int rl = G == kWave ? r : (r < Lp ? r : Lp - 1);   // four waves per query: the rows hold round_up(L, 4) entries
            const float ys = r < n ? Y_sys[rl] : 0.0f, yi = r < n ? Y_id[rl] : 0.0f;
            const float rdisc = __builtin_amdgcn_rcpf(__builtin_amdgcn_logf((float)r + 2.0f));   // 1 / log2(rank + 2): v_log_f32, v_rcp_f32 (1 ulp each)
            // DCG gain: 2^l - 1 for graded labels, the raw label for LABEL_TYPE.Permutation (adhoc_metric.py:207-212,225-230)
            const float gs = in ? (linear_gain ? ys : gain_fast(ys)) : 0.0f, gi = in ? (linear_gain ? yi : gain_fast(yi)) : 0.0f;
            float sdcg = 0.0f, idcg = 1.0f;
            if (w_ndcg) {
                sdcg = gs * rdisc; idcg = gi * rdisc;
                wave_incl_sum2(sdcg, idcg);                                      // adhoc_metric.py:233-234
                sdcg += c_sdcg; idcg += c_idcg;
            }
            const float rel = in ? fminf(fmaxf(ys, 0.0f), 1.0f) : 0.0f;                // binary relevance (:106)
            const float cumrel = (w_ap || w_p) ? wave_incl_sum(rel, lane) + c_rel : 0.0f;
            const float rpos = (float)r + 1.0f;
            const float rr = __builtin_amdgcn_rcpf(rpos);                // 1 ulp: the nERR cascade below
            // rank-wise precision cumrel / (r + 1) (:111), correctly rounded like the reference's division — a perfect prefix gives P@k = AP =
            // 1.0 exactly, not 0.99999994 — by the Newton steps of the IEEE expansion WITHOUT its range scaling (v_div_scale /
            // v_div_fmas / v_div_fixup): the divisor is an integer in [1, 4096] and 0 <= cumrel <= r + 1, nothing scales, nothing is special
            const float y0 = fmaf(fmaf(-rpos, rr, 1.0f), rr, rr);
            const float q0 = cumrel * y0;
            const float q1 = fmaf(fmaf(-rpos, q0, cumrel), y0, q0);
            const float pr = fmaf(fmaf(-rpos, q1, cumrel), y0, q1);
            float cumprec = 0.0f, cumideal = 1.0f;
            if (w_ap) {
                cumprec = pr * rel; cumideal = yi;                              // (:112); GRADED ideal labels (:114)
                wave_incl_sum2(cumprec, cumideal);
                cumprec += c_prec; cumideal += c_ideal;
            }
            float serr = 0.0f, ierr = 1.0f, s_incl = 1.0f, i_incl = 1.0f;
            if (w_nerr) {
                const float ssat = gs * rpow_max, isat = gi * rpow_max;                       // (:133)
                // cascade: product of (1 - sat) over EARLIER ranks (:135-143) = exclusive prefix product
                s_incl = 1.0f - ssat; i_incl = 1.0f - isat;
                wave_incl_prod2(s_incl, i_incl);
                // lane t <- lane t - 1 (wave_shr:1), lane 0 keeps the identity
                const int one = __builtin_bit_cast(int, 1.0f);
                const float s_excl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(one, __builtin_bit_cast(int, s_incl), 0x138, 0xF, 0xF, false));
                const float i_excl = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(one, __builtin_bit_cast(int, i_incl), 0x138, 0xF, 0xF, false));
                serr = rr * ssat * (s_excl * c_sun); ierr = rr * isat * (i_excl * c_iun);
                wave_incl_sum2(serr, ierr);
                serr += c_serr; ierr += c_ierr;
            }
            {
                const int src = kslot - 1 - ch * 64;                       // lane of this chunk that holds rank kslot - 1
                const bool mine = lane < used && src >= 0 && src < 64;
                const int sel = (src & 63) << 2;
                if (r_ndcg) { const float o = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, __builtin_bit_cast(int, sdcg / idcg))); if (mine) r_ndcg[lane] = o; }
                if (r_nerr) { const float o = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, __builtin_bit_cast(int, serr / ierr))); if (mine) r_nerr[lane] = o; }
                if (r_ap) { const float o = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, __builtin_bit_cast(int, cumprec / cumideal))); if (mine) r_ap[lane] = o; }
                if (r_p) { const float o = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(sel, __builtin_bit_cast(int, pr))); if (mine) r_p[lane] = o; }
            }
            if (ch + 1 < nchunk) {                                         // carries into the next chunk: lane 63's totals (v_readlane)
                auto last = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63)); };
                c_sdcg = last(sdcg); c_idcg = last(idcg); c_rel = last(cumrel);
                c_prec = last(cumprec); c_ideal = last(cumideal);
                c_serr = last(serr); c_ierr = last(ierr);
                c_sun *= last(s_incl); c_iun *= last(i_incl);
            }
        }
    }
}

}  // namespace ptr

extern "C" int ptr_sort_desc(const float *preds, const int32_t *lens, int B, int L, float *vals, int64_t *idx, void *stream) {
    using namespace ptr;
    const char *who = "ptr_sort_desc";
    if (int rc = check_batch(preds, vals, B, L, who)) return rc;
    if (B > 0 && !idx) { set_error("%s: NULL output pointer", who); return PTR_ERR_INVALID_ARG; }
    if (B == 0) return 0;
    return dispatch_wave_tiling(L, [&]<int G, int DPT>() -> int {
        constexpr int QPB = kBlock / G;
        const int Lp = G == kWave ? kWave * DPT : round_up(L, 4);      // one wavefront per query sorts 64*DPT padded keys
        auto kern = sort_desc_kernel<G, DPT>;
        const size_t lds = (size_t)QPB * 3 * Lp * sizeof(float);
        if (int e = allow_lds(kern, lds)) return e;
        hipLaunchKernelGGL(kern, dim3((B + QPB - 1) / QPB), dim3(kBlock), lds, as_stream(stream), preds, lens, B, L, Lp, vals, idx,
                           (int)(((reinterpret_cast<uintptr_t>(preds) | reinterpret_cast<uintptr_t>(vals) | reinterpret_cast<uintptr_t>(idx)) & 15) == 0));
        return check_hip(hipGetLastError(), who);
    });
}

extern "C" int ptr_metrics_at_ks(const float *preds, const float *labels, const int32_t *lens, int B, int L, const int32_t *ks,
                                 int nk, int presort, int label_type, float max_label, float *max_label_ws, float *ndcg, float *nerr, float *ap,
                                 float *prec, void *stream) {
    using namespace ptr;
    const char *who = "ptr_metrics_at_ks";
    if (int rc = check_batch(preds, labels, B, L, who)) return rc;
    if (nk < 0 || (nk > 0 && !ks)) { set_error("%s: bad cut-off list", who); return PTR_ERR_INVALID_ARG; }
    if (nk > PTR_MAX_CUTOFFS) { set_error("%s: %d cut-offs exceed PTR_MAX_CUTOFFS=%d", who, nk, PTR_MAX_CUTOFFS); return PTR_ERR_UNSUPPORTED; }
    if (label_type != PTR_LABEL_MULTILABEL && label_type != PTR_LABEL_PERMUTATION) { set_error("%s: unknown label_type %d", who, label_type); return PTR_ERR_INVALID_ARG; }
    if (nerr && label_type != PTR_LABEL_MULTILABEL) {   // adhoc_metric.py:157-164 raises NotImplementedError for anything else
        set_error("%s: nERR is only defined for graded (MultiLabel) labels", who);
        return PTR_ERR_UNSUPPORTED;
    }
    if (nerr && max_label < 0.0f && !max_label_ws) {
        set_error("%s: nERR with max_label < 0 needs the max_label_ws device scalar", who);
        return PTR_ERR_INVALID_ARG;
    }
    if (B == 0 || nk == 0) return 0;
    Cutoffs ck;
    ck.nk = nk;
    for (int c = 0; c < PTR_MAX_CUTOFFS; ++c) ck.k[c] = c < nk ? ks[c] : 0;
    hipStream_t st = as_stream(stream);
    const float *ml_dev = nullptr;
    if (nerr && max_label < 0.0f) {
        if (int rc = check_hip(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(max_label_ws), (int)0x80000000, 1, st), who)) return rc;
        const size_t total = (size_t)B * L;
        const size_t cells = (total + kBlock - 1) / kBlock;
        if (!lens && (total & 3) == 0 && (reinterpret_cast<uintptr_t>(labels) & 15) == 0) {
            const size_t c4 = (total / 4 + kBlock - 1) / kBlock;
            hipLaunchKernelGGL(batch_max_vec_kernel, dim3((unsigned)(c4 < 512 ? c4 : 512)), dim3(kBlock), 0, st,       // 512 atomics on the one key
                               reinterpret_cast<const float4 *>(labels), total / 4, reinterpret_cast<int *>(max_label_ws));
        } else
            hipLaunchKernelGGL(batch_max_kernel, dim3((unsigned)(cells < 2048 ? cells : 2048)), dim3(kBlock), 0, st, labels, lens, B, L,
                               reinterpret_cast<int *>(max_label_ws));
        if (int rc = check_hip(hipGetLastError(), who)) return rc;
        ml_dev = max_label_ws;
    }
    const int which = (ndcg ? 1 : 0) | (nerr ? 2 : 0) | (ap ? 4 : 0) | (prec ? 8 : 0);
    return dispatch_wave_tiling(L, [&]<int G, int DPT>() -> int {
        constexpr int QPB = kBlock / G;
        const int Lp = G == kWave ? kWave * DPT : round_up(L, 4);
        auto go = [&](auto kern) -> int {
            const size_t lds = (size_t)QPB * 3 * Lp * sizeof(float);
            if (int e = allow_lds(kern, lds)) return e;
            hipLaunchKernelGGL(kern, dim3((B + QPB - 1) / QPB), dim3(kBlock), lds, st, preds, labels, lens, B, L, Lp, ck, presort, label_type == PTR_LABEL_PERMUTATION ? 1 : 0, max_label,
                               ml_dev, ndcg, nerr, ap, prec, (int)(((reinterpret_cast<uintptr_t>(preds) | reinterpret_cast<uintptr_t>(labels)) & 15) == 0));
            return check_hip(hipGetLastError(), who);
        };
        // the Evaluator's calls: one metric (ndcg_at_k(s), nerr_at_k, ap_at_k, p_at_k) or all four (adhoc_performance_at_ks)
        switch (which) {
            case 1: return go(metrics_kernel<G, DPT, 1>);
            case 2: return go(metrics_kernel<G, DPT, 2>);
            case 4: return go(metrics_kernel<G, DPT, 4>);
            case 8: return go(metrics_kernel<G, DPT, 8>);
            case 15: return go(metrics_kernel<G, DPT, 15>);
            default: return go(metrics_kernel<G, DPT, -1>);
        }
    });
}
