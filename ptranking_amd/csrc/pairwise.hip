// Fused pairwise-BCE kernels: LambdarRank (delta-NDCG weighted, predicted order) and RankNet (unweighted, input order).
//
// One kernel per batch does, per query and entirely out of LDS/registers:
//   load scores+labels (coalesced) -> rank by score (counting sort, (score desc, index asc)) -> IDCG ->
//   normalised gains G, discounts D -> all L(L-1)/2 pairs -> per-document gradient -> scatter back through the sort
//   permutation -> coalesced store of grad[B,L] + one loss slot per query.
// HBM traffic is the algorithmic minimum: 4L (scores) + 4L (labels) + 4L (grad) + 4 (loss) bytes per query.
//
// Pair schedule ("circulant"): with n documents at rank positions 0..n-1, step d = 1..floor((n-1)/2) visits the pair
// (a, (a+d) mod n) for every a — each unordered pair exactly once (for even n the step d = n/2 is visited by
// a < n/2 only).  Lane a keeps its own gradient in a register; the partner's share is accumulated in a per-WAVE LDS row
// with a plain ds_read / add / ds_write (within one step all lanes of a wave hit distinct positions, a wave's LDS
// operations execute in program order, and no other wave touches the row), so there are no float atomics at all —
// LDS fp32 atomics measured ~10x slower than the whole rest of the pair body — and results are run-to-run bit-stable.
//
// Arithmetic follows what the reference executes in ATen (SURVEY.md §7 i-ii):
//   p = sigmoid(sigma*(s_i - s_j)) rounded to fp32; BCE with the -100 log clamp; backward w*(p-t)/max(p(1-p),1e-12)
//   times sigmoid' = p(1-p) (=> gradient exactly 0 once p rounds to 1.0f).
// Reference: ptranking/ltr_adhoc/listwise/lambdarank.py:39-56, ptranking/ltr_adhoc/pairwise/ranknet.py:32-36,
//            ptranking/ltr_adhoc/util/lambda_utils.py:5-23, ptranking/metric/metric_utils.py:19-45.
#include <stdlib.h>

#include "ptr_device.h"
#include "ptr_ring.h"

namespace ptr {

// LDS carve per group (floats): pk float4[Lp] | keys float[Lp] | gacc float[NW][Lp] | red float[4]
__host__ __device__ constexpr size_t pairwise_group_floats(int Lp, int NW) { return (size_t)Lp * (4 + 1 + NW) + 4; }

template <int G, int DPT, bool WEIGHTED>
__global__ void __launch_bounds__(kBlock)
pairwise_bce_kernel(const float *__restrict__ preds, const float *__restrict__ labels, const int32_t *__restrict__ lens,
                    int B, int L, int Lp, float sigma, float *__restrict__ loss_q, float *__restrict__ grad) {
    // G == 32 (RankNet, lists up to 32 documents — BASELINE config 1): TWO queries per wavefront, one per 32-lane half; each half owns its
    // LDS rows, so "all lanes of a wave hit distinct positions within a step" still holds and nothing else changes
    constexpr int QPB = kBlock / G, NW = G >= kWave ? G / kWave : 1;
    static_assert(G >= kWave || !WEIGHTED, "the half-wave form has no rank count");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, grp = tid / G, t = tid % G, wv = t >> 6;
    const int q = blockIdx.x * QPB + grp;
    const bool valid = q < B;
    const int n = valid ? query_len(lens, q, L) : 0;

    float *base = smem + (size_t)grp * pairwise_group_floats(Lp, NW);
    float4 *pk = reinterpret_cast<float4 *>(base);          // {score, gain-or-label, discount, -} by rank position
    float *keys = base + 4 * (size_t)Lp;                     // raw scores for the counting sort; later: sorted grads
    float *gacc = keys + Lp;                                 // [NW][Lp] partner-gradient accumulators (one row per wave)
    float *red = gacc + (size_t)NW * Lp;

    // ---- phase A: coalesced load of the query tile
    float si[DPT], li[DPT];
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int i = t + m * G;
        const bool in = i < n;
        si[m] = in ? preds[(size_t)q * L + i] : -INFINITY;
        li[m] = in ? labels[(size_t)q * L + i] : 0.0f;
        if (i < Lp) {
            keys[i] = si[m];
#pragma unroll
            for (int w = 0; w < NW; ++w) gacc[(size_t)w * Lp + i] = 0.0f;
        }
    }
    __syncthreads();

    // ---- phase B: rank positions, IDCG, per-position tile
    int rk[DPT];
    if constexpr (WEIGHTED) {
        count_ranks_fast<G, DPT>(keys, reinterpret_cast<int *>(pk), n, t, si, rk);      // pk (filled below) is the check scratch
        float part = 0.0f;
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int i = t + m * G;
            // labels arrive in ideal order (lambdarank.py:36), so DCG of the input order IS the IDCG (metric_utils.py:27)
            if (i < n) part += gain_of(li[m]) / log2f((float)i + 2.0f);
        }
        const float idcg = group_sum<G>(part, red, t);
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int i = t + m * G;
            if (i < n) {
                float *dst = reinterpret_cast<float *>(pk + rk[m]);
                dst[0] = si[m];
                dst[1] = gain_of(li[m]) / idcg;                                       // metric_utils.py:35
                reinterpret_cast<float *>(pk + i)[2] = 1.0f / log2f((float)i + 2.0f);  // metric_utils.py:39
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int i = t + m * G;
            rk[m] = i;
            if (i < n) pk[i] = make_float4(si[m], li[m], 0.0f, 0.0f);
        }
    }
    __syncthreads();

    // ---- phase C: all pairs
    float4 me[DPT];
    float ga[DPT];
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int a = t + m * G;
        me[m] = a < n ? pk[a] : make_float4(0.f, 0.f, 0.f, 0.f);
        ga[m] = 0.0f;
    }
    float lacc = 0.0f;
    float *gw = gacc + (size_t)wv * Lp;

    // The pair body is branch-free: lanes without a document (a >= n) or outside the half step run on a clamped index
    // with activity factor 0, so the DPT independent chains of a lane can be interleaved by the scheduler.
    auto pair = [&](int m, int a_eff, int d, float act) {
        int b = a_eff + d;
        b -= b >= n ? n : 0;
        const bool fwd = b > a_eff;                   // lo = a, hi = b (no wrap) — else lo = b, hi = a
        const float4 o = pk[b];
        const float gold = gw[b];                               // partner accumulator: plain read-modify-write, see header
        const float ds = fwd ? me[m].x - o.x : o.x - me[m].x;   // s_lo - s_hi
        const float dy = fwd ? me[m].y - o.y : o.y - me[m].y;   // gain (or label) of lo minus hi
        float lam;
        if constexpr (WEIGHTED) {
            const float w = (fabsf(dy) * fabsf(me[m].z - o.z)) * act;   // |G_i - G_j| * |D_i - D_j|, metric_utils.py:43
            const float x = sigma * ds;                         // >= 0 in predicted order
            const float e = __expf(-x);
            const float dd = 1.0f + e;
            float p = __builtin_amdgcn_rcpf(dd);
            p = fmaf(p, fmaf(-dd, p, 1.0f), p);                 // one Newton step: correctly-rounded-class 1/(1+e)
            const float qv = 1.0f - p;
            const bool t1 = dy > 0.0f;                          // target 1 (lo has the higher grade) or 0; ties: w == 0
            const float lg = fmaxf(fast_ln(t1 ? p : qv), -100.0f);  // argument is 0 or a normal float <= 1
            lacc = fmaf(-w, lg, lacc);
            const float pm = t1 ? -qv : (qv == 0.0f ? 0.0f : p);   // (p - t), zero once p(1-p) underflows
            lam = (sigma * w) * pm;
        } else {
            const float S = fminf(fmaxf(dy, -1.0f), 1.0f);      // lambda_utils.py:20
            const float tt = 0.5f * (1.0f + S);
            const float x = sigma * ds;
            float p, qv, l1, l0, gp;                            // gp = (p - t) * sigmoid'(x) / max(p (1 - p), 1e-12)
            if (__any(!(fabsf(x) <= 80.0f))) {
                // score gaps this wide reach the reference's edge arithmetic (p denormal or rounded to 0 / 1, BCE's -100 clamp, the
                // 1e-12 floor under p(1-p)): the library functions and IEEE divisions, as the reference's ATen kernels compute it
                p = 1.0f / (1.0f + expf(-x));
                qv = 1.0f - p;
                l1 = fmaxf(logf(p), -100.0f); l0 = fmaxf(logf(qv), -100.0f);
                const float den = qv * p;
                gp = ((p - tt) / fmaxf(den, 1e-12f)) * den;
            } else {
                // |x| <= 80 for every pair of this wavefront step: e = exp(-|x|) is a normal float, the larger probability is 1/(1+e) by
                // v_rcp + one Newton step (the LambdaRank kernels' form), the smaller one its product with e; logs on the transcendental
                // pipe (1 ulp: 1e-7 absolute near 1, against per-query losses of 10^2); the 1e-12 floor under p (1 - p) (reached from
                // |x| > 27.6 on) becomes a factor instead of a division
                const float e = __expf(-fabsf(x));
                const float dd = 1.0f + e;
                float pb = __builtin_amdgcn_rcpf(dd);
                pb = fmaf(pb, fmaf(-dd, pb, 1.0f), pb);
                p = x >= 0.0f ? pb : e * pb;
                qv = 1.0f - p;
                l1 = fmaxf(fast_ln(p), -100.0f); l0 = fmaxf(fast_ln(qv), -100.0f);
                const float den = qv * p;
                gp = (p - tt) * (den >= 1e-12f ? 1.0f : den * 1e12f);
            }
            lacc += ((tt - 1.0f) * l0 - tt * l1) * act;
            lam = (sigma * gp) * act;
        }
        const float sl = fwd ? lam : -lam;
        ga[m] += sl;
        if (act != 0.0f) gw[b] = gold - sl;                     // active lanes hit distinct b: race-free; idle lanes (clamped
                                                                // index) must not write back a stale value
    };

    int aeff[DPT];
    float act[DPT];
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int a = t + m * G;
        act[m] = a < n ? 1.0f : 0.0f;
        aeff[m] = a < n ? a : 0;
    }
    const int half = (n - 1) >> 1;
    if (n > 1) {
        for (int d = 1; d <= half; ++d) {
#pragma unroll
            for (int m = 0; m < DPT; ++m) pair(m, aeff[m], d, act[m]);
        }
        if ((n & 1) == 0) {
            const int d = n >> 1;
#pragma unroll
            for (int m = 0; m < DPT; ++m) {
                const int a = t + m * G;
                pair(m, a < d ? a : 0, d, a < d ? 1.0f : 0.0f);
            }
        }
    }
    __syncthreads();

    // ---- phase D: combine, un-sort, store
#pragma unroll
    for (int m = 0; m < DPT; ++m) {
        const int a = t + m * G;
        if (a < n) {
            float tot = ga[m];
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += gacc[(size_t)w * Lp + a];
            keys[a] = tot;
        }
    }
    __syncthreads();
    const float loss = group_sum<G>(lacc, red, t);
    if (valid) {
#pragma unroll
        for (int m = 0; m < DPT; ++m) {
            const int i = t + m * G;
            if (i < L) grad[(size_t)q * L + i] = i < n ? keys[rk[m]] : 0.0f;
        }
        if (t == 0) loss_q[q] = loss;
    }
}


static int ring_waves() {                         // PTR_RING_WAVES=1/2/4/8/16 pins the waves per workgroup (measurements)
    const char *e = getenv("PTR_RING_WAVES");
    const int v = e ? atoi(e) : 0;
    return v == 1 || v == 2 || v == 4 || v == 8 || v == 16 ? v : 0;
}
static int ring_num_cus() {
    static int n = 0;
    if (!n) { int dev = 0; hipDeviceProp_t pr; n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }
    return n;
}
static int ring_enabled() {                       // PTR_LAMBDARANK_RING=0 selects the LDS kernel (A/B measurements, tests)
    const char *e = getenv("PTR_LAMBDARANK_RING");
    return e ? (atoi(e) != 0) : 1;
}

template <bool WEIGHTED>
static int launch_pairwise(const float *preds, const float *labels, const int32_t *lens, int B, int L, float sigma,
                           float *loss_out, float *loss_q, float *grad, void *stream, const char *who) {
    if (int rc = check_batch(preds, labels, B, L, who)) return rc;
    if (B > 0 && (!loss_q || !grad)) { set_error("%s: NULL output pointer", who); return PTR_ERR_INVALID_ARG; }
    if (WEIGHTED && !(sigma >= 0.0f)) { set_error("%s: sigma must be >= 0 (got %g)", who, (double)sigma); return PTR_ERR_INVALID_ARG; }
    hipStream_t st = as_stream(stream);
    if (WEIGHTED && B > 0 && L <= 512 && sigma > 0.0f && ring_enabled()) {
        const int dpt = L <= 64 ? 1 : L <= 128 ? 2 : L <= 256 ? 4 : 8;
        int QPB = ring_waves();
        if (!QPB) { QPB = kRingBlock / kWave; while (QPB > 1 && B < QPB * ring_num_cus()) QPB >>= 1; }
        if (dpt >= 4 && QPB > PTR_RING4_WAVES) QPB = PTR_RING4_WAVES;
        if (dpt >= 8 && QPB > 4) QPB = 4;
        if (dpt < 8) {                            // pairwise_ring.hip
            if (int e = launch_lambdarank_ring_small(dpt, QPB, preds, labels, lens, B, L, sigma, loss_q, grad, st)) return check_hip((hipError_t)e, who);
        } else {
            const size_t lds = (size_t)QPB * 2 * 64 * dpt * sizeof(float);
            hipLaunchKernelGGL(lambdarank_ring_kernel<8>, dim3((B + QPB - 1) / QPB), dim3(QPB * kWave), lds, st, preds, labels, lens, B, L, sigma, loss_q, grad);
            if (int rc = check_hip(hipGetLastError(), who)) return rc;
        }
    } else if (!WEIGHTED && B > 0 && L <= 32) {
        if constexpr (!WEIGHTED) {               // RankNet on short lists: two queries per wavefront
            const int Lp = round_up(L, 4);
            constexpr int QPB = kBlock / 32;
            auto kern = pairwise_bce_kernel<32, 1, false>;
            const size_t lds = QPB * pairwise_group_floats(Lp, 1) * sizeof(float);
            hipLaunchKernelGGL(kern, dim3((B + QPB - 1) / QPB), dim3(kBlock), lds, st, preds, labels, lens, B, L, Lp, sigma, loss_q, grad);
            if (int rc = check_hip(hipGetLastError(), who)) return rc;
        }
    } else if (B > 0) {
        const int Lp = round_up(L, 4);
        int rc = dispatch_tiling(L, [&]<int G, int DPT>() -> int {
            constexpr int QPB = kBlock / G, NW = G / kWave;
            auto kern = pairwise_bce_kernel<G, DPT, WEIGHTED>;
            const size_t lds = QPB * pairwise_group_floats(Lp, NW) * sizeof(float);
            if (int e = allow_lds(kern, lds)) return e;
            hipLaunchKernelGGL(kern, dim3((B + QPB - 1) / QPB), dim3(kBlock), lds, st, preds, labels, lens, B, L, Lp, sigma,
                               loss_q, grad);
            return check_hip(hipGetLastError(), who);
        });
        if (rc) return rc;
    }
    return loss_out ? ptr_sum_f32(loss_q, B, 1.0f, loss_out, stream) : 0;
}

}  // namespace ptr

extern "C" int ptr_ranknet_fwd_bwd(const float *preds, const float *labels, const int32_t *lens, int B, int L, float sigma,
                                   float *loss_out, float *loss_q, float *grad, void *stream) {
    return ptr::launch_pairwise<false>(preds, labels, lens, B, L, sigma, loss_out, loss_q, grad, stream, "ptr_ranknet_fwd_bwd");
}

extern "C" int ptr_lambdarank_fwd_bwd(const float *preds, const float *labels, const int32_t *lens, int B, int L, float sigma,
                                      float *loss_out, float *loss_q, float *grad, void *stream) {
    return ptr::launch_pairwise<true>(preds, labels, lens, B, L, sigma, loss_out, loss_q, grad, stream, "ptr_lambdarank_fwd_bwd");
}
