/*
 * ORACLE — test infrastructure only.  Never linked into, imported by or called from the product
 * path (ptranking_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Plain-C, single-threaded, fp32 restatement of wildltr/ptranking's ltr_adhoc loss + metric hot
 * path with CLOSED-FORM gradients, independent of torch/autograd.  Layout
 * is the padded (B, L) row-major batch used by the HIP kernels, with an optional per-query length
 * array `lens` (NULL => every list has L documents; the reference itself never pads,
 * ptranking/data/data_utils.py:683-742).
 *
 * Parity pin: tests/test_oracle_golden.py::test_c_oracle_* compares every function below with
 * tests/golden/{losses,metrics}.npz, which were produced by RUNNING THE REFERENCE
 * (tests/golden/make_golden.py), including the five known-answer vectors of the reference's own
 * testing/metric/testing_metric.py:17-61.
 *
 * Arithmetic notes (all inherited from what the reference executes in ATen on CPU):
 *   - torch.sigmoid:               y = 1/(1+exp(-x)) rounded to fp32
 *   - F.binary_cross_entropy:      -(t*max(log(y),-100) + (1-t)*max(log(1-y),-100)) * weight
 *     backward:                    w*(y-t)/max((1-y)*y, 1e-12), then sigmoid' = (1-y)*y
 *   - sort order:                  (score descending, original index ascending)
 *   - torch.sum:                   cascade / vectorised summation, far more accurate than a sequential fp32 loop, so
 *                                  the per-query LOSS totals below are accumulated in double and rounded once; the
 *                                  per-pair terms themselves stay fp32
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_EINVAL 1
#define ORC_ENOMEM 2

static inline int qlen(const int32_t *lens, int q, int L) {
    if (!lens) return L;
    int n = lens[q];
    return n < 0 ? 0 : (n > L ? L : n);
}

/* ---- stable descending argsort: (value desc, index asc) -------------------------------------- */
typedef struct { float v; int32_t i; } kv_t;
static int kv_cmp_desc(const void *a, const void *b) {
    const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
    if (x->v > y->v) return -1;
    if (x->v < y->v) return 1;
    return (x->i > y->i) - (x->i < y->i);
}
static void argsort_desc(const float *v, int n, kv_t *tmp, int32_t *idx) {
    for (int i = 0; i < n; ++i) { tmp[i].v = v[i]; tmp[i].i = i; }
    qsort(tmp, (size_t)n, sizeof(kv_t), kv_cmp_desc);
    for (int i = 0; i < n; ++i) idx[i] = tmp[i].i;
}

/* torch.sort(preds, descending=True) as ptranking/base/ranker.py:50 uses it. */
int orc_sort_desc(const float *preds, const int32_t *lens, int B, int L, float *vals, int64_t *idx) {
    kv_t *tmp = (kv_t *)malloc(sizeof(kv_t) * (size_t)(L > 0 ? L : 1));
    int32_t *ix = (int32_t *)malloc(sizeof(int32_t) * (size_t)(L > 0 ? L : 1));
    if (!tmp || !ix) { free(tmp); free(ix); return ORC_ENOMEM; }
    for (int q = 0; q < B; ++q) {
        int n = qlen(lens, q, L);
        argsort_desc(preds + (size_t)q * L, n, tmp, ix);
        for (int r = 0; r < L; ++r) {
            if (r < n) { vals[(size_t)q * L + r] = preds[(size_t)q * L + ix[r]]; idx[(size_t)q * L + r] = ix[r]; }
            else       { vals[(size_t)q * L + r] = 0.0f;                         idx[(size_t)q * L + r] = r; }
        }
    }
    free(tmp); free(ix);
    return ORC_OK;
}

static inline float gain(float l) { return exp2f(l) - 1.0f; }           /* adhoc_metric.py:208-209 */
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* DCG over the whole list, ptranking/metric/adhoc/adhoc_metric.py:197-217 (cutoff=None). */
static float dcg_all(const float *labels, int n) {
    float acc = 0.0f;
    for (int r = 0; r < n; ++r) acc += gain(labels[r]) / log2f((float)r + 2.0f);
    return acc;
}

/* One BCE pair as torch evaluates it; returns the loss term and dL/dx (x = sigma*s_ij is the sigmoid
 * input, the caller multiplies by sigma).  ptranking/ltr_adhoc/pairwise/ranknet.py:34-35,
 * ptranking/ltr_adhoc/listwise/lambdarank.py:52-54. */
static inline void bce_pair(float x, float t, float w, float *loss, float *dx) {
    float y = sigmoidf_(x);
    float l1 = logf(y);        if (l1 < -100.0f) l1 = -100.0f;
    float l0 = logf(1.0f - y); if (l0 < -100.0f) l0 = -100.0f;
    *loss = ((t - 1.0f) * l0 - t * l1) * w;
    float den = (1.0f - y) * y;
    float g = w * (y - t) / (den > 1e-12f ? den : 1e-12f);
    *dx = g * den;
}

/* RankNet — ptranking/ltr_adhoc/pairwise/ranknet.py:25-42 + util/lambda_utils.py:5-23.
 * Input order, pairs i<j, ties contribute with target 0.5. */
int orc_ranknet(const float *preds, const float *labels, const int32_t *lens, int B, int L, float sigma,
                float *loss_q, float *grad) {
    for (int q = 0; q < B; ++q) {
        const float *s = preds + (size_t)q * L, *y = labels + (size_t)q * L;
        float *g = grad + (size_t)q * L;
        int n = qlen(lens, q, L);
        memset(g, 0, sizeof(float) * (size_t)L);
        double loss = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) {
                float S = y[i] - y[j]; S = S > 1.0f ? 1.0f : (S < -1.0f ? -1.0f : S);
                float t = 0.5f * (1.0f + S), l, dx;
                bce_pair(sigma * (s[i] - s[j]), t, 1.0f, &l, &dx);
                loss += l; g[i] += sigma * dx; g[j] -= sigma * dx;
            }
        loss_q[q] = (float)loss;
    }
    return ORC_OK;
}

/* LambdaRank — ptranking/ltr_adhoc/listwise/lambdarank.py:27-62 with get_delta_ndcg
 * (ptranking/metric/metric_utils.py:19-45).  Labels are taken as the ideal ranking in input order
 * (the reference asserts presort=True, lambdarank.py:36). */
int orc_lambdarank(const float *preds, const float *labels, const int32_t *lens, int B, int L, float sigma,
                   float *loss_q, float *grad) {
    size_t Ls = (size_t)(L > 0 ? L : 1);
    kv_t *tmp = (kv_t *)malloc(sizeof(kv_t) * Ls);
    int32_t *ix = (int32_t *)malloc(sizeof(int32_t) * Ls);
    float *buf = (float *)malloc(sizeof(float) * Ls * 4);
    if (!tmp || !ix || !buf) { free(tmp); free(ix); free(buf); return ORC_ENOMEM; }
    float *ss = buf, *G = buf + Ls, *D = buf + 2 * Ls, *gs = buf + 3 * Ls;
    for (int q = 0; q < B; ++q) {
        const float *s = preds + (size_t)q * L, *y = labels + (size_t)q * L;
        float *g = grad + (size_t)q * L;
        int n = qlen(lens, q, L);
        memset(g, 0, sizeof(float) * (size_t)L);
        argsort_desc(s, n, tmp, ix);
        float idcg = dcg_all(y, n);
        for (int r = 0; r < n; ++r) {
            ss[r] = s[ix[r]];
            G[r] = gain(y[ix[r]]) / idcg;                       /* metric_utils.py:35 */
            D[r] = 1.0f / log2f((float)r + 2.0f);               /* metric_utils.py:39 */
            gs[r] = 0.0f;
        }
        double loss = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) {
                float li = y[ix[i]], lj = y[ix[j]];
                float S = li - lj; S = S > 1.0f ? 1.0f : (S < -1.0f ? -1.0f : S);
                float t = 0.5f * (1.0f + S);
                float w = fabsf(G[i] - G[j]) * fabsf(D[i] - D[j]);
                float l, dx;
                bce_pair(sigma * (ss[i] - ss[j]), t, w, &l, &dx);
                loss += l; gs[i] += sigma * dx; gs[j] -= sigma * dx;
            }
        for (int r = 0; r < n; ++r) g[ix[r]] = gs[r];
        loss_q[q] = (float)loss;
    }
    free(tmp); free(ix); free(buf);
    return ORC_OK;
}

/* LambdaLoss NDCG_Loss2 (loss_type 1) / NDCG_Loss2++ (loss_type 2) —
 * ptranking/ltr_adhoc/listwise/lambdaloss.py:36-58,83-132; epsilon = 1e-8 (ptranking/ltr_global.py:10).
 * The reference's discount table is inverted twice (SURVEY.md §7 iii): inv[r] = (1/log2(r+2))^-1. */
int orc_lambdaloss(const float *preds, const float *labels, const int32_t *lens, int B, int L, int k, float sigma,
                   float mu, int loss_type, int presort, float *loss_q, float *grad) {
    if (loss_type != 0 && loss_type != 1 && loss_type != 2) return ORC_EINVAL;
    size_t Ls = (size_t)(L > 0 ? L : 1);
    kv_t *tmp = (kv_t *)malloc(sizeof(kv_t) * Ls);
    int32_t *il = (int32_t *)malloc(sizeof(int32_t) * Ls), *ip = (int32_t *)malloc(sizeof(int32_t) * Ls);
    float *buf = (float *)malloc(sizeof(float) * Ls * 6);
    if (!tmp || !il || !ip || !buf) { free(tmp); free(il); free(ip); free(buf); return ORC_ENOMEM; }
    float *tp = buf, *ideal = buf + Ls, *ss = buf + 2 * Ls, *G = buf + 3 * Ls, *inv = buf + 4 * Ls, *gs = buf + 5 * Ls;
    const float eps = 1e-8f, ln2 = 0.6931471805599453f;
    for (int q = 0; q < B; ++q) {
        const float *s = preds + (size_t)q * L, *y = labels + (size_t)q * L;
        float *g = grad + (size_t)q * L;
        int n = qlen(lens, q, L);
        memset(g, 0, sizeof(float) * (size_t)L);
        if (presort) { for (int r = 0; r < n; ++r) il[r] = r; }      /* lambdaloss.py:83-84 */
        else argsort_desc(y, n, tmp, il);                            /* lambdaloss.py:86-87 */
        for (int r = 0; r < n; ++r) { tp[r] = s[il[r]]; ideal[r] = y[il[r]]; }
        argsort_desc(tp, n, tmp, ip);                                /* lambdaloss.py:89 */
        float idcg = dcg_all(ideal, n);
        for (int r = 0; r < n; ++r) {
            ss[r] = tp[ip[r]];
            G[r] = gain(ideal[ip[r]]) / idcg;
            inv[r] = powf(1.0f / log2f((float)r + 2.0f), -1.0f);
            gs[r] = 0.0f;
        }
        int kk = k < n ? k : n;
        double loss = 0.0;
        for (int i = 0; i < kk; ++i)
            for (int j = 0; j < kk; ++j) {
                float w;
                if (loss_type == 0) {
                    /* NDCG_Loss1 (lambdaloss.py:33-34,108-109,130): every entry of the k x k block, diagonal included, weight
                     * of the column: G_j / discount_j (the reference's [B,L] weights broadcast onto the last axis at B = 1) */
                    w = G[j] / (1.0f / log2f((float)j + 2.0f));
                } else {
                    if (i == j) continue;
                    if (!(ideal[ip[i]] - ideal[ip[j]] > 0.0f)) continue;   /* lambdaloss.py:127-128 */
                    int d = i > j ? i - j : j - i;
                    float delta = fabsf(inv[d - 1] - inv[d]);
                    w = delta * fabsf(G[i] - G[j]);
                    if (loss_type == 2) w = (fabsf(inv[i] - inv[j]) + mu * delta) * fabsf(G[i] - G[j]);
                }
                float df = ss[i] - ss[j];
                if (df > 1e8f) df = 1e8f; if (df < -1e8f) df = -1e8f; if (df != df) df = 0.0f;
                float p0 = sigmoidf_(sigma * df);
                float p = p0 > eps ? p0 : eps;
                float wp0 = powf(p, w);
                float wp = wp0 > eps ? wp0 : eps;
                loss += -log2f(wp);
                if (i != j && p0 >= eps && wp0 >= eps) {             /* clamp(min) passes gradient at equality */
                    float dls = -(1.0f / (wp * ln2)) * (w * powf(p, w - 1.0f)) * ((1.0f - p0) * p0) * sigma;
                    gs[i] += dls; gs[j] -= dls;
                }
            }
        for (int r = 0; r < n; ++r) g[il[ip[r]]] = gs[r];
        loss_q[q] = (float)loss;
    }
    free(tmp); free(il); free(ip); free(buf);
    return ORC_OK;
}

/* SoftRank — ptranking/ltr_adhoc/listwise/softrank.py:47-69 (labels in ideal order, the reference asserts presort).
 * E[rank_i] = 1 + sum_{j != i} 0.5*erfc((s_i - s_j)/sqrt(2*2*delta^2)); loss_q = -sum_{i<k} g_i/(log2(E_i + 1)*IDCG_q).
 * Closed-form gradient: c_i = g_i/(IDCG*ln2*(1+E_i)*log2(1+E_i)^2) (i < k, else 0), phi_ij = exp(-x_ij^2)/(sqrt(pi)*den),
 * dL/ds_m = sum_{i != m} phi_im*(c_i - c_m). */
int orc_softrank(const float *preds, const float *labels, const int32_t *lens, int B, int L, float delta, int top_k,
                 float *loss_q, float *grad) {
    size_t Ls = (size_t)(L > 0 ? L : 1);
    float *E = (float *)malloc(sizeof(float) * Ls * 2);
    if (!E) return ORC_ENOMEM;
    float *c = E + Ls;
    const float var = 2.0f * (delta * delta);
    const float den = sqrtf(2.0f * var);
    const float ln2 = 0.6931471805599453f, inv_sqrt_pi = 0.5641895835477563f;
    for (int q = 0; q < B; ++q) {
        const float *s = preds + (size_t)q * L, *y = labels + (size_t)q * L;
        float *g = grad + (size_t)q * L;
        int n = qlen(lens, q, L);
        memset(g, 0, sizeof(float) * (size_t)L);
        int k = (top_k > 0 && top_k < n) ? top_k : n;
        float idcg = dcg_all(y, n);
        double loss = 0.0;
        for (int i = 0; i < n; ++i) {
            double acc = 0.0;
            for (int j = 0; j < n; ++j) if (j != i) acc += 0.5f * erfcf((s[i] - s[j]) / den);
            E[i] = (float)acc + 1.0f;
            c[i] = 0.0f;
            if (i < k) {
                float lg = log2f(E[i] + 1.0f);
                loss -= (double)((1.0f / lg) * gain(y[i]) / idcg);
                c[i] = gain(y[i]) / (idcg * ln2 * (1.0f + E[i]) * lg * lg);
            }
        }
        for (int m = 0; m < n; ++m) {
            double acc = 0.0;
            for (int i = 0; i < n; ++i) {
                if (i == m) continue;
                float x = (s[i] - s[m]) / den;
                acc += (double)(inv_sqrt_pi / den * expf(-x * x)) * (double)(c[i] - c[m]);
            }
            g[m] = (float)acc;
        }
        loss_q[q] = (float)loss;
    }
    free(E);
    return ORC_OK;
}

/* MDPRank — ptranking/ltr_adhoc/listwise/mdprank.py:46-75 on a given sampled ranking perm (first len_q entries of row q).
 * G_t = gamma^(t+1) * sum_{t'=t}^{top-1} (2^l - 1)/log2(2 + t'); loss_q = sum_{t<top} G_t*(log sum_{j>=t} exp(u_j) - u_t);
 * dL/du_j = e^{u_j - m} * sum_{i <= min(j, top-1)} G_i/T_i - (j < top ? G_j : 0). */
int orc_mdprank(const float *preds, const float *labels, const int64_t *perm, const int32_t *lens, int B, int L, int top_k, float gamma,
                float *loss_q, float *grad) {
    size_t Ls = (size_t)(L > 0 ? L : 1);
    double *T = (double *)malloc(sizeof(double) * Ls * 2);
    if (!T) return ORC_ENOMEM;
    double *W = T + Ls;
    for (int q = 0; q < B; ++q) {
        const float *s = preds + (size_t)q * L, *y = labels + (size_t)q * L;
        const int64_t *p = perm + (size_t)q * L;
        float *g = grad + (size_t)q * L;
        int n = qlen(lens, q, L);
        memset(g, 0, sizeof(float) * (size_t)L);
        int top = (top_k <= 0 || top_k > n) ? n : top_k;
        float m = -INFINITY;
        for (int k = 0; k < n; ++k) if (s[p[k]] > m) m = s[p[k]];
        double tail = 0.0, rtail = 0.0, loss = 0.0;
        for (int k = n - 1; k >= 0; --k) {
            tail += (double)expf(s[p[k]] - m);
            T[k] = tail;
            if (k < top) rtail += (double)(gain(y[p[k]]) / log2f(2.0f + (float)k));
            W[k] = k < top ? (double)((float)rtail * (gamma == 1.0f ? 1.0f : powf(gamma, (float)(k + 1)))) : 0.0;
            if (k < top) loss += W[k] * (double)((logf((float)T[k]) + m) - s[p[k]]);
        }
        double pre = 0.0;
        for (int k = 0; k < n; ++k) {
            pre += W[k] / T[k];
            g[p[k]] = (float)((double)expf(s[p[k]] - m) * pre - W[k]);
        }
        loss_q[q] = (float)loss;
    }
    free(T);
    return ORC_OK;
}

/* Robust_Sigmoid forward — ptranking/base/utils.py:57-81. */
static inline float robust_sigmoid(float x_in, float sigma) {
    float x = sigma * x_in;
    if (x_in > 0.0f) return 1.0f / (1.0f + expf(-x));
    if (x_in < 0.0f) { float e = expf(x); return e / (1.0f + e); }
    return 0.5f;
}

/* ApproxNDCG — ptranking/ltr_adhoc/listwise/approxNDCG.py:19-27,45-62,83-109.
 * couple_batch != 0 reproduces the reference's [B]/[B,1] broadcast: loss = -(sum_b DCG_b)*(sum_a 1/IDCG_a) and
 * every gradient is scaled by S = sum_a 1/IDCG_a (SURVEY.md §7 vi).  couple_batch == 0 is the per-query form.
 * Outputs: dcg_q[B] (approximate DCG per query), inv_idcg_q[B], loss_total[1], grad[B,L]. */
int orc_approxndcg(const float *preds, const float *labels, const int32_t *lens, int B, int L, float alpha,
                   int presort, int couple_batch, float *loss_total, float *dcg_q, float *inv_idcg_q, float *grad) {
    size_t Ls = (size_t)(L > 0 ? L : 1);
    kv_t *tmp = (kv_t *)malloc(sizeof(kv_t) * Ls);
    int32_t *il = (int32_t *)malloc(sizeof(int32_t) * Ls);
    float *buf = (float *)malloc(sizeof(float) * Ls * 4);
    if (!tmp || !il || !buf) { free(tmp); free(il); free(buf); return ORC_ENOMEM; }
    float *tp = buf, *ideal = buf + Ls, *c = buf + 2 * Ls, *gs = buf + 3 * Ls;
    const float ln2 = 0.6931471805599453f;
    float S = 0.0f;
    for (int q = 0; q < B; ++q) {
        int n = qlen(lens, q, L);
        const float *y = labels + (size_t)q * L;
        if (presort) { inv_idcg_q[q] = 1.0f / dcg_all(y, n); }
        else {
            argsort_desc(y, n, tmp, il);
            for (int r = 0; r < n; ++r) ideal[r] = y[il[r]];
            inv_idcg_q[q] = 1.0f / dcg_all(ideal, n);
        }
        S += inv_idcg_q[q];
    }
    double sum_dcg = 0.0, sum_ndcg = 0.0;
    for (int q = 0; q < B; ++q) {
        const float *s = preds + (size_t)q * L, *y = labels + (size_t)q * L;
        float *g = grad + (size_t)q * L;
        int n = qlen(lens, q, L);
        memset(g, 0, sizeof(float) * (size_t)L);
        if (presort) { for (int r = 0; r < n; ++r) il[r] = r; }
        else argsort_desc(y, n, tmp, il);
        for (int r = 0; r < n; ++r) { tp[r] = s[il[r]]; ideal[r] = y[il[r]]; }
        float scale = couple_batch ? S : inv_idcg_q[q];
        double dcg = 0.0;
        for (int i = 0; i < n; ++i) {
            double pid = 0.0;
            for (int j = 0; j < n; ++j) pid += robust_sigmoid(tp[j] - tp[i], alpha);   /* approxNDCG.py:21-25 */
            float pi = (float)pid + 0.5f;
            float lg = log2f(pi + 1.0f);
            float gi = gain(ideal[i]);
            dcg += gi / lg;
            /* d(-g/log2(1+pi) * scale)/d pi = +g*scale / (ln2 (1+pi) log2(1+pi)^2) */
            c[i] = gi * scale / (ln2 * (1.0f + pi) * lg * lg);
            gs[i] = 0.0f;
        }
        /* pi_i depends on s_j (+) and s_i (-) through sigma(alpha (s_j - s_i)); derivative alpha*y*(1-y) */
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                if (i == j) continue;
                float yv = robust_sigmoid(tp[j] - tp[i], alpha);
                float d = alpha * yv * (1.0f - yv);
                gs[j] += c[i] * d; gs[i] -= c[i] * d;
            }
        for (int r = 0; r < n; ++r) g[il[r]] = gs[r];
        dcg_q[q] = (float)dcg;
        sum_dcg += dcg_q[q]; sum_ndcg += dcg_q[q] * inv_idcg_q[q];
    }
    *loss_total = couple_batch ? -((float)sum_dcg * S) : -(float)sum_ndcg;
    free(tmp); free(il); free(buf);
    return ORC_OK;
}

/* ListNet — ptranking/ltr_adhoc/listwise/listnet.py:39: -sum softmax(labels)*log_softmax(preds);
 * gradient softmax(preds) - softmax(labels). */
int orc_listnet(const float *preds, const float *labels, const int32_t *lens, int B, int L, float *loss_q, float *grad) {
    for (int q = 0; q < B; ++q) {
        const float *s = preds + (size_t)q * L, *y = labels + (size_t)q * L;
        float *g = grad + (size_t)q * L;
        int n = qlen(lens, q, L);
        memset(g, 0, sizeof(float) * (size_t)L);
        if (n == 0) { loss_q[q] = 0.0f; continue; }
        float ms = s[0], my = y[0];
        for (int i = 1; i < n; ++i) { if (s[i] > ms) ms = s[i]; if (y[i] > my) my = y[i]; }
        double zsd = 0.0, zyd = 0.0, loss = 0.0;
        for (int i = 0; i < n; ++i) { zsd += expf(s[i] - ms); zyd += expf(y[i] - my); }
        float zs = (float)zsd, zy = (float)zyd;
        float lzs = logf(zs);
        for (int i = 0; i < n; ++i) {
            float py = expf(y[i] - my) / zy;
            float lsm = (s[i] - ms) - lzs;
            loss -= py * lsm;
            g[i] = expf(lsm) - py;
        }
        loss_q[q] = (float)loss;
    }
    return ORC_OK;
}

/* STListNet — ptranking/ltr_adhoc/listwise/st_listnet.py:41-49 with the uniform draws `unif` supplied by the caller. */
int orc_stlistnet(const float *preds, const float *labels, const float *unif, const int32_t *lens, int B, int L, float temperature,
                  float *loss_q, float *grad) {
    float *z = (float *)malloc(sizeof(float) * (size_t)(L > 0 ? L : 1));
    if (!z) return ORC_ENOMEM;
    for (int q = 0; q < B; ++q) {
        const float *s = preds + (size_t)q * L, *y = labels + (size_t)q * L, *u = unif + (size_t)q * L;
        float *g = grad + (size_t)q * L;
        int n = qlen(lens, q, L);
        memset(g, 0, sizeof(float) * (size_t)L);
        if (n == 0) { loss_q[q] = 0.0f; continue; }
        float mz = -INFINITY, my = -INFINITY;
        for (int i = 0; i < n; ++i) {
            z[i] = (s[i] + -logf(-logf(u[i] + 1e-20f) + 1e-20f)) / temperature;
            if (z[i] > mz) mz = z[i];
            if (y[i] > my) my = y[i];
        }
        double zsd = 0.0, zyd = 0.0, loss = 0.0;
        for (int i = 0; i < n; ++i) { zsd += expf(z[i] - mz); zyd += expf(y[i] - my); }
        float lzs = logf((float)zsd), zy = (float)zyd;
        for (int i = 0; i < n; ++i) {
            float py = expf(y[i] - my) / zy, lsm = (z[i] - mz) - lzs;
            loss -= py * lsm;
            g[i] = (expf(lsm) - py) / temperature;
        }
        loss_q[q] = (float)loss;
    }
    free(z);
    return ORC_OK;
}

/* RankMSE — ptranking/ltr_adhoc/pointwise/rank_mse.py:13-22: loss_q = per-query sum of squared errors (the batch loss is
 * their MEAN over the B queries), grad = 2 (s - y) / B. */
int orc_rankmse(const float *preds, const float *labels, const int32_t *lens, int B, int L, float *loss_q, float *grad) {
    for (int q = 0; q < B; ++q) {
        int n = qlen(lens, q, L);
        double acc = 0.0;
        for (int i = 0; i < L; ++i) {
            size_t o = (size_t)q * L + i;
            grad[o] = 0.0f;
            if (i < n) { float d = preds[o] - labels[o]; acc += (double)d * d; grad[o] = 2.0f * d / (float)B; }
        }
        loss_q[q] = (float)acc;
    }
    return ORC_OK;
}

/* RankCosine — ptranking/ltr_adhoc/listwise/rank_cosine.py:15,32; nn.CosineSimilarity(dim=1, eps=1e-8). */
int orc_rankcosine(const float *preds, const float *labels, const int32_t *lens, int B, int L, float *loss_q, float *grad) {
    const float eps = 1e-8f;
    for (int q = 0; q < B; ++q) {
        const float *s = preds + (size_t)q * L, *y = labels + (size_t)q * L;
        float *g = grad + (size_t)q * L;
        int n = qlen(lens, q, L);
        double sy = 0.0, ss = 0.0, yy = 0.0;
        for (int i = 0; i < n; ++i) { sy += (double)s[i] * y[i]; ss += (double)s[i] * s[i]; yy += (double)y[i] * y[i]; }
        float ns = sqrtf((float)ss), ny = sqrtf((float)yy);
        float ds = ns > eps ? ns : eps, dy = ny > eps ? ny : eps;
        float c = (float)sy / (ds * dy);
        for (int i = 0; i < L; ++i)
            g[i] = i < n ? -2.0f * (y[i] / (ds * dy) - (ns > eps ? c * s[i] / (float)ss : 0.0f)) : 0.0f;
        loss_q[q] = (1.0f - c) / 0.5f;
    }
    return ORC_OK;
}

/* ListMLE — ptranking/ltr_adhoc/listwise/listmle.py:81-97 with the tie-shuffled permutation `perm`
 * (int64, what arg_shuffle_ties returns, ptranking/ltr_adhoc/util/sampling_utils.py:13-28) supplied by the caller. */
int orc_listmle(const float *preds, const int64_t *perm, const int32_t *lens, int B, int L, float *loss_q, float *grad) {
    size_t Ls = (size_t)(L > 0 ? L : 1);
    float *buf = (float *)malloc(sizeof(float) * Ls * 3);
    if (!buf) return ORC_ENOMEM;
    float *u = buf, *e = buf + Ls, *T = buf + 2 * Ls;
    for (int q = 0; q < B; ++q) {
        const float *s = preds + (size_t)q * L;
        const int64_t *pi = perm + (size_t)q * L;
        float *g = grad + (size_t)q * L;
        int n = qlen(lens, q, L);
        memset(g, 0, sizeof(float) * (size_t)L);
        if (n == 0) { loss_q[q] = 0.0f; continue; }
        float m = -INFINITY;
        for (int i = 0; i < n; ++i) {
            if (pi[i] < 0 || pi[i] >= n) { free(buf); return ORC_EINVAL; }
            u[i] = s[pi[i]]; if (u[i] > m) m = u[i];
        }
        float acc = 0.0f;
        for (int i = n - 1; i >= 0; --i) { e[i] = expf(u[i] - m); acc += e[i]; T[i] = acc; }   /* flip-cumsum-flip */
        double loss = 0.0; float invsum = 0.0f;
        for (int i = 0; i < n; ++i) {
            loss += (logf(T[i]) + m) - u[i];
            invsum += 1.0f / T[i];
            g[pi[i]] = e[i] * invsum - 1.0f;
        }
        loss_q[q] = (float)loss;
    }
    free(buf);
    return ORC_OK;
}

/* Evaluator prologue (sort by prediction, gather labels, ideal ranking) + nDCG / nERR / AP / P at cut-offs ks —
 * ptranking/base/ranker.py:202-263 and ptranking/metric/adhoc/adhoc_metric.py:36-62,91-123,127-193,219-260.
 * Outputs are [B, nk] each (any may be NULL); cut-offs > list length are zero-filled AT THE END of the row, exactly
 * like the reference's padded_*_at_ks.  max_label < 0 => use the batch maximum (adhoc_metric.py:174-175). */
int orc_metrics_at_ks(const float *preds, const float *labels, const int32_t *lens, int B, int L, const int32_t *ks,
                      int nk, int presort, int linear_gain, float max_label, float *ndcg, float *nerr, float *ap, float *prec) {
    size_t Ls = (size_t)(L > 0 ? L : 1);
    kv_t *tmp = (kv_t *)malloc(sizeof(kv_t) * Ls);
    int32_t *ix = (int32_t *)malloc(sizeof(int32_t) * Ls);
    float *buf = (float *)malloc(sizeof(float) * Ls * 2);
    if (!tmp || !ix || !buf) { free(tmp); free(ix); free(buf); return ORC_ENOMEM; }
    float *sys = buf, *ideal = buf + Ls;
    int32_t *slot = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nk > 0 ? nk : 1));
    if (!slot) { free(tmp); free(ix); free(buf); return ORC_ENOMEM; }
    if (max_label < 0.0f) {
        max_label = -INFINITY;
        for (int q = 0; q < B; ++q) { int n = qlen(lens, q, L); for (int i = 0; i < n; ++i) if (labels[(size_t)q * L + i] > max_label) max_label = labels[(size_t)q * L + i]; }
    }
    float pow_max = exp2f(max_label);
    for (int q = 0; q < B; ++q) {
        const float *s = preds + (size_t)q * L, *y = labels + (size_t)q * L;
        int n = qlen(lens, q, L);
        argsort_desc(s, n, tmp, ix);
        for (int r = 0; r < n; ++r) sys[r] = y[ix[r]];
        if (presort) memcpy(ideal, y, sizeof(float) * (size_t)n);
        else { argsort_desc(y, n, tmp, ix); for (int r = 0; r < n; ++r) ideal[r] = y[ix[r]]; }
        int used = 0;   /* used_ks keep their relative order in the output row, the zero padding goes last */
        for (int c = 0; c < nk; ++c) {
            size_t o = (size_t)q * nk;
            if (ndcg) ndcg[o + c] = 0.0f; if (nerr) nerr[o + c] = 0.0f; if (ap) ap[o + c] = 0.0f; if (prec) prec[o + c] = 0.0f;
            slot[c] = (ks[c] >= 1 && ks[c] <= n) ? used++ : -1;
        }
        float sdcg = 0.0f, idcg = 0.0f, cumrel = 0.0f, cumprec = 0.0f, cumideal = 0.0f;
        float serr = 0.0f, ierr = 0.0f, sun = 1.0f, iun = 1.0f;
        for (int r = 0; r < n; ++r) {
            float disc = log2f((float)r + 2.0f);
            /* LABEL_TYPE.Permutation: the label is the gain (adhoc_metric.py:207-212) */
            sdcg += (linear_gain ? sys[r] : gain(sys[r])) / disc; idcg += (linear_gain ? ideal[r] : gain(ideal[r])) / disc;
            float rel = sys[r] < 0.0f ? 0.0f : (sys[r] > 1.0f ? 1.0f : sys[r]);
            cumrel += rel;
            float pr = cumrel / ((float)r + 1.0f);
            cumprec += pr * rel; cumideal += ideal[r];
            float ssat = gain(sys[r]) / pow_max, isat = gain(ideal[r]) / pow_max;
            serr += (1.0f / ((float)r + 1.0f)) * ssat * sun; sun *= (1.0f - ssat);
            ierr += (1.0f / ((float)r + 1.0f)) * isat * iun; iun *= (1.0f - isat);
            for (int c = 0; c < nk; ++c)
                if (ks[c] == r + 1) {
                    size_t o = (size_t)q * nk + (size_t)slot[c];
                    if (ndcg) ndcg[o] = sdcg / idcg; if (nerr) nerr[o] = serr / ierr;
                    if (ap) ap[o] = cumprec / cumideal; if (prec) prec[o] = pr;
                }
        }
    }
    free(tmp); free(ix); free(buf); free(slot);
    return ORC_OK;
}
