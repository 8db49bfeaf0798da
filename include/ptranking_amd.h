/*
 * ptranking_amd — C ABI of the MI355X-native ltr_adhoc loss / metric hot path.
 *
 * This is the drop-in boundary.  The reference (wildltr/ptranking) is pure Python and has no
 * FFI of its own; each entry point below replaces the ATen op sequence that one reference function executes, and
 * is what a `ctypes` binding on the reference side would bind (INTEGRATION.md shows that stub).  The Python host
 * layer in ptranking_amd/ mirrors the reference's plugin surface (`NeuralRanker.custom_loss_function`,
 * `Evaluator.*`) on top of these calls.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into HBM unless it is marked "host"; the caller owns every buffer,
 *     nothing is allocated, freed or retained by the library; no state survives a call;
 *   - batches are padded row-major (B, L) fp32: `preds[q*L + i]`, `labels[q*L + i]`; `lens` (int32[B], nullable)
 *     gives the number of real documents per query (NULL => L for every query); padded documents are excluded
 *     from every sum and receive gradient 0;
 *   - calls only ENQUEUE work on `stream` (a hipStream_t passed as void*, NULL = the legacy default stream) and
 *     never synchronise the host; results are run-to-run bit-stable (no floating-point atomics across waves);
 *   - return value: 0 on success, a hipError_t (> 0, < 1000) if the HIP runtime failed, or one of PTR_ERR_*;
 *     ptr_last_error() returns a thread-local message for the last non-zero return on this thread;
 *   - list lengths up to PTR_MAX_LIST_LEN are supported (per-query tiles live in LDS).
 */
#ifndef PTRANKING_AMD_H
#define PTRANKING_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTR_ABI_VERSION 5
#define PTR_MAX_LIST_LEN 4096
#define PTR_MAX_CUTOFFS 32
#define PTR_MLP_ACT_LD 112

#define PTR_ERR_INVALID_ARG 1001   /* NULL pointer, negative size, bad enum value               */
#define PTR_ERR_UNSUPPORTED 1002   /* L > PTR_MAX_LIST_LEN, nk > PTR_MAX_CUTOFFS, ...            */

#define PTR_LAMBDALOSS_NDCG_LOSS1 0    /* 'NDCG_Loss1'   (lambdaloss.py:33-34; the reference only runs it at batch size 1: its
                                         [B,L] weights broadcast against [B,L,L]; here every query uses its own w_j)  */
#define PTR_LAMBDALOSS_NDCG_LOSS2 1    /* 'NDCG_Loss2'   (ptranking/ltr_adhoc/listwise/lambdaloss.py:36-45) */
#define PTR_LAMBDALOSS_NDCG_LOSS2PP 2  /* 'NDCG_Loss2++' (ptranking/ltr_adhoc/listwise/lambdaloss.py:47-58) */

int ptr_abi_version(void);
const char *ptr_last_error(void);

/* RankNet — replaces ptranking/ltr_adhoc/pairwise/ranknet.py:32-36 (+ ltr_adhoc/util/lambda_utils.py:5-23) and
 * its autograd backward.  Pairs i<j in INPUT order, unweighted BCE, ties have target 0.5.
 *   loss_q[B] = per-query loss; grad[B,L]; loss_out[1] = sum over queries (nullable: NULL skips the reduction launch,
 *   the caller can run ptr_sum_f32 over loss_q itself).  Same convention for every *_fwd_bwd below. */
int ptr_ranknet_fwd_bwd(const float *preds, const float *labels, const int32_t *lens, int B, int L, float sigma,
                        float *loss_out, float *loss_q, float *grad, void *stream);

/* LambdaRank — replaces ptranking/ltr_adhoc/listwise/lambdarank.py:39-56 (torch.sort, gather,
 * get_pairwise_comp_probs, get_delta_ndcg = ptranking/metric/metric_utils.py:19-45, weighted BCE over the upper
 * triangle) and its backward, fused into one kernel.  `labels` must be in ideal (descending) order per query, as
 * the reference asserts (lambdarank.py:36).  sigma must be >= 0.  Ties in `preds` are ranked by original index. */
int ptr_lambdarank_fwd_bwd(const float *preds, const float *labels, const int32_t *lens, int B, int L, float sigma,
                           float *loss_out, float *loss_q, float *grad, void *stream);

/* LambdaLoss — replaces ptranking/ltr_adhoc/listwise/lambdaloss.py:83-132 and its backward.
 * loss_type: PTR_LAMBDALOSS_*; k = truncation (pairs with both ranks < k); mu only used by NDCG_Loss2++;
 * presort != 0 => labels already in ideal order (lambdaloss.py:83-84), else they are sorted first (:86-87). */
int ptr_lambdaloss_fwd_bwd(const float *preds, const float *labels, const int32_t *lens, int B, int L, int k,
                           float sigma, float mu, int loss_type, int presort, float *loss_out, float *loss_q,
                           float *grad, void *stream);

/* SoftRank — replaces ptranking/ltr_adhoc/listwise/softrank.py:47-69 and its backward: expected ranks from
 * 0.5*erfc((s_i-s_j)/sqrt(4*delta^2)), loss = -sum_q sum_{i<top_k} (2^l_i-1)/(log2(E[rank_i]+1)*IDCG_q).  Labels must be
 * in ideal order (the reference asserts presort).  top_k <= 0: no truncation (top_k=None).  delta > 0. */
int ptr_softrank_fwd_bwd(const float *preds, const float *labels, const int32_t *lens, int B, int L, float delta, int top_k,
                         float *loss_out, float *loss_q, float *grad, void *stream);

/* ApproxNDCG — replaces ptranking/ltr_adhoc/listwise/approxNDCG.py:19-27,45-62 (+ Robust_Sigmoid,
 * ptranking/base/utils.py:57-95) and its backward.  alpha must be > 0.
 * couple_batch != 0 reproduces the reference: loss = -(sum_b DCG_b) * S, S = sum_a 1/IDCG_a, gradients scaled by
 * S (SURVEY.md §7 vi); couple_batch == 0 gives the per-query normalised form -sum_b DCG_b/IDCG_b.
 * Outputs: loss_out[1]; dcg_q[B] (approximate DCG per query); inv_idcg_q[B]; grad[B,L] (fully scaled);
 * scale_out[2] = {factor applied to the gradients, local S}.  With couple_batch != 0 and grad_scale_override > 0 the
 * gradients and loss are scaled with that value instead of the local S (data parallel: pass 1.0, all-reduce the
 * local S with the parameter gradients and rescale them afterwards — they are linear in S). */
int ptr_approxndcg_fwd_bwd(const float *preds, const float *labels, const int32_t *lens, int B, int L, float alpha,
                           int presort, int couple_batch, float grad_scale_override, float *loss_out, float *dcg_q,
                           float *inv_idcg_q, float *scale_out, float *grad, void *stream);

/* ListNet — replaces ptranking/ltr_adhoc/listwise/listnet.py:39 and its backward. */
int ptr_listnet_fwd_bwd(const float *preds, const float *labels, const int32_t *lens, int B, int L, float *loss_out,
                        float *loss_q, float *grad, void *stream);

/* ---- sibling losses served by the same listwise machinery ----
 * STListNet — replaces ptranking/ltr_adhoc/listwise/st_listnet.py:41-49: ListNet on (preds + gumbel)/temperature with
 * gumbel = -log(-log(u + 1e-20) + 1e-20); `unif` [B,L] holds the uniform draws u (the reference: torch.rand). */
int ptr_stlistnet_fwd_bwd(const float *preds, const float *labels, const float *unif, const int32_t *lens, int B, int L,
                          float temperature, float *loss_out, float *loss_q, float *grad, void *stream);
/* RankMSE — replaces ptranking/ltr_adhoc/pointwise/rank_mse.py:13-22: mean over the B queries of sum_i (s_i - y_i)^2
 * (loss_q holds the per-query sums; loss_out and grad carry the 1/B). */
int ptr_rankmse_fwd_bwd(const float *preds, const float *labels, const int32_t *lens, int B, int L, float *loss_out,
                        float *loss_q, float *grad, void *stream);
/* RankCosine — replaces ptranking/ltr_adhoc/listwise/rank_cosine.py:15,32: sum_q (1 - cos(s_q, y_q)) / 0.5,
 * nn.CosineSimilarity(dim=1, eps=1e-8). */
int ptr_rankcosine_fwd_bwd(const float *preds, const float *labels, const int32_t *lens, int B, int L, float *loss_out,
                           float *loss_q, float *grad, void *stream);

/* ListMLE — replaces ptranking/ltr_adhoc/listwise/listmle.py:82,92-97 and its backward.  `perm` (int64[B,L]) is
 * the tie-shuffled label-descending order the reference obtains from arg_shuffle_ties
 * (ptranking/ltr_adhoc/util/sampling_utils.py:13-28); row q holds a permutation of 0..len_q-1 in its first len_q
 * entries. */
int ptr_listmle_fwd_bwd(const float *preds, const int64_t *perm, const int32_t *lens, int B, int L, float *loss_out,
                        float *loss_q, float *grad, void *stream);

/* MDPRank — replaces ptranking/ltr_adhoc/listwise/mdprank.py:46-75 and its backward: ListMLE on a SAMPLED ranking `perm`
 * (int64[B,L], drawn by the caller from the Plackett-Luce model, sampling_utils.py:32-83) whose first top_k positions are
 * weighted with the discounted long-term return G_t = gamma^(t+1) * sum_{t'=t}^{top_k-1} (2^l - 1)/log2(2 + t').
 * top_k <= 0: the whole list (top_k=None).  The reference only accepts batch size 1; here every query is independent. */
int ptr_mdprank_fwd_bwd(const float *preds, const float *labels, const int64_t *perm, const int32_t *lens, int B, int L, int top_k,
                        float gamma, float *loss_out, float *loss_q, float *grad, void *stream);

/* Device tie-shuffled label-descending order (the role of arg_shuffle_ties, sampling_utils.py:13-28) from a
 * counter-based RNG: same distribution, NOT the torch.randperm stream (not parity-checked, statistically tested). */
int ptr_shuffle_ties_order(const float *labels, const int32_t *lens, int B, int L, uint64_t seed, int64_t *perm,
                           void *stream);

/* torch.sort(preds, dim=1, descending=True) as ptranking/base/ranker.py:50 and lambdarank.py:39 call it:
 * vals[B,L] fp32, idx[B,L] int64; order = (value descending, original index ascending); padded tail: 0 / identity. */
int ptr_sort_desc(const float *preds, const int32_t *lens, int B, int L, float *vals, int64_t *idx, void *stream);

/* Evaluator prologue + metrics — replaces ptranking/base/ranker.py:46-60,220-243 (sort, gather, ideal sort) and
 * ptranking/metric/adhoc/adhoc_metric.py:36-62 (P@ks), :91-123 (AP@ks), :127-193 (nERR@ks), :219-260 (nDCG@ks).
 *   ks: HOST int32[nk] cut-offs (nk <= PTR_MAX_CUTOFFS); presort != 0 => labels already ideal-ordered;
 *   max_label: nERR's 2^max_label normaliser; < 0 => the batch maximum is computed on device into max_label_ws[1];
 *   label_type: PTR_LABEL_* (nDCG's gain, adhoc_metric.py:207-212; nERR exists for MultiLabel only, as in the reference);
 *   ndcg/nerr/ap/prec: [B,nk] outputs, each nullable.  Cut-offs larger than the list are zero-filled at the END
 *   of the row exactly like the reference's padded_*_at_ks. */
#define PTR_LABEL_MULTILABEL 0    /* LABEL_TYPE.MultiLabel: graded labels, DCG gain 2^l - 1 (data_utils.py:120-126)          */
#define PTR_LABEL_PERMUTATION 1   /* LABEL_TYPE.Permutation: labels = n - rank position, DCG gain = the label itself        */
int ptr_metrics_at_ks(const float *preds, const float *labels, const int32_t *lens, int B, int L, const int32_t *ks,
                      int nk, int presort, int label_type, float max_label, float *max_label_ws, float *ndcg, float *nerr,
                      float *ap, float *prec, void *stream);

/* Deterministic sum of n floats (fixed reduction tree): out[0] = scale * sum(x).  Used for the per-query loss slots
 * and for Evaluator running sums. */
int ptr_sum_f32(const float *x, int n, float scale, float *out, void *stream);

/* ---- pointwise MLP scorer (`pointsf`) --------------------------------------------------------------------------------
 * Replaces ptranking/base/point_ranker.py:45-55 (forward of the stacked feed-forward scorer built by
 * ptranking/base/utils.py:288-356 with AF='R', BN=False, apply_tl_af=False: (Dropout -> Linear -> ReLU) x NL -> Linear), its
 * autograd backward, and the Adam update of ptranking/base/ranker.py:512-525.  Hidden width is 100 (point_ranker.py:30).
 * `params` / `grad` are ONE flat fp32 buffer in PyTorch's own order and layouts:
 *   W1[100][F] b1[100] | W2[100][100] b2[100] | ... (NL hidden layers) | w_out[100] b_out[1]      (ptr_mlp_num_params floats)
 * X is [R][F] row-major (R = B*L documents), preds [R].  train != 0: dropout p_drop from the counter-based generator
 * seeded by `seed`, and the post-dropout activations needed by backward are written to `acts`: ptr_mlp_acts_floats(R, NL) floats,
 * opaque to the caller (ABI v4: TILE-MAJOR [NL][ceil(R / 16)][7][16][16] — every block of 16 rows x 16 of the PTR_MLP_ACT_LD = 112
 * padded features is one contiguous KB, the unit a forward store instruction writes and a backward LDS-DMA piece reads; element
 * (layer, row, col) at layer * ceil16(R) * 112 + (row / 16) * 1792 + (col / 16) * 256 + (row % 16) * 16 + col % 16).
 * SIZE IT WITH ptr_mlp_acts_floats(R, NL): the forward writes WHOLE 16-row tiles, so a buffer allocated as the pre-v4 [NL][R][112] is up to
 * 15 rows per layer too small whenever R % 16 != 0 (an out-of-bounds write).
 * ptr_mlp_backward: dpreds [R] -> grad (every entry overwritten); ws (ptr_mlp_backward_ws_floats) and dz (ptr_mlp_backward_dz_floats
 * floats: [NL][R][PTR_MLP_ACT_LD] for the layer-wise kernels, 0 => may be NULL when the single-pass fused backward serves the
 * configuration — NL = 3, F in {132, 136, 140}) are caller-provided scratch; p_drop / seed must be the forward call's.
 * All calls are deterministic. */
size_t ptr_mlp_num_params(int F, int NL);
size_t ptr_mlp_backward_ws_floats(int F, int NL);
size_t ptr_mlp_backward_dz_floats(int R, int F, int NL);
size_t ptr_mlp_acts_floats(int R, int NL);                       /* ABI v4: NL * ceil16(R) * 112 */
int ptr_mlp_forward(const float *X, const float *params, int R, int F, int NL, int train, float p_drop, uint64_t seed,
                    float *preds, float *acts, void *stream);
int ptr_mlp_backward(const float *X, const float *params, const float *acts, const float *dpreds, int R, int F, int NL,
                     float p_drop, uint64_t seed, float *dz, float *ws, float *grad, void *stream);
/* ptr_mlp_backward + the optimiser step + the loss-slot sum in the SAME launches (ABI v2): the partial-gradient reduction applies the
 * update to each element it has just reduced, and one extra block sums loss_q[nq] into loss_out (nullable) exactly like ptr_sum_f32 — for
 * single-device training, where no all-reduce sits between the gradient and the step (ptranking/base/ranker.py:589-603 does
 * backward -> optimizer.step back to back).  `grad` still receives the gradient.  opt_kind / hyper-parameters:
 *   PTR_OPT_ADAM     hyper1 = beta1, hyper2 = beta2, state1 = exp_avg, state2 = exp_avg_sq      (arithmetic of ptr_adam_step)
 *   PTR_OPT_ADAGRAD  hyper1 = lr_decay, state1 = sum, state2 unused                             (ptr_adagrad_step)
 *   PTR_OPT_RMSPROP  hyper1 = alpha, state1 = square_avg, state2 unused                          (ptr_rmsprop_step)
 * Results are bit-identical to the separate calls. */
#define PTR_OPT_ADAM 1
#define PTR_OPT_ADAGRAD 2
#define PTR_OPT_RMSPROP 3
int ptr_mlp_backward_step(const float *X, float *params, const float *acts, const float *dpreds, int R, int F, int NL, float p_drop,
                          uint64_t seed, float *dz, float *ws, float *grad, int opt_kind, float lr, float hyper1, float hyper2, float eps,
                          float weight_decay, int step, float *state1, float *state2, const float *loss_q, int nq, float *loss_out,
                          void *stream);
/* ABI v4.  The optimiser step of ptr_mlp_backward_step and its loss-slot sum as ONE launch on a gradient that is already complete — the
 * data-parallel step: ptr_mlp_backward -> RCCL all-reduce of `grad` -> this (no reference counterpart: ptranking is single-device,
 * ptranking/ltr_adhoc/eval/ltr.py:44-48; the step itself is ptranking/base/ranker.py:512-525 + the loss accumulation of :589-603).  Same
 * arithmetic as ptr_mlp_backward_step, bit for bit.  loss_out (optional) = sum of the nq loss slots. */
int ptr_opt_step_loss(float *params, const float *grad, int64_t n, int opt_kind, float lr, float hyper1, float hyper2, float eps,
                      float weight_decay, int step, float *state1, float *state2, const float *loss_q, int nq, float *loss_out, void *stream);
/* torch.optim.Adam step (L2 weight decay added to the gradient, bias correction with `step` >= 1) on flat buffers. */
int ptr_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, void *stream);
/* torch.optim.Adagrad (clr = lr / (1 + (step - 1) lr_decay); sum += g^2; p -= clr g / (sqrt(sum) + eps)) and torch.optim.RMSprop
 * (sq = alpha sq + (1 - alpha) g^2; p -= lr g / (sqrt(sq) + eps); no momentum, not centered) on flat buffers, g = grad + weight_decay p:
 * the 'Adagrad' / 'RMS' choices of ptranking/base/ranker.py:518-521. */
int ptr_adagrad_step(float *param, const float *grad, float *state_sum, int64_t n, float lr, float lr_decay, float eps,
                     float weight_decay, int step, void *stream);
int ptr_rmsprop_step(float *param, const float *grad, float *square_avg, int64_t n, float lr, float alpha, float eps,
                     float weight_decay, void *stream);
/* ---- the same scorer on bf16 matrix instructions with fp32 results (ABI v3, csrc/scorer_x6.hip) ---------------------------------
 * Replaces the same reference code as ptr_mlp_forward (ptranking/base/point_ranker.py:45-55, ptranking/base/utils.py:288-356).
 * Every fp32 operand is split exactly into three bf16 pieces and a product is the sum of the six piece products above 2^-24 (six
 * v_mfma_f32_16x16x32_bf16 with fp32 accumulation per 32-deep slice): results agree with the fp32-MFMA entry points to fp32 rounding
 * (error against float64 equal or lower), the arithmetic type of the path stays fp32.  Same operands and `acts` layout as
 * ptr_mlp_forward; `wimg` is caller-provided scratch of ptr_mlp_x6_ws_bytes(F, NL) bytes (16-byte aligned) that receives the
 * pre-split weight image (rebuilt by every call: the weights change every step).  Served: F % 4 == 0, 2 <= NL <= 8 (ws_bytes returns
 * 0 otherwise and the call fails with PTR_ERR_UNSUPPORTED). */
size_t ptr_mlp_x6_ws_bytes(int F, int NL);
int ptr_mlp_forward_x6(const float *X, const float *params, int R, int F, int NL, int train, float p_drop, uint64_t seed,
                       float *preds, float *acts, void *wimg, void *stream);
/* ---- one train step as ONE call (ABI v5, csrc/train_step.hip) ------------------------------------------------------------------------
 * Replaces ptranking/base/ranker.py:589-603 (`NeuralRanker.train_op`: forward -> custom_loss_function = loss, zero_grad, backward,
 * optimizer.step) for the pointsf scorer (ptranking/base/point_ranker.py:45-55) with one of the single-kernel losses: the SAME three
 * entry points a caller would chain — ptr_mlp_forward[_x6] -> ptr_<loss>_fwd_bwd -> ptr_mlp_backward_step — enqueued from C on
 * `stream`, with the same arguments in the same order, so the parameters after the step are bit-identical to the separate calls.
 * What it removes is the host work between the launches (three foreign-function calls with ~60 marshalled arguments: the step was
 * host-bound below ~1024 queries).  The descriptor is caller-owned and may be kept across steps (only seed / step / lr change).
 *   loss_kind      PTR_LOSS_*; loss_f / loss_i carry that loss's parameters:
 *                    RANKNET, LAMBDARANK  loss_f[0] = sigma
 *                    LAMBDALOSS           loss_i[0] = k, loss_i[1] = loss_type (PTR_LAMBDALOSS_*), loss_i[2] = presort; loss_f[0] = sigma, loss_f[1] = mu
 *                    LISTNET              none
 *   X [B*L][F], labels [B][L], lens [B] or NULL; params / grad / state1 / state2: the flat buffers of ptr_mlp_backward_step
 *   scratch        preds [B*L], acts (ptr_mlp_acts_floats), loss_q [B], dpreds [B*L], ws (ptr_mlp_backward_ws_floats), dz
 *                  (ptr_mlp_backward_dz_floats, NULL when 0), wimg (ptr_mlp_x6_ws_bytes; NULL selects the fp32-MFMA forward)
 *   loss_out [1]   sum of the per-query losses (written by the backward's reduction launch)
 *   wimg_current   with the bf16x6 forward the step is FOUR launches: the optimiser launch also writes the updated weights' bf16 planes into `wimg`
 *                  (bit-identical to what the prep launch of ptr_mlp_forward_x6 would build), so the next step needs no prep launch
 * Errors: the first failing stage's code; ptr_last_error() names the stage's entry point. */
#define PTR_LOSS_RANKNET 1
#define PTR_LOSS_LAMBDARANK 2
#define PTR_LOSS_LAMBDALOSS 3
#define PTR_LOSS_LISTNET 4
typedef struct ptr_train_step_desc {
    int32_t struct_bytes;                 /* sizeof(ptr_train_step_desc): a binding built against another layout is refused */
    int32_t loss_kind;
    int32_t B, L, F, NL;
    int32_t opt_kind, step;
    int32_t loss_i[4];
    float loss_f[4];
    float p_drop, lr, hyper1, hyper2, eps, weight_decay;
    uint64_t seed;
    const float *X, *labels;
    const int32_t *lens;
    float *params, *grad, *state1, *state2;
    float *preds, *acts, *loss_q, *dpreds, *dz, *ws;
    void *wimg;
    float *loss_out;
    void *events[4];                      /* optional hipEvent_t handles (NULL: none) recorded on `stream` in front of the forward, between the stages
                                             and behind the backward: events[0..1] bracket the forward, [1..2] the loss kernel, [2..3] the backward + step
                                             (bench.py times the stages of the product path with them) */
    int32_t wimg_current;                 /* != 0: `wimg` already holds the bf16 planes of `params` (left there by the previous ptr_train_step on these
                                             buffers): the forward skips its prep launch.  The call ALWAYS leaves the image current for the updated
                                             parameters — the optimiser launch rewrites it element by element.  0 when in doubt (first call, parameters or
                                             image touched by anything else in between) */
    int32_t reserved;
} ptr_train_step_desc;
int ptr_train_step(const ptr_train_step_desc *d, void *stream);
/* Test helper: the dropout keep-mask (1.0 / 0.0) of dropout site `site` for an [R][n_feat] activation. */
int ptr_mlp_dropout_mask(int R, int n_feat, int site, float p_drop, uint64_t seed, float *out, void *stream);

/* ---- linear layers of the scoring functions (hand-written fp32-MFMA kernels, csrc/linear.hip) -------------------------------
 * Replace the library GEMMs behind the reference's nn.Linear modules: the stacked feed-forward nets of
 * ptranking/base/utils.py:288-356 (pointsf with any activation / batch norm; the listsf head / tail stacks, ff_dims 128/256/512)
 * and the Q|K|V / fc projections of ptranking/base/list_ranker.py:176-254.  Row-major operands with explicit leading dimensions
 * (so packed buffers such as [R][3F] are read / written in place); W is nn.Linear's [N][K] weight.
 *   ptr_linear_forward          Y[R][N] = epi(X[R][K] W^T + bias)   epi: PTR_LINEAR_NONE | _RELU | _RELU_DROPOUT (ReLU, then the
 *                               NEXT layer's dropout from the counter-based generator (seed, site): the stored value is that
 *                               layer's input and `a > 0` encodes "ReLU active and kept")
 *   ptr_linear_backward_input   dX[R][K] = dY[R][N] W, optionally gated: dX *= [gate > 0] / (1 - p_drop)  (gate = the stored
 *                               _RELU_DROPOUT output of the layer below, NULL = no gate)
 *   ptr_linear_backward_weight  dW[N][K] = dY^T X, db[N] = column sums of dY (NULL = skip); ws = ptr_linear_backward_weight_ws_floats
 *                               floats of scratch; deterministic (fixed-order reduction of row chunks)                          */
#define PTR_LINEAR_NONE 0
#define PTR_LINEAR_RELU 1
#define PTR_LINEAR_RELU_DROPOUT 2
#define PTR_LINEAR_GATE 3          /* internal: the epilogue of ptr_linear_backward_input */
int ptr_linear_forward(const float *X, int ldx, const float *W, const float *bias, int R, int K, int N, int act, float p_drop,
                       uint64_t seed, int site, float *Y, int ldy, void *stream);
int ptr_linear_backward_input(const float *dY, int ldy, const float *W, int R, int K, int N, const float *gate, int ldg, float p_drop,
                              float *dX, int ldx, void *stream);
size_t ptr_linear_backward_weight_ws_floats(int R, int K, int N);
int ptr_linear_backward_weight(const float *X, int ldx, const float *dY, int ldy, int R, int K, int N, float *ws, float *dW, float *db,
                               void *stream);
/* nn.Dropout in front of a stack's first Linear (utils.py:299): out = x * keep(seed, site, row, col) / (1 - p); the backward is the
 * same call on the incoming gradient (the mask is recomputed).  C, ldx, ldo multiples of 4, 16-byte aligned pointers. */
int ptr_dropout_apply(const float *x, int ldx, int R, int C, float p_drop, uint64_t seed, int site, float *out, int ldo, void *stream);
/* out = dy * [y > 0]: backward of a trailing ReLU (the `apply_tl_af` activation of the listsf head stack, list_ranker.py:318). */
int ptr_relu_gate(const float *dy, const float *y, int64_t n, float *out, void *stream);

/* ---- batch normalisation + activation + dropout of the stacked feed-forward nets (csrc/bnact.hip) --------------------------
 * One hidden layer of get_stacked_FFNet (ptranking/base/utils.py:296-315) is Dropout -> Linear -> [LTRBatchNorm] -> AF; the default
 * pointsf (ptranking/ltr_adhoc/eval/parameter.py:145-146) is 5 x [.. -> BN(affine) -> GELU] -> Linear -> BN -> Sigmoid.  Around
 * ptr_linear_*:  ptr_bn_stats = LTRBatchNorm's batch statistics (utils.py:201-223: BatchNorm1d without running statistics — batch
 * statistics in training and evaluation, biased variance, two-pass), ptr_bnact_forward = dropout_next(AF(gamma * xhat + beta)),
 * ptr_bnact_backward = the backward of all three (dropout mask and activation derivative recomputed from the stored
 * pre-normalisation z; BatchNorm's two column sums reduced in a fixed order), also yielding dgamma / dbeta.
 * Activations = the working entries of get_AF (utils.py:100-143).  mean == NULL: no batch norm; gamma / beta NULL: no affine. */
#define PTR_AF_NONE 0
#define PTR_AF_RELU 1      /* 'R'  */
#define PTR_AF_LEAKY 2     /* 'LR' */
#define PTR_AF_ELU 3       /* 'E' and 'CE' (alpha = 1) */
#define PTR_AF_SELU 4      /* 'SE' */
#define PTR_AF_GELU 5      /* 'GE' (erf form) */
#define PTR_AF_SIGMOID 6   /* 'S'  */
#define PTR_AF_TANH 7      /* 'T'  */
/* group_rows: 0 = statistics over all R rows (LTRBatchNorm 'BN'); L > 0 = per group of L consecutive rows, i.e. per query (LTRBatchNorm2
 * 'BN2', utils.py:227-286; R % L == 0) — mean / rstd then are [R / L][N]. */
/* lens / rows_per_query (ABI v2; lens nullable): PADDED query batches.  Row r of the [B * rows_per_query] rows is a
 * real document iff (r % rows_per_query) < lens[r / rows_per_query]; only real rows enter the statistics (mean / variance, and the two
 * column sums of the backward, divided by the number of REAL rows — of the batch for 'BN', of the query for 'BN2'), and a padded row's dz
 * is 0, so a padded batch scores and trains exactly like the unpadded lists (the reference batches equal-length lists only,
 * data_utils.py:683-742, so its LTRBatchNorm never sees a padded row).  group_rows > 0 requires group_rows == rows_per_query. */
size_t ptr_bn_ws_floats(int R, int N, int group_rows);
int ptr_bn_stats(const float *z, int ld, int R, int N, int group_rows, const int32_t *lens, int rows_per_query, float eps, float *ws,
                 float *mean, float *rstd, void *stream);
int ptr_bnact_forward(const float *z, int ld, int R, int N, int group_rows, const int32_t *lens, int rows_per_query, const float *mean,
                      const float *rstd, const float *gamma, const float *beta, int af, float p_drop, uint64_t seed, int site, float *out,
                      void *stream);
/* ws: ptr_bn_ws_floats(R, N, group_rows) floats (only read / written with batch norm); dgamma / dbeta: sums over all REAL rows */
int ptr_bnact_backward(const float *z, const float *da, int ld, int R, int N, int group_rows, const int32_t *lens, int rows_per_query,
                       const float *mean, const float *rstd, const float *gamma, const float *beta, int af, float p_drop, uint64_t seed,
                       int site, float *ws, float *dz, float *dgamma, float *dbeta, void *stream);

/* ---- listsf: the permutation-equivariant scorer's fused pieces (fp32 MFMA attention core, the reference's LayerNorm) ----
 * ptr_mhsa_forward replaces ptranking/base/list_ranker.py:216-240 (Q K^T / sqrt(d_h) -> softmax -> Dropout -> . V, heads = column
 * blocks of width F / n_heads of the [B][L][F] projections Q, K, V; the output O has the same layout, i.e. what
 * `x.permute(0,2,1,3).contiguous().view(bsz,-1,F)` yields at :243-247).  Nothing of size L^2 is written: lse [B*H*L] (log-sum-exp
 * of every score row) is the only side output and lets ptr_mhsa_backward recompute the probabilities.
 * lens (nullable): keys >= lens[b] are excluded from the softmax (padded batches; the reference has no padding).
 * p_drop > 0: dropout on the attention probabilities from the counter generator (seed, site, b, h, row, key); p_drop = 0: eval.
 * ptr_mhsa_backward: dO [B][L][F] -> dQ, dK, dV [B][L][F] (every element written); dvec [B*H*L] is scratch; O, lse, p_drop,
 * seed, site must be the forward call's.  Head dimension F / n_heads <= PTR_MHSA_MAX_HEAD_DIM.
 * ld_qkv = row stride in floats of Q, K, V (and dQ, dK, dV): F for three separate tensors, 3F when they are the column blocks
 * of ONE packed [B][L][3F] projection (pass base, base + F, base + 2F) — one GEMM then produces all three and dQ|dK|dV is
 * directly the gradient of that projection.  O, dO are always [B][L][F]. */
#define PTR_MHSA_MAX_HEAD_DIM 128
int ptr_mhsa_forward(const float *Q, const float *K, const float *V, int ld_qkv, const int32_t *lens, int B, int L, int F,
                     int n_heads, float p_drop, uint64_t seed, int site, float *O, float *lse, void *stream);
/* ds_ws (ABI v2, nullable): B * n_heads * L * L floats of scratch.  When given, the dK / dV kernel stores the scaled dS it forms and the dQ
 * kernel is ONE GEMM unit dS . K instead of recomputing S = Q K^T and dP = dO V^T (7 -> 5 GEMM units for the backward, at 4 L^2 bytes per
 * (query, head) through HBM); NULL keeps the recomputing dQ kernel (no L^2 scratch). */
int ptr_mhsa_backward(const float *Q, const float *K, const float *V, int ld_qkv, const float *O, const float *dO, const float *lse,
                      const int32_t *lens, int B, int L, int F, int n_heads, float p_drop, uint64_t seed, int site, float *dvec,
                      float *dQ, float *dK, float *dV, float *ds_ws, void *stream);
/* Test helper: the attention dropout keep-mask (1.0 / 0.0), out [B][n_heads][L][L]. */
int ptr_mhsa_dropout_mask(int B, int L, int n_heads, float p_drop, uint64_t seed, int site, float *out, void *stream);
/* LayerNorm of ptranking/base/list_ranker.py:152-174: y = a_2 * (x - mean) / (std + eps) + b_2 over the last axis of X [R][F],
 * std UNBIASED (divides by F-1) and eps added to the std.  stats [R][3] = {mean, 1/(std+eps), std} feeds the backward.
 * ptr_layernorm_backward: dY -> dX [R][F], da2 [F], db2 [F]; ws = ptr_layernorm_backward_ws_floats(F) floats of scratch. */
int ptr_layernorm_forward(const float *X, const float *a2, const float *b2, int64_t R, int F, float eps, float *Y, float *stats,
                          void *stream);
size_t ptr_layernorm_backward_ws_floats(int F);
int ptr_layernorm_backward(const float *X, const float *a2, const float *dY, const float *stats, int64_t R, int F, float *ws,
                           float *dX, float *da2, float *db2, void *stream);

/* ---- LETOR / libsvm text input (HOST buffers; the data format feeding the path) ---------------------------------------
 * Replaces the pure-Python tokenizer ptranking/data/data_utils.py:276-387 (iter_lines / parse_letor):
 *   "<label> qid:<id> <fid>:<val> ... [# comment]", feature ids one-indexed unless one_indexed == 0 (Yahoo! sets,
 *   data_utils.py:495-496), absent features = `missing`, width = largest feature id in the file, values rounded
 *   text -> double -> float (the reference's float() + FloatTensor cast, data_utils.py:610).
 * Stateless two-call protocol: ptr_letor_scan sizes the file, the caller allocates, ptr_letor_load fills
 *   X [n_docs][n_features] (float, or double when x_is_f64 — parse_letor's own precision, wanted before feature scaling),
 *   y [n_docs], qids [n_queries], qoff [n_queries+1] (row range of every run of equal qids, in file
 *   order; a non-numeric qid token is reported as a 63-bit FNV-1a hash).  Multi-threaded on the host; no GPU involved. */
int ptr_letor_scan(const char *path, int one_indexed, int64_t *n_docs, int32_t *n_features, int64_t *n_queries);
int ptr_letor_load(const char *path, int one_indexed, float missing, int64_t n_docs, int32_t n_features, int64_t n_queries,
                   void *X, int x_is_f64, float *y, int64_t *qids, int64_t *qoff);

#ifdef __cplusplus
}
#endif
#endif /* PTRANKING_AMD_H */
