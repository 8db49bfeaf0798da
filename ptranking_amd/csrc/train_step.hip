// One train step of the pointsf scorer as ONE C-ABI call: scorer forward -> fused loss + gradient kernel -> scorer backward + optimiser
// step + loss-slot sum, enqueued back to back on the caller's stream (ABI v5).
//
// Reference: ptranking/base/ranker.py:589-603 (NeuralRanker.train_op -> custom_loss_function, e.g. ptranking/ltr_adhoc/listwise/
// lambdarank.py:39-62: loss, zero_grad, backward, optimizer.step).  The entry points chained here are the ones a caller chains itself
// (ptranking_amd/rankers.py `_direct_train_op`, r2-r5): identical launches with identical arguments, so the results are bit-identical —
// what goes away is the host work between them.  r5 measured the five-launch step at 80 us for 64 queries (GPU work: ~25 us) and 23 %
// above a quarter of the 4096-query step at 1024 queries: three ctypes calls with ~60 marshalled arguments per step.
#include "ptr_mlp.h"

extern "C" int ptr_train_step(const ptr_train_step_desc *d, void *stream) {
    using namespace ptr;
    const char *who = "ptr_train_step";
    if (!d) { set_error("%s: NULL descriptor", who); return PTR_ERR_INVALID_ARG; }
    if (d->struct_bytes != (int32_t)sizeof(ptr_train_step_desc)) {
        set_error("%s: descriptor of %d bytes, this library expects %d (ABI v%d)", who, (int)d->struct_bytes, (int)sizeof(ptr_train_step_desc), PTR_ABI_VERSION);
        return PTR_ERR_INVALID_ARG;
    }
    if (d->B < 0 || d->L <= 0 || (int64_t)d->B * d->L > 0x7fffffffll) { set_error("%s: bad shape B=%d L=%d", who, d->B, d->L); return PTR_ERR_INVALID_ARG; }
    if (d->loss_kind < PTR_LOSS_RANKNET || d->loss_kind > PTR_LOSS_LISTNET) { set_error("%s: unknown loss %d", who, d->loss_kind); return PTR_ERR_INVALID_ARG; }
    if (!d->loss_q || !d->dpreds || !d->preds) { set_error("%s: NULL scratch pointer", who); return PTR_ERR_INVALID_ARG; }
    const int R = d->B * d->L;
    hipStream_t st = as_stream(stream);
    auto mark = [&](int k) -> int { return d->events[k] ? check_hip(hipEventRecord(reinterpret_cast<hipEvent_t>(d->events[k]), st), who) : 0; };
    if (int rc = mark(0)) return rc;
    // 1. forward (training mode: the activations the backward reads are stored).  bf16x6 forward: no prep launch when the image is current
    if (int rc = d->wimg ? mlp_forward_x6_impl(d->X, d->params, R, d->F, d->NL, 1, d->p_drop, d->seed, d->preds, d->acts, d->wimg, d->wimg_current == 0, stream)
                         : ptr_mlp_forward(d->X, d->params, R, d->F, d->NL, 1, d->p_drop, d->seed, d->preds, d->acts, stream)) return rc;
    if (int rc = mark(1)) return rc;
    // 2. loss + dLoss/dscore.  loss_out = NULL: the per-query slots are summed by the backward's reduction launch (3)
    int rc = 0;
    switch (d->loss_kind) {
    case PTR_LOSS_RANKNET: rc = ptr_ranknet_fwd_bwd(d->preds, d->labels, d->lens, d->B, d->L, d->loss_f[0], nullptr, d->loss_q, d->dpreds, stream); break;
    case PTR_LOSS_LAMBDARANK: rc = ptr_lambdarank_fwd_bwd(d->preds, d->labels, d->lens, d->B, d->L, d->loss_f[0], nullptr, d->loss_q, d->dpreds, stream); break;
    case PTR_LOSS_LAMBDALOSS:
        rc = ptr_lambdaloss_fwd_bwd(d->preds, d->labels, d->lens, d->B, d->L, d->loss_i[0], d->loss_f[0], d->loss_f[1], d->loss_i[1], d->loss_i[2], nullptr,
                                    d->loss_q, d->dpreds, stream);
        break;
    default: rc = ptr_listnet_fwd_bwd(d->preds, d->labels, d->lens, d->B, d->L, nullptr, d->loss_q, d->dpreds, stream); break;
    }
    if (rc) return rc;
    if (int rc2 = mark(2)) return rc2;
    // 3. backward -> flat gradient; the partial reduction applies the optimiser step, sums the loss slots and (bf16x6 forward) refreshes the weight image
    if (int rc3 = mlp_backward_step_impl(d->X, d->params, d->acts, d->dpreds, R, d->F, d->NL, d->p_drop, d->seed, d->dz, d->ws, d->grad, d->opt_kind, d->lr, d->hyper1,
                                         d->hyper2, d->eps, d->weight_decay, d->step, d->state1, d->state2, d->loss_q, d->B, d->loss_out, d->wimg, stream)) return rc3;
    return mark(3);
}
