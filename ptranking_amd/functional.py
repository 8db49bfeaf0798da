"""Tensor-level API of the fused HIP kernels: differentiable losses and device metrics.

Every loss is a `torch.autograd.Function` whose forward launches ONE fused kernel that produces both the loss and
dLoss/dpreds; backward only multiplies by the incoming scalar.  Inputs must be CUDA float32
tensors — there is no CPU path (the reference's CPU op sequences are what `oracle/` restates for the tests).

Shapes: preds / labels `[B, L]` (padded, row-major), optional `lens` int32 `[B]`.
"""
import ctypes as C

import torch

from . import _lib

__all__ = ["ranknet_loss", "lambdarank_loss", "lambdaloss_loss", "approxndcg_loss", "listnet_loss", "listmle_loss",
           "stlistnet_loss", "rankmse_loss", "rankcosine_loss",
           "softrank_loss", "mdprank_loss", "shuffle_ties_order", "sort_desc", "metrics_at_ks", "sum_f32", "LAMBDALOSS_TYPES"]

LAMBDALOSS_TYPES = {"NDCG_Loss1": 0, "NDCG_Loss2": 1, "NDCG_Loss2++": 2}   # ptranking/ltr_adhoc/listwise/lambdaloss.py:27


def _check(name, t, dtype=torch.float32, shape=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: ptranking_amd runs on the MI355X HIP path only (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
    return t.contiguous()


def _batch(preds, second, lens, second_dtype=torch.float32, second_name="labels"):
    preds = _check("preds", preds)
    if preds.dim() != 2:
        raise ValueError(f"preds must be [batch, ranking_size], got {tuple(preds.shape)}")
    B, L = preds.shape
    if L > _lib.MAX_LIST_LEN:
        raise ValueError(f"ranking_size {L} exceeds the supported maximum {_lib.MAX_LIST_LEN}")
    second = _check(second_name, second, second_dtype, (B, L))
    if second.device != preds.device:
        raise RuntimeError("preds and labels live on different devices")
    if lens is not None:
        lens = _check("lens", lens, torch.int32, (B,))
    return preds, second, lens, B, L


class _FusedLoss(torch.autograd.Function):
    """forward(preds, launch) where launch(preds) -> (loss 0-d tensor, grad [B,L]); backward scales grad."""

    @staticmethod
    def forward(ctx, preds, launch):
        loss, grad = launch(preds)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        (grad,) = ctx.saved_tensors
        return grad * grad_out, None


def _reduce(loss_q, B, dev):
    out = torch.empty(1, device=dev, dtype=torch.float32)
    _lib.call("ptr_sum_f32", _lib.ptr(loss_q), B, C.c_float(1.0), _lib.ptr(out), _lib.current_stream(dev))
    return out.reshape(())


def _simple(entry, preds, labels, lens, *params):
    """Shared driver of the (preds, labels, lens, B, L, <params>, loss_out, loss_q, grad, stream) entry points: one launch
    of the fused loss+gradient kernel, then the deterministic reduction of the per-query loss slots."""
    preds_c, labels, lens, B, L = _batch(preds.detach(), labels, lens)
    dev = preds_c.device

    def launch(p):
        loss_q = torch.empty(max(B, 1), device=dev, dtype=torch.float32)
        grad = torch.empty((B, L), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.call(entry, _lib.ptr(p), _lib.ptr(labels), _lib.ptr(lens), B, L, *params, None, _lib.ptr(loss_q),
                      _lib.ptr(grad), _lib.current_stream(dev))
            return _reduce(loss_q, B, dev), grad

    if preds.requires_grad:
        return _FusedLoss.apply(preds if preds.is_contiguous() else preds.contiguous(), lambda p: launch(p.detach()))
    return launch(preds_c)[0]


def ranknet_loss(preds, labels, sigma=1.0, lens=None):
    """RankNet, ptranking/ltr_adhoc/pairwise/ranknet.py:32-36."""
    return _simple("ptr_ranknet_fwd_bwd", preds, labels, lens, C.c_float(float(sigma)))


def lambdarank_loss(preds, labels, sigma=1.0, lens=None):
    """LambdaRank, ptranking/ltr_adhoc/listwise/lambdarank.py:39-56.  `labels` in ideal (descending) order per query."""
    return _simple("ptr_lambdarank_fwd_bwd", preds, labels, lens, C.c_float(float(sigma)))


def lambdaloss_loss(preds, labels, k=5, sigma=1.0, mu=5.0, loss_type="NDCG_Loss2", presort=True, lens=None):
    """LambdaLoss NDCG_Loss1 / NDCG_Loss2 / NDCG_Loss2++, ptranking/ltr_adhoc/listwise/lambdaloss.py:83-132.
    NDCG_Loss1: the reference's [B,L] weights only broadcast against its [B,L,L] tensors at batch size 1; here every query
    of a batch uses its own weights (identical to the reference at B = 1)."""
    if loss_type not in LAMBDALOSS_TYPES:
        raise NotImplementedError(f"LambdaLoss type {loss_type!r} (supported: {sorted(LAMBDALOSS_TYPES)})")
    return _simple("ptr_lambdaloss_fwd_bwd", preds, labels, lens, int(k), C.c_float(float(sigma)), C.c_float(float(mu)),
                   LAMBDALOSS_TYPES[loss_type], int(bool(presort)))


def softrank_loss(preds, labels, delta=2.0, top_k=None, lens=None):
    """SoftRank, ptranking/ltr_adhoc/listwise/softrank.py:47-69.  `labels` in ideal (descending) order per query."""
    return _simple("ptr_softrank_fwd_bwd", preds, labels, lens, C.c_float(float(delta)), int(top_k) if top_k else 0)


def listnet_loss(preds, labels, lens=None):
    """ListNet, ptranking/ltr_adhoc/listwise/listnet.py:39."""
    return _simple("ptr_listnet_fwd_bwd", preds, labels, lens)


def rankmse_loss(preds, labels, lens=None):
    """RankMSE, ptranking/ltr_adhoc/pointwise/rank_mse.py:13-22 (mean over queries of the per-query squared error sums)."""
    preds_c, labels, lens, B, L = _batch(preds.detach(), labels, lens)
    dev = preds_c.device

    def launch(p):
        loss_q = torch.empty(max(B, 1), device=dev, dtype=torch.float32)
        grad = torch.empty((B, L), device=dev, dtype=torch.float32)
        out = torch.empty(1, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.call("ptr_rankmse_fwd_bwd", _lib.ptr(p), _lib.ptr(labels), _lib.ptr(lens), B, L, _lib.ptr(out), _lib.ptr(loss_q),
                      _lib.ptr(grad), _lib.current_stream(dev))
        return out.reshape(()), grad

    if preds.requires_grad:
        return _FusedLoss.apply(preds if preds.is_contiguous() else preds.contiguous(), lambda p: launch(p.detach()))
    return launch(preds_c)[0]


def rankcosine_loss(preds, labels, lens=None):
    """RankCosine, ptranking/ltr_adhoc/listwise/rank_cosine.py:32."""
    return _simple("ptr_rankcosine_fwd_bwd", preds, labels, lens)


def stlistnet_loss(preds, labels, temperature=1.0, unif=None, lens=None):
    """STListNet, ptranking/ltr_adhoc/listwise/st_listnet.py:41-49.  `unif` = the uniform draws the Gumbel noise is made of
    (default: torch.rand on the device, as the reference does)."""
    preds_c, labels, lens, B, L = _batch(preds.detach(), labels, lens)
    dev = preds_c.device
    if unif is None:
        unif = torch.rand((B, L), device=dev)
    unif = _check("unif", unif, torch.float32, (B, L))

    def launch(p):
        loss_q = torch.empty(max(B, 1), device=dev, dtype=torch.float32)
        grad = torch.empty((B, L), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.call("ptr_stlistnet_fwd_bwd", _lib.ptr(p), _lib.ptr(labels), _lib.ptr(unif), _lib.ptr(lens), B, L,
                      C.c_float(float(temperature)), None, _lib.ptr(loss_q), _lib.ptr(grad), _lib.current_stream(dev))
            return _reduce(loss_q, B, dev), grad

    if preds.requires_grad:
        return _FusedLoss.apply(preds if preds.is_contiguous() else preds.contiguous(), lambda p: launch(p.detach()))
    return launch(preds_c)[0]


def listmle_loss(preds, perm, lens=None):
    """ListMLE, ptranking/ltr_adhoc/listwise/listmle.py:82,92-97; `perm` int64 [B,L] from arg_shuffle_ties / shuffle_ties_order."""
    preds_c, perm, lens, B, L = _batch(preds.detach(), perm, lens, torch.int64, "perm")
    dev = preds_c.device

    def launch(p):
        loss_q = torch.empty(max(B, 1), device=dev, dtype=torch.float32)
        grad = torch.empty((B, L), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.call("ptr_listmle_fwd_bwd", _lib.ptr(p), _lib.ptr(perm), _lib.ptr(lens), B, L, None, _lib.ptr(loss_q),
                      _lib.ptr(grad), _lib.current_stream(dev))
            return _reduce(loss_q, B, dev), grad

    if preds.requires_grad:
        return _FusedLoss.apply(preds if preds.is_contiguous() else preds.contiguous(), lambda p: launch(p.detach()))
    return launch(preds_c)[0]


def mdprank_loss(preds, labels, perm, top_k=10, gamma=1.0, lens=None):
    """MDPRank, ptranking/ltr_adhoc/listwise/mdprank.py:46-75: return-weighted ListMLE on the sampled ranking `perm` (int64 [B,L]).
    `preds` are the action scores by ORIGINAL document index (for 'STPL' the caller passes (preds + gumbel) / temperature)."""
    preds_c, perm, lens, B, L = _batch(preds.detach(), perm, lens, torch.int64, "perm")
    labels = _check("labels", labels, shape=(B, L)).contiguous()
    dev = preds_c.device

    def launch(p):
        loss_q = torch.empty(max(B, 1), device=dev, dtype=torch.float32)
        grad = torch.empty((B, L), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.call("ptr_mdprank_fwd_bwd", _lib.ptr(p), _lib.ptr(labels), _lib.ptr(perm), _lib.ptr(lens), B, L,
                      int(top_k) if top_k else 0, C.c_float(float(gamma)), None, _lib.ptr(loss_q), _lib.ptr(grad),
                      _lib.current_stream(dev))
            return _reduce(loss_q, B, dev), grad

    if preds.requires_grad:
        return _FusedLoss.apply(preds if preds.is_contiguous() else preds.contiguous(), lambda p: launch(p.detach()))
    return launch(preds_c)[0]


def approxndcg_loss(preds, labels, alpha=10.0, presort=True, couple_batch=True, lens=None, grad_scale_override=0.0,
                    return_parts=False):
    """ApproxNDCG, ptranking/ltr_adhoc/listwise/approxNDCG.py:45-62.  couple_batch=True reproduces the reference's batch
    coupling (loss = -(sum DCG_b)(sum 1/IDCG_a)).  With return_parts also returns (dcg_q [B], inv_idcg_q [B], scale [2])."""
    preds_c, labels, lens, B, L = _batch(preds.detach(), labels, lens)
    dev = preds_c.device
    parts = {}

    def launch(p):
        out = torch.empty(1, device=dev, dtype=torch.float32)
        dcg = torch.empty(max(B, 1), device=dev, dtype=torch.float32)
        inv = torch.empty(max(B, 1), device=dev, dtype=torch.float32)
        scale = torch.empty(2, device=dev, dtype=torch.float32)
        grad = torch.empty((B, L), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.call("ptr_approxndcg_fwd_bwd", _lib.ptr(p), _lib.ptr(labels), _lib.ptr(lens), B, L, C.c_float(float(alpha)),
                      int(bool(presort)), int(bool(couple_batch)), C.c_float(float(grad_scale_override)), _lib.ptr(out),
                      _lib.ptr(dcg), _lib.ptr(inv), _lib.ptr(scale), _lib.ptr(grad), _lib.current_stream(dev))
        parts.update(dcg_q=dcg[:B], inv_idcg_q=inv[:B], scale=scale)
        return out.reshape(()), grad

    if preds.requires_grad:
        loss = _FusedLoss.apply(preds if preds.is_contiguous() else preds.contiguous(), lambda p: launch(p.detach()))
    else:
        loss = launch(preds_c)[0]
    return (loss, parts) if return_parts else loss


def shuffle_ties_order(labels, seed, lens=None):
    """Device replacement for arg_shuffle_ties (ptranking/ltr_adhoc/util/sampling_utils.py:13-28): int64 [B,L] random
    tie-broken label-descending order.  Same distribution, different random stream (not the torch.randperm one)."""
    labels = _check("labels", labels)
    B, L = labels.shape
    if lens is not None:
        lens = _check("lens", lens, torch.int32, (B,))
    perm = torch.empty((B, L), device=labels.device, dtype=torch.int64)
    with torch.cuda.device(labels.device):
        _lib.call("ptr_shuffle_ties_order", _lib.ptr(labels), _lib.ptr(lens), B, L, C.c_uint64(int(seed) & (2 ** 64 - 1)),
                  _lib.ptr(perm), _lib.current_stream(labels.device))
    return perm


def sort_desc(preds, lens=None):
    """torch.sort(preds, dim=1, descending=True) on device -> (values, int64 indices); ties by original index."""
    preds = _check("preds", preds)
    B, L = preds.shape
    if lens is not None:
        lens = _check("lens", lens, torch.int32, (B,))
    vals = torch.empty_like(preds)
    idx = torch.empty((B, L), device=preds.device, dtype=torch.int64)
    with torch.cuda.device(preds.device):
        _lib.call("ptr_sort_desc", _lib.ptr(preds), _lib.ptr(lens), B, L, _lib.ptr(vals), _lib.ptr(idx),
                  _lib.current_stream(preds.device))
    return vals, idx


def metrics_at_ks(preds, labels, ks, presort=False, max_label=None, lens=None, which=("ndcg", "nerr", "ap", "p"),
                  permutation_labels=False):
    """Evaluator prologue + metrics at cut-offs `ks` for one batch -> dict name -> [B, len(ks)] float32 on device.
    Replaces ptranking/base/ranker.py:220-243 + ptranking/metric/adhoc/adhoc_metric.py @ks functions.
    permutation_labels: LABEL_TYPE.Permutation — nDCG's gain is the label itself; nERR is undefined (NotImplementedError)."""
    if permutation_labels and "nerr" in which:
        raise NotImplementedError("nERR is only defined for LABEL_TYPE.MultiLabel (adhoc_metric.py:157-164)")
    preds, labels, lens, B, L = _batch(preds.detach(), labels, lens)
    ks = [int(k) for k in ks]
    if len(ks) > _lib.MAX_CUTOFFS:
        raise ValueError(f"at most {_lib.MAX_CUTOFFS} cut-offs")
    dev = preds.device
    out = {m: torch.empty((B, len(ks)), device=dev, dtype=torch.float32) for m in which}
    ws = torch.empty(1, device=dev, dtype=torch.float32) if ("nerr" in which and max_label is None) else None
    ks_arr = (C.c_int32 * max(len(ks), 1))(*ks)
    ml = -1.0 if max_label is None else float(max_label)
    with torch.cuda.device(dev):
        _lib.call("ptr_metrics_at_ks", _lib.ptr(preds), _lib.ptr(labels), _lib.ptr(lens), B, L, ks_arr, len(ks),
                  int(bool(presort)), int(bool(permutation_labels)), C.c_float(ml), _lib.ptr(ws), _lib.ptr(out.get("ndcg")), _lib.ptr(out.get("nerr")),
                  _lib.ptr(out.get("ap")), _lib.ptr(out.get("p")), _lib.current_stream(dev))
    return out


def sum_f32(x, scale=1.0):
    """Deterministic device sum -> 1-element tensor."""
    x = _check("x", x).reshape(-1)
    out = torch.empty(1, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.call("ptr_sum_f32", _lib.ptr(x), x.numel(), C.c_float(float(scale)), _lib.ptr(out), _lib.current_stream(x.device))
    return out
