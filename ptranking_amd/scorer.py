"""The reference's pointwise MLP scorer (`pointsf`) on the fused fp32-MFMA HIP kernels.

`FusedPointScorer` is a drop-in for the `nn.Sequential` that ptranking/base/utils.py:288-356 (`get_stacked_FFNet`) builds for
the configuration AF='R', BN=False, apply_tl_af=False, h_dim=100, out_dim=1 — (Dropout -> Linear -> ReLU) x num_layers -> Linear:
same initialisation (xavier_normal_ weights, nn.Linear's default bias init), same `state_dict` keys (`ff_2.weight`, ...), same
call signature `scorer(X[B, L, F]) -> [B, L, 1]`-shaped scores that `PointNeuralRanker.forward` views as [B, L].
All parameters live in ONE flat tensor (PyTorch order / layouts), which is what `FlatAdam` and the data-parallel all-reduce see.
Dropout uses a counter-based generator inside the kernels (seeded per call from torch's CPU generator, so `torch.manual_seed`
still makes runs reproducible); it is statistically, not bit-wise, the same as `nn.Dropout`.
"""
import ctypes as C
import math
import os

import torch
import torch.nn as nn

from . import _lib
from . import dp

HIDDEN = 100   # ptranking/base/point_ranker.py:30
ACT_LD = 112   # PTR_MLP_ACT_LD: padded features per row of the stored activations / leading dimension of the dZ scratch


def acts_floats(R, NL):
    """Floats of the stored-activation buffer of an R-row, NL-layer training forward: NL * ceil16(R) * 112 (tile-major, include/ptranking_amd.h)."""
    return NL * ((R + 15) // 16) * 16 * ACT_LD


def alloc_acts(R, NL, device):
    return torch.empty(acts_floats(R, NL), device=device, dtype=torch.float32)


def acts_rowmajor(acts, R, NL):
    """[NL, R, 112] row-major copy of a tile-major activation buffer (tests / debugging)."""
    T = (R + 15) // 16
    return acts[:NL * T * 16 * ACT_LD].view(NL, T, 7, 16, 16).permute(0, 1, 3, 2, 4).reshape(NL, T * 16, ACT_LD)[:, :R]


_STACK_AF = {'R', 'LR', 'E', 'SE', 'CE', 'GE', 'S', 'T'}      # the working entries of get_AF (utils.py:100-143) minus the random RReLU


def fused_kind(num_features=None, h_dim=100, out_dim=1, num_layers=3, AF='R', TL_AF='S', apply_tl_af=False, BN=True, bn_type=None,
               bn_affine=False, dropout=0.1, **_):
    """Which hand-written path serves a pointsf configuration:
      'single'  the single-kernel scorer (forward = one kernel, backward = one kernel + reduction): AF='R', no batch norm, no tail
                activation — the benchmark configuration;
      'stack'   the layer-wise fused stack (ptranking_amd/linear.py FusedStack: hand-written GEMMs + batch-norm / activation / dropout
                kernels): every other working activation, bn_type='BN' / 'BN2', tail activation — incl. the reference's DEFAULT pointsf
                (5 layers, GELU, BN affine, Sigmoid tail; ptranking/ltr_adhoc/eval/parameter.py:145-146);
      None      not covered (RReLU and the broken get_AF entries): torch modules (their Linear layers still run the hand-written GEMMs)."""
    if AF == 'R' and not BN and not apply_tl_af and h_dim == HIDDEN and out_dim == 1 and 1 <= num_layers <= 8:
        hidden = (num_layers - 1) * 112 * 100 + num_layers * 112 + 112 + 16
        w1 = 112 * ((num_features + 3) // 4 * 4 + 4)
        if (w1 + hidden) * 4 <= 160 * 1024:
            return 'single'                  # all weights in LDS
        # large F (Yahoo: 700): W1 streams from L2 — needs 16-byte aligned rows and at most 48 in-feature tiles
        if hidden * 4 <= 160 * 1024 and num_features % 4 == 0 and (num_features + 15) // 16 <= 48:
            return 'single'
    if AF in _STACK_AF and (not apply_tl_af or TL_AF in _STACK_AF) and (not BN or bn_type in ('BN', 'BN2')) and \
            (dropout == 0.0 or num_features % 4 == 0 or num_layers == 0):
        return 'stack'
    return None


def fusable(**kw):
    """True when a hand-written path (single-kernel scorer or fused stack) implements the pointsf configuration."""
    return fused_kind(**kw) is not None


X6_MIN_ROWS = 98304     # below ~1.5 passes of 256 documents per CU the 8-wave lockstep tiles leave CUs idle: the fp32-MFMA kernels are faster


def x6_mode():
    """PTR_MLP_X6: "1" (default) the bf16x6 scorer kernels (csrc/scorer_x6.hip) from X6_MIN_ROWS rows on, "0" never (the fp32-MFMA kernels),
    "2" always (tests).  Read per call."""
    return os.environ.get("PTR_MLP_X6", "1")


X6_BUFFER_LIMIT = 0xFFFFC000      # bytes a buffer resource of ptr_mlp_forward_x6 addresses (scorer_x6.hip: X and the stored activations)


def x6_serves(X2d, R, F, NL, train):
    """The run-time limits of ptr_mlp_forward_x6 beyond (F, NL): 16-byte aligned X, R * F * 4 and (training) NL * R * 448 bytes below 4 GB.
    Inputs outside take ptr_mlp_forward, which has none of them (big evaluation batches, e.g. F = 700 x 1.6 M rows)."""
    if X2d.data_ptr() % 16:
        return False
    if R * F * 4 >= X6_BUFFER_LIMIT:
        return False
    if train and acts_floats(R, NL) * 4 >= X6_BUFFER_LIMIT:
        return False
    return True


_X6_WS = {}
# r6: whose parameters an image buffer currently holds.  ptr_train_step leaves the image current for the UPDATED parameters (its optimiser launch
# rewrites it), so the next step's forward can skip the prep launch — but only when NOTHING touched the parameters or the shared image in between.
# The tag is (parameter storage address, torch's in-place version counter, optimiser step count): every torch-level write bumps the version
# (load_state_dict, no_grad updates), every optimiser step of ours bumps the count, every other forward through a shared image drops the tag.
_X6_IMG_TAG = {}


def x6_image_tag(dev, F, NL):
    return _X6_IMG_TAG.get((torch.device(dev), F, NL))


def x6_set_image_tag(dev, F, NL, tag):
    _X6_IMG_TAG[(torch.device(dev), F, NL)] = tag


def x6_workspace(dev, F, NL):
    """Scratch for the pre-split weight image of the bf16x6 kernels (None when the configuration is outside their range)."""
    n = _lib.query("ptr_mlp_x6_ws_bytes", F, NL)
    if n == 0:
        return None
    key = (torch.device(dev), F, NL)
    ws = _X6_WS.get(key)
    if ws is None:
        ws = _X6_WS[key] = torch.empty(n, device=dev, dtype=torch.uint8)
    return ws


def x6_wimg_for(X2d, R, F, NL, train, dev):
    """The weight-image scratch when the bf16x6 forward serves this call (PTR_MLP_X6 mode, row threshold, run-time limits), else None (= the
    fp32-MFMA forward).  The one place that makes the choice: mlp_forward and the single-call train step (ptr_train_step) both ask here."""
    mode = x6_mode()
    ws = x6_workspace(dev, F, NL) if (mode == "2" or (mode != "0" and R >= X6_MIN_ROWS)) else None
    if ws is not None and not x6_serves(X2d, R, F, NL, train):
        ws = None                            # the fp32-MFMA entry point serves what the bf16x6 one refuses (4 GB buffer resources, alignment)
    return ws


def mlp_forward(X2d, flat, R, F, NL, train, p, seed, preds, acts, dev):
    """Scorer forward through the C ABI: the bf16x6 entry point when it serves (F, NL), the fp32-MFMA one otherwise."""
    ws = x6_wimg_for(X2d, R, F, NL, train, dev)
    if ws is not None:
        x6_set_image_tag(dev, F, NL, None)       # this call's prep launch rebuilds the (shared) image for `flat` as it is NOW: no train step's tag survives it
        _lib.call("ptr_mlp_forward_x6", _lib.ptr(X2d), _lib.ptr(flat), R, F, NL, int(train), C.c_float(p), C.c_uint64(seed),
                  _lib.ptr(preds), _lib.ptr(acts), _lib.ptr(ws), _lib.current_stream(dev))
    else:
        _lib.call("ptr_mlp_forward", _lib.ptr(X2d), _lib.ptr(flat), R, F, NL, int(train), C.c_float(p), C.c_uint64(seed),
                  _lib.ptr(preds), _lib.ptr(acts), _lib.current_stream(dev))


class _ScorerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X2d, flat, F, NL, p, seed, store):
        R = X2d.shape[0]
        dev = X2d.device
        preds = torch.empty(R, device=dev, dtype=torch.float32)
        train = bool(store or p > 0.0)
        acts = alloc_acts(R, NL, dev) if train else None
        with torch.cuda.device(dev):
            mlp_forward(X2d, flat, R, F, NL, train, p, seed, preds, acts, dev)
        if store:
            ctx.save_for_backward(X2d, flat, acts)
            ctx.meta = (R, F, NL, p, seed)
        return preds

    @staticmethod
    def backward(ctx, dpreds):
        X2d, flat, acts = ctx.saved_tensors
        R, F, NL, p, seed = ctx.meta
        dev = X2d.device
        dpreds = dpreds.contiguous()
        ndz = _lib.query("ptr_mlp_backward_dz_floats", R, F, NL)     # 0: the single-pass fused backward needs no dZ scratch
        dz = torch.empty(ndz, device=dev, dtype=torch.float32) if ndz else None
        ws = torch.empty(_lib.query("ptr_mlp_backward_ws_floats", F, NL), device=dev, dtype=torch.float32)
        grad = torch.empty_like(flat)
        with torch.cuda.device(dev):
            _lib.call("ptr_mlp_backward", _lib.ptr(X2d), _lib.ptr(flat), _lib.ptr(acts), _lib.ptr(dpreds), R, F, NL, C.c_float(p),
                      C.c_uint64(seed), _lib.ptr(dz), _lib.ptr(ws), _lib.ptr(grad), _lib.current_stream(dev))
        return None, grad, None, None, None, None, None


class FusedPointScorer(nn.Module):
    def __init__(self, num_features, num_layers=3, dropout=0.1):
        super().__init__()
        self.num_features, self.num_layers, self.dropout = int(num_features), int(num_layers), float(dropout)
        n = 100 * self.num_features + 100 + (self.num_layers - 1) * (100 * 100 + 100) + 100 + 1
        self.flat = nn.Parameter(torch.empty(n, dtype=torch.float32))
        self.reset_parameters()

    # ---- parameter views, named like the reference's Sequential (base/utils.py:299-322)
    def layout(self):
        """[(state_dict key, offset, shape)] in flat order."""
        F, NL = self.num_features, self.num_layers
        out, off = [], 0
        for l in range(NL):
            k = F if l == 0 else HIDDEN
            out.append((f"ff_{l + 2}.weight", off, (HIDDEN, k))); off += HIDDEN * k
            out.append((f"ff_{l + 2}.bias", off, (HIDDEN,))); off += HIDDEN
        out.append((f"ff_{NL + 2}.weight", off, (1, HIDDEN))); off += HIDDEN
        out.append((f"ff_{NL + 2}.bias", off, (1,))); off += 1
        assert off == self.flat.numel()
        return out

    def views(self, grad=False):
        src = self.flat.grad if grad else self.flat.data
        return {k: src[o:o + math.prod(s)].view(s) for k, o, s in self.layout()}

    def reset_parameters(self):
        with torch.no_grad():
            for k, v in self.views().items():
                if k.endswith("weight"):
                    nn.init.xavier_normal_(v)                      # nr_init, base/utils.py:13,303,321
                else:
                    fan_in = self.num_features if k == "ff_2.bias" else HIDDEN
                    bound = 1.0 / math.sqrt(fan_in)                # nn.Linear's default bias init
                    nn.init.uniform_(v, -bound, bound)

    # ---- checkpoints interchangeable with the reference's state_dict (point_ranker.py:63-71)
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for k, v in self.views().items():
            destination[prefix + k] = v if keep_vars else v.detach().clone()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        with torch.no_grad():
            for k, v in self.views().items():
                key = prefix + k
                if key in state_dict:
                    v.copy_(state_dict[key].reshape(v.shape))
                elif strict:
                    missing_keys.append(key)
        known = {prefix + k for k, _, _ in self.layout()}
        for key in state_dict:
            if key.startswith(prefix) and key not in known and strict:
                unexpected_keys.append(key)

    def forward(self, X):
        if not X.is_cuda:
            raise RuntimeError(f"FusedPointScorer input is on {X.device}: the fused scorer runs on the MI355X HIP path only")
        lead = X.shape[:-1]
        if X.shape[-1] != self.num_features:
            raise ValueError(f"expected {self.num_features} features, got {X.shape[-1]}")
        X2d = X.reshape(-1, self.num_features)
        if X2d.dtype != torch.float32:
            X2d = X2d.float()
        X2d = X2d.contiguous()
        p = self.dropout if self.training else 0.0
        store = torch.is_grad_enabled() and self.flat.requires_grad
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0.0 else 0     # CPU generator: no device sync
        if p > 0.0:
            seed = dp.local_dropout_seed(seed, X2d.shape[0], local_queries=lead[0] if len(lead) == 2 else None)   # replicas draw the masks of THEIR global rows (dp.py)
        preds = _ScorerFn.apply(X2d, self.flat, self.num_features, self.num_layers, p, seed, store)
        return preds.view(*lead, 1)

    def dropout_mask(self, R, site, seed):
        """Keep-mask [R, n_feat] of dropout site `site` (0 = on the input features) for the given call seed — test helper."""
        n_feat = self.num_features if site == 0 else HIDDEN
        out = torch.empty((R, n_feat), device=self.flat.device, dtype=torch.float32)
        _lib.call("ptr_mlp_dropout_mask", R, n_feat, site, C.c_float(self.dropout), C.c_uint64(seed), _lib.ptr(out),
                  _lib.current_stream(out.device))
        return out


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (L2 weight decay folded into the gradient, bias correction) as ONE kernel per flat
    parameter tensor — the update the reference configures in ptranking/base/ranker.py:516-517."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    def _step_param(self, p, group):
        if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
            raise RuntimeError("FlatAdam needs contiguous CUDA float32 parameters / gradients")
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p)
            st["exp_avg_sq"] = torch.zeros_like(p)
        st["step"] += 1
        b1, b2 = group["betas"]
        with torch.cuda.device(p.device):
            _lib.call("ptr_adam_step", _lib.ptr(p), _lib.ptr(p.grad), _lib.ptr(st["exp_avg"]), _lib.ptr(st["exp_avg_sq"]),
                      C.c_int64(p.numel()), C.c_float(group["lr"]), C.c_float(b1), C.c_float(b2), C.c_float(group["eps"]),
                      C.c_float(group["weight_decay"]), int(st["step"]), _lib.current_stream(p.device))

    def fused_step_args(self, p):
        """(opt_kind, lr, hyper1, hyper2, eps, weight_decay, step, state1, state2) of THIS step for ptr_mlp_backward_step — the state /
        step bookkeeping of _step_param without its kernel launch."""
        group, st = self.param_groups[0], self.state[p]
        if not st:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p)
            st["exp_avg_sq"] = torch.zeros_like(p)
        st["step"] += 1
        b1, b2 = group["betas"]
        self._opt_called = True
        return 1, group["lr"], b1, b2, group["eps"], group["weight_decay"], int(st["step"]), st["exp_avg"], st["exp_avg_sq"]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    self._step_param(p, group)
        return loss

    def step_flat(self, p):
        """The update of `step()` for one parameter without torch.optim's per-call wrappers (profiler hooks, no_grad scope) —
        the direct train step of rankers.FusedStepMixin.  Keeps `state` and the lr-scheduler's bookkeeping consistent."""
        self._step_param(p, self.param_groups[0])
        self._opt_called = True           # what torch.optim.lr_scheduler's wrapper of step() records


class FlatAdagrad(FlatAdam):
    """torch.optim.Adagrad (lr_decay 0, eps 1e-10, accumulator from 0) as one kernel per flat parameter tensor (ranker.py:520-521)."""

    def __init__(self, params, lr=1e-2, lr_decay=0.0, eps=1e-10, weight_decay=0.0):
        torch.optim.Optimizer.__init__(self, params, dict(lr=lr, lr_decay=lr_decay, eps=eps, weight_decay=weight_decay))

    def _step_param(self, p, group):
        if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
            raise RuntimeError("FlatAdagrad needs contiguous CUDA float32 parameters / gradients")
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["sum"] = torch.zeros_like(p)
        st["step"] += 1
        with torch.cuda.device(p.device):
            _lib.call("ptr_adagrad_step", _lib.ptr(p), _lib.ptr(p.grad), _lib.ptr(st["sum"]), C.c_int64(p.numel()), C.c_float(group["lr"]),
                      C.c_float(group["lr_decay"]), C.c_float(group["eps"]), C.c_float(group["weight_decay"]), int(st["step"]),
                      _lib.current_stream(p.device))


    def fused_step_args(self, p):
        group, st = self.param_groups[0], self.state[p]
        if not st:
            st["step"] = 0
            st["sum"] = torch.zeros_like(p)
        st["step"] += 1
        self._opt_called = True
        return 2, group["lr"], group["lr_decay"], 0.0, group["eps"], group["weight_decay"], int(st["step"]), st["sum"], None


class FlatRMSprop(FlatAdam):
    """torch.optim.RMSprop (alpha 0.99, eps 1e-8, no momentum, not centered) as one kernel per flat parameter tensor (ranker.py:518-519)."""

    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0):
        torch.optim.Optimizer.__init__(self, params, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay))

    def _step_param(self, p, group):
        if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()):
            raise RuntimeError("FlatRMSprop needs contiguous CUDA float32 parameters / gradients")
        st = self.state[p]
        if not st:
            st["step"] = 0
            st["square_avg"] = torch.zeros_like(p)
        st["step"] += 1
        with torch.cuda.device(p.device):
            _lib.call("ptr_rmsprop_step", _lib.ptr(p), _lib.ptr(p.grad), _lib.ptr(st["square_avg"]), C.c_int64(p.numel()), C.c_float(group["lr"]),
                      C.c_float(group["alpha"]), C.c_float(group["eps"]), C.c_float(group["weight_decay"]), _lib.current_stream(p.device))


    def fused_step_args(self, p):
        group, st = self.param_groups[0], self.state[p]
        if not st:
            st["step"] = 0
            st["square_avg"] = torch.zeros_like(p)
        st["step"] += 1
        self._opt_called = True
        return 3, group["lr"], group["alpha"], 0.0, group["eps"], group["weight_decay"], int(st["step"]), st["square_avg"], None


FLAT_OPTIMIZERS = {'Adam': FlatAdam, 'Adagrad': FlatAdagrad, 'RMS': FlatRMSprop}      # the `opt` strings of ranker.py:516-521


class _FlatViewMixin:
    """A flat optimiser over the ONE flat buffer a FusedStack's parameters are views of (linear.FusedStack.flatten_parameters): one
    kernel for the whole scoring function instead of torch.optim's multi-tensor update (0.34 ms of host time per step for the 22 tensors
    of the default pointsf), zero_grad() = one memset that also tells the stack that its next backward may write the gradients in place."""

    def __init__(self, stack, **kw):
        flat, gflat = stack.flatten_parameters()
        self.stack = stack
        self.flat_param = torch.nn.Parameter(flat)          # shares the storage the module's parameters view
        self.flat_param.grad = gflat
        super().__init__([self.flat_param], **kw)

    def zero_grad(self, set_to_none=True):
        self.flat_param.grad.zero_()
        self.stack._grads_fresh = True

    @torch.no_grad()
    def step(self, closure=None):
        # the stack's gradients only reach flat_param.grad while every parameter's .grad aliases its slice of it: re-home
        # anything that module.zero_grad(), a foreign zero_grad(set_to_none=True) or a gradient bucket re-pointed
        self.stack.reattach_grads()
        return super().step(closure)

    def grad_bucket(self, extra):
        """dp.FlatGradBucket's interface over the flat gradient buffer itself (+ `extra` <= 4 scalars in its spare tail): the one
        all-reduce of a loss that ships scalars with its gradients (ApproxNDCG's batch coupling, RankMSE's batch mean)."""
        n = self.flat_param.numel()
        return dp.ViewGradBucket(self.stack._gbuf, n, extra, self.zero_grad, before_reduce=self.stack.reattach_grads)


class FlatViewAdam(_FlatViewMixin, FlatAdam):
    def __init__(self, stack, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(stack, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)


class FlatViewAdagrad(_FlatViewMixin, FlatAdagrad):
    def __init__(self, stack, lr=1e-2, lr_decay=0.0, eps=1e-10, weight_decay=0.0):
        super().__init__(stack, lr=lr, lr_decay=lr_decay, eps=eps, weight_decay=weight_decay)


class FlatViewRMSprop(_FlatViewMixin, FlatRMSprop):
    def __init__(self, stack, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0.0):
        super().__init__(stack, lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay)


FLAT_VIEW_OPTIMIZERS = {'Adam': FlatViewAdam, 'Adagrad': FlatViewAdagrad, 'RMS': FlatViewRMSprop}


class FusedScorerMixin:
    """Makes a ranker build the fused scorer + FlatAdam whenever its pointsf configuration allows it (otherwise the base
    class's own torch modules are used).  Set `use_fused_scorer = False` on the class or instance to opt out."""

    use_fused_scorer = True
    use_flat_stack_optimizer = True     # FusedStack + Adam: parameters re-homed in one flat buffer, FlatViewAdam

    def ini_pointsf(self, **kw):
        on_gpu = bool(getattr(self, "gpu", False)) and torch.cuda.is_available()
        kind = fused_kind(**kw) if (self.use_fused_scorer and on_gpu) else None
        if kind == 'single':
            return FusedPointScorer(kw["num_features"], num_layers=kw.get("num_layers", 3), dropout=kw.get("dropout", 0.1))
        if kind == 'stack':
            from .host import build_pointsf          # our modules with the reference's names / initialisation (FusedStack)
            allowed = ("num_features", "h_dim", "out_dim", "num_layers", "AF", "TL_AF", "apply_tl_af", "BN", "bn_type", "bn_affine", "dropout")
            return build_pointsf(**{k: v for k, v in kw.items() if k in allowed})
        return super().ini_pointsf(**kw)

    def config_optimizer(self):
        sf = getattr(self, "point_sf", None)
        if isinstance(sf, FusedPointScorer) and self.opt in FLAT_OPTIMIZERS:
            self.optimizer = FLAT_OPTIMIZERS[self.opt](self.get_parameters(), lr=self.lr, weight_decay=self.weight_decay)
            self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=20, gamma=0.5)   # ranker.py:525
        elif (self.opt in FLAT_VIEW_OPTIMIZERS and type(sf).__name__ == 'FusedStack' and self.use_flat_stack_optimizer
              and all(p.is_cuda and p.dtype == torch.float32 for p in sf.parameters())):
            self.optimizer = FLAT_VIEW_OPTIMIZERS[self.opt](sf, lr=self.lr, weight_decay=self.weight_decay)
            self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=20, gamma=0.5)   # ranker.py:525
            self._dp_single = self.optimizer.flat_param     # data parallelism: ONE all-reduce of the flat gradient (rankers.FusedStepMixin)
        else:
            super().config_optimizer()
