"""The ltr_adhoc rankers of the hot path (the six losses of the north star + the sibling losses of SURVEY.md §8 f-4) as drop-in replacements for the reference classes of the same name.

Each class keeps the reference's constructor signature and `custom_loss_function` contract (consume `batch_preds`
attached to the scorer graph + `batch_std_labels`, do zero_grad -> backward -> optimizer.step itself, return the scalar
loss tensor; ptranking/base/ranker.py:605-613) but computes the loss and dLoss/dpreds with ONE fused HIP kernel.

`make_ranker_classes(base)` builds the classes on top of any base that provides the reference's AdhocNeuralRanker
interface: ptranking_amd.host.PointScorerRanker stand-alone, or the reference's own
ptranking.base.adhoc_ranker.AdhocNeuralRanker when `ptranking_amd.install()` drops them into
ptranking.ltr_adhoc.eval.ltr.
"""
import ctypes as C
import os

import torch

from . import _lib
from . import dp
from . import functional as F_
from .host import DeviceEvaluator, DeviceTrainLoop, PointScorerRanker, is_multilabel
from .listsf import FusedListScorerMixin
from .scorer import (FlatAdagrad, FlatAdam, FlatRMSprop, FusedPointScorer, FusedScorerMixin, alloc_acts, mlp_forward, x6_wimg_for, x6_image_tag,
                     x6_set_image_tag)

RANKER_NAMES = ("RankNet", "LambdaRank", "LambdaLoss", "ApproxNDCG", "ListNet", "ListMLE", "STListNet", "RankCosine", "RankMSE", "SoftRank")
# SURVEY.md 2 marks these OUT OF SCOPE (the reference's driver cannot reach them): kept as classes for whoever asks for them by name
# (install(extras=True), pa.DASALC / pa.MDPRank), not part of the default drop-in surface
EXTRA_RANKER_NAMES = ("DASALC", "MDPRank")

# default hyper-parameters = the reference's `default_para_dict()`s
DEFAULT_PARAS = {
    "RankNet": dict(model_id="RankNet", sigma=1.0),                                    # pairwise/ranknet.py:57
    "LambdaRank": dict(model_id="LambdaRank", sigma=1.0),                              # listwise/lambdarank.py:78
    "LambdaLoss": dict(model_id="LambdaLoss", k=5, sigma=1.0, mu=5.0, loss_type="NDCG_Loss2"),   # listwise/lambdaloss.py:155-158
    "ApproxNDCG": dict(model_id="ApproxNDCG", alpha=10),                               # listwise/approxNDCG.py:131
    "ListNet": dict(model_id="ListNet"),
    "ListMLE": dict(model_id="ListMLE"),
    "STListNet": dict(model_id="STListNet", temperature=1.0),                          # listwise/st_listnet.py:70
    "RankCosine": dict(model_id="RankCosine"),
    "RankMSE": dict(model_id="RankMSE"),
    "SoftRank": dict(model_id="SoftRank", delta=2.0, metric='nDCG', top_k=None),       # listwise/softrank.py:97
    "DASALC": dict(model_id="DASALC"),
    "MDPRank": dict(model_id="MDPRank", temperature=1.0, gamma=1.0, top_k=10, distribution='PL'),   # listwise/mdprank.py:95-96
}


class FusedStepMixin:
    """zero_grad -> backward -> [one all_reduce(SUM) of the flat gradient under data parallelism] -> optimizer.step,
    the tail every reference loss ends with (e.g. lambdarank.py:58-60)."""

    data_parallel = True          # only takes effect when torch.distributed is initialised with world_size > 1
    _grad_bucket = None
    _dp_single = None

    # ---- direct train step: scorer forward -> fused loss kernel -> scorer backward -> [all-reduce] -> Adam as five C-ABI calls
    # One step of the autograd path costs ~0.37 ms of host time (autograd engine, tensor bookkeeping, optimizer hooks): at
    # <= 1024 queries per step that is more than the GPU needs.  When the scorer is the fused pointsf scorer, the optimiser is
    # FlatAdam and the ranker's loss is one of the single-kernel losses below, `DeviceTrainLoop.train_op` takes this path
    # instead: the SAME kernels with the SAME arguments in the SAME order (bit-identical parameters), no autograd graph.
    # A subclass that overrides custom_loss_function (the reference's plugin surface) is detected and keeps the autograd path.
    use_direct_step = True
    fuse_optimizer_step = True    # direct step on one device: optimiser step + loss-slot sum inside ptr_mlp_backward_step
    single_call_step = True       # r6: ... and the whole step as ONE C-ABI call (ptr_train_step: the same three entry points chained in C)
    reuse_weight_image = True     # r6: ... whose optimiser launch refreshes the bf16x6 forward's weight image, so the next step skips the prep launch
    _direct_entry = None          # (C-ABI entry point, lambda self, kwargs: [loss parameters]) — set by the loss mixins that qualify
    _direct_owner = None          # the class whose custom_loss_function the entry point implements

    def _direct_check(self, kwargs):
        """The reference's asserts of the loss (run on the direct path too)."""

    def _direct_train_op(self, X, Y, kwargs):
        spec = self._direct_entry
        sf = getattr(self, "point_sf", None)
        if (spec is None or not self.use_direct_step or not isinstance(sf, FusedPointScorer) or not isinstance(self.optimizer, FlatAdam)
                or type(self).custom_loss_function is not self._direct_owner.custom_loss_function or not sf.training
                or not torch.is_grad_enabled() or not (X.is_cuda and X.dim() == 3 and X.dtype == torch.float32 and X.is_contiguous())
                or not (Y.is_cuda and Y.dtype == torch.float32 and Y.is_contiguous() and Y.shape == X.shape[:2])
                or X.size(2) != sf.num_features or len(self.optimizer.param_groups) != 1):
            return None
        od = self.optimizer.__dict__       # an optimiser whose step / zero_grad were replaced on the instance (gradient accumulation hacks,
        if 'zero_grad' in od or ('step' in od and not getattr(od['step'], '_wrapped_by_lr_sched', False)):   # hooks) keeps its semantics:
            return None                    # autograd path (torch's lr schedulers wrap step() themselves — that wrapper is fine)
        self._direct_check(kwargs)
        lens = kwargs.get('lens')
        B, L, Fd = X.shape
        if L > _lib.MAX_LIST_LEN or (lens is not None and not (lens.is_cuda and lens.dtype == torch.int32 and lens.is_contiguous()
                                                               and lens.shape == (B,))):
            return None                    # the autograd path validates (functional._batch) and raises
        R, NL, dev = B * L, sf.num_layers, X.device
        cache = self.__dict__.setdefault("_direct_buffers", {})
        buf = cache.get((B, L))
        if buf is None:
            if len(cache) >= 4:                      # a few length buckets at most: do not pin scratch for every shape ever seen
                cache.pop(next(iter(cache)))
            ndz = _lib.query("ptr_mlp_backward_dz_floats", R, Fd, NL)
            buf = dict(preds=torch.empty((B, L), device=dev), acts=alloc_acts(R, NL, dev),
                       loss_q=torch.empty(max(B, 1), device=dev), dpreds=torch.empty((B, L), device=dev),
                       ws=torch.empty(_lib.query("ptr_mlp_backward_ws_floats", Fd, NL), device=dev),
                       dz=torch.empty(ndz, device=dev) if ndz else None)
            cache[(B, L)] = buf
        flat = sf.flat
        if flat.grad is None or flat.grad.shape != flat.shape or not flat.grad.is_contiguous():
            flat.grad = torch.empty_like(flat)
        p = sf.dropout
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0.0 else 0      # same CPU-generator draw as FusedPointScorer.forward
        if p > 0.0:
            seed = dp.local_dropout_seed(seed, R, local_queries=B)
        dp.end_step()                            # the recorded query slice described this batch only — with or without dropout (as _fused_step does)
        loss = torch.empty(1, device=dev)
        entry, params = spec
        distributed = self.data_parallel and dp.is_distributed()
        # single device: the optimiser step and the loss-slot sum ride in the backward's partial reduction (three launches fewer per step,
        # bit-identical results); under data parallelism the all-reduce sits between the gradient and the step
        fuse_step = self.fuse_optimizer_step and not distributed and type(self.optimizer) in (FlatAdam, FlatAdagrad, FlatRMSprop)
        if fuse_step and self.single_call_step:
            # r6: forward -> loss -> backward + step enqueued by ONE foreign call (ptr_train_step chains the very entry points used below:
            # bit-identical parameters); bench.py's per-stage timings (_lib.TIMING) come from events the call records between its stages
            return self._single_call_step(X, Y, lens, buf, flat, B, L, Fd, NL, R, p, seed, spec, kwargs, loss, dev)
        with torch.cuda.device(dev):
            st = _lib.current_stream(dev)
            mlp_forward(X, flat, R, Fd, NL, 1, p, seed, buf["preds"], buf["acts"], dev)
            stop_training = False
            if 'epoch_k' in kwargs and kwargs['epoch_k'] % self.stop_check_freq == 0:
                stop_training = self.stop_training(buf["preds"])
            # data parallel with a flat optimiser: backward -> flat gradient -> all-reduce -> ONE launch (optimiser step + loss-slot sum,
            # ptr_opt_step_loss): the single-device launch sequence + one kernel + one collective, nothing returns to autograd in between
            dp_fused = distributed and self.fuse_optimizer_step and type(self.optimizer) in (FlatAdam, FlatAdagrad, FlatRMSprop)
            _lib.call(entry, _lib.ptr(buf["preds"]), _lib.ptr(Y), _lib.ptr(lens), B, L, *params(self, kwargs),
                      None if (fuse_step or dp_fused) else _lib.ptr(loss), _lib.ptr(buf["loss_q"]), _lib.ptr(buf["dpreds"]), st)
            if fuse_step:
                kind, lr, h1, h2, eps, wd, step, s1, s2 = self.optimizer.fused_step_args(flat)
                try:
                    self._launch_backward_step(X, flat, buf, R, Fd, NL, p, seed, kind, lr, h1, h2, eps, wd, step, s1, s2, B, loss, st)
                except Exception:
                    self.optimizer.state[flat]["step"] -= 1      # the launch failed: the bias-correction counter must not run ahead
                    raise
            else:
                _lib.call("ptr_mlp_backward", _lib.ptr(X), _lib.ptr(flat), _lib.ptr(buf["acts"]), _lib.ptr(buf["dpreds"]), R, Fd, NL, C.c_float(p),
                          C.c_uint64(seed), _lib.ptr(buf["dz"]), _lib.ptr(buf["ws"]), _lib.ptr(flat.grad), st)
                if distributed:
                    dp.all_reduce_sum(flat.grad)
                if dp_fused:
                    kind, lr, h1, h2, eps, wd, step, s1, s2 = self.optimizer.fused_step_args(flat)
                    try:
                        _lib.call("ptr_opt_step_loss", _lib.ptr(flat), _lib.ptr(flat.grad), C.c_int64(flat.numel()), kind, C.c_float(lr), C.c_float(h1),
                                  C.c_float(h2), C.c_float(eps), C.c_float(wd), step, _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(buf["loss_q"]), B,
                                  _lib.ptr(loss), st)
                    except Exception:
                        self.optimizer.state[flat]["step"] -= 1
                        raise
                else:
                    self.optimizer.step_flat(flat)
        return loss.reshape(()), stop_training

    def _single_call_step(self, X, Y, lens, buf, flat, B, L, Fd, NL, R, p, seed, spec, kwargs, loss, dev):
        entry, params = spec
        d = buf.get("desc")
        if d is None:
            d = buf["desc"] = _lib.TrainStepDesc()
            d.struct_bytes = C.sizeof(_lib.TrainStepDesc)
            d.loss_kind = _lib.LOSS_KINDS[entry]
            d.B, d.L, d.F, d.NL = B, L, Fd, NL
            for k in ("preds", "acts", "loss_q", "dpreds", "dz", "ws"):
                setattr(d, k, None if buf[k] is None else buf[k].data_ptr())
        pl = params(self, kwargs)                      # the loss parameters, as the separate entry point takes them
        if d.loss_kind == 3:                           # LambdaLoss: k, sigma, mu, loss_type, presort
            d.loss_i[0], d.loss_i[1], d.loss_i[2] = pl[0], pl[3], pl[4]
            d.loss_f[0], d.loss_f[1] = pl[1].value, pl[2].value
        elif pl:
            d.loss_f[0] = pl[0].value                  # RankNet / LambdaRank: sigma
        wimg = x6_wimg_for(X, R, Fd, NL, True, dev)    # the forward mlp_forward would choose for this call
        d.wimg = None if wimg is None else wimg.data_ptr()
        d.X, d.labels, d.lens = X.data_ptr(), Y.data_ptr(), (None if lens is None else lens.data_ptr())
        # is the weight image still the one the previous step left for exactly these parameters?  (tag: scorer.x6_image_tag)
        st_ = self.optimizer.state.get(flat)
        step_before = int(st_["step"]) if st_ else 0
        d.wimg_current = int(wimg is not None and self.reuse_weight_image and os.environ.get("PTR_REUSE_IMG", "1") != "0" and
                             x6_image_tag(dev, Fd, NL) == (flat.data_ptr(), flat._version, step_before, wimg.data_ptr()))
        timing = _lib.TIMING
        evs = None
        if timing is not None:                         # bench.py: HIP events recorded by the call itself between its stages
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            for k, ev in enumerate(evs):
                ev.record()                            # creates the handle; ptr_train_step re-records it at its place in the stream
                d.events[k] = ev.cuda_event
        else:
            for k in range(4):
                d.events[k] = None
        kind, lr, h1, h2, eps, wd, step, s1, s2 = self.optimizer.fused_step_args(flat)
        d.opt_kind, d.step, d.lr, d.hyper1, d.hyper2, d.eps, d.weight_decay = kind, step, lr, h1, h2, eps, wd
        d.p_drop, d.seed = p, seed
        d.params, d.grad, d.state1, d.state2 = flat.data_ptr(), flat.grad.data_ptr(), s1.data_ptr(), (None if s2 is None else s2.data_ptr())
        d.loss_out = loss.data_ptr()
        if wimg is not None:
            x6_set_image_tag(dev, Fd, NL, None)        # until the call has gone through, nobody may trust the image
        try:
            with torch.cuda.device(dev):
                _lib.TIMING = None                     # (the call's own events time its stages; no bracket around the whole call)
                try:
                    _lib.call("ptr_train_step", C.addressof(d), _lib.current_stream(dev))
                finally:
                    _lib.TIMING = timing
        except Exception:
            self.optimizer.state[flat]["step"] -= 1      # the launch failed: the bias-correction counter must not run ahead
            raise
        if wimg is not None:                           # the optimiser launch rewrote the image for the parameters it has just produced
            x6_set_image_tag(dev, Fd, NL, (flat.data_ptr(), flat._version, step, wimg.data_ptr()))
        if evs is not None:
            timing.setdefault("ptr_mlp_forward_x6" if wimg is not None else "ptr_mlp_forward", []).append((evs[0], evs[1]))
            timing.setdefault(entry, []).append((evs[1], evs[2]))
            timing.setdefault("ptr_mlp_backward_step", []).append((evs[2], evs[3]))
        stop_training = False
        if 'epoch_k' in kwargs and kwargs['epoch_k'] % self.stop_check_freq == 0:
            stop_training = self.stop_training(buf["preds"])      # the scores of THIS step's forward (the reference checks them before the loss; it steps either way)
        return loss.reshape(()), stop_training

    def _launch_backward_step(self, X, flat, buf, R, Fd, NL, p, seed, kind, lr, h1, h2, eps, wd, step, s1, s2, B, loss, st):
        _lib.call("ptr_mlp_backward_step", _lib.ptr(X), _lib.ptr(flat), _lib.ptr(buf["acts"]), _lib.ptr(buf["dpreds"]), R, Fd, NL, C.c_float(p),
                  C.c_uint64(seed), _lib.ptr(buf["dz"]), _lib.ptr(buf["ws"]), _lib.ptr(flat.grad), kind, C.c_float(lr), C.c_float(h1),
                  C.c_float(h2), C.c_float(eps), C.c_float(wd), step, _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(buf["loss_q"]), B, _lib.ptr(loss), st)

    def _bucket(self, extra=0):
        if self._grad_bucket is None or self._grad_bucket.extra != extra:
            if hasattr(self.optimizer, "grad_bucket"):
                # FlatViewAdam: the stack's gradients already live in ONE flat buffer that the optimiser steps on — a FlatGradBucket
                # would re-point every .grad away from it and Adam would step on zeros
                self._grad_bucket = self.optimizer.grad_bucket(extra)
            else:
                self._grad_bucket = dp.FlatGradBucket(list(self.get_parameters()), extra=extra)
        return self._grad_bucket

    def _fused_step(self, loss):
        if self.data_parallel and dp.is_distributed():
            if self._dp_single is None:
                ps = [p for p in self.get_parameters() if p.requires_grad]
                self._dp_single = ps[0] if len(ps) == 1 else False
            if self._dp_single is not False:
                # fused scorer: ONE flat parameter -> backward hands back one flat gradient, reduced in place
                # (no bucket memset, no accumulate-into-view copy)
                self.optimizer.zero_grad()
                loss.backward()
                if self._dp_single.grad is None:          # no gradient on this rank (e.g. an all-padding shard): every rank
                    self._dp_single.grad = torch.zeros_like(self._dp_single)   # must still join the collective
                stack = getattr(self.optimizer, "stack", None)
                if stack is not None:                     # flat-view optimiser: gradients that something re-pointed (module.zero_grad(),
                    stack.reattach_grads()                # a foreign bucket) must be back in the flat buffer BEFORE it is reduced
                dp.all_reduce_sum(self._dp_single.grad)
            else:
                bucket = self._bucket()
                bucket.zero()
                loss.backward()
                bucket.all_reduce()
        else:
            self.optimizer.zero_grad()
            loss.backward()
        self.optimizer.step()
        dp.end_step()                            # a query slice recorded by dp.shard_queries() described this step's batch only
        return loss


class RankNetLoss(FusedStepMixin):
    _direct_entry = ("ptr_ranknet_fwd_bwd", lambda self, kw: [C.c_float(float(self.sigma))])

    def custom_loss_function(self, batch_preds, batch_std_labels, **kwargs):
        """ptranking/ltr_adhoc/pairwise/ranknet.py:25-42"""
        return self._fused_step(F_.ranknet_loss(batch_preds, batch_std_labels, sigma=self.sigma, lens=kwargs.get('lens')))


class LambdaRankLoss(FusedStepMixin):
    _direct_entry = ("ptr_lambdarank_fwd_bwd", lambda self, kw: [C.c_float(float(self.sigma))])

    def _direct_check(self, kwargs):
        assert 'label_type' in kwargs and is_multilabel(kwargs['label_type'])
        assert 'presort' in kwargs and kwargs['presort'] is True  # aiming for direct usage of ideal ranking

    def custom_loss_function(self, batch_preds, batch_std_labels, **kwargs):
        """ptranking/ltr_adhoc/listwise/lambdarank.py:27-62"""
        assert 'label_type' in kwargs and is_multilabel(kwargs['label_type'])
        assert 'presort' in kwargs and kwargs['presort'] is True  # aiming for direct usage of ideal ranking
        return self._fused_step(F_.lambdarank_loss(batch_preds, batch_std_labels, sigma=self.sigma, lens=kwargs.get('lens')))


class LambdaLossLoss(FusedStepMixin):
    _direct_entry = ("ptr_lambdaloss_fwd_bwd",
                     lambda self, kw: [int(self.k), C.c_float(float(self.sigma)), C.c_float(float(getattr(self, 'mu', 5.0))),
                                       F_.LAMBDALOSS_TYPES[self.loss_type], int(bool('presort' in kw and kw['presort']))])

    def _direct_check(self, kwargs):
        assert is_multilabel(kwargs['label_type'])
        if self.loss_type not in F_.LAMBDALOSS_TYPES:
            raise NotImplementedError(f"LambdaLoss type {self.loss_type!r} (supported: {sorted(F_.LAMBDALOSS_TYPES)})")

    def custom_loss_function(self, batch_preds, batch_std_labels, **kwargs):
        """ptranking/ltr_adhoc/listwise/lambdaloss.py:73-138"""
        assert is_multilabel(kwargs['label_type'])
        presort = bool('presort' in kwargs and kwargs['presort'])
        loss = F_.lambdaloss_loss(batch_preds, batch_std_labels, k=self.k, sigma=self.sigma, mu=getattr(self, 'mu', 5.0),
                                  loss_type=self.loss_type, presort=presort, lens=kwargs.get('lens'))
        return self._fused_step(loss)


class ApproxNDCGLoss(FusedStepMixin):
    couple_batch = True   # the reference's [B]/[B,1] broadcast (SURVEY.md §7 vi); False = per-query normalisation

    def uniform_eval_setting(self, **kwargs):
        """ptranking/ltr_adhoc/listwise/approxNDCG.py:78-81"""
        eval_dict = kwargs['eval_dict']
        if eval_dict["do_validation"] and not eval_dict['vali_metric'] == 'nDCG':
            eval_dict['vali_metric'] = "nDCG"

    def custom_loss_function(self, batch_preds, batch_std_labels, **kwargs):
        """ptranking/ltr_adhoc/listwise/approxNDCG.py:83-109"""
        assert is_multilabel(kwargs['label_type'])
        presort = bool('presort' in kwargs and kwargs['presort'])
        if not (self.data_parallel and dp.is_distributed() and self.couple_batch):
            loss = F_.approxndcg_loss(batch_preds, batch_std_labels, alpha=self.alpha, presort=presort,
                                      couple_batch=self.couple_batch, lens=kwargs.get('lens'))
            return self._fused_step(loss)
        # Data parallel + batch coupling: the global loss is -(sum_all DCG)(sum_all 1/IDCG).  Every rank back-propagates
        # with scale 1, ships its local S = sum 1/IDCG and D = sum DCG in the gradient bucket, and rescales by the global S
        # after the single all-reduce (gradients are linear in S) — SURVEY.md §8e.
        loss1, parts = F_.approxndcg_loss(batch_preds, batch_std_labels, alpha=self.alpha, presort=presort,
                                          couple_batch=True, grad_scale_override=1.0, return_parts=True, lens=kwargs.get('lens'))
        bucket = self._bucket(extra=2)
        bucket.zero()
        loss1.backward()
        bucket.extras[0] = parts["scale"][1]
        bucket.extras[1] = -loss1.detach()            # = local sum of DCG (scale 1)
        bucket.all_reduce()
        S, D = bucket.extras[0].clone(), bucket.extras[1].clone()
        bucket.flat[:bucket.numel].mul_(S)
        self.optimizer.step()
        return -(D * S)


class SoftRankLoss(FusedStepMixin):
    def custom_loss_function(self, batch_preds, batch_std_labels, **kwargs):
        """ptranking/ltr_adhoc/listwise/softrank.py:33-78"""
        assert 'presort' in kwargs and kwargs['presort'] is True  # aiming for direct usage of ideal ranking
        assert 'nDCG' == self.metric
        assert is_multilabel(kwargs['label_type'])
        return self._fused_step(F_.softrank_loss(batch_preds, batch_std_labels, delta=self.delta_value, top_k=self.top_k,
                                                 lens=kwargs.get('lens')))


class MDPRankLoss(FusedStepMixin):
    def custom_loss_function(self, batch_preds, batch_std_labels, **kwargs):
        """ptranking/ltr_adhoc/listwise/mdprank.py:24-78.  The reference asserts batch size 1 ("aiming for meaningful
        batch-normalization"); every query is an independent episode here, so any batch size works and B = 1 reproduces it."""
        assert 'presort' in kwargs and kwargs['presort'] is True  # aiming for direct usage of ideal ranking
        lens = kwargs.get('lens')
        with torch.no_grad():
            det = batch_preds.detach()
            if lens is not None:      # padded documents must be sampled last: they get (numerically) zero probability mass
                pad = torch.arange(det.size(1), device=det.device)[None, :] >= lens[:, None].to(det.device)
                det = det.masked_fill(pad, -1e30)
            if 'PL' == self.distribution:           # sampling_utils.py:32-58 (torch.multinomial without replacement)
                t = det / self.temperature if 1.0 != self.temperature else det
                probs = torch.exp(t - torch.max(t, dim=1, keepdim=True)[0]).clamp_min(1e-38)
                perm = torch.multinomial(probs, num_samples=det.size(1), replacement=False)
                if lens is not None:  # real documents whose exp() underflowed to the same clamp as the padding may be drawn AFTER a
                    # padded one: stable-partition every sampled ranking so that all real documents come first
                    is_pad = (perm >= lens[:, None].to(perm.device)).to(torch.int8)
                    perm = torch.gather(perm, 1, torch.sort(is_pad, dim=1, stable=True)[1])
                noise = None
            elif 'STPL' == self.distribution:       # sampling_utils.py:61-83 (Gumbel perturbation, then sort)
                unif = torch.rand(det.size(), device=det.device)
                noise = -torch.log(-torch.log(unif + 1e-20) + 1e-20)
                logits = det + noise if 1.0 == self.temperature else (det + noise) / self.temperature
                perm = torch.sort(logits, dim=1, descending=True)[1]
            else:
                raise NotImplementedError
        if noise is None:
            action_preds = batch_preds                                   # gathered in sample order inside the kernel
        else:
            action_preds = batch_preds + noise if 1.0 == self.temperature else (batch_preds + noise) / self.temperature
        loss = F_.mdprank_loss(action_preds, batch_std_labels, perm, top_k=self.top_k, gamma=self.gamma, lens=lens)
        return self._fused_step(loss)


class ListNetLoss(FusedStepMixin):
    _direct_entry = ("ptr_listnet_fwd_bwd", lambda self, kw: [])

    def custom_loss_function(self, batch_preds, batch_std_labels, **kwargs):
        """ptranking/ltr_adhoc/listwise/listnet.py:22-45"""
        return self._fused_step(F_.listnet_loss(batch_preds, batch_std_labels, lens=kwargs.get('lens')))


class STListNetLoss(FusedStepMixin):
    def custom_loss_function(self, batch_preds, batch_std_labels, **kwargs):
        """ptranking/ltr_adhoc/listwise/st_listnet.py:33-55"""
        unif = torch.rand(batch_preds.size(), device=batch_preds.device)  # [batch_size, ranking_size]
        return self._fused_step(F_.stlistnet_loss(batch_preds, batch_std_labels, temperature=self.temperature, unif=unif,
                                                  lens=kwargs.get('lens')))


class RankCosineLoss(FusedStepMixin):
    def custom_loss_function(self, batch_preds, batch_std_labels, **kwargs):
        """ptranking/ltr_adhoc/listwise/rank_cosine.py:24-38"""
        return self._fused_step(F_.rankcosine_loss(batch_preds, batch_std_labels, lens=kwargs.get('lens')))


class RankMSELoss(FusedStepMixin):
    def custom_loss_function(self, batch_preds, batch_std_labels, **kwargs):
        """ptranking/ltr_adhoc/pointwise/rank_mse.py:29-40.  The loss is a MEAN over the batch (the one in-scope loss that is not
        a sum over queries): under data parallelism the local gradient (scaled 1/B_local) is turned back into a sum, shipped with
        B_local in the same bucket, and divided by the global batch size after the single all-reduce."""
        loss = F_.rankmse_loss(batch_preds, batch_std_labels, lens=kwargs.get('lens'))
        if not (self.data_parallel and dp.is_distributed()):
            return self._fused_step(loss)
        b_local = float(batch_preds.size(0))
        bucket = self._bucket(extra=2)
        bucket.zero()
        loss.backward()
        bucket.flat[:bucket.numel].mul_(b_local)
        bucket.extras[0] = b_local
        bucket.extras[1] = loss.detach() * b_local
        bucket.all_reduce()
        b_global = bucket.extras[0].clone()
        bucket.flat[:bucket.numel].div_(b_global)
        self.optimizer.step()
        return bucket.extras[1] / b_global


class ListMLELoss(FusedStepMixin):
    # "device" (default): one counter-based HIP kernel shuffles the ties of the whole batch; "torch": the reference's construction, one
    # torch.randperm per query from the global torch RNG (B host-driven calls per step).  Neither reproduces the reference's CPU
    # random stream on a GPU; both draw tie orders uniformly (tests/test_parity_gpu.py::test_tie_shuffle_is_uniform_enough).
    tie_shuffle = "device"
    _tie_seed = 137           # ptranking/ltr_global.py:7
    _tie_calls = 0

    def _shuffle_ties(self, batch_std_labels, lens=None):
        if self.tie_shuffle == "device" or lens is not None:     # padded batches: the device kernel honours `lens`
            self._tie_calls += 1
            return F_.shuffle_ties_order(batch_std_labels, seed=self._tie_seed * 0x9E3779B1 + self._tie_calls, lens=lens)
        # ptranking/ltr_adhoc/util/sampling_utils.py:13-28, batched: one randperm per query from the global torch RNG
        B, L = batch_std_labels.shape
        dev = batch_std_labels.device
        if B > 1:
            rperms = torch.stack([torch.randperm(L, device=dev) for _ in range(B)], dim=0)
        else:
            rperms = torch.randperm(L, device=dev).view(1, -1)
        shuffled = torch.gather(batch_std_labels, dim=1, index=rperms)
        desc = torch.argsort(shuffled, descending=True)
        return torch.gather(rperms, dim=1, index=desc)

    def custom_loss_function(self, batch_preds, batch_std_labels, **kwargs):
        """ptranking/ltr_adhoc/listwise/listmle.py:73-104"""
        lens = kwargs.get('lens')
        perm = self._shuffle_ties(batch_std_labels, lens)   # shuffle per epoch rather than using the same order for a query
        return self._fused_step(F_.listmle_loss(batch_preds, perm, lens=lens))


for _cls in (RankNetLoss, LambdaRankLoss, LambdaLossLoss, ListNetLoss):
    _cls._direct_owner = _cls


def make_ranker_classes(base=PointScorerRanker):
    """Return {name: class} built on `base` (anything with the reference's AdhocNeuralRanker interface)."""

    class RankNet(RankNetLoss, FusedScorerMixin, FusedListScorerMixin, DeviceTrainLoop, DeviceEvaluator, base):
        def __init__(self, sf_para_dict=None, model_para_dict=None, gpu=False, device=None):
            base.__init__(self, id='RankNet', sf_para_dict=sf_para_dict, gpu=gpu, device=device)
            self.sigma = model_para_dict['sigma']

    class LambdaRank(LambdaRankLoss, FusedScorerMixin, FusedListScorerMixin, DeviceTrainLoop, DeviceEvaluator, base):
        def __init__(self, sf_para_dict=None, model_para_dict=None, gpu=False, device=None):
            base.__init__(self, id='LambdaRank', sf_para_dict=sf_para_dict, gpu=gpu, device=device)
            self.sigma = model_para_dict['sigma']

    class LambdaLoss(LambdaLossLoss, FusedScorerMixin, FusedListScorerMixin, DeviceTrainLoop, DeviceEvaluator, base):
        def __init__(self, sf_para_dict=None, model_para_dict=None, gpu=False, device=None):
            base.__init__(self, id='LambdaLoss', sf_para_dict=sf_para_dict, gpu=gpu, device=device)
            self.lambdaloss_dict = model_para_dict
            self.k, self.sigma, self.loss_type = model_para_dict['k'], model_para_dict['sigma'], model_para_dict['loss_type']
            if 'NDCG_Loss2++' == self.loss_type:
                self.mu = model_para_dict['mu']
            if self.loss_type not in F_.LAMBDALOSS_TYPES:
                raise NotImplementedError(f"loss_type {self.loss_type!r}")

    class ApproxNDCG(ApproxNDCGLoss, FusedScorerMixin, FusedListScorerMixin, DeviceTrainLoop, DeviceEvaluator, base):
        def __init__(self, sf_para_dict=None, model_para_dict=None, gpu=False, device=None):
            base.__init__(self, id='ApproxNDCG', sf_para_dict=sf_para_dict, gpu=gpu, device=device)
            self.alpha = model_para_dict['alpha']

    class ListNet(ListNetLoss, FusedScorerMixin, FusedListScorerMixin, DeviceTrainLoop, DeviceEvaluator, base):
        def __init__(self, sf_para_dict=None, gpu=False, device=None):
            base.__init__(self, id='ListNet', sf_para_dict=sf_para_dict, gpu=gpu, device=device)

    class ListMLE(ListMLELoss, FusedScorerMixin, FusedListScorerMixin, DeviceTrainLoop, DeviceEvaluator, base):
        def __init__(self, sf_para_dict=None, model_para_dict=None, gpu=False, device=None):
            base.__init__(self, id='ListMLE', sf_para_dict=sf_para_dict, gpu=gpu, device=device)

    class STListNet(STListNetLoss, FusedScorerMixin, FusedListScorerMixin, DeviceTrainLoop, DeviceEvaluator, base):
        def __init__(self, sf_para_dict=None, model_para_dict=None, gpu=False, device=None):
            base.__init__(self, id='STListNet', sf_para_dict=sf_para_dict, gpu=gpu, device=device)
            self.temperature = model_para_dict['temperature']

    class RankCosine(RankCosineLoss, FusedScorerMixin, FusedListScorerMixin, DeviceTrainLoop, DeviceEvaluator, base):
        def __init__(self, sf_para_dict=None, gpu=False, device=None):
            base.__init__(self, id='RankCosine', sf_para_dict=sf_para_dict, gpu=gpu, device=device)

    class RankMSE(RankMSELoss, FusedScorerMixin, FusedListScorerMixin, DeviceTrainLoop, DeviceEvaluator, base):
        def __init__(self, sf_para_dict=None, gpu=False, device=None):
            base.__init__(self, id='RankMSE', sf_para_dict=sf_para_dict, gpu=gpu, device=device)

    class SoftRank(SoftRankLoss, FusedScorerMixin, FusedListScorerMixin, DeviceTrainLoop, DeviceEvaluator, base):
        def __init__(self, sf_para_dict=None, model_para_dict=None, gpu=False, device=None):
            base.__init__(self, id='SoftRank', sf_para_dict=sf_para_dict, gpu=gpu, device=device)
            self.delta_value = float(model_para_dict['delta'])
            self.delta = torch.tensor([self.delta_value], device=self.device)     # softrank.py:28-29
            self.top_k = model_para_dict['top_k']
            self.metric = model_para_dict['metric']

    class DASALC(ListNetLoss, FusedScorerMixin, FusedListScorerMixin, DeviceTrainLoop, DeviceEvaluator, base):
        """ptranking/ltr_adhoc/listwise/dasalc.py:7-41: the top-1 ListNet loss on the listwise (listsf) scorer."""

        def __init__(self, sf_para_dict=None, gpu=False, device=None):
            base.__init__(self, id='DASALC', sf_para_dict=sf_para_dict, gpu=gpu, device=device)
            assert 'listsf' == sf_para_dict['sf_id']

    class MDPRank(MDPRankLoss, FusedScorerMixin, FusedListScorerMixin, DeviceTrainLoop, DeviceEvaluator, base):
        def __init__(self, sf_para_dict=None, model_para_dict=None, gpu=False, device=None):
            base.__init__(self, id='MDPRank', sf_para_dict=sf_para_dict, gpu=gpu, device=device)
            self.gamma = model_para_dict['gamma']
            self.top_k = model_para_dict['top_k']
            self.temperature = model_para_dict['temperature']
            self.distribution = model_para_dict['distribution']  # 'PL', 'STPL'
            self.pg_checking = False

    out = dict(RankNet=RankNet, LambdaRank=LambdaRank, LambdaLoss=LambdaLoss, ApproxNDCG=ApproxNDCG, ListNet=ListNet,
               ListMLE=ListMLE, STListNet=STListNet, RankCosine=RankCosine, RankMSE=RankMSE, SoftRank=SoftRank, DASALC=DASALC, MDPRank=MDPRank)
    for name, cls in out.items():
        cls.__name__ = cls.__qualname__ = name
        cls.__module__ = __name__
    return out


_standalone = make_ranker_classes(PointScorerRanker)
RankNet, LambdaRank, LambdaLoss = _standalone["RankNet"], _standalone["LambdaRank"], _standalone["LambdaLoss"]
ApproxNDCG, ListNet, ListMLE = _standalone["ApproxNDCG"], _standalone["ListNet"], _standalone["ListMLE"]
STListNet, RankCosine, RankMSE = _standalone["STListNet"], _standalone["RankCosine"], _standalone["RankMSE"]
SoftRank, DASALC, MDPRank = _standalone["SoftRank"], _standalone["DASALC"], _standalone["MDPRank"]
