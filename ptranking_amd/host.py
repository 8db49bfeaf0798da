"""Host-side mirror of the reference's plugin interface for the ltr_adhoc hot path (names, arguments, return types and
error behaviour follow wildltr/ptranking; the arithmetic runs in the HIP kernels behind ptranking_amd.functional).

  LABEL_TYPE              ptranking/data/data_utils.py:88-92
  DeviceEvaluator         ptranking/base/ranker.py:28-263   (class Evaluator; metrics computed on the GPU instead of
                          predict -> .cpu() -> torch.sort -> gather -> metric on the host)
  DeviceTrainLoop         ptranking/base/ranker.py:565-603  (train / train_op; no per-batch .item() host sync)
  PointScorerRanker       ptranking/base/ranker.py:479-561 + ptranking/base/point_ranker.py:9-74 +
                          ptranking/base/adhoc_ranker.py:7-87 for sf_id == 'pointsf' (the scorer the benchmark configs use)

When the reference package itself is importable, `ptranking_amd.install()` builds the same ranker classes on top of
the reference's own `AdhocNeuralRanker` instead of PointScorerRanker, so everything outside the hot path (listsf
scorer, BN variants, checkpoint naming...) is literally the reference's code.
"""
import os
from enum import Enum, unique, auto

import torch
import torch.nn as nn
import torch.optim as optim
from torch.optim.lr_scheduler import StepLR

from . import dp
from .linear import FusedLinear, FusedStack
from . import functional as F_
from .batching import unpack_batch


@unique
class LABEL_TYPE(Enum):
    """The types of labels of supported datasets (ptranking/data/data_utils.py:88-92)."""
    MultiLabel = auto()
    Permutation = auto()


def is_permutation(label_type):
    """True for LABEL_TYPE.Permutation (ours or the reference's enum member of the same name)."""
    return getattr(label_type, "name", None) == "Permutation"


def is_multilabel(label_type):
    """True for our LABEL_TYPE.MultiLabel and for the reference's enum member of the same name."""
    return getattr(label_type, "name", label_type) == "MultiLabel"


# ------------------------------------------------------------------------------------------------ evaluation
class DeviceEvaluator:
    """In-built evaluation APIs with the signature of ptranking/base/ranker.py:28-263.  Results are CPU float32 tensors
    of shape [1] (single cut-off) or [len(ks)], as in the reference; `device` is accepted and ignored (the reference
    passes 'cpu' because it evaluates on the host)."""

    def _to_dev(self, t):
        return t.to(self.device, non_blocking=True) if t.device != torch.device(self.device) else t

    def _run_eval(self, test_data, ks, presort, which, max_label=None, min_len=None, need_per_q=False, permutation_labels=False):
        """Shared loop of every Evaluator method.  Accepts the reference's (ids, X, Y) batches (all lists of a batch have the
        same length) and PaddedQueryBatches' (ids, X, Y, lens).  min_len: the single-cut-off methods skip queries with
        fewer than k documents (ranker.py:41-42) — per batch for the reference's loaders, per query for padded batches."""
        self.eval_mode()
        num_queries = 0
        sums = None
        per_q = {m: [] for m in which} if need_per_q else None
        for batch in test_data:
            batch_ids, batch_q_doc_vectors, batch_std_labels, lens = unpack_batch(batch)
            if min_len is not None and lens is None and batch_std_labels.size(1) < min_len:
                continue  # skip if the number of documents is smaller than k (ranker.py:41-42)
            lens_d = None if lens is None else self._to_dev(lens).to(torch.int32)
            self._batch_lens = lens_d
            Xd = self._to_dev(batch_q_doc_vectors)
            try:
                with scorer_lens(self, lens_d, Xd):
                    batch_preds = self.predict(Xd)
            finally:
                self._batch_lens = None
            out = F_.metrics_at_ks(batch_preds.detach(), self._to_dev(batch_std_labels).float(), ks, presort=presort,
                                   max_label=max_label, which=which, lens=lens_d, permutation_labels=permutation_labels)
            if sums is None:
                sums = {m: torch.zeros(len(ks), device=batch_preds.device) for m in which}
            if min_len is not None and lens_d is not None:
                keep = (lens_d >= min_len)
                n_kept = keep.sum()
                for m in which:
                    sums[m] += (out[m] * keep.unsqueeze(1)).sum(dim=0)
                num_queries = num_queries + n_kept          # stays on the device until the end
            else:
                num_queries = num_queries + len(batch_ids)
                for m in which:
                    sums[m] += out[m].sum(dim=0)
            if need_per_q:
                for m in which:
                    per_q[m].append(out[m].cpu())
        if getattr(self, 'distributed_eval', False) and dp.is_distributed():
            # every rank evaluated ITS shard of the queries: one all_reduce(SUM) of [sum metric@ks ..., num_queries] per pass
            # (SURVEY.md 8e); per-query lists stay local
            dev = torch.device(self.device)
            if sums is None:
                sums = {m: torch.zeros(len(ks), device=dev) for m in which}
            nq = num_queries if torch.is_tensor(num_queries) else torch.tensor(float(num_queries), device=dev)
            pack = torch.cat([sums[m].float() for m in which] + [nq.reshape(1).float().to(dev)])
            dp.all_reduce_sum(pack)
            for i, m in enumerate(which):
                sums[m] = pack[i * len(ks):(i + 1) * len(ks)]
            num_queries = pack[-1]
            if float(num_queries) == 0.0:
                sums = None
        if sums is None:   # nothing evaluated: the reference divides 0 by 0 here
            avg = {m: torch.zeros(len(ks)) / 0.0 for m in which}
        else:
            avg = {m: (sums[m] / num_queries).cpu() for m in which}
        return avg, per_q

    def ndcg_at_k(self, test_data=None, k=10, label_type=LABEL_TYPE.MultiLabel, presort=False, device='cpu'):
        """ranker.py:31-65; LABEL_TYPE.Permutation uses the label itself as gain (adhoc_metric.py:207-212)"""
        if not (is_multilabel(label_type) or is_permutation(label_type)):
            raise NotImplementedError
        return self._run_eval(test_data, [k], presort, ("ndcg",), min_len=k, permutation_labels=is_permutation(label_type))[0]["ndcg"]

    def ndcg_at_ks(self, test_data=None, ks=[1, 5, 10], label_type=LABEL_TYPE.MultiLabel, presort=False, device='cpu'):
        """ranker.py:67-95"""
        if not (is_multilabel(label_type) or is_permutation(label_type)):
            raise NotImplementedError
        return self._run_eval(test_data, ks, presort, ("ndcg",), permutation_labels=is_permutation(label_type))[0]["ndcg"]

    def nerr_at_k(self, test_data=None, k=10, label_type=LABEL_TYPE.MultiLabel, max_label=None, presort=False, device='cpu'):
        """ranker.py:97-128"""
        if not is_multilabel(label_type):
            raise NotImplementedError
        return self._run_eval(test_data, [k], presort, ("nerr",), max_label=max_label, min_len=k)[0]["nerr"]

    def ap_at_k(self, test_data=None, k=10, presort=False, device='cpu'):
        """ranker.py:130-160"""
        return self._run_eval(test_data, [k], presort, ("ap",), min_len=k)[0]["ap"]

    def p_at_k(self, test_data=None, k=10, device='cpu'):
        """ranker.py:162-187"""
        return self._run_eval(test_data, [k], True, ("p",), min_len=k)[0]["p"]

    def validation(self, vali_data=None, vali_metric=None, k=5, presort=False, max_label=None,
                   label_type=LABEL_TYPE.MultiLabel, device='cpu'):
        """ranker.py:189-200"""
        if 'nDCG' == vali_metric:
            return self.ndcg_at_k(test_data=vali_data, k=k, label_type=label_type, presort=presort, device=device)
        elif 'nERR' == vali_metric:
            return self.nerr_at_k(test_data=vali_data, k=k, label_type=label_type, max_label=max_label, presort=presort,
                                  device=device)
        elif 'AP' == vali_metric:
            return self.ap_at_k(test_data=vali_data, k=k, presort=presort, device=device)
        elif 'P' == vali_metric:
            return self.p_at_k(test_data=vali_data, k=k, device=device)
        else:
            raise NotImplementedError

    def adhoc_performance_at_ks(self, test_data=None, ks=[1, 5, 10], label_type=LABEL_TYPE.MultiLabel, max_label=None,
                                presort=False, device='cpu', need_per_q=False):
        """ranker.py:202-263 -> (avg nDCG, avg nERR, avg AP, avg P)[, per-query lists in the same order]"""
        if not is_multilabel(label_type):
            raise NotImplementedError
        avg, per_q = self._run_eval(test_data, ks, presort, ("ndcg", "nerr", "ap", "p"), max_label=max_label,
                                    need_per_q=need_per_q)
        if need_per_q:
            return (avg["ndcg"], avg["nerr"], avg["ap"], avg["p"], per_q["ndcg"], per_q["nerr"], per_q["ap"], per_q["p"])
        return avg["ndcg"], avg["nerr"], avg["ap"], avg["p"]


_BN_NAMES = ("_BatchNormOverDocs", "_BatchNormPerQuery", "LTRBatchNorm", "LTRBatchNorm2", "BatchNorm1d")


def _scorer_modules(ranker):
    mods = []
    for name in ("point_sf", "list_sf"):
        sf = getattr(ranker, name, None)
        if isinstance(sf, nn.Module):
            mods.append(sf)
        elif isinstance(sf, dict):                          # listsf: {"head_ffnns": ..., "encoder": ..., "tail_ffnns": ...}
            mods += [m for m in sf.values() if isinstance(m, nn.Module)]
    return mods


class scorer_lens:
    """Context around a scorer forward on a PADDED query batch (`lens` given).  Zero-padded rows must not enter batch-norm statistics —
    the reference never pads (data_utils.py:683-742 batches equal-length lists only), and BN=True is its DEFAULT scorer
    (eval/parameter.py:145-146).  The fused stacks (linear.FusedStack on the GPU) take `batch_lens` and keep padded rows out of the 'BN'
    (batch x docs) and 'BN2' (per-query) statistics and their backward (csrc/bnact.hip), so a padded batch scores and trains like the
    unpadded lists.  A batch-norm scorer that is NOT on that path (CPU tensors, module-by-module fallbacks) fails loudly instead of
    silently training on different statistics."""

    def __init__(self, ranker, lens, X=None):
        self.ranker, self.lens, self.X, self.stacks = ranker, lens, X, []

    def __enter__(self):
        if self.lens is None:
            return self
        from .linear import FusedStack

        # Two passes: first VALIDATE every stack, then arm them — a refusal half way through the walk must not leave the stacks
        # already visited with these lens (the next unpadded forward with the same batch size would mask its BN statistics with them).
        found = []

        def walk(m):
            if isinstance(m, FusedStack):
                has_bn = any(type(c).__name__ in _BN_NAMES for c in m.children())
                # probe every stack with ITS OWN input width (a nested stack — the listsf tail behind the encoder — never sees the raw
                # features, and FusedStack.forward decides its module-by-module fallback from its own input)
                lin0 = next((c for c in m.modules() if isinstance(c, torch.nn.Linear)), None)
                width = lin0.in_features if lin0 is not None else 4
                probe = torch.empty((1, 1, width), device=self.lens.device if self.X is None else self.X.device)
                if has_bn and not m.handles_padding(probe):
                    self._refuse()
                found.append(m)
            elif type(m).__name__ in _BN_NAMES:
                self._refuse()
            else:
                for c in m.children():
                    walk(c)

        for top in _scorer_modules(self.ranker):
            walk(top)
        for m in found:
            m.batch_lens = self.lens
        self.stacks = found
        return self

    @staticmethod
    def _refuse():
        raise NotImplementedError("padded query batches (lens) cannot be scored by a scoring function with batch normalisation on this "
                                  "path (CPU tensors / unfused modules): the padded rows would enter the BN statistics; score on the GPU, "
                                  "batch equal-length lists or build the scorer with BN=False")

    def __exit__(self, *exc):
        for m in self.stacks:
            m.batch_lens = None
        return False


def _reject_padding_with_batchnorm(ranker, lens, X=None):
    """Raises when a padded batch would reach batch-norm statistics unmasked (see scorer_lens)."""
    with scorer_lens(ranker, lens, X):
        pass


# ------------------------------------------------------------------------------------------------ training loop
class DeviceTrainLoop:
    """train / train_op with the reference's contract (ranker.py:565-603) minus the per-batch `.item()` host sync:
    the running loss stays on the device; the caller still receives `(epoch_loss [1] tensor on self.device, stop)`."""

    def train(self, train_data, epoch_k=None, **kwargs):
        self.train_mode()
        assert 'label_type' in kwargs and 'presort' in kwargs
        label_type, presort = kwargs['label_type'], kwargs['presort']
        num_queries = 0
        epoch_loss = torch.zeros(1, device=self.device)
        stop_training = False
        for batch in train_data:
            batch_ids, batch_q_doc_vectors, batch_std_labels, lens = unpack_batch(batch)   # lens: padded batches only
            num_queries += len(batch_ids)
            batch_q_doc_vectors = batch_q_doc_vectors.to(self.device, non_blocking=True)
            batch_std_labels = batch_std_labels.to(self.device, non_blocking=True)
            extra = {} if lens is None else {"lens": lens.to(self.device, non_blocking=True).to(torch.int32)}
            batch_loss, stop_training = self.train_op(batch_q_doc_vectors, batch_std_labels, batch_ids=batch_ids,
                                                      epoch_k=epoch_k, presort=presort, label_type=label_type, **extra)
            if stop_training:
                break
            epoch_loss += batch_loss.detach().reshape(-1)[:1]
        epoch_loss = epoch_loss / num_queries
        return epoch_loss, stop_training

    def train_op(self, batch_q_doc_vectors, batch_std_labels, **kwargs):
        direct = getattr(self, "_direct_train_op", None)     # rankers.FusedStepMixin: five C-ABI calls instead of the autograd graph
        if direct is not None:
            out = direct(batch_q_doc_vectors, batch_std_labels, kwargs)
            if out is not None:
                return out
        stop_training = False
        self._batch_lens = kwargs.get('lens')          # padded batches: the listwise scorer masks padded documents as keys,
        try:                                           # batch norm keeps padded rows out of its statistics (scorer_lens)
            with scorer_lens(self, self._batch_lens, batch_q_doc_vectors):
                batch_preds = self.forward(batch_q_doc_vectors)
        finally:
            self._batch_lens = None
        if 'epoch_k' in kwargs and kwargs['epoch_k'] % self.stop_check_freq == 0:
            stop_training = self.stop_training(batch_preds)
        return self.custom_loss_function(batch_preds, batch_std_labels, **kwargs), stop_training


# ------------------------------------------------------------------------------------------------ standalone base ranker
_AF = {'R': nn.ReLU, 'LR': nn.LeakyReLU, 'RR': nn.RReLU, 'E': nn.ELU, 'SE': nn.SELU, 'CE': nn.CELU, 'GE': nn.GELU,
       'S': nn.Sigmoid, 'T': nn.Tanh}


def get_AF(af_str):
    """String identifier -> activation module (ptranking/base/utils.py:101-143)."""
    if af_str in _AF:
        return _AF[af_str]()
    raise NotImplementedError(af_str)


class _BatchNormOverDocs(nn.Module):
    """'BN': statistics over batch x docs (ptranking/base/utils.py:201-223)."""

    def __init__(self, num_features, momentum=0.1, affine=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, momentum=momentum, affine=affine, track_running_stats=False)

    def forward(self, X):
        return self.bn(X.permute(0, 2, 1)).permute(0, 2, 1) if X.dim() == 3 else self.bn(X)


class _BatchNormPerQuery(nn.Module):
    """'BN2': statistics over the documents of each query (ptranking/base/utils.py:227-286), quirks included: training vs
    inference is decided by torch.is_grad_enabled() (:229), the moving statistics are averaged over the batch (:242-245),
    gamma / beta are always parameters and `affine` adds a second weight / bias pair (:266-284)."""

    def __init__(self, num_features, momentum=0.1, affine=True, device=None):
        super().__init__()
        shape = (1, 1, num_features)
        self.gamma = nn.Parameter(torch.ones(shape, device=device))
        self.beta = nn.Parameter(torch.zeros(shape, device=device))
        self.moving_mean = torch.zeros(shape, device=device)
        self.moving_var = torch.ones(shape, device=device)
        self.momentum, self.affine = momentum, affine
        if affine:
            self.weight = nn.Parameter(torch.ones(shape, device=device))
            self.bias = nn.Parameter(torch.zeros(shape, device=device))

    def forward(self, X):
        eps = 1e-5
        if not torch.is_grad_enabled():
            X_hat = (X - self.moving_mean.to(X.device)) / torch.sqrt(self.moving_var.to(X.device) + eps)
        else:
            mean = X.mean(dim=1, keepdim=True)
            var = ((X - mean) ** 2).mean(dim=1, keepdim=True)
            X_hat = (X - mean) / torch.sqrt(var + eps)
            mm = (1.0 - self.momentum) * self.moving_mean.to(X.device) + self.momentum * mean
            mv = (1.0 - self.momentum) * self.moving_var.to(X.device) + self.momentum * var
            self.moving_mean, self.moving_var = mm.mean(dim=0, keepdim=True).data, mv.mean(dim=0, keepdim=True).data
        Y = self.gamma * X_hat + self.beta
        return Y * self.weight + self.bias if self.affine else Y


# nn.Linear on the hand-written fp32-MFMA kernels (csrc/linear.hip): forward, backward-input and a split-row backward-weight with a
# fixed-order reduction.  (Round 1 ran library GEMMs here, with a split-K wrapper around the library's 25-workgroup dW kernel.)
SplitKLinear = FusedLinear


def build_stacked_ffnet(ff_dims, AF=None, TL_AF=None, apply_tl_af=False, dropout=0.1, BN=True, bn_type=None, bn_affine=False,
                        device=None):
    """get_stacked_FFNet (ptranking/base/utils.py:288-356): (Dropout -> Linear[xavier_normal] -> [BN] -> AF) per hidden layer,
    then Linear [-> [BN] -> TL_AF].  Module names match the reference so state_dicts are interchangeable
    (dr_i / ff_{i+1} / bn_{i+1} / act_{i+1})."""
    assert ff_dims is not None and len(ff_dims) >= 2

    def bn(dim):
        if bn_type == 'BN':
            return _BatchNormOverDocs(dim, momentum=0.1, affine=bn_affine)
        if bn_type == 'BN2':
            return _BatchNormPerQuery(dim, momentum=0.1, affine=bn_affine, device=device)
        raise NotImplementedError(f"bn_type={bn_type!r}")

    # On the GPU the whole stack runs as one autograd node on the hand-written kernels (ptranking_amd/linear.py: FusedStack): GEMM
    # epilogues for ReLU / dropout, layer-wise statistics for bn_type='BN', every working activation of get_AF
    net = FusedStack()
    n = len(ff_dims)
    for i in range(1, n - 1):
        net.add_module(f'dr_{i}', nn.Dropout(dropout))
        lin = SplitKLinear(ff_dims[i - 1], ff_dims[i])
        nn.init.xavier_normal_(lin.weight)
        net.add_module(f'ff_{i + 1}', lin)
        if BN:
            net.add_module(f'bn_{i + 1}', bn(ff_dims[i]))
        net.add_module(f'act_{i + 1}', get_AF(AF))
    last = SplitKLinear(ff_dims[-2], ff_dims[-1])
    nn.init.xavier_normal_(last.weight)
    net.add_module(f'ff_{n}', last)
    if apply_tl_af:
        if BN:
            net.add_module(f'bn_{n}', bn(ff_dims[-1]))
        net.add_module(f'act_{n}', get_AF(TL_AF))
    return net


def build_pointsf(num_features=None, h_dim=100, out_dim=1, num_layers=3, AF='R', TL_AF='S', apply_tl_af=False, BN=True,
                  bn_type=None, bn_affine=False, dropout=0.1):
    """The stacked feed-forward scorer of ptranking/base/point_ranker.py:30-42."""
    return build_stacked_ffnet([num_features] + [h_dim] * num_layers + [out_dim], AF=AF, TL_AF=TL_AF, apply_tl_af=apply_tl_af,
                               dropout=dropout, BN=BN, bn_type=bn_type, bn_affine=bn_affine)


class PointScorerRanker:
    """Stand-alone equivalent of NeuralRanker + PointNeuralRanker + ListNeuralRanker + AdhocNeuralRanker
    (ptranking/base/ranker.py:499-545, point_ranker.py:17-71, list_ranker.py:284-400, adhoc_ranker.py:8-77)."""

    def __init__(self, id='AdhocNeuralRanker', sf_para_dict=None, weight_decay=1e-3, gpu=False, device=None):
        self.id = id
        self.gpu, self.device = gpu, device
        self.sf_para_dict = sf_para_dict
        self.sf_id = sf_para_dict['sf_id']
        assert self.sf_id in ['pointsf', 'listsf']                      # base/adhoc_ranker.py:18
        if 'listsf' == self.sf_id:
            self.encoder_type = self.sf_para_dict[self.sf_id]['encoder_type']
        self.opt, self.lr = sf_para_dict['opt'], sf_para_dict['lr']
        self.weight_decay = weight_decay
        self.stop_check_freq = 10

    # ---- ranker.py:499-545 / point_ranker.py:17-71
    def init(self):
        if 'listsf' == self.sf_id:
            self.list_sf = self.ini_listsf(**self.sf_para_dict[self.sf_id])
        else:
            self.point_sf = self.config_point_neural_scoring_function()
        self.config_optimizer()

    def ini_listsf(self, **kw):
        from .listsf import build_listsf
        list_sf = build_listsf(device=self.device, **kw)
        return {k: m.to(self.device) for k, m in list_sf.items()} if self.gpu else list_sf

    def config_point_neural_scoring_function(self):
        point_sf = self.ini_pointsf(**self.sf_para_dict[self.sf_para_dict['sf_id']])
        if self.gpu:
            point_sf = point_sf.to(self.device)
        return point_sf

    def ini_pointsf(self, **kw):
        return build_pointsf(**kw)

    def get_parameters(self):
        if 'listsf' == self.sf_id:                                       # list_ranker.py:296-300
            return list(self.list_sf['head_ffnns'].parameters()) + list(self.list_sf['encoder'].parameters()) + \
                list(self.list_sf['tail_ffnns'].parameters())
        return self.point_sf.parameters()

    def config_optimizer(self):
        if 'Adam' == self.opt:
            self.optimizer = optim.Adam(self.get_parameters(), lr=self.lr, weight_decay=self.weight_decay)
        elif 'RMS' == self.opt:
            self.optimizer = optim.RMSprop(self.get_parameters(), lr=self.lr, weight_decay=self.weight_decay)
        elif 'Adagrad' == self.opt:
            self.optimizer = optim.Adagrad(self.get_parameters(), lr=self.lr, weight_decay=self.weight_decay)
        else:
            raise NotImplementedError
        self.scheduler = StepLR(self.optimizer, step_size=20, gamma=0.5)

    def forward(self, batch_q_doc_vectors):
        if 'listsf' == self.sf_id:
            from .listsf import listsf_forward
            return listsf_forward(self.list_sf, self.encoder_type, batch_q_doc_vectors, getattr(self, '_batch_lens', None))
        batch_size, num_docs, num_features = batch_q_doc_vectors.size()
        _batch_preds = self.point_sf(batch_q_doc_vectors)
        return _batch_preds.view(-1, num_docs)

    def predict(self, batch_q_doc_vectors):
        return self.forward(batch_q_doc_vectors)

    def _modules(self):
        return list(self.list_sf.values()) if 'listsf' == self.sf_id else [self.point_sf]

    def eval_mode(self):
        for m in self._modules():
            m.eval()

    def train_mode(self):
        for m in self._modules():
            m.train(mode=True)

    def save(self, dir, name):
        if not os.path.exists(dir):
            os.makedirs(dir)
        if 'listsf' == self.sf_id:                                       # list_ranker.py:390-396
            torch.save({k: m.state_dict() for k, m in self.list_sf.items()}, dir + name)
        else:
            torch.save(self.point_sf.state_dict(), dir + name)

    def load(self, file_model, **kwargs):
        device = kwargs['device']
        if 'listsf' == self.sf_id:                                       # list_ranker.py:398-402
            checkpoint = torch.load(file_model, map_location=device)
            for k, m in self.list_sf.items():
                m.load_state_dict(checkpoint[k])
        else:
            self.point_sf.load_state_dict(torch.load(file_model, map_location=device))

    def get_tl_af(self):
        return self.sf_para_dict[self.sf_para_dict['sf_id']]['TL_AF']

    def uniform_eval_setting(self, **kwargs):
        pass

    def stop_training(self, batch_preds):
        """ranker.py:547-561"""
        if torch.nonzero(batch_preds, as_tuple=False).size(0) <= 0:
            print('All zero error.\n')
            return True
        if torch.isnan(batch_preds).any():
            print('Including NaN error.')
            return True
        return False

    def custom_loss_function(self, batch_preds, batch_std_labels, **kwargs):
        raise NotImplementedError

