#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE ITSELF.

Run here (the build container), never on the GPU box:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports wildltr/ptranking read-only from /root/reference, feeds seeded synthetic
batches through the reference's own loss classes / metric functions on CPU and stores
inputs + outputs as small .npz files.  The reference cannot travel to the GPU box, the
fixtures do: every parity test (oracle vs golden on CPU, HIP vs golden on the GPU)
reads these files and nothing else.

How (loss, dL/dpreds) is extracted without the reference's optimizer side effects
(SURVEY.md §8c): the ranker class is constructed WITHOUT init() (so no scorer / no real
optimizer), `ranker.optimizer` is replaced by a stub whose zero_grad()/step() are
no-ops, and `custom_loss_function` is called on a leaf `preds` tensor; after the call
`preds.grad` is the reference gradient.

Reference entry points exercised:
  ptranking/ltr_adhoc/pairwise/ranknet.py:25-42
  ptranking/ltr_adhoc/listwise/lambdarank.py:27-62
  ptranking/ltr_adhoc/listwise/lambdaloss.py:73-138
  ptranking/ltr_adhoc/listwise/approxNDCG.py:83-109
  ptranking/ltr_adhoc/listwise/listnet.py:22-45
  ptranking/ltr_adhoc/listwise/listmle.py:73-104
  ptranking/metric/adhoc/adhoc_metric.py (P / AP / nERR / nDCG @k(s))
  ptranking/base/ranker.py:31-263 (Evaluator sort->gather prologue)
  testing/metric/testing_metric.py:17-61 (the reference's only known-answer vectors)
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference"
if not os.path.isdir(REF):
    raise SystemExit("the reference tree is only mounted in the build container")
sys.path.insert(0, REF)

import numpy as np
import torch

from ptranking.data.data_utils import LABEL_TYPE
from ptranking.ltr_adhoc.pairwise.ranknet import RankNet
from ptranking.ltr_adhoc.listwise.lambdarank import LambdaRank
from ptranking.ltr_adhoc.listwise.lambdaloss import LambdaLoss
from ptranking.ltr_adhoc.listwise.approxNDCG import ApproxNDCG
from ptranking.ltr_adhoc.listwise.listnet import ListNet
from ptranking.ltr_adhoc.listwise.listmle import ListMLE
from ptranking.ltr_adhoc.listwise.st_listnet import STListNet
from ptranking.ltr_adhoc.listwise.rank_cosine import RankCosine
from ptranking.ltr_adhoc.pointwise.rank_mse import RankMSE
import ptranking.ltr_adhoc.listwise.listmle as ref_listmle_mod
from ptranking.metric.adhoc.adhoc_metric import (
    torch_ndcg_at_k, torch_ndcg_at_ks, torch_ap_at_k, torch_ap_at_ks,
    torch_nerr_at_k, torch_nerr_at_ks, torch_precision_at_k, torch_precision_at_ks)

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 137  # ptranking/ltr_global.py:7
MSLR_P = [0.5147, 0.3250, 0.1339, 0.0183, 0.0081]   # testing/data/testing_data_utils.py:326
YAHOO_P = [0.2609, 0.3580, 0.2855, 0.0767, 0.0189]  # testing/data/testing_data_utils.py:342

SF = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3,
      "pointsf": dict(num_layers=3, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None, bn_affine=False)}


class _StubOptimizer:
    def zero_grad(self):
        pass

    def step(self):
        pass


def synth(rng, B, L, p=MSLR_P, presort=True):
    preds = rng.standard_normal((B, L)).astype(np.float32)
    labels = rng.choice(len(p), size=(B, L), p=np.asarray(p) / np.sum(p)).astype(np.float32)
    for b in range(B):               # min_rele=1 (ptranking/data/data_utils.py:398-400)
        if labels[b].max() < 1:
            labels[b, rng.integers(L)] = 1.0
    if presort:
        labels = -np.sort(-labels, axis=1)
    return preds, labels


def run_loss(ranker, preds, labels, presort=True, **extra):
    ranker.optimizer = _StubOptimizer()
    p = torch.from_numpy(preds).clone().requires_grad_(True)
    y = torch.from_numpy(labels).clone()
    loss = ranker.custom_loss_function(p, y, presort=presort, label_type=LABEL_TYPE.MultiLabel, **extra)
    return np.float32(loss.detach().item()), p.grad.detach().numpy().astype(np.float32)


def pred_sort_idx(preds):
    return torch.sort(torch.from_numpy(preds), dim=1, descending=True)[1].numpy().astype(np.int64)


def add(store, name, **arrays):
    for k, v in arrays.items():
        store[f"{name}/{k}"] = np.asarray(v)


SHAPES = [(3, 8), (4, 32), (3, 64), (2, 128), (2, 200), (2, 256)]


def gen_losses():
    store = {}
    rng = np.random.default_rng(SEED)
    torch.manual_seed(SEED)
    names = []
    for ci, (B, L) in enumerate(SHAPES):
        preds, labels = synth(rng, B, L)
        for sigma in ((1.0, 2.5) if ci < 2 else (1.0,)):
            tag = f"c{ci}_s{sigma:g}"
            # RankNet
            loss, grad = run_loss(RankNet(sf_para_dict=SF, model_para_dict={"sigma": sigma}, device="cpu"), preds, labels)
            add(store, f"ranknet/{tag}", preds=preds, labels=labels, sigma=np.float32(sigma), loss=loss, grad=grad)
            names.append(f"ranknet/{tag}")
            # LambdaRank
            loss, grad = run_loss(LambdaRank(sf_para_dict=SF, model_para_dict={"sigma": sigma}, device="cpu"), preds, labels)
            add(store, f"lambdarank/{tag}", preds=preds, labels=labels, sigma=np.float32(sigma), loss=loss, grad=grad,
                sort_idx=pred_sort_idx(preds))
            names.append(f"lambdarank/{tag}")
        # LambdaLoss: type code 1 = NDCG_Loss2, 2 = NDCG_Loss2++ (NDCG_Loss1 only broadcasts for B == 1, SURVEY §7 iii)
        for lt, code in (("NDCG_Loss2", 1), ("NDCG_Loss2++", 2)):
            for k in sorted({5, min(L, 40), L}):
                mpd = dict(k=k, sigma=1.0, loss_type=lt, mu=5.0)
                loss, grad = run_loss(LambdaLoss(sf_para_dict=SF, model_para_dict=mpd, device="cpu"), preds, labels)
                tag = f"c{ci}_t{code}_k{k}"
                add(store, f"lambdaloss/{tag}", preds=preds, labels=labels, sigma=np.float32(1.0), k=np.int32(k),
                    mu=np.float32(5.0), loss_type=np.int32(code), loss=loss, grad=grad)
                names.append(f"lambdaloss/{tag}")
        # ApproxNDCG (batch-coupled reference semantics, SURVEY §7 vi); alpha 10 is the default, 1.0 exercises the
        # sigma==1 branch of Robust_Sigmoid (ptranking/base/utils.py:68,76)
        for alpha in (10.0, 1.0):
            loss, grad = run_loss(ApproxNDCG(sf_para_dict=SF, model_para_dict={"alpha": alpha}, device="cpu"), preds, labels)
            tag = f"c{ci}_a{alpha:g}"
            add(store, f"approxndcg/{tag}", preds=preds, labels=labels, alpha=np.float32(alpha), loss=loss, grad=grad)
            names.append(f"approxndcg/{tag}")
        # ListNet
        loss, grad = run_loss(ListNet(sf_para_dict=SF, device="cpu"), preds, labels)
        add(store, f"listnet/c{ci}", preds=preds, labels=labels, loss=loss, grad=grad)
        names.append(f"listnet/c{ci}")
        # ListMLE: capture the tie-shuffled permutation the reference drew
        captured = {}
        orig = ref_listmle_mod.arg_shuffle_ties

        def spy(batch_rankings, descending=True, device=None):
            out = orig(batch_rankings=batch_rankings, descending=descending, device=device)
            captured["perm"] = out.numpy().astype(np.int64)
            return out

        ref_listmle_mod.arg_shuffle_ties = spy
        try:
            loss, grad = run_loss(ListMLE(sf_para_dict=SF, device="cpu"), preds, labels)
        finally:
            ref_listmle_mod.arg_shuffle_ties = orig
        add(store, f"listmle/c{ci}", preds=preds, labels=labels, perm=captured["perm"], loss=loss, grad=grad)
        names.append(f"listmle/c{ci}")

    # Yahoo-shaped labels, one case per pairwise/listwise family
    preds, labels = synth(rng, 3, 48, p=YAHOO_P)
    loss, grad = run_loss(LambdaRank(sf_para_dict=SF, model_para_dict={"sigma": 1.0}, device="cpu"), preds, labels)
    add(store, "lambdarank/yahoo", preds=preds, labels=labels, sigma=np.float32(1.0), loss=loss, grad=grad,
        sort_idx=pred_sort_idx(preds))
    loss, grad = run_loss(ApproxNDCG(sf_para_dict=SF, model_para_dict={"alpha": 10.0}, device="cpu"), preds, labels)
    add(store, "approxndcg/yahoo", preds=preds, labels=labels, alpha=np.float32(10.0), loss=loss, grad=grad)
    names += ["lambdarank/yahoo", "approxndcg/yahoo"]

    # ---- edge cases pinned in SURVEY.md Appendix A ----
    # saturation: sigmoid rounds to 1.0f, BCE log clamps at -100, gradient is exactly zero
    sp = np.array([[-60, -40, -20, 0, 20, 40]], np.float32)
    sl = np.array([[2, 2, 1, 1, 0, 0]], np.float32)
    loss, grad = run_loss(LambdaRank(sf_para_dict=SF, model_para_dict={"sigma": 1.0}, device="cpu"), sp, sl)
    add(store, "lambdarank/saturated", preds=sp, labels=sl, sigma=np.float32(1.0), loss=loss, grad=grad, sort_idx=pred_sort_idx(sp))
    loss, grad = run_loss(RankNet(sf_para_dict=SF, model_para_dict={"sigma": 1.0}, device="cpu"), sp, sl)
    add(store, "ranknet/saturated", preds=sp, labels=sl, sigma=np.float32(1.0), loss=loss, grad=grad)
    # moderately large score gaps (fp32 rounding of 1-p matters but nothing clamps)
    mp = np.array([[-6, 4.5, -2.25, 0.5, 3, 7.75, -9, 1.125]], np.float32)
    ml = np.array([[4, 3, 2, 2, 1, 0, 0, 0]], np.float32)
    loss, grad = run_loss(LambdaRank(sf_para_dict=SF, model_para_dict={"sigma": 1.0}, device="cpu"), mp, ml)
    add(store, "lambdarank/wide", preds=mp, labels=ml, sigma=np.float32(1.0), loss=loss, grad=grad, sort_idx=pred_sort_idx(mp))
    loss, grad = run_loss(RankNet(sf_para_dict=SF, model_para_dict={"sigma": 1.0}, device="cpu"), mp, ml)
    add(store, "ranknet/wide", preds=mp, labels=ml, sigma=np.float32(1.0), loss=loss, grad=grad)
    loss, grad = run_loss(ApproxNDCG(sf_para_dict=SF, model_para_dict={"alpha": 10.0}, device="cpu"), mp, ml)
    add(store, "approxndcg/wide", preds=mp, labels=ml, alpha=np.float32(10.0), loss=loss, grad=grad)
    mpd = dict(k=8, sigma=1.0, loss_type="NDCG_Loss2", mu=5.0)
    loss, grad = run_loss(LambdaLoss(sf_para_dict=SF, model_para_dict=mpd, device="cpu"), mp, ml)
    add(store, "lambdaloss/wide", preds=mp, labels=ml, sigma=np.float32(1.0), k=np.int32(8), mu=np.float32(5.0),
        loss_type=np.int32(1), loss=loss, grad=grad)
    # all-tied scores, L=6: CPU torch.sort keeps the input order (== our (score desc, index asc) rule)
    tp = np.zeros((1, 6), np.float32)
    tl = np.array([[2, 1, 1, 0, 0, 0]], np.float32)
    loss, grad = run_loss(LambdaRank(sf_para_dict=SF, model_para_dict={"sigma": 1.0}, device="cpu"), tp, tl)
    add(store, "lambdarank/tied", preds=tp, labels=tl, sigma=np.float32(1.0), loss=loss, grad=grad, sort_idx=pred_sort_idx(tp))
    # B == 1 (ApproxNDCG batch coupling degenerates to the per-query form)
    preds, labels = synth(rng, 1, 24)
    loss, grad = run_loss(ApproxNDCG(sf_para_dict=SF, model_para_dict={"alpha": 10.0}, device="cpu"), preds, labels)
    add(store, "approxndcg/b1", preds=preds, labels=labels, alpha=np.float32(10.0), loss=loss, grad=grad)
    # LambdaLoss / ApproxNDCG with presort=False (labels arrive unsorted; the reference sorts them first)
    preds, labels = synth(rng, 3, 40, presort=False)
    mpd = dict(k=10, sigma=1.0, loss_type="NDCG_Loss2", mu=5.0)
    loss, grad = run_loss(LambdaLoss(sf_para_dict=SF, model_para_dict=mpd, device="cpu"), preds, labels, presort=False)
    add(store, "lambdaloss/unsorted", preds=preds, labels=labels, sigma=np.float32(1.0), k=np.int32(10), mu=np.float32(5.0),
        loss_type=np.int32(1), loss=loss, grad=grad, presort=np.int32(0))
    loss, grad = run_loss(ApproxNDCG(sf_para_dict=SF, model_para_dict={"alpha": 10.0}, device="cpu"), preds, labels, presort=False)
    add(store, "approxndcg/unsorted", preds=preds, labels=labels, alpha=np.float32(10.0), loss=loss, grad=grad, presort=np.int32(0))
    # ListNet / ListMLE never look at presort; RankNet uses the input order as is
    loss, grad = run_loss(RankNet(sf_para_dict=SF, model_para_dict={"sigma": 1.0}, device="cpu"), preds, labels, presort=False)
    add(store, "ranknet/unsorted", preds=preds, labels=labels, sigma=np.float32(1.0), loss=loss, grad=grad)

    np.savez_compressed(os.path.join(HERE, "losses.npz"), **store)
    print(f"losses.npz: {len(store)} arrays, {len(set(k.rsplit('/', 1)[0] for k in store))} cases")


def gen_siblings():
    """STListNet (st_listnet.py:33-55), RankCosine (rank_cosine.py:24-38), RankMSE (rank_mse.py:13-40)."""
    store = {}
    rng = np.random.default_rng(SEED + 7)
    torch.manual_seed(SEED + 7)          # STListNet's torch.rand draws and MDPRank's multinomial samples: reproducible fixtures
    for ci, (B, L) in enumerate([(3, 8), (4, 32), (2, 128), (2, 300)]):
        preds, labels = synth(rng, B, L)
        loss, grad = run_loss(RankMSE(sf_para_dict=SF, device="cpu"), preds, labels)
        add(store, f"rankmse/c{ci}", preds=preds, labels=labels, loss=loss, grad=grad)
        loss, grad = run_loss(RankCosine(sf_para_dict=SF, device="cpu"), preds, labels)
        add(store, f"rankcosine/c{ci}", preds=preds, labels=labels, loss=loss, grad=grad)
        for T in (1.0, 2.5):
            captured = {}
            orig_rand = torch.rand

            def spy(*a, **k):
                out = orig_rand(*a, **k)
                captured["unif"] = out.numpy().astype(np.float32).copy()
                return out

            torch.rand = spy
            try:
                loss, grad = run_loss(STListNet(sf_para_dict=SF, model_para_dict={"temperature": T}, device="cpu"), preds, labels)
            finally:
                torch.rand = orig_rand
            add(store, f"stlistnet/c{ci}_t{T:g}", preds=preds, labels=labels, unif=captured["unif"], temperature=np.float32(T),
                loss=loss, grad=grad)
    # SoftRank (softrank.py:33-78) and LambdaLoss NDCG_Loss1 (lambdaloss.py:33-34; only runs at batch size 1 in the reference)
    from ptranking.ltr_adhoc.listwise.softrank import SoftRank
    for ci, (B, L) in enumerate([(3, 8), (4, 32), (2, 128), (2, 300)]):
        preds, labels = synth(rng, B, L)
        for delta, top_k in ((2.0, None), (1.0, None), (0.5, 5)):
            mpd = dict(delta=delta, top_k=top_k, metric="nDCG")
            loss, grad = run_loss(SoftRank(sf_para_dict=SF, model_para_dict=mpd, device="cpu"), preds, labels)
            add(store, f"softrank/c{ci}_d{delta:g}_k{top_k or 0}", preds=preds, labels=labels, delta=np.float32(delta),
                top_k=np.int32(top_k or 0), loss=loss, grad=grad)
    for ci, L in enumerate([8, 32, 128, 300]):
        preds, labels = synth(rng, 1, L)
        for k in (5, L):
            mpd = dict(k=k, sigma=1.0, loss_type="NDCG_Loss1", mu=5.0)
            loss, grad = run_loss(LambdaLoss(sf_para_dict=SF, model_para_dict=mpd, device="cpu"), preds, labels)
            add(store, f"lambdaloss1/c{ci}_k{k}", preds=preds, labels=labels, sigma=np.float32(1.0), k=np.int32(k), loss=loss,
                grad=grad, sort_idx=pred_sort_idx(preds))
    # nDCG with LABEL_TYPE.Permutation (labels = n - rank position, the MSLETOR "list" collections, data_utils.py:520-523):
    # the label itself is the gain (adhoc_metric.py:207-212,225-230)
    from ptranking.metric.adhoc.adhoc_metric import torch_ndcg_at_ks as _ndcg_ks, torch_ndcg_at_k as _ndcg_k
    for ci, (B, L) in enumerate([(3, 8), (2, 40), (2, 130)]):
        preds = rng.standard_normal((B, L)).astype(np.float32)
        labels = np.stack([rng.permutation(L) + 1 for _ in range(B)]).astype(np.float32)
        tp, tl = torch.from_numpy(preds), torch.from_numpy(labels)
        _, idx = torch.sort(tp, dim=1, descending=True)
        sys_sorted = torch.gather(tl, 1, idx)
        ideal = torch.sort(tl, dim=1, descending=True)[0]
        ks = [1, 3, 5, 10, 20, 50]
        add(store, f"permndcg/c{ci}", preds=preds, labels=labels, ks=np.asarray(ks, np.int32),
            ndcg=_ndcg_ks(sys_sorted, ideal, ks=ks, label_type=LABEL_TYPE.Permutation).numpy().astype(np.float32),
            ndcg_k=_ndcg_k(sys_sorted, ideal, k=min(5, L), label_type=LABEL_TYPE.Permutation).numpy().astype(np.float32))
    # MDPRank (mdprank.py:24-78; the reference only accepts batch size 1): the sampled ranking is captured by wrapping the
    # module-level sampling function, the fixture stores it next to the loss / gradient
    import ptranking.ltr_adhoc.listwise.mdprank as _mdp
    for ci, L in enumerate([12, 40, 150]):
        preds, labels = synth(rng, 1, L)
        for top_k, gamma in ((10, 1.0), (None, 0.9)):
            captured = {}
            orig = _mdp.sample_ranking_PL

            def spy(*a, **k):
                out = orig(*a, **k)
                captured["perm"] = out[0].numpy().astype(np.int64).copy()
                return out

            _mdp.sample_ranking_PL = spy
            try:
                mpd = dict(gamma=gamma, top_k=top_k, temperature=1.0, distribution='PL')
                loss, grad = run_loss(_mdp.MDPRank(sf_para_dict=SF, model_para_dict=mpd, device="cpu"), preds, labels)
            finally:
                _mdp.sample_ranking_PL = orig
            add(store, f"mdprank/c{ci}_k{top_k or 0}", preds=preds, labels=labels, perm=captured["perm"], top_k=np.int32(top_k or 0),
                gamma=np.float32(gamma), loss=loss, grad=grad)
    zp = np.zeros((2, 5), np.float32)                                  # zero score vector: CosineSimilarity's eps path
    zl = np.array([[2, 1, 1, 0, 0], [1, 0, 0, 0, 0]], np.float32)
    loss, grad = run_loss(RankCosine(sf_para_dict=SF, device="cpu"), zp, zl)
    add(store, "rankcosine/zero", preds=zp, labels=zl, loss=loss, grad=grad)
    np.savez_compressed(os.path.join(HERE, "siblings.npz"), **store)
    print(f"siblings.npz: {len(store)} arrays")


def gen_big():
    """Reference outputs at BASELINE.json's config shapes: C1 RankNet 8x32, C2 LambdaRank 4x128, the north-star
    kernel at 4x256, C3 ListNet / ListMLE 4x256 (label ties -> the reference's tie shuffle, captured), C4 ApproxNDCG 2x512 with the
    Yahoo label mix, C5's loss LambdaLoss NDCG_Loss2 2x256.  Kept in a file of its own so that losses.npz stays byte-identical."""
    store = {}
    rng = np.random.default_rng(SEED + 11)
    torch.manual_seed(SEED + 11)
    preds, labels = synth(rng, 8, 32)
    loss, grad = run_loss(RankNet(sf_para_dict=SF, model_para_dict={"sigma": 1.0}, device="cpu"), preds, labels)
    add(store, "ranknet/C1_8x32", preds=preds, labels=labels, sigma=np.float32(1.0), loss=loss, grad=grad)
    for tag, (B, L) in (("C2_4x128", (4, 128)), ("NS_4x256", (4, 256))):
        preds, labels = synth(rng, B, L)
        loss, grad = run_loss(LambdaRank(sf_para_dict=SF, model_para_dict={"sigma": 1.0}, device="cpu"), preds, labels)
        add(store, f"lambdarank/{tag}", preds=preds, labels=labels, sigma=np.float32(1.0), loss=loss, grad=grad,
            sort_idx=pred_sort_idx(preds))
    preds, labels = synth(rng, 4, 256)
    loss, grad = run_loss(ListNet(sf_para_dict=SF, device="cpu"), preds, labels)
    add(store, "listnet/C3_4x256", preds=preds, labels=labels, loss=loss, grad=grad)
    captured = {}
    orig = ref_listmle_mod.arg_shuffle_ties

    def spy(batch_rankings, descending=True, device=None):
        out = orig(batch_rankings=batch_rankings, descending=descending, device=device)
        captured["perm"] = out.numpy().astype(np.int64)
        return out

    ref_listmle_mod.arg_shuffle_ties = spy
    try:
        loss, grad = run_loss(ListMLE(sf_para_dict=SF, device="cpu"), preds, labels)
    finally:
        ref_listmle_mod.arg_shuffle_ties = orig
    add(store, "listmle/C3_4x256", preds=preds, labels=labels, perm=captured["perm"], loss=loss, grad=grad)
    preds, labels = synth(rng, 2, 512, p=YAHOO_P)
    loss, grad = run_loss(ApproxNDCG(sf_para_dict=SF, model_para_dict={"alpha": 10.0}, device="cpu"), preds, labels)
    add(store, "approxndcg/C4_2x512_yahoo", preds=preds, labels=labels, alpha=np.float32(10.0), loss=loss, grad=grad)
    preds, labels = synth(rng, 2, 256)
    mpd = dict(k=5, sigma=1.0, loss_type="NDCG_Loss2", mu=5.0)
    loss, grad = run_loss(LambdaLoss(sf_para_dict=SF, model_para_dict=mpd, device="cpu"), preds, labels)
    add(store, "lambdaloss/C5_2x256_k5", preds=preds, labels=labels, sigma=np.float32(1.0), k=np.int32(5), mu=np.float32(5.0),
        loss_type=np.int32(1), loss=loss, grad=grad)
    np.savez_compressed(os.path.join(HERE, "losses_big.npz"), **store)
    print(f"losses_big.npz: {len(store)} arrays")


def gen_knife():
    """The sigmoid-saturation knife edge: pairwise gaps with sigma * s_ij swept over [14, 19] in steps of 2^-6, both target
    orientations.  In this band ATen's fp32 sigmoid rounds to exactly 1.0 (from sigma * s_ij ~ 16.64 on: 1 - p < 2^-25), where the reference's BCE
    jumps from -log(1 - p) ~ 16.6 to the -100 clamp and its gradient to exactly 0 (lambdarank.py:52, lambda_utils.py:15, lambdaloss.py:118-119;
    Robust_Sigmoid, base/utils.py:57-95, for ApproxNDCG with alpha * (s_i - s_j)).  One query per gap value x: six documents with the scores
    (0, x, 1/4, x + 1/4, 1/2, x + 1/2) / sigma under the sorted labels (2, 2, 1, 1, 0, 0), so that every gap x + {-1/2 .. 1/2} occurs with the
    better document above AND below.  Kept in a file of its own so that losses.npz stays byte-identical."""
    store = {}
    xs = np.arange(14.0, 19.0 + 1e-9, 2.0 ** -6, dtype=np.float64)

    def lists(scale):
        base = np.stack([np.zeros_like(xs), xs, 0.25 + 0 * xs, xs + 0.25, 0.5 + 0 * xs, xs + 0.5], axis=1) / scale
        return base.astype(np.float32), np.tile(np.array([2, 2, 1, 1, 0, 0], np.float32), (len(xs), 1))

    for sigma in (1.0, 2.5):
        preds, labels = lists(sigma)
        tag = f"knife_s{sigma:g}"
        loss, grad = run_loss(LambdaRank(sf_para_dict=SF, model_para_dict={"sigma": sigma}, device="cpu"), preds, labels)
        add(store, f"lambdarank/{tag}", preds=preds, labels=labels, sigma=np.float32(sigma), loss=loss, grad=grad, sort_idx=pred_sort_idx(preds))
        loss, grad = run_loss(RankNet(sf_para_dict=SF, model_para_dict={"sigma": sigma}, device="cpu"), preds, labels)
        add(store, f"ranknet/{tag}", preds=preds, labels=labels, sigma=np.float32(sigma), loss=loss, grad=grad)
    preds, labels = lists(1.0)
    for lt, code in (("NDCG_Loss2", 1), ("NDCG_Loss2++", 2)):
        mpd = dict(k=6, sigma=1.0, loss_type=lt, mu=5.0)
        loss, grad = run_loss(LambdaLoss(sf_para_dict=SF, model_para_dict=mpd, device="cpu"), preds, labels)
        add(store, f"lambdaloss/knife_t{code}", preds=preds, labels=labels, sigma=np.float32(1.0), k=np.int32(6), mu=np.float32(5.0),
            loss_type=np.int32(code), loss=loss, grad=grad)
    # ApproxNDCG: the pair term is Robust_Sigmoid(alpha * (s_i - s_j)); alpha = 10 (the default) and alpha = 1 (the sigma == 1 branch)
    for alpha in (10.0, 1.0):
        preds, labels = lists(alpha)
        loss, grad = run_loss(ApproxNDCG(sf_para_dict=SF, model_para_dict={"alpha": alpha}, device="cpu"), preds, labels)
        add(store, f"approxndcg/knife_a{alpha:g}", preds=preds, labels=labels, alpha=np.float32(alpha), loss=loss, grad=grad)
    # "hug": the band's actual discontinuities.  For x in [14.6, 17.4] the reference's 1 - p is k * 2^-24 with k = .., 4, 2, 0 (p is an fp32 in
    # [0.5, 1]); -log(1 - p) and the gradient factor jump wherever k changes, and k = 0 is the -100 clamp / the exactly-zero gradient.  The four
    # thresholds x_k (smallest fp32 x at which ATen's sigmoid returns 1 - k * 2^-24) are found HERE by bisection on the reference's own
    # torch.sigmoid; the lists put gaps at x_k -/+ {2^-9, 2^-11, 2^-13} — 10^3 times further out than the ~1e-7 window in which two correctly
    # rounded exp implementations may disagree about k, close enough that a kernel whose p is off by more than a few ulp takes the wrong side
    def threshold(k):
        target = np.float32(1.0) - np.float32(k) * np.float32(2.0 ** -24)
        lo, hi = np.float32(14.0), np.float32(19.0)
        while np.nextafter(lo, hi) < hi:
            mid = np.float32((np.float64(lo) + np.float64(hi)) / 2)
            if torch.sigmoid(torch.tensor([mid]))[0].item() >= target: hi = mid
            else: lo = mid
        return float(hi)
    th = sorted({threshold(k) for k in range(0, 9)}, reverse=True)[:4]     # ATen evaluates 1 / (1 + exp(-x)): 1 + e is a multiple of 2^-23, so k is even
    hug = np.array([t + sg * 2.0 ** -e for t in th for e in (9, 11, 13) for sg in (-1.0, 1.0)], np.float64)

    def hug_lists(scale):
        base = np.stack([np.zeros_like(hug), hug, 0.25 + 0 * hug, hug + 0.25, 0.5 + 0 * hug, hug + 0.5], axis=1) / scale
        return base.astype(np.float32), np.tile(np.array([2, 2, 1, 1, 0, 0], np.float32), (len(hug), 1))
    preds, labels = hug_lists(1.0)
    loss, grad = run_loss(LambdaRank(sf_para_dict=SF, model_para_dict={"sigma": 1.0}, device="cpu"), preds, labels)
    add(store, "lambdarank/knife_hug", preds=preds, labels=labels, sigma=np.float32(1.0), loss=loss, grad=grad, sort_idx=pred_sort_idx(preds),
        thresholds=np.asarray(th, np.float64))
    loss, grad = run_loss(RankNet(sf_para_dict=SF, model_para_dict={"sigma": 1.0}, device="cpu"), preds, labels)
    add(store, "ranknet/knife_hug", preds=preds, labels=labels, sigma=np.float32(1.0), loss=loss, grad=grad)
    mpd = dict(k=6, sigma=1.0, loss_type="NDCG_Loss2", mu=5.0)
    loss, grad = run_loss(LambdaLoss(sf_para_dict=SF, model_para_dict=mpd, device="cpu"), preds, labels)
    add(store, "lambdaloss/knife_hug", preds=preds, labels=labels, sigma=np.float32(1.0), k=np.int32(6), mu=np.float32(5.0),
        loss_type=np.int32(1), loss=loss, grad=grad)
    loss, grad = run_loss(ApproxNDCG(sf_para_dict=SF, model_para_dict={"alpha": 1.0}, device="cpu"), preds, labels)
    add(store, "approxndcg/knife_hug", preds=preds, labels=labels, alpha=np.float32(1.0), loss=loss, grad=grad)
    # a longer list: 64 documents whose neighbours are 17/63 apart, so the gaps k * 17/63 cross the band at many (i, j) with mixed labels
    rng = np.random.default_rng(SEED + 23)
    L = 64
    preds = (np.arange(L, dtype=np.float64)[None, :] * (17.0 / 63.0) + rng.uniform(-0.02, 0.02, (4, L))).astype(np.float32)
    preds = np.take_along_axis(preds, np.stack([rng.permutation(L) for _ in range(4)]), axis=1)
    labels = -np.sort(-rng.choice(5, size=(4, L), p=MSLR_P).astype(np.float32), axis=1)
    labels[:, 0] = np.maximum(labels[:, 0], 1)
    loss, grad = run_loss(LambdaRank(sf_para_dict=SF, model_para_dict={"sigma": 1.0}, device="cpu"), preds, labels)
    add(store, "lambdarank/knife_L64", preds=preds, labels=labels, sigma=np.float32(1.0), loss=loss, grad=grad, sort_idx=pred_sort_idx(preds))
    loss, grad = run_loss(RankNet(sf_para_dict=SF, model_para_dict={"sigma": 1.0}, device="cpu"), preds, labels)
    add(store, "ranknet/knife_L64", preds=preds, labels=labels, sigma=np.float32(1.0), loss=loss, grad=grad)
    mpd = dict(k=64, sigma=1.0, loss_type="NDCG_Loss2", mu=5.0)
    loss, grad = run_loss(LambdaLoss(sf_para_dict=SF, model_para_dict=mpd, device="cpu"), preds, labels)
    add(store, "lambdaloss/knife_L64", preds=preds, labels=labels, sigma=np.float32(1.0), k=np.int32(64), mu=np.float32(5.0),
        loss_type=np.int32(1), loss=loss, grad=grad)
    loss, grad = run_loss(ApproxNDCG(sf_para_dict=SF, model_para_dict={"alpha": 1.0}, device="cpu"), preds, labels)
    add(store, "approxndcg/knife_L64", preds=preds, labels=labels, alpha=np.float32(1.0), loss=loss, grad=grad)
    np.savez_compressed(os.path.join(HERE, "losses_knife.npz"), **store)
    print(f"losses_knife.npz: {len(store)} arrays")


def gen_metrics():
    store = {}
    # --- the reference's own known-answer vectors, testing/metric/testing_metric.py:17-61 ---
    kat = [
        ("ap1", [1, 0, 1, 0, 1], [1, 1, 1, 1, 1], [1, 3, 5], "ap", [1.0000, 0.5556, 0.4533]),        # :20-23
        ("ap2", [1, 0, 1, 0, 1], [1, 1, 1, 0, 0], [1, 3, 5], "ap", [1.0000, 0.5556, 0.7556]),        # :27-30
        ("ap3", [1, 1, 0, 1, 0, 0, 1], [1, 1, 1, 1, 0, 0, 0], [1, 2, 3, 5, 7], "ap",
         [1.0000, 1.0000, 0.6667, 0.6875, 0.8304]),                                                  # :33-36
        ("ndcg1", [1, 1, 0, 1, 0, 0, 1], [1, 1, 1, 1, 0, 0, 0], [1, 2, 3, 4, 5, 6, 7], "ndcg",
         [1.0000, 1.0000, 0.7654, 0.8048, 0.8048, 0.8048, 0.9349]),                                  # :43-46
        ("nerr1", [3, 2, 4], [4, 3, 2], [1, 2, 3], "nerr", [0.4667, 0.5154, 0.6640]),                 # :54-58
    ]
    fn = {"ap": lambda s, i, ks: torch_ap_at_ks(s, i, ks=ks),
          "ndcg": lambda s, i, ks: torch_ndcg_at_ks(s, i, ks=ks),
          "nerr": lambda s, i, ks: torch_nerr_at_ks(s, i, ks=ks)}
    for name, sys_l, std_l, ks, kind, commented in kat:
        s = torch.tensor([sys_l], dtype=torch.float32)
        i = torch.tensor([std_l], dtype=torch.float32)
        out = fn[kind](s, i, ks).numpy().astype(np.float32)
        assert np.allclose(out[0], commented, atol=5e-5), (name, out, commented)
        add(store, f"kat/{name}", sys_sorted=s.numpy(), ideal_sorted=i.numpy(), ks=np.asarray(ks, np.int32),
            kind=np.array(kind), expected=out, commented=np.asarray(commented, np.float32))

    # --- Evaluator prologue + all four metrics on random batches (ptranking/base/ranker.py:202-263) ---
    rng = np.random.default_rng(SEED + 1)
    KS = [1, 3, 5, 10, 20, 50]  # ptranking/ltr_adhoc/eval/parameter.py:456
    for ci, (B, L, presort, p) in enumerate([(4, 8, True, MSLR_P), (3, 30, True, MSLR_P), (3, 64, False, MSLR_P),
                                             (2, 128, True, MSLR_P), (2, 256, True, YAHOO_P), (2, 50, False, YAHOO_P)]):
        preds, labels = synth(rng, B, L, p=p, presort=presort)
        tp, tl = torch.from_numpy(preds), torch.from_numpy(labels)
        _, idx = torch.sort(tp, dim=1, descending=True)
        sys_sorted = torch.gather(tl, 1, idx)
        ideal = tl if presort else torch.sort(tl, dim=1, descending=True)[0]
        out = dict(
            ndcg=torch_ndcg_at_ks(sys_sorted, ideal, ks=KS, label_type=LABEL_TYPE.MultiLabel),
            nerr=torch_nerr_at_ks(sys_sorted, ideal, ks=KS, label_type=LABEL_TYPE.MultiLabel, max_label=None),
            ap=torch_ap_at_ks(sys_sorted, ideal, ks=KS),
            p=torch_precision_at_ks(sys_sorted, ks=KS))
        k1 = min(5, L)
        single = dict(
            ndcg_k=torch_ndcg_at_k(sys_sorted, ideal, k=k1, label_type=LABEL_TYPE.MultiLabel),
            nerr_k=torch_nerr_at_k(sys_sorted, ideal, k=k1, label_type=LABEL_TYPE.MultiLabel, device="cpu"),
            ap_k=torch_ap_at_k(sys_sorted, ideal, k=k1),
            p_k=torch_precision_at_k(sys_sorted, k=k1))
        add(store, f"rand/c{ci}", preds=preds, labels=labels, presort=np.int32(presort), ks=np.asarray(KS, np.int32),
            sort_idx=idx.numpy().astype(np.int64), sorted_vals=torch.sort(tp, dim=1, descending=True)[0].numpy(),
            max_label=np.float32(ideal.max().item()), k1=np.int32(k1),
            **{k: v.numpy().astype(np.float32) for k, v in out.items()},
            **{k: v.numpy().astype(np.float32) for k, v in single.items()})
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **store)
    print(f"metrics.npz: {len(store)} arrays")


def gen_long():
    """r6: reference outputs of LambdaRank on LONG lists — 3x384 and 2x512 (the ring kernel's one-wave-per-SIMD form, lambdarank_ring_kernel<8>),
    2x700 and 1x1251 (the LDS kernel; 1 251 documents is MSLR-WEB30K's longest list, data_utils.py:118-123) — on the MSLR label mix, whose
    grade-0 tail fills whole 64-document slots (the blocks the ring kernel skips).  A file of its own: the older files stay byte-identical."""
    store = {}
    rng = np.random.default_rng(SEED + 23)
    torch.manual_seed(SEED + 23)
    for tag, (B, L) in (("L_3x384", (3, 384)), ("L_2x512", (2, 512)), ("L_2x700", (2, 700)), ("L_1x1251", (1, 1251))):
        preds, labels = synth(rng, B, L)
        loss, grad = run_loss(LambdaRank(sf_para_dict=SF, model_para_dict={"sigma": 1.0}, device="cpu"), preds, labels)
        add(store, f"lambdarank/{tag}", preds=preds, labels=labels, sigma=np.float32(1.0), loss=loss, grad=grad,
            sort_idx=pred_sort_idx(preds))
    np.savez_compressed(os.path.join(HERE, "losses_long.npz"), **store)
    print(f"losses_long.npz: {len(store)} arrays")


if __name__ == "__main__":
    if "--only-long" in sys.argv:
        gen_long()
    elif "--only-big" in sys.argv:
        gen_big()
    elif "--only-knife" in sys.argv:
        gen_knife()
    elif "--only-siblings" in sys.argv:
        gen_siblings()
    else:
        gen_losses()
        gen_metrics()
        gen_siblings()
        gen_big()
        gen_knife()
        gen_long()
    print("torch", torch.__version__, "numpy", np.__version__)
