"""GPU tests of the ApproxNDCG ring kernel (csrc/approxndcg.hip approxndcg_ring_kernel: one wavefront per query, both pair passes out of
registers, lists up to 512 documents) against the oracle and against the LDS kernel it replaces (PTR_APPROX_RING=0): every ring size and
its boundaries, ragged lengths, unsorted labels (the value-only ideal sort), coupled / un-coupled batch normalisation, ties in the scores.

Reference: ptranking/ltr_adhoc/listwise/approxNDCG.py:19-27, :45-62, :83-109; Robust_Sigmoid ptranking/base/utils.py:57-95.
"""
import os

import numpy as np
import pytest
import torch

import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from ptranking_amd import functional
    return functional


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _lg(F, preds, labels, ring, **kw):
    old = os.environ.get("PTR_APPROX_RING")
    os.environ["PTR_APPROX_RING"] = "1" if ring else "0"
    try:
        p = dev(preds).requires_grad_(True)
        loss = F.approxndcg_loss(p, dev(labels), **kw)
        loss.backward()
        return float(loss.detach().cpu()), p.grad.detach().cpu().numpy()
    finally:
        if old is None:
            os.environ.pop("PTR_APPROX_RING", None)
        else:
            os.environ["PTR_APPROX_RING"] = old


def _data(seed, B, L, presort, lens):
    rng = np.random.default_rng(seed)
    preds = rng.standard_normal((B, L)).astype(np.float32)
    labels = rng.choice(5, size=(B, L), p=[0.5147, 0.3250, 0.1339, 0.0183, 0.0081]).astype(np.float32)
    labels[:, 0] = np.maximum(labels[:, 0], 1.0)
    ln = None
    if lens:
        ln = rng.integers(1, L + 1, B).astype(np.int32)
        ln[0] = L
    if presort:
        for q in range(B):
            n = L if ln is None else int(ln[q])
            labels[q, :n] = -np.sort(-labels[q, :n], kind="stable")
    return preds, labels, ln


@pytest.mark.parametrize("L", [1, 2, 5, 64, 65, 128, 129, 190, 192, 193, 256, 257, 300, 384, 385, 500, 512])
@pytest.mark.parametrize("presort", [True, False])
def test_ring_matches_the_oracle_and_the_lds_kernel(F, L, presort):
    from oracle import c_oracle as CO
    B = 11
    for lens in (False, True):
        preds, labels, ln = _data(7000 + L, B, L, presort, lens)
        lens_t = None if ln is None else dev(ln)
        for couple in (True, False):
            kw = dict(alpha=10.0, presort=presort, couple_batch=couple, lens=lens_t)
            loss, grad = _lg(F, preds, labels, True, **kw)
            ol, dcg, inv, g = CO.approxndcg(preds, labels, 10.0, presort, couple, lens=ln)
            G.assert_close(loss, ol, f"loss couple={couple} lens={lens}")
            G.assert_close(grad, g, f"grad couple={couple} lens={lens}")
            l2, g2 = _lg(F, preds, labels, False, **kw)
            G.assert_close(loss, l2, "ring vs LDS kernel: loss")
            G.assert_close(grad, g2, "ring vs LDS kernel: grad")


@pytest.mark.parametrize("alpha", [1.0, 10.0, 100.0])
def test_ring_tied_scores_and_alpha(F, alpha):
    from oracle import c_oracle as CO
    B, L = 9, 200
    preds, labels, ln = _data(31, B, L, True, True)
    preds = np.round(preds * 2).astype(np.float32) / 2                       # many exactly tied scores: delta == 0 pairs contribute 0.5 each way
    loss, grad = _lg(F, preds, labels, True, alpha=alpha, presort=True, couple_batch=True, lens=dev(ln))
    ol, dcg, inv, g = CO.approxndcg(preds, labels, alpha, True, True, lens=ln)
    G.assert_close(loss, ol, "loss")
    G.assert_close(grad, g, "grad")


def test_ring_full_size_properties(F):
    """BASELINE config 4's list length at a batch the oracle does not reach: run-to-run bit stability, shift invariance (gradients of a
    query sum to zero), queries independent of their batch neighbours (un-coupled mode)."""
    B, L = 2048, 512
    preds, labels, _ = _data(5, B, L, True, False)
    l1, g1 = _lg(F, preds, labels, True, alpha=10.0, presort=True, couple_batch=False)
    l2, g2 = _lg(F, preds, labels, True, alpha=10.0, presort=True, couple_batch=False)
    assert l1 == l2 and np.array_equal(g1, g2)
    assert np.abs(g1.sum(axis=1)).max() <= 2e-4 * max(1.0, np.abs(g1).max())
    la, ga = _lg(F, preds[:1000], labels[:1000], True, alpha=10.0, presort=True, couple_batch=False)
    assert np.array_equal(ga, g1[:1000])
    l3, g3 = _lg(F, preds, labels, False, alpha=10.0, presort=True, couple_batch=False)
    G.assert_close(g1, g3, "ring vs LDS kernel at 2048 x 512")
    assert abs(l1 - l3) <= 1e-5 * abs(l3)
