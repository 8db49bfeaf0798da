"""GPU tests of the one-wavefront-per-query sort / metric kernels (register bitonic network + binary-search ranks, csrc/ptr_device.h
wave_sort_desc / count_ranks_wave; lists up to 1024 documents) against the oracle: every tiling boundary, ragged lengths, ties (the exact
fallback), signed zeros, infinities, cut-offs beyond one 64-position chunk.  Indices are bit-exact: (score descending, index ascending).

Reference: ptranking/base/ranker.py:46-60 (torch.sort of the predictions, gather of the labels, ideal sort of the labels),
ptranking/metric/adhoc/adhoc_metric.py:36-260.
"""
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


def _check_sort(F, preds, ln=None):
    from oracle import c_oracle as CO
    lens_t = None if ln is None else dev(ln)
    vals, idx = F.sort_desc(dev(preds), lens=lens_t)
    rv, ri = CO.sort_desc(preds, lens=ln)
    assert np.array_equal(idx.cpu().numpy(), ri)
    assert np.array_equal(vals.cpu().numpy().view(np.uint32), rv.view(np.uint32))      # bit for bit (signed zeros included)


def _check_metrics(F, preds, labels, ks, ln=None):
    from oracle import c_oracle as CO
    lens_t = None if ln is None else dev(ln)
    for presort in (False, True):
        lab = labels
        if presort:   # the caller's promise: labels already in ideal (descending) order, inside the valid prefix of each row
            lab = labels.copy()
            for q in range(lab.shape[0]):
                n = lab.shape[1] if ln is None else int(ln[q])
                lab[q, :n] = -np.sort(-lab[q, :n], kind="stable")
        out = F.metrics_at_ks(dev(preds), dev(lab), ks, presort=presort, lens=lens_t)
        ref = CO.metrics_at_ks(preds, lab, ks, presort, lens=ln)
        for m in ("ndcg", "nerr", "ap", "p"):
            G.assert_close(out[m].cpu().numpy(), ref[m], f"{m} presort={presort}")


LENGTHS = [1, 2, 3, 63, 64, 65, 100, 128, 129, 200, 255, 256, 257, 400, 512, 513, 1000, 1024, 1025]


@pytest.mark.parametrize("L", LENGTHS)
def test_sort_every_tiling_boundary(F, L):
    rng = np.random.default_rng(L)
    B = 37
    preds = rng.standard_normal((B, L)).astype(np.float32)
    _check_sort(F, preds)
    ln = rng.integers(0, L + 1, B).astype(np.int32)
    ln[0], ln[1] = L, 0
    _check_sort(F, preds, ln)


@pytest.mark.parametrize("L", [5, 64, 200, 256, 700, 1024])
def test_sort_with_ties_and_special_values(F, L):
    rng = np.random.default_rng(100 + L)
    B = 24
    preds = np.round(rng.standard_normal((B, L)) * 2).astype(np.float32)      # heavy ties: the exact fallback
    _check_sort(F, preds)
    one = rng.standard_normal((B, L)).astype(np.float32)
    if L >= 5:
        one[0, 1] = one[0, L - 1]                                              # a single tied pair
        one[1, 0], one[1, L - 1] = 0.0, -0.0                                   # signed zeros compare equal: index order
        one[2, 2] = np.inf; one[3, 3] = -np.inf                                # lone infinities are ordinary keys
        one[4, 0] = one[4, 4] = np.inf                                         # tied infinities
        one[5, 1] = one[5, 3] = -np.inf
        one[6, :] = 1.5                                                        # constant row
    _check_sort(F, one)
    ln = rng.integers(1, L + 1, B).astype(np.int32)
    two = one.copy()
    two[7, :] = np.arange(L, dtype=np.float32)
    if ln[7] < L:
        two[7, ln[7]:] = two[7, 0]                                             # ties only with entries BEYOND the valid length: not ties
    _check_sort(F, two, ln)
    _check_sort(F, preds, ln)


@pytest.mark.parametrize("L", [7, 64, 130, 256, 300, 1024])
def test_metrics_every_wave_tiling(F, L):
    rng = np.random.default_rng(200 + L)
    B = 33
    preds = rng.standard_normal((B, L)).astype(np.float32)
    labels = rng.integers(0, 5, (B, L)).astype(np.float32)
    labels[:, 0] = np.maximum(labels[:, 0], 1.0)                               # every valid prefix holds a relevant document (ideal DCG > 0)
    ks = [k for k in (1, 3, 5, 10, 20, 50, 64, 65, 100, 200, 256, 1000, 1024) if k <= max(L, 10)][:12]
    _check_metrics(F, preds, labels, ks)
    ln = rng.integers(1, L + 1, B).astype(np.int32)
    ln[0] = L
    _check_metrics(F, preds, labels, ks, ln)
    tied = np.round(preds * 2).astype(np.float32)                              # tied scores: ranks by original index
    _check_metrics(F, tied, labels, ks, ln)


def test_sort_large_batch_is_a_sorted_permutation(F):
    B, L = 65536, 256
    g = torch.Generator(device="cuda").manual_seed(3)
    p = torch.randn(B, L, device="cuda", generator=g)
    vals, idx = F.sort_desc(p)
    tv, ti = torch.sort(p, dim=1, descending=True, stable=True)
    assert torch.equal(vals, tv) and torch.equal(idx, ti)
