"""GPU tests of the two-queries-per-wavefront form of the RankNet kernel (lists up to 32 documents — BASELINE config 1; csrc/pairwise.hip
pairwise_bce_kernel<32, 1, false>) against the C oracle: every length 1 .. 32, ragged batches whose two queries of a wavefront differ in
length, odd batch sizes (a half-filled last wavefront), sigma values, and agreement with the one-query-per-wavefront kernel (L = 33 .. 64
padded with `lens`).

Reference: ptranking/ltr_adhoc/pairwise/ranknet.py:25-42, ptranking/ltr_adhoc/util/lambda_utils.py:5-23.
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


def _lg(F, preds, labels, sigma, ln):
    p = dev(preds).requires_grad_(True)
    loss = F.ranknet_loss(p, dev(labels), sigma=sigma, lens=None if ln is None else dev(ln))
    loss.backward()
    return float(loss.detach().cpu()), p.grad.detach().cpu().numpy()


@pytest.mark.parametrize("L", list(range(1, 33)))
def test_every_length_up_to_32(F, L):
    from oracle import c_oracle as CO
    rng = np.random.default_rng(L)
    B = 13                                                     # odd: the last wavefront holds one query
    preds = rng.standard_normal((B, L)).astype(np.float32)
    labels = rng.integers(0, 5, (B, L)).astype(np.float32)
    for ln in (None, rng.integers(0, L + 1, B).astype(np.int32)):
        loss, grad = _lg(F, preds, labels, 1.0, ln)
        lq, g = CO.ranknet(preds, labels, 1.0, lens=ln)
        G.assert_close(loss, lq.astype(np.float64).sum(), "loss")
        G.assert_close(grad, g, "grad")


@pytest.mark.parametrize("sigma", [0.5, 1.0, 3.0])
def test_large_batch_sigma_and_wide_scores(F, sigma):
    from oracle import c_oracle as CO
    rng = np.random.default_rng(7)
    B, L = 4096, 32
    preds = (rng.standard_normal((B, L)) * 8).astype(np.float32)        # score gaps of tens: p rounds to 0 / 1, the -100 clamp is reached
    labels = rng.integers(0, 5, (B, L)).astype(np.float32)
    ln = rng.integers(1, L + 1, B).astype(np.int32)
    loss, grad = _lg(F, preds, labels, sigma, ln)
    lq, g = CO.ranknet(preds, labels, sigma, lens=ln)
    G.assert_close(loss, lq.astype(np.float64).sum(), "loss")
    G.assert_close(grad, g, "grad")
    l2, g2 = _lg(F, preds, labels, sigma, ln)
    assert loss == l2 and np.array_equal(grad, g2)                      # run-to-run bit stability


def test_packed_form_equals_the_one_query_per_wavefront_kernel(F):
    """The same queries as rows of width 32 (packed form) and as rows of width 48 padded through `lens` (one query per wavefront)."""
    rng = np.random.default_rng(3)
    B, L = 257, 32
    preds = rng.standard_normal((B, L)).astype(np.float32)
    labels = rng.integers(0, 5, (B, L)).astype(np.float32)
    ln = rng.integers(1, L + 1, B).astype(np.int32)
    la, ga = _lg(F, preds, labels, 1.0, ln)
    wide_p = np.zeros((B, 48), np.float32); wide_y = np.zeros((B, 48), np.float32)
    wide_p[:, :L] = preds; wide_y[:, :L] = labels
    lb, gb = _lg(F, wide_p, wide_y, 1.0, ln)
    assert abs(la - lb) <= 1e-6 * max(1.0, abs(lb))
    G.assert_close(ga, gb[:, :L], "packed vs one query per wavefront")
    assert np.all(gb[:, L:] == 0.0)
