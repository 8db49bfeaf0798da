"""CPU: pin oracle/torch_ref.py (the torch-CPU restatement) against the golden fixtures produced by the reference."""
import numpy as np
import pytest
import torch

import golden_util as G
from oracle import torch_ref as T


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _presort(c):
    return bool(int(c["presort"])) if "presort" in c else True


@pytest.mark.parametrize("name", G.case_ids("ranknet"))
def test_ranknet(name):
    c = G.losses()["ranknet"][name]
    loss, grad = T.loss_and_grad(T.ranknet_loss, _t(c["preds"]), _t(c["labels"]), sigma=float(c["sigma"]))
    G.assert_close(loss.numpy(), c["loss"], "loss")
    G.assert_close(grad.numpy(), c["grad"], "grad")


@pytest.mark.parametrize("name", G.case_ids("lambdarank"))
def test_lambdarank(name):
    c = G.losses()["lambdarank"][name]
    loss, grad = T.loss_and_grad(T.lambdarank_loss, _t(c["preds"]), _t(c["labels"]), sigma=float(c["sigma"]))
    G.assert_close(loss.numpy(), c["loss"], "loss")
    G.assert_close(grad.numpy(), c["grad"], "grad")
    assert np.array_equal(T.sort_desc(_t(c["preds"]))[1].numpy(), c["sort_idx"])


@pytest.mark.parametrize("name", G.case_ids("lambdaloss"))
def test_lambdaloss(name):
    c = G.losses()["lambdaloss"][name]
    loss, grad = T.loss_and_grad(T.lambdaloss_loss, _t(c["preds"]), _t(c["labels"]), k=int(c["k"]),
                                 sigma=float(c["sigma"]), mu=float(c["mu"]), loss_type=int(c["loss_type"]),
                                 presort=_presort(c))
    G.assert_close(loss.numpy(), c["loss"], "loss")
    G.assert_close(grad.numpy(), c["grad"], "grad")


@pytest.mark.parametrize("name", G.case_ids("approxndcg"))
def test_approxndcg(name):
    c = G.losses()["approxndcg"][name]
    loss, grad = T.loss_and_grad(T.approxndcg_loss, _t(c["preds"]), _t(c["labels"]), alpha=float(c["alpha"]),
                                 presort=_presort(c))
    G.assert_close(loss.numpy(), c["loss"], "loss")
    G.assert_close(grad.numpy(), c["grad"], "grad")


@pytest.mark.parametrize("name", G.case_ids("listnet"))
def test_listnet(name):
    c = G.losses()["listnet"][name]
    loss, grad = T.loss_and_grad(T.listnet_loss, _t(c["preds"]), _t(c["labels"]))
    G.assert_close(loss.numpy(), c["loss"], "loss")
    G.assert_close(grad.numpy(), c["grad"], "grad")


@pytest.mark.parametrize("name", G.case_ids("listmle"))
def test_listmle(name):
    c = G.losses()["listmle"][name]
    loss, grad = T.loss_and_grad(T.listmle_loss, _t(c["preds"]), _t(c["perm"]))
    G.assert_close(loss.numpy(), c["loss"], "loss")
    G.assert_close(grad.numpy(), c["grad"], "grad")


def test_arg_shuffle_ties_is_a_tie_respecting_permutation():
    labels = torch.tensor([[2., 2., 1., 1., 1., 0., 0., 0.]]).repeat(5, 1)
    perm = T.arg_shuffle_ties(labels, generator=torch.Generator().manual_seed(3))
    assert sorted(perm[0].tolist()) == list(range(8))
    assert torch.equal(torch.gather(labels, 1, perm), labels)       # still label-descending
    assert len({tuple(p.tolist()) for p in perm}) > 1               # ties really are shuffled


# ------------------------------------------------------------------ metrics: the reference's known-answer vectors
@pytest.mark.parametrize("name", G.case_ids("kat", "metrics"))
def test_known_answer_vectors(name):
    c = G.metrics()["kat"][name]
    fn = {"ap": T.ap_at_ks, "ndcg": T.ndcg_at_ks, "nerr": T.nerr_at_ks}[str(c["kind"])]
    out = fn(_t(c["sys_sorted"]), _t(c["ideal_sorted"]), [int(k) for k in c["ks"]]).numpy()
    G.assert_close(out, c["expected"], "vs reference output")
    assert np.allclose(out[0], c["commented"], atol=5e-5), "vs the value commented in testing_metric.py"


@pytest.mark.parametrize("name", G.case_ids("rand", "metrics"))
def test_evaluator_metrics(name):
    c = G.metrics()["rand"][name]
    ks = [int(k) for k in c["ks"]]
    out = T.evaluate_at_ks(_t(c["preds"]), _t(c["labels"]), ks, presort=bool(int(c["presort"])))
    assert np.array_equal(out["sort_idx"].numpy(), c["sort_idx"])
    for m in ("ndcg", "nerr", "ap", "p"):
        G.assert_close(out[m].numpy(), c[m], m)
