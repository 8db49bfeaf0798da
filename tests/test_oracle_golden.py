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


# =================================================================== the plain-C oracle (closed-form gradients)
from oracle import c_oracle as CO


@pytest.mark.parametrize("name", G.case_ids("ranknet"))
def test_c_oracle_ranknet(name):
    c = G.losses()["ranknet"][name]
    lq, grad = CO.ranknet(c["preds"], c["labels"], float(c["sigma"]))
    G.assert_close(lq.astype(np.float64).sum(), c["loss"], "loss")
    G.assert_close(grad, c["grad"], "grad")


@pytest.mark.parametrize("name", G.case_ids("lambdarank"))
def test_c_oracle_lambdarank(name):
    c = G.losses()["lambdarank"][name]
    lq, grad = CO.lambdarank(c["preds"], c["labels"], float(c["sigma"]))
    G.assert_close(lq.astype(np.float64).sum(), c["loss"], "loss")
    G.assert_close(grad, c["grad"], "grad")
    assert np.array_equal(CO.sort_desc(c["preds"])[1], c["sort_idx"])


@pytest.mark.parametrize("name", G.case_ids("lambdaloss"))
def test_c_oracle_lambdaloss(name):
    c = G.losses()["lambdaloss"][name]
    lq, grad = CO.lambdaloss(c["preds"], c["labels"], int(c["k"]), float(c["sigma"]), float(c["mu"]),
                             int(c["loss_type"]), _presort(c))
    G.assert_close(lq.astype(np.float64).sum(), c["loss"], "loss")
    G.assert_close(grad, c["grad"], "grad")


@pytest.mark.parametrize("name", G.case_ids("approxndcg"))
def test_c_oracle_approxndcg(name):
    c = G.losses()["approxndcg"][name]
    loss, dcg, inv, grad = CO.approxndcg(c["preds"], c["labels"], float(c["alpha"]), _presort(c), True)
    G.assert_close(loss, c["loss"], "loss")
    G.assert_close(grad, c["grad"], "grad")
    G.assert_close(-(dcg.astype(np.float64).sum() * inv.astype(np.float64).sum()), c["loss"], "loss from slots")


@pytest.mark.parametrize("name", G.case_ids("listnet"))
def test_c_oracle_listnet(name):
    c = G.losses()["listnet"][name]
    lq, grad = CO.listnet(c["preds"], c["labels"])
    G.assert_close(lq.astype(np.float64).sum(), c["loss"], "loss")
    G.assert_close(grad, c["grad"], "grad")


@pytest.mark.parametrize("name", G.case_ids("listmle"))
def test_c_oracle_listmle(name):
    c = G.losses()["listmle"][name]
    lq, grad = CO.listmle(c["preds"], c["perm"])
    G.assert_close(lq.astype(np.float64).sum(), c["loss"], "loss")
    G.assert_close(grad, c["grad"], "grad")


@pytest.mark.parametrize("name", G.case_ids("rand", "metrics"))
def test_c_oracle_metrics(name):
    c = G.metrics()["rand"][name]
    out = CO.metrics_at_ks(c["preds"], c["labels"], c["ks"], bool(int(c["presort"])))
    for m in ("ndcg", "nerr", "ap", "p"):
        G.assert_close(out[m], c[m], m)
    vals, idx = CO.sort_desc(c["preds"])
    assert np.array_equal(idx, c["sort_idx"])
    assert np.array_equal(vals, c["sorted_vals"])


def test_c_oracle_known_answers_via_identity_scores():
    """Feed the reference's hand vectors through the full prologue: scores that already rank the list as given."""
    checked = 0
    for name in G.case_ids("kat", "metrics"):
        c = G.metrics()["kat"][name]
        sys_sorted = c["sys_sorted"]
        L = sys_sorted.shape[1]
        preds = -np.arange(L, dtype=np.float32)[None]          # keeps the given order
        kind = str(c["kind"])
        # only vectors whose 'ideal' row is the sorted 'system' row can go through the sort->gather prologue
        if sorted(sys_sorted[0].tolist(), reverse=True) == c["ideal_sorted"][0].tolist():
            out = CO.metrics_at_ks(preds, sys_sorted, c["ks"], presort=False)
            G.assert_close(out[kind][0], c["expected"][0], name)
            checked += 1
    assert checked >= 3


def test_c_oracle_padding_equals_unpadded():
    rng = np.random.default_rng(5)
    B, L = 5, 40
    lens = np.array([40, 17, 1, 33, 8], np.int32)
    preds = rng.standard_normal((B, L)).astype(np.float32)
    labels = -np.sort(-rng.integers(0, 5, (B, L)).astype(np.float32), axis=1)
    labels[:, 0] = np.maximum(labels[:, 0], 1)
    for fn, args in ((CO.lambdarank, (1.0,)), (CO.ranknet, (1.0,)), (CO.listnet, ())):
        lq, g = fn(preds, labels, *args, lens=lens)
        for b in range(B):
            n = lens[b]
            lq1, g1 = fn(preds[b:b + 1, :n], labels[b:b + 1, :n], *args)
            assert np.allclose(lq[b], lq1[0], rtol=1e-6, atol=1e-6)
            assert np.allclose(g[b, :n], g1[0], rtol=1e-6, atol=1e-6)
            assert np.all(g[b, n:] == 0)


# =================================================================== sibling losses
@pytest.mark.parametrize("name", G.case_ids("rankmse", "siblings"))
def test_rankmse_oracles(name):
    c = G.siblings()["rankmse"][name]
    loss, grad = T.loss_and_grad(T.rankmse_loss, _t(c["preds"]), _t(c["labels"]))
    G.assert_close(loss.numpy(), c["loss"], "torch loss"); G.assert_close(grad.numpy(), c["grad"], "torch grad")
    lq, g = CO.rankmse(c["preds"], c["labels"])
    G.assert_close(lq.astype(np.float64).mean(), c["loss"], "C loss"); G.assert_close(g, c["grad"], "C grad")


@pytest.mark.parametrize("name", G.case_ids("rankcosine", "siblings"))
def test_rankcosine_oracles(name):
    c = G.siblings()["rankcosine"][name]
    loss, grad = T.loss_and_grad(T.rankcosine_loss, _t(c["preds"]), _t(c["labels"]))
    G.assert_close(loss.numpy(), c["loss"], "torch loss"); G.assert_close(grad.numpy(), c["grad"], "torch grad")
    lq, g = CO.rankcosine(c["preds"], c["labels"])
    G.assert_close(lq.astype(np.float64).sum(), c["loss"], "C loss"); G.assert_close(g, c["grad"], "C grad")


@pytest.mark.parametrize("name", G.case_ids("stlistnet", "siblings"))
def test_stlistnet_oracles(name):
    c = G.siblings()["stlistnet"][name]
    loss, grad = T.loss_and_grad(T.stlistnet_loss, _t(c["preds"]), _t(c["labels"]), _t(c["unif"]), temperature=float(c["temperature"]))
    G.assert_close(loss.numpy(), c["loss"], "torch loss"); G.assert_close(grad.numpy(), c["grad"], "torch grad")
    lq, g = CO.stlistnet(c["preds"], c["labels"], c["unif"], float(c["temperature"]))
    G.assert_close(lq.astype(np.float64).sum(), c["loss"], "C loss"); G.assert_close(g, c["grad"], "C grad")


@pytest.mark.parametrize("name", G.case_ids("softrank", "siblings"))
def test_softrank_oracles(name):
    c = G.siblings()["softrank"][name]
    tk = int(c["top_k"]) or None
    loss, grad = T.loss_and_grad(T.softrank_loss, _t(c["preds"]), _t(c["labels"]), delta=float(c["delta"]), top_k=tk)
    G.assert_close(loss.numpy(), c["loss"], "torch loss"); G.assert_close(grad.numpy(), c["grad"], "torch grad")
    lq, g = CO.softrank(c["preds"], c["labels"], float(c["delta"]), tk)
    G.assert_close(lq.astype(np.float64).sum(), c["loss"], "C loss"); G.assert_close(g, c["grad"], "C grad")


@pytest.mark.parametrize("name", G.case_ids("lambdaloss1", "siblings"))
def test_lambdaloss1_oracles(name):
    """NDCG_Loss1 — golden at B = 1 (the only batch size the reference's broadcast accepts)."""
    c = G.siblings()["lambdaloss1"][name]
    loss, grad = T.loss_and_grad(T.lambdaloss_loss, _t(c["preds"]), _t(c["labels"]), k=int(c["k"]), sigma=float(c["sigma"]), loss_type=0)
    G.assert_close(loss.numpy(), c["loss"], "torch loss"); G.assert_close(grad.numpy(), c["grad"], "torch grad")
    lq, g = CO.lambdaloss(c["preds"], c["labels"], k=int(c["k"]), sigma=float(c["sigma"]), loss_type=0)
    G.assert_close(lq.astype(np.float64).sum(), c["loss"], "C loss"); G.assert_close(g, c["grad"], "C grad")


@pytest.mark.parametrize("name", G.case_ids("permndcg", "siblings"))
def test_ndcg_with_permutation_labels_oracles(name):
    """LABEL_TYPE.Permutation: nDCG's gain is the label itself (adhoc_metric.py:207-212,225-230)."""
    c = G.siblings()["permndcg"][name]
    tp, tl = _t(c["preds"]), _t(c["labels"])
    _, idx = torch.sort(tp, dim=1, descending=True)
    sys_sorted = torch.gather(tl, 1, idx)
    ideal = torch.sort(tl, dim=1, descending=True)[0]
    ks = [int(k) for k in c["ks"]]
    G.assert_close(T.ndcg_at_ks(sys_sorted, ideal, ks, permutation_labels=True).numpy(), c["ndcg"], "torch ndcg")
    out = CO.metrics_at_ks(c["preds"], c["labels"], ks, presort=False, max_label=1.0, permutation_labels=True)
    G.assert_close(out["ndcg"], c["ndcg"], "C ndcg")


@pytest.mark.parametrize("name", G.case_ids("mdprank", "siblings"))
def test_mdprank_oracles(name):
    """MDPRank (mdprank.py:46-75) on the ranking the reference sampled (captured in the fixture)."""
    c = G.siblings()["mdprank"][name]
    tk = int(c["top_k"]) or None
    perm = torch.from_numpy(c["perm"])
    loss, grad = T.loss_and_grad(lambda p, y: T.mdprank_loss(p, y, perm, top_k=tk, gamma=float(c["gamma"])), _t(c["preds"]), _t(c["labels"]))
    G.assert_close(loss.numpy(), c["loss"], "torch loss"); G.assert_close(grad.numpy(), c["grad"], "torch grad")
    lq, g = CO.mdprank(c["preds"], c["labels"], c["perm"], top_k=tk, gamma=float(c["gamma"]))
    G.assert_close(lq.astype(np.float64).sum(), c["loss"], "C loss"); G.assert_close(g, c["grad"], "C grad")


@pytest.mark.parametrize("name", sorted(G.STEP_CASES))
def test_train_step_restatement_reproduces_the_reference_itself(name):
    """r6: oracle/torch_ref.py's restatement of the WHOLE train step (scorer forward -> loss -> zero_grad / backward / Adam step) against three steps of
    the reference's own ranker objects (tests/golden/step.npz: NeuralRanker.init + train_op, ranker.py:512-525,589-603, run by make_golden_step.py)."""
    c = G.steps()[name]
    _, _, loss_fn, okw = G.STEP_CASES[name]
    X, Y = torch.from_numpy(c["X"]), torch.from_numpy(c["Y"])
    net = T.build_pointsf(X.shape[2], dropout=0.0)
    net.load_state_dict({k[len("sd0/"):]: torch.from_numpy(v) for k, v in c.items() if k.startswith("sd0/")})
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-3)
    for step in range(3):
        got = T.cpu_train_step(net, opt, X, Y, getattr(T, loss_fn), **okw)
        G.assert_close(got, c["losses"][step], f"loss of step {step}")
    for k, v in net.state_dict().items():
        if k == "ff_5.bias":
            continue        # every loss here is shift-invariant: this gradient is identically 0 and Adam turns rounding noise into +-lr moves
        assert np.allclose(v.numpy(), c[f"sd3/{k}"], rtol=1e-4, atol=2e-5), k
