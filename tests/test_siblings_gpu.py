"""GPU: STListNet / RankCosine / RankMSE — HIP vs the reference's golden outputs and vs the oracle."""
import copy

import numpy as np
import pytest
import torch

import golden_util as G
from test_parity_gpu import dev, loss_and_grad, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def F():
    from ptranking_amd import functional
    return functional


@pytest.mark.parametrize("name", G.case_ids("rankmse", "siblings"))
def test_golden_rankmse(F, name):
    c = G.siblings()["rankmse"][name]
    loss, grad = loss_and_grad(F.rankmse_loss, c["preds"], dev(c["labels"]))
    G.assert_close(loss, c["loss"], "loss"); G.assert_close(grad, c["grad"], "grad")


@pytest.mark.parametrize("name", G.case_ids("rankcosine", "siblings"))
def test_golden_rankcosine(F, name):
    c = G.siblings()["rankcosine"][name]
    loss, grad = loss_and_grad(F.rankcosine_loss, c["preds"], dev(c["labels"]))
    G.assert_close(loss, c["loss"], "loss"); G.assert_close(grad, c["grad"], "grad")


@pytest.mark.parametrize("name", G.case_ids("stlistnet", "siblings"))
def test_golden_stlistnet(F, name):
    c = G.siblings()["stlistnet"][name]
    loss, grad = loss_and_grad(F.stlistnet_loss, c["preds"], dev(c["labels"]), temperature=float(c["temperature"]), unif=dev(c["unif"]))
    G.assert_close(loss, c["loss"], "loss"); G.assert_close(grad, c["grad"], "grad")


@pytest.mark.parametrize("name", G.case_ids("softrank", "siblings"))
def test_golden_softrank(F, name):
    c = G.siblings()["softrank"][name]
    loss, grad = loss_and_grad(F.softrank_loss, c["preds"], dev(c["labels"]), delta=float(c["delta"]), top_k=int(c["top_k"]) or None)
    G.assert_close(loss, c["loss"], "loss"); G.assert_close(grad, c["grad"], "grad")


@pytest.mark.parametrize("name", G.case_ids("lambdaloss1", "siblings"))
def test_golden_lambdaloss1(F, name):
    c = G.siblings()["lambdaloss1"][name]
    loss, grad = loss_and_grad(F.lambdaloss_loss, c["preds"], dev(c["labels"]), k=int(c["k"]), sigma=float(c["sigma"]),
                               loss_type="NDCG_Loss1")
    G.assert_close(loss, c["loss"], "loss"); G.assert_close(grad, c["grad"], "grad")


@pytest.mark.parametrize("B,L", [(7, 5), (33, 128), (9, 700), (3, 2048)])
@pytest.mark.parametrize("use_lens", [False, True])
def test_oracle_softrank_lambdaloss1(F, B, L, use_lens):
    from oracle import c_oracle as CO
    preds, labels, ln = synth(7000 + L, B, L, lens=use_lens)
    lens_t = None if ln is None else dev(ln)
    for delta, tk in ((2.0, None), (0.3, 10)):
        loss, grad = loss_and_grad(F.softrank_loss, preds, dev(labels), delta=delta, top_k=tk, lens=lens_t)
        lq, g = CO.softrank(preds, labels, delta, tk, lens=ln)
        G.assert_close(loss, lq.astype(np.float64).sum(), "softrank loss"); G.assert_close(grad, g, "softrank grad")
    for k in (5, 64, L):
        loss, grad = loss_and_grad(F.lambdaloss_loss, preds, dev(labels), k=k, sigma=1.3, loss_type="NDCG_Loss1", lens=lens_t)
        lq, g = CO.lambdaloss(preds, labels, k=k, sigma=1.3, loss_type=0, lens=ln)
        G.assert_close(loss, lq.astype(np.float64).sum(), "loss1 loss"); G.assert_close(grad, g, "loss1 grad")


@pytest.mark.parametrize("B,L", [(7, 5), (33, 128), (9, 700), (2, 4096)])
@pytest.mark.parametrize("use_lens", [False, True])
def test_oracle_siblings(F, B, L, use_lens):
    from oracle import c_oracle as CO
    preds, labels, ln = synth(5000 + L, B, L, lens=use_lens)
    lens_t = None if ln is None else dev(ln)
    unif = np.random.default_rng(L).random((B, L)).astype(np.float32)
    loss, grad = loss_and_grad(F.rankmse_loss, preds, dev(labels), lens=lens_t)
    lq, g = CO.rankmse(preds, labels, lens=ln)
    G.assert_close(loss, lq.astype(np.float64).mean(), "rankmse loss"); G.assert_close(grad, g, "rankmse grad")
    loss, grad = loss_and_grad(F.rankcosine_loss, preds, dev(labels), lens=lens_t)
    lq, g = CO.rankcosine(preds, labels, lens=ln)
    G.assert_close(loss, lq.astype(np.float64).sum(), "rankcosine loss"); G.assert_close(grad, g, "rankcosine grad")
    loss, grad = loss_and_grad(F.stlistnet_loss, preds, dev(labels), temperature=2.0, unif=dev(unif), lens=lens_t)
    lq, g = CO.stlistnet(preds, labels, unif, 2.0, lens=ln)
    G.assert_close(loss, lq.astype(np.float64).sum(), "stlistnet loss"); G.assert_close(grad, g, "stlistnet grad")


def test_sibling_rankers_train():
    import ptranking_amd as pa
    sf = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3,
          "pointsf": dict(num_features=24, num_layers=3, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None, bn_affine=False)}
    X = torch.randn(6, 40, 24, device="cuda")
    Y = torch.sort(torch.randint(0, 5, (6, 40), device="cuda").float(), dim=1, descending=True)[0].contiguous()
    for name in ("STListNet", "RankCosine", "RankMSE", "SoftRank"):
        cls = getattr(pa, name)
        r = cls(sf_para_dict=copy.deepcopy(sf), model_para_dict=dict(pa.DEFAULT_PARAS[name]), gpu=True, device="cuda:0") \
            if name in ("STListNet", "SoftRank") else cls(sf_para_dict=copy.deepcopy(sf), gpu=True, device="cuda:0")
        r.init(); r.train_mode()
        before = r.point_sf.flat.detach().clone()
        loss, stop = r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
        assert torch.isfinite(loss) and not stop and not torch.equal(before, r.point_sf.flat), name


@pytest.mark.parametrize("name", G.case_ids("permndcg", "siblings"))
def test_golden_ndcg_permutation_labels(F, name):
    """nDCG with LABEL_TYPE.Permutation (gain = label) through the metric kernel and the Evaluator surface."""
    import ptranking_amd as pa
    c = G.siblings()["permndcg"][name]
    ks = [int(k) for k in c["ks"]]
    out = F.metrics_at_ks(dev(c["preds"]), dev(c["labels"]), ks, presort=False, which=("ndcg", "ap", "p"), permutation_labels=True)
    G.assert_close(out["ndcg"].cpu().numpy(), c["ndcg"], "ndcg")
    with pytest.raises(NotImplementedError):
        F.metrics_at_ks(dev(c["preds"]), dev(c["labels"]), ks, presort=False, permutation_labels=True)     # nERR undefined

    class Fixed(pa.host.DeviceEvaluator):
        device = "cuda:0"
        def eval_mode(self): pass
        def predict(self, X): return X[:, :, 0]
    ev = Fixed()
    X = dev(c["preds"]).unsqueeze(2).contiguous()
    got = ev.ndcg_at_ks(test_data=[(list(range(X.size(0))), X, dev(c["labels"]))], ks=ks, label_type=pa.LABEL_TYPE.Permutation, presort=False)
    G.assert_close(got.numpy(), c["ndcg"].mean(axis=0), "Evaluator.ndcg_at_ks")


@pytest.mark.parametrize("name", G.case_ids("mdprank", "siblings"))
def test_golden_mdprank(F, name):
    c = G.siblings()["mdprank"][name]
    perm = torch.from_numpy(c["perm"]).cuda()
    loss, grad = loss_and_grad(lambda p, y: F.mdprank_loss(p, y, perm, top_k=int(c["top_k"]) or None, gamma=float(c["gamma"])),
                               c["preds"], dev(c["labels"]))
    G.assert_close(loss, c["loss"], "loss"); G.assert_close(grad, c["grad"], "grad")


@pytest.mark.parametrize("B,L", [(9, 7), (33, 128), (5, 700), (2, 4096)])
@pytest.mark.parametrize("use_lens", [False, True])
def test_oracle_mdprank(F, B, L, use_lens):
    from oracle import c_oracle as CO
    preds, labels, ln = synth(9000 + L, B, L, lens=use_lens)
    rng = np.random.default_rng(L)
    perm = np.stack([np.concatenate([rng.permutation(L if ln is None else int(ln[b])), np.arange(L if ln is None else int(ln[b]), L)])
                     for b in range(B)]).astype(np.int64)
    lens_t = None if ln is None else dev(ln)
    for top_k, gamma in ((10, 1.0), (None, 0.95), (3, 0.5)):
        loss, grad = loss_and_grad(lambda p, y: F.mdprank_loss(p, y, torch.from_numpy(perm).cuda(), top_k=top_k, gamma=gamma, lens=lens_t),
                                   preds, dev(labels))
        lq, g = CO.mdprank(preds, labels, perm, top_k=top_k, gamma=gamma, lens=ln)
        G.assert_close(loss, lq.astype(np.float64).sum(), "loss"); G.assert_close(grad, g, "grad")


@pytest.mark.parametrize("dist", ["PL", "STPL"])
def test_mdprank_ranker_trains(dist):
    import ptranking_amd as pa
    sf = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3,
          "pointsf": dict(num_features=24, num_layers=3, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None, bn_affine=False)}
    X = torch.randn(6, 40, 24, device="cuda")
    Y = torch.sort(torch.randint(0, 5, (6, 40), device="cuda").float(), dim=1, descending=True)[0].contiguous()
    paras = dict(pa.DEFAULT_PARAS["MDPRank"], distribution=dist, temperature=1.0 if dist == "PL" else 2.0)
    r = pa.MDPRank(sf_para_dict=copy.deepcopy(sf), model_para_dict=paras, gpu=True, device="cuda:0")
    r.init(); r.train_mode()
    before = r.point_sf.flat.detach().clone()
    lens = torch.tensor([40, 40, 17, 40, 3, 40], dtype=torch.int32, device="cuda")
    for kw in ({}, {"lens": lens}):
        loss, stop = r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel, **kw)
        assert torch.isfinite(loss) and not stop
    assert not torch.equal(before, r.point_sf.flat)
