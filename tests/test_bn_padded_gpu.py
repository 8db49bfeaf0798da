"""GPU: batch-norm scorers on PADDED query batches (missing item 1 / SURVEY.md 8 f-1) — the reference's DEFAULT scoring
function is BN=True (ptranking/ltr_adhoc/eval/parameter.py:145-146).  `FusedStack.batch_lens` keeps padded rows out of the 'BN' / 'BN2'
statistics and their backward (csrc/bnact.hip), so a padded batch reproduces what the REFERENCE's get_stacked_FFNet computes on the
unpadded lists: tests/golden/ffnet_padded.npz, made by tests/golden/make_golden_ffnet_padded.py from ptranking/base/utils.py:201-356."""
import copy
import os

import numpy as np
import pytest
import torch

from test_ffnet_cpu import close

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ffnet_padded.npz"))
CASES = {
    "bn_default": dict(num_layers=5, AF='GE', TL_AF='S', apply_tl_af=True, BN=True, bn_type='BN', bn_affine=True),
    "bn_tanh":    dict(num_layers=2, AF='T', TL_AF='T', apply_tl_af=False, BN=True, bn_type='BN', bn_affine=False),
    "bn2_gelu":   dict(num_layers=3, AF='GE', TL_AF='S', apply_tl_af=True, BN=True, bn_type='BN2', bn_affine=True),
    "bn2_relu":   dict(num_layers=2, AF='R', TL_AF='S', apply_tl_af=False, BN=True, bn_type='BN2', bn_affine=False),
}


def _load(tag):
    from ptranking_amd.host import build_pointsf
    F = int(G[f"{tag}/cfg"][0])
    net = build_pointsf(num_features=F, dropout=0.0, **CASES[tag])
    sd = {k[len(tag) + 4:]: torch.from_numpy(G[k]) for k in G.files if k.startswith(f"{tag}/sd/")}
    assert set(sd) == set(net.state_dict())
    net.load_state_dict(sd)
    for _, m in net.named_modules():
        if hasattr(m, "moving_mean"):
            m.moving_mean = torch.zeros_like(m.moving_mean)
            m.moving_var = torch.ones_like(m.moving_var)
    return net.cuda().train()


@pytest.mark.parametrize("tag", sorted(CASES))
def test_padded_batch_reproduces_the_reference_on_the_unpadded_lists(tag):
    net = _load(tag)
    x = torch.from_numpy(G[f"{tag}/x"]).cuda().requires_grad_(True)
    R = torch.from_numpy(G[f"{tag}/R"]).cuda()
    lens_np = G[f"{tag}/lens"]
    net.batch_lens = torch.from_numpy(lens_np).cuda()
    y = net(x)
    (y.reshape(R.shape) * R).sum().backward()
    real = (np.arange(x.shape[1])[None, :] < lens_np[:, None])
    tol = 2e-5
    close((y.detach().cpu().numpy() * real[..., None]), G[f"{tag}/y"], f"{tag} output", rtol=tol)
    close(x.grad.cpu().numpy(), G[f"{tag}/dx"], f"{tag} dX (0 at padded rows)", rtol=tol)
    gscale = max(float(np.abs(G[f]).max()) for f in G.files if f.startswith(f"{tag}/grad/"))
    for k, p in net.named_parameters():
        close(p.grad.cpu().numpy(), G[f"{tag}/grad/{k}"], f"{tag} grad {k}", rtol=tol, floor=1e-1 * gscale)
    if CASES[tag]["bn_type"] == 'BN2':
        for name, m in net.named_modules():
            if hasattr(m, "moving_mean"):
                close(m.moving_mean.cpu().numpy(), G[f"{tag}/moving_after/{name}/mean"], f"{tag} moving mean {name}", rtol=tol)
                close(m.moving_var.cpu().numpy(), G[f"{tag}/moving_after/{name}/var"], f"{tag} moving var {name}", rtol=tol)
    # the junk in the padding really was out of reach: other junk, same result on the real rows
    x2 = x.detach().clone()
    for b, n in enumerate(lens_np):
        x2[b, n:] = -1e3
    y2 = net(x2)
    assert torch.equal((y2.detach().cpu() * torch.from_numpy(real[..., None])), (y.detach().cpu() * torch.from_numpy(real[..., None])))
    # without lens the padded rows DO enter the statistics (what the reference would compute on a zero-padded tensor)
    net.batch_lens = None
    y3 = net(x.detach())
    assert not torch.allclose(y3.detach().cpu() * torch.from_numpy(real[..., None]), torch.from_numpy(G[f"{tag}/y"]), atol=1e-3)


def _data(lens, L, F, seed):
    rng = np.random.default_rng(seed)
    B = len(lens)
    X = rng.standard_normal((B, L, F)).astype(np.float32)
    Y = rng.choice(5, size=(B, L), p=[0.5147, 0.3250, 0.1339, 0.0183, 0.0081]).astype(np.float32)
    for b, n in enumerate(lens):
        Y[b, :n] = -np.sort(-Y[b, :n])
        Y[b, 0] = max(Y[b, 0], 1.0)
        X[b, n:] = 0.0
        Y[b, n:] = 0.0
    return torch.from_numpy(X), torch.from_numpy(Y)


@pytest.mark.parametrize("bn_type", ["BN", "BN2"])
def test_default_pointsf_trains_and_evaluates_on_padded_batches(bn_type):
    """The default scorer (5 x [Linear -> BN(affine) -> GELU] -> Linear -> BN -> Sigmoid) through the plugin surface with `lens`:
    train_op no longer refuses; with per-query statistics ('BN2', batch composition is irrelevant) the padded step's gradient equals the
    sum of the gradients of the per-length batches, and padded evaluation equals the per-length evaluation; with 'BN' a padded batch
    equals the same queries with every other padding width."""
    import ptranking_amd as pa
    from ptranking_amd.batching import PaddedQueryBatches  # noqa: F401  (the loader that produces such batches)
    lens = [40, 17, 40, 5, 17, 40, 5, 29]
    F, L = 136, 40
    sf = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3,
          "pointsf": dict(num_features=F, num_layers=5, AF='GE', TL_AF='S', apply_tl_af=True, BN=True, bn_type=bn_type, bn_affine=True, dropout=0.0)}

    def make():
        torch.manual_seed(3)
        r = pa.LambdaRank(sf_para_dict=copy.deepcopy(sf), model_para_dict={"sigma": 1.0}, gpu=True, device="cuda:0")
        r.init(); r.train_mode()
        return r

    X, Y = _data(lens, L, F, 11)
    lens_t = torch.tensor(lens, dtype=torch.int32, device="cuda")
    kw = dict(epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
    a = make()
    loss_a, _ = a.train_op(X.cuda(), Y.cuda(), lens=lens_t, **kw)
    ga = a.optimizer.flat_param.grad.detach().clone()
    assert torch.isfinite(loss_a) and float(ga.abs().max()) > 0
    # wider padding, same queries: identical statistics
    b = make()
    Xw = torch.zeros(len(lens), L + 13, F); Xw[:, :L] = X
    Yw = torch.zeros(len(lens), L + 13); Yw[:, :L] = Y
    loss_b, _ = b.train_op(Xw.cuda(), Yw.cuda(), lens=lens_t, **kw)
    gb = b.optimizer.flat_param.grad.detach()
    scale = max(1.0, float(ga.abs().max()))
    assert abs(float(loss_a) - float(loss_b)) <= 1e-5 * max(1.0, abs(float(loss_a)))
    assert float((ga - gb).abs().max()) <= 2e-5 * scale
    if bn_type == "BN2":
        # per-length batches (what the reference's sampler would feed), gradients accumulated without stepping
        c = make()
        c.optimizer.zero_grad()
        tot = 0.0
        gsum = torch.zeros_like(ga)
        for n in sorted(set(lens)):
            idx = [i for i, v in enumerate(lens) if v == n]
            c.optimizer.zero_grad()
            preds = c.forward(X[idx, :n].contiguous().cuda())
            l = pa.functional.lambdarank_loss(preds, Y[idx, :n].contiguous().cuda(), sigma=1.0)
            l.backward()
            gsum += c.optimizer.flat_param.grad
            tot += float(l)
        assert abs(tot - float(loss_a)) <= 1e-5 * max(1.0, abs(tot))
        assert float((gsum - ga).abs().max()) <= 2e-5 * scale
    # evaluation on the padded loader vs query by query
    e = make()
    padded = [(list(range(len(lens))), X, Y, torch.tensor(lens, dtype=torch.int32))]
    ks = [1, 3, 5, 10]
    got = e.ndcg_at_ks(test_data=padded, ks=ks, label_type=pa.LABEL_TYPE.MultiLabel, presort=True)
    assert got.shape == (len(ks),) and torch.isfinite(got).all()
    if bn_type == "BN2":
        single = [([i], X[i:i + 1, :n].contiguous(), Y[i:i + 1, :n].contiguous()) for i, n in enumerate(lens)]
        ref = e.ndcg_at_ks(test_data=single, ks=ks, label_type=pa.LABEL_TYPE.MultiLabel, presort=True)
        assert torch.allclose(got, ref, atol=2e-6), (got, ref)


def test_batch_norm_on_the_unfused_path_still_refuses_padding():
    import ptranking_amd as pa
    from ptranking_amd import host
    r = type("R", (), {})()
    r.point_sf = torch.nn.Sequential(torch.nn.Linear(4, 4), host._BatchNormOverDocs(4)).cuda()
    with pytest.raises(NotImplementedError, match="batch normalisation"):
        with host.scorer_lens(r, torch.tensor([3, 2], dtype=torch.int32, device="cuda"), torch.zeros(2, 3, 4, device="cuda")):
            pass
