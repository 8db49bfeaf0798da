"""listsf on CPU: the oracle's restatement of the reference's LayerNorm / MultiheadAttention / ListNeuralRanker.forward against
fixtures produced by the reference itself (tests/golden/make_golden_listsf.py), and the host-side module mirror (parameter
names, state_dict interchange, no CPU fallback)."""
import numpy as np
import pytest
import torch

import golden_util as G
from oracle import torch_ref as T


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize("name", G.case_ids("layernorm", "listsf"))
def test_oracle_layernorm(name):
    c = G.listsf()["layernorm"][name]
    sd = {k: _t(v).requires_grad_(True) for k, v in G.sub(c, "sd").items()}
    x = _t(c["x"]).requires_grad_(True)
    y = T.layer_norm_ref(x, sd["a_2"], sd["b_2"])
    (y * _t(c["R"])).sum().backward()
    G.assert_close(y.detach().numpy(), c["y"], "y"); G.assert_close(x.grad.numpy(), c["dx"], "dx")
    for k, v in G.sub(c, "grad").items():
        G.assert_close(sd[k].grad.numpy(), v, k)


@pytest.mark.parametrize("name", G.case_ids("mhsa", "listsf"))
def test_oracle_mhsa(name):
    c = G.listsf()["mhsa"][name]
    sd = {k: _t(v).requires_grad_(True) for k, v in G.sub(c, "sd").items()}
    x = _t(c["x"]).requires_grad_(True)
    y = T.mhsa_ref(x, sd, int(c["n_heads"]))
    (y * _t(c["R"])).sum().backward()
    G.assert_close(y.detach().numpy(), c["y"], "y"); G.assert_close(x.grad.numpy(), c["dx"], "dx")
    for k, v in G.sub(c, "grad").items():
        G.assert_close(sd[k].grad.numpy(), v, k)


@pytest.mark.parametrize("enc", ["DASALC", "AllRank", "AttnDIN"])
def test_oracle_listsf_forward(enc):
    c = G.listsf()["listsf"][enc]
    sd = {k: _t(v).requires_grad_(True) for k, v in G.sub(c, "sd").items()}
    preds = T.listsf_ref(_t(c["x"]), sd, enc, n_heads=2, encoder_layers=2, n_ff=2)
    (preds * _t(c["R"])).sum().backward()
    G.assert_close(preds.detach().numpy(), c["preds"], "preds")
    for k, v in G.sub(c, "grad").items():
        got = sd[k].grad.numpy() if sd[k].grad is not None else np.zeros_like(v)
        G.assert_close(got, v, k)


@pytest.mark.parametrize("enc", ["DASALC", "AllRank", "AttnDIN"])
def test_module_mirror_loads_reference_state_dict(enc):
    from ptranking_amd import listsf as LS
    c = G.listsf()["listsf"][enc]
    mods = LS.build_listsf(num_features=24, ff_dims=[16, 32], AF='R', TL_AF='GE', apply_tl_af=False, BN=False, bn_type='BN2',
                           bn_affine=False, n_heads=2, encoder_layers=2, encoder_type=enc)
    for part, m in mods.items():
        ref_sd = {k: _t(v) for k, v in G.sub(G.sub(c, "sd"), part).items()}
        assert set(m.state_dict().keys()) == set(ref_sd.keys()), part
        m.load_state_dict(ref_sd)                                   # shapes match too
    assert [l.mhsa.site for l in mods["encoder"].layers] == [0, 1]


def test_standalone_ranker_listsf_parameters_and_no_cpu_fallback():
    import ptranking_amd as pa
    listsf = dict(num_features=24, ff_dims=[16, 32], AF='R', TL_AF='GE', apply_tl_af=False, BN=False, bn_type='BN2', bn_affine=False,
                  n_heads=2, encoder_layers=3, encoder_type='DASALC')
    sf = dict(sf_id='listsf', opt='Adagrad', lr=0.001, listsf=listsf)
    r = pa.LambdaLoss(sf_para_dict=sf, model_para_dict=dict(pa.DEFAULT_PARAS["LambdaLoss"]), gpu=False, device="cpu")
    r.init()
    n_par = sum(p.numel() for p in r.get_parameters())
    enc = 3 * (4 * (24 * 24 + 24) + 2 * 24)
    ff = lambda dims: sum(a * b + b for a, b in zip(dims[:-1], dims[1:]))  # noqa: E731
    assert n_par == ff([24, 16, 32, 24]) + enc + ff([24, 16, 32, 1])
    assert isinstance(r.optimizer, torch.optim.Adagrad)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r.forward(torch.randn(2, 5, 24))


@pytest.mark.parametrize("rows", [(3, 5), (9000,), (20, 500)])
def test_splitk_linear_matches_nn_linear(rows):
    """Same forward and gradients as nn.Linear (also through the chunked dW path: >= 8 chunks of 1024 rows + a tail)."""
    from ptranking_amd.host import SplitKLinear
    torch.manual_seed(1)
    a = torch.nn.Linear(24, 17).double()
    b = SplitKLinear(24, 17).double()
    b.load_state_dict(a.state_dict())
    x = torch.randn(*rows, 24, dtype=torch.float64)
    R = torch.randn(*rows, 17, dtype=torch.float64)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    (a(xa) * R).sum().backward(); (b(xb) * R).sum().backward()
    assert torch.allclose(a(xa), b(xb), atol=1e-12)
    assert torch.allclose(xa.grad, xb.grad, atol=1e-12)
    assert torch.allclose(a.weight.grad, b.weight.grad, atol=1e-10) and torch.allclose(a.bias.grad, b.bias.grad, atol=1e-10)
