"""GPU: the layer-wise fused stack (hand-written GEMMs + batch-norm / activation / dropout kernels) on the reference's DEFAULT
pointsf configuration — 5 layers, GELU, LTRBatchNorm 'BN' affine, Sigmoid tail (ptranking/ltr_adhoc/eval/parameter.py:145-146) — and
on the other activations of get_AF, against the same modules evaluated by torch on the CPU (fp32), incl. the kernels' own dropout
masks fed to the torch side."""
import copy
import ctypes as C

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
CPU_REFERENCE_MODULES = True      # tests/conftest.py: this module evaluates build_pointsf() module objects on the CPU as its reference (torch ops, not our kernels)

DEFAULT = dict(num_layers=5, AF='GE', TL_AF='S', apply_tl_af=True, BN=True, bn_type='BN', bn_affine=True)


def close(a, b, tol=3e-5, what=""):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{what}: max|diff|={err:.3e} > {tol * scale:.3e}"


def _mask(R, width, p, seed, site):
    from ptranking_amd import _lib
    ones = torch.ones(R, width, device="cuda")
    m = torch.empty_like(ones)
    _lib.call("ptr_dropout_apply", _lib.ptr(ones), width, R, width, C.c_float(p), C.c_uint64(seed), site, _lib.ptr(m), width,
              _lib.current_stream(ones.device))
    return (m > 0).float().cpu()


def _cpu_forward_with_masks(net_cpu, x, p, seed, R):
    """nn.Sequential.forward on the CPU with every nn.Dropout replaced by the kernel's keep mask of that site."""
    site = 0
    for m in net_cpu:
        if isinstance(m, nn.Dropout):
            if p > 0:
                x = x * _mask(R, x.shape[-1], p, seed, site).view(x.shape) / (1 - p)
            site += 1
        else:
            x = m(x)
    return x


@pytest.mark.parametrize("cfg", [DEFAULT,
                                 dict(num_layers=3, AF='GE', TL_AF='S', apply_tl_af=False, BN=False, bn_type=None, bn_affine=False),
                                 dict(num_layers=2, AF='T', TL_AF='T', apply_tl_af=True, BN=True, bn_type='BN', bn_affine=False),
                                 dict(num_layers=3, AF='SE', TL_AF='S', apply_tl_af=True, BN=False, bn_type=None, bn_affine=False),
                                 dict(num_layers=2, AF='LR', TL_AF='E', apply_tl_af=True, BN=True, bn_type='BN', bn_affine=True),
                                 dict(num_layers=4, AF='CE', TL_AF='S', apply_tl_af=False, BN=True, bn_type='BN', bn_affine=True),
                                 dict(num_layers=3, AF='S', TL_AF='R', apply_tl_af=True, BN=False, bn_type=None, bn_affine=False),
                                 dict(num_layers=3, AF='GE', TL_AF='S', apply_tl_af=True, BN=True, bn_type='BN2', bn_affine=True),
                                 dict(num_layers=2, AF='R', TL_AF='S', apply_tl_af=False, BN=True, bn_type='BN2', bn_affine=False)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_fused_stack_matches_torch_cpu_modules(cfg, p):
    from ptranking_amd.host import build_pointsf
    from ptranking_amd.linear import FusedStack
    from ptranking_amd.scorer import fused_kind
    assert fused_kind(num_features=136, dropout=p, **cfg) == 'stack'
    torch.manual_seed(3 + cfg["num_layers"])
    net = build_pointsf(num_features=136, dropout=p, **cfg)
    assert isinstance(net, FusedStack)
    with torch.no_grad():                                   # non-trivial affine parameters / biases
        for n_, prm in net.named_parameters():
            if "bn" in n_:
                prm.add_(0.3 * torch.randn_like(prm))
        for m_ in net.modules():                            # LTRBatchNorm2: non-trivial moving statistics
            if hasattr(m_, "moving_mean"):
                m_.moving_mean = 0.2 * torch.randn_like(m_.moving_mean)
                m_.moving_var = 1.0 + 0.3 * torch.rand_like(m_.moving_var)
    ref = copy.deepcopy(net)                                # CPU tensors -> FusedStack.forward = the plain torch modules
    net = net.cuda()
    net.train(); ref.train()
    B, L = 6, 129
    R = B * L
    x = torch.randn(B, L, 136)
    xg = x.cuda().requires_grad_(True)
    out = net(xg)
    assert net._plan and not net._plan["relu_only"]
    g = torch.randn_like(out)
    out.backward(g)
    xr = x.clone().requires_grad_(True)                      # [B, L, F]: LTRBatchNorm2 normalises over dim 1
    outr = _cpu_forward_with_masks(ref, xr, p, net.last_seed, R)
    outr.backward(g.cpu())
    close(out, outr, what="out")
    close(xg.grad, xr.grad, tol=1e-4, what="dx")
    got = dict(net.named_parameters())
    for n_, prm in ref.named_parameters():
        close(got[n_].grad, prm.grad, tol=1e-4, what=n_)
    assert list(net.state_dict()) == list(ref.state_dict())
    if cfg["bn_type"] == 'BN2':                             # the moving statistics were updated like the reference's (utils.py:242-245)
        for (na, ma), (nb, mb) in zip(net.named_modules(), ref.named_modules()):
            if hasattr(ma, "moving_mean"):
                if p == 0.0:                                # (with dropout the CPU side above bypassed nn.Sequential.forward's own masks)
                    close(ma.moving_mean.reshape(-1), mb.moving_mean.reshape(-1), tol=1e-4, what=f"{na}.moving_mean")
                    close(ma.moving_var.reshape(-1), mb.moving_var.reshape(-1), tol=1e-4, what=f"{na}.moving_var")
                mb.moving_mean, mb.moving_var = ma.moving_mean.cpu().clone(), ma.moving_var.cpu().clone()
    # LTRBatchNorm has no running statistics: evaluation uses the batch statistics too (utils.py:214); no dropout in eval.
    # LTRBatchNorm2 decides by torch.is_grad_enabled() (utils.py:229): under no_grad it normalises with the moving statistics
    net.eval(); ref.eval()
    with torch.no_grad():
        close(net(xg), ref(x), what="eval (no_grad)")
        assert torch.equal(net(xg), net(xg))
    close(net(xg), ref(x), what="eval (grad enabled: batch statistics)")


def test_default_pointsf_ranker_trains_on_the_fused_stack():
    import ptranking_amd as pa
    from ptranking_amd.linear import FusedStack
    sf = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3, "pointsf": dict(num_features=136, **DEFAULT)}
    r = pa.LambdaRank(sf_para_dict=sf, model_para_dict={"sigma": 1.0}, gpu=True, device="cuda:0")
    r.init()
    assert isinstance(r.point_sf, FusedStack) and pa.scorer.fusable(num_features=136, **DEFAULT)
    r.train_mode()
    X = torch.randn(16, 64, 136, device="cuda")
    Y = torch.sort(torch.randint(0, 5, (16, 64), device="cuda").float(), dim=1, descending=True)[0]
    Y[:, 0] = 2.0
    before = [p_.detach().clone() for p_ in r.point_sf.parameters()]
    losses = []
    for _ in range(3):
        loss, stop = r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
        losses.append(float(loss))
    assert all(torch.isfinite(torch.tensor(losses))) and not stop
    assert any(not torch.equal(a, b) for a, b in zip(before, r.point_sf.parameters()))
    m = r.adhoc_performance_at_ks(test_data=[(list(range(16)), X, Y)], ks=[1, 5, 10], label_type=pa.LABEL_TYPE.MultiLabel, presort=True)
    assert all(torch.isfinite(t).all() for t in m)


def test_flat_stack_optimizer_matches_torch_adam():
    """Default pointsf ranker: parameters re-homed in one flat buffer + FlatViewAdam + gradients written in place by the stack's
    backward == the same ranker with torch.optim.Adam over the separate parameter tensors (same seeds, 4 train steps)."""
    import ptranking_amd as pa
    from ptranking_amd.scorer import FlatViewAdam
    B, L, F = 16, 40, 136
    sf = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3, "pointsf": dict(num_features=F, dropout=0.1, **DEFAULT)}
    g = torch.Generator().manual_seed(5)
    X = torch.randn(B, L, F, generator=g).cuda()
    Y = torch.sort(torch.randint(0, 5, (B, L), generator=g).float(), dim=1, descending=True)[0].cuda()
    Y[:, 0] = 2.0
    states, losses = [], []
    for flat in (True, False):
        torch.manual_seed(11)
        r = pa.LambdaRank(sf_para_dict=sf, model_para_dict={"sigma": 1.0}, gpu=True, device="cuda:0")
        r.use_flat_stack_optimizer = flat
        r.init(); r.train_mode()
        assert isinstance(r.optimizer, FlatViewAdam) == flat
        ls = [float(r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)[0].detach()) for _ in range(4)]
        losses.append(ls)
        states.append({k: v.detach().cpu().clone() for k, v in r.point_sf.state_dict().items()})
        r.point_sf.eval()
        with torch.no_grad():
            states[-1]["__scores__"] = r.point_sf(X).cpu().clone()
        r.point_sf.train()
        if flat:                                            # a second backward before zero_grad() accumulates (no overwrite)
            r.optimizer.zero_grad()
            out = r.point_sf(X).sum(); out.backward()
            g1 = r.optimizer.flat_param.grad.clone()
            r.point_sf.eval(); r.point_sf.train()
            torch.manual_seed(1); r.optimizer.zero_grad(); o1 = r.point_sf(X).sum(); o1.backward()
            ga = r.optimizer.flat_param.grad.clone()
            torch.manual_seed(1); o2 = r.point_sf(X).sum(); o2.backward()
            close(r.optimizer.flat_param.grad, 2 * ga, 1e-5, "accumulated gradient")
            assert g1.abs().max() > 0
    assert states[0].keys() == states[1].keys()
    for a, b in zip(*losses):
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (losses)
    for k in states[0]:
        # the bias of a Linear in front of a batch norm has a mathematically zero gradient: Adam turns its rounding noise into
        # +-lr steps that differ between any two implementations and do not influence the scores
        if k.startswith("ff_") and k.endswith(".bias"):
            continue
        close(states[0][k], states[1][k], 5e-5, f"{k} after 4 steps")


@pytest.mark.parametrize("sf_id", ["pointsf", "listsf"])
def test_widths_that_are_not_multiples_of_4_train_with_dropout(sf_id):
    """46-feature data (MQ2007 / MQ2008) with dropout 0.1.  The fused input dropout works on float4 feature groups; these
    stacks run module by module (FusedLinear GEMMs take any K) instead of raising — GELU pointsf and the listsf head / tail stacks."""
    import ptranking_amd as pa
    if sf_id == "pointsf":
        sf = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3,
              "pointsf": dict(num_features=46, num_layers=3, AF='GE', TL_AF='S', apply_tl_af=False, BN=False, bn_type=None, bn_affine=False,
                              dropout=0.1)}
        r = pa.LambdaRank(sf_para_dict=sf, model_para_dict={"sigma": 1.0}, gpu=True, device="cuda:0")
    else:
        listsf = dict(num_features=46, ff_dims=[32, 64], AF='R', TL_AF='GE', apply_tl_af=False, BN=False, bn_type='BN2', bn_affine=False,
                      n_heads=2, encoder_layers=2, encoder_type='DASALC', dropout=0.1)
        sf = dict(sf_id='listsf', opt='Adagrad', lr=0.001, listsf=listsf)
        r = pa.LambdaLoss(sf_para_dict=copy.deepcopy(sf), model_para_dict=dict(pa.DEFAULT_PARAS["LambdaLoss"]), gpu=True, device="cuda:0")
    r.init()
    r.train_mode()
    X = torch.randn(6, 40, 46, device="cuda")
    Y = torch.sort(torch.randint(0, 5, (6, 40), device="cuda").float(), dim=1, descending=True)[0].contiguous()
    Y[:, 0] = 2.0
    before = [p_.detach().clone() for p_ in r.get_parameters()]
    for _ in range(2):
        loss, stop = r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
        assert torch.isfinite(loss) and not stop
    assert any(not torch.equal(a, b) for a, b in zip(before, r.get_parameters()))
    r.eval_mode()
    with torch.no_grad():
        a, b = r.predict(X), r.predict(X)
    assert torch.equal(a, b)


def test_flat_view_adam_rehomes_gradients_that_were_repointed():
    """module.zero_grad() / a foreign zero_grad(set_to_none=True) detach the stack's .grad views from the flat gradient
    buffer; FlatViewAdam.step() re-homes them instead of silently stepping on zeros."""
    import ptranking_amd as pa
    sf = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3,
          "pointsf": dict(num_features=136, num_layers=2, AF='GE', TL_AF='S', apply_tl_af=False, BN=False, bn_type=None, bn_affine=False,
                          dropout=0.0)}
    r = pa.LambdaRank(sf_para_dict=sf, model_para_dict={"sigma": 1.0}, gpu=True, device="cuda:0")
    r.init(); r.train_mode()
    opt = r.optimizer
    assert type(opt).__name__ == "FlatViewAdam"
    X = torch.randn(4, 16, 136, device="cuda")
    r.point_sf.zero_grad(set_to_none=True)               # every .grad is None now
    (r.forward(X) ** 2).sum().backward()                 # autograd allocates fresh .grad tensors outside the flat buffer
    want = torch.cat([p_.grad.reshape(-1) for p_ in r.point_sf.parameters()]).clone()
    before = opt.flat_param.detach().clone()
    opt.step()
    got = torch.cat([opt.flat_param.grad[o:o + n] for o, n in _slices(r.point_sf)])
    assert torch.equal(got, want) and not torch.equal(before, opt.flat_param)
    assert r.point_sf.reattach_grads() == 0


def _slices(stack):
    off = 0
    for p_ in stack.parameters():
        yield off, p_.numel()
        off += (p_.numel() + 3) // 4 * 4
