"""GPU: the fused fp32-MFMA pointsf scorer (forward / backward / FlatAdam) against plain PyTorch fp32 modules ON THE CPU, built the
way the reference builds them (ptranking/base/utils.py:288-356; pinned to the reference's own outputs by tests/test_ffnet_cpu.py),
with identical weights and — in training mode — the identical dropout masks exported from the kernel's counter-based generator."""
import pytest
import torch

from ptranking_amd import scorer as _scorer

pytestmark = pytest.mark.gpu
CPU_REFERENCE_MODULES = True      # tests/conftest.py: this module evaluates build_pointsf() module objects on the CPU as its reference (torch ops, not our kernels)


@pytest.fixture(autouse=True, params=["0", "2"], ids=["fp32mfma", "bf16x6"])
def _forward_kernel(request, monkeypatch):
    """Every test of this file runs with the fp32-MFMA forward pinned and with the bf16x6 forward forced (ptranking_amd/scorer.py x6_mode;
    shapes the bf16x6 kernels do not serve — F % 4 != 0, one hidden layer — take the fp32-MFMA kernels in both runs)."""
    monkeypatch.setenv("PTR_MLP_X6", request.param)


_REAL_RANDINT = torch.randint


def torch_scorer(F, NL):
    from ptranking_amd.host import build_pointsf
    return build_pointsf(num_features=F, num_layers=NL, AF="R", BN=False, apply_tl_af=False, dropout=0.0)     # CPU: plain torch ops


def make_pair(F, NL, dropout=0.1, seed=0):
    from ptranking_amd.scorer import FusedPointScorer
    torch.manual_seed(seed)
    fused = FusedPointScorer(F, num_layers=NL, dropout=dropout).cuda()
    ref = torch_scorer(F, NL)
    ref.load_state_dict({k: v.cpu() for k, v in fused.state_dict().items()})   # reference-named keys: ff_2.weight ... ff_{NL+2}.bias
    return fused, ref


def close(a, b, tol=2e-5):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"max|diff|={err:.3e} > {tol * scale:.3e}"


@pytest.mark.parametrize("F,NL", [(136, 3), (24, 3), (46, 2), (8, 1), (24, 4), (180, 2), (700, 3), (256, 3)])
@pytest.mark.parametrize("shape", [(5, 7), (64, 128), (3, 341)])
def test_eval_forward_matches_torch(F, NL, shape):
    fused, ref = make_pair(F, NL)
    fused.eval(); ref.eval()
    X = torch.randn(*shape, F, device="cuda")
    with torch.no_grad():
        out = fused(X)
        exp = ref(X.cpu())
    assert out.shape == exp.shape == (*shape, 1)
    close(out, exp)


def _train_reference(ref, fused, X2d, seed, p, NL):
    """Re-run the forward in torch with the kernel's own dropout masks."""
    R = X2d.shape[0]
    lin = [m for m in ref if isinstance(m, torch.nn.Linear)]
    a = X2d.cpu() * fused.dropout_mask(R, 0, seed).cpu() / (1 - p)
    for l in range(NL):
        h = torch.relu(lin[l](a))
        a = h * fused.dropout_mask(R, l + 1, seed).cpu() / (1 - p) if l < NL - 1 else h
    return lin[NL](a)


# F = 132 / 140: the other widths the single-pass fused backward serves; F = 144 fills all nine 16-feature tiles, leaves no column for the
# ones column that carries db_0 and therefore takes the layer-wise kernels — its ff_2.bias gradient is checked like the rest
@pytest.mark.parametrize("F,NL,R", [(136, 3, 2048 + 37), (132, 3, 1500), (140, 3, 777), (144, 3, 1111), (128, 3, 900), (24, 2, 100), (46, 3, 515), (136, 1, 64), (40, 4, 1000), (180, 2, 300), (700, 3, 1111), (256, 3, 640),
                                    (400, 2, 333), (200, 4, 500), (700, 3, 8 * 32 * 3 + 5)])
def test_train_forward_backward_match_torch_with_same_masks(F, NL, R, monkeypatch):
    p = 0.1
    fused, ref = make_pair(F, NL, dropout=p)
    fused.train()
    X = torch.randn(R, F, device="cuda")
    seed = 123456789 + R
    monkeypatch.setattr(torch, "randint", lambda *a, **k: torch.tensor([seed]))
    out = fused(X)
    monkeypatch.setattr(torch, "randint", _REAL_RANDINT)
    exp = _train_reference(ref, fused, X, seed, p, NL)
    close(out, exp)
    w = torch.randn(R, 1, device="cuda")
    (out * w).sum().backward()
    (exp * w.cpu()).sum().backward()
    got = fused.views(grad=True)
    for name, prm in ref.named_parameters():
        close(got[name], prm.grad, tol=5e-5)


@pytest.mark.parametrize("F,R", [(136, 2048 + 37), (8, 333), (132, 900), (24, 100), (4, 77)])
def test_layer1_k_tail_form_matches_torch(F, R, monkeypatch):
    """PTR_FWD_TQ=1: the first layer's contraction tail (F mod 16 = 4 or 8) in 1 / 2 MFMAs per output tile instead of a zero-padded
    super-step (opt-in: it costs registers in the 16-wave form).  Training and eval forward against the CPU modules."""
    monkeypatch.setenv("PTR_FWD_TQ", "1")
    p = 0.1
    fused, ref = make_pair(F, 3, dropout=p)
    fused.train()
    X = torch.randn(R, F, device="cuda")
    seed = 55 + R
    monkeypatch.setattr(torch, "randint", lambda *a, **k: torch.tensor([seed]))
    out = fused(X)
    monkeypatch.setattr(torch, "randint", torch.randint) if False else None
    exp = _train_reference(ref, fused, X, seed, p, 3)
    close(out, exp)
    fused.eval(); ref.eval()
    with torch.no_grad():
        close(fused(X), ref(X.cpu()))


@pytest.mark.parametrize("F,NL,R", [(700, 3, 1111), (180, 2, 300), (200, 4, 257)])
@pytest.mark.parametrize("stream", ["0", "1"])
def test_large_first_layer_slab_and_streaming_forms_agree_with_torch(F, NL, R, stream, monkeypatch):
    """F above ~157: W1 does not fit in LDS next to the hidden weights.  Default: the workgroup stages 48-column slabs of it (waves in
    lockstep, static tile groups); four hidden layers leave no room for the slabs and every wave streams its fragments from L2 —
    PTR_FWD_W1_STREAM=1 selects that form everywhere.  Both against the CPU modules, training (same masks) and eval."""
    monkeypatch.setenv("PTR_FWD_W1_STREAM", stream)
    p = 0.1
    fused, ref = make_pair(F, NL, dropout=p)
    fused.train()
    X = torch.randn(R, F, device="cuda")
    seed = 99 + R
    monkeypatch.setattr(torch, "randint", lambda *a, **k: torch.tensor([seed]))
    out = fused(X)
    exp = _train_reference(ref, fused, X, seed, p, NL)
    close(out, exp)
    fused.eval(); ref.eval()
    with torch.no_grad():
        close(fused(X), ref(X.cpu()))


def test_gradients_without_dropout_and_in_eval_mode():
    fused, ref = make_pair(136, 3, dropout=0.0)
    fused.train(); ref.train()
    X = torch.randn(777, 136, device="cuda")
    w = torch.randn(777, 1, device="cuda")
    (fused(X) * w).sum().backward()
    (ref(X.cpu()) * w.cpu()).sum().backward()
    got = fused.views(grad=True)
    for name, prm in ref.named_parameters():
        close(got[name], prm.grad, tol=5e-5)


def test_dropout_generator_statistics():
    from ptranking_amd.scorer import FusedPointScorer
    f = FusedPointScorer(136, 3, dropout=0.1).cuda()
    m0 = f.dropout_mask(4096, 0, 42)
    m1 = f.dropout_mask(4096, 1, 42)
    m0b = f.dropout_mask(4096, 0, 43)
    assert abs(m0.mean().item() - 0.9) < 2e-3 and abs(m1.mean().item() - 0.9) < 2e-3
    assert abs((m0[:, :100] * m1).mean().item() - 0.81) < 3e-3          # sites are independent
    assert abs((m0 * m0b).mean().item() - 0.81) < 3e-3                  # seeds are independent
    assert (m0.mean(dim=0) - 0.9).abs().max() < 0.03 and (m0.mean(dim=1) - 0.9).abs().max() < 0.12
    assert torch.equal(m0, f.dropout_mask(4096, 0, 42))                 # counter-based: reproducible
    X = torch.randn(64, 136, device="cuda")
    f.eval()
    with torch.no_grad():
        a, b = f(X), f(X)
    assert torch.equal(a, b)                                            # eval: no dropout
    f.train()
    with torch.no_grad():
        c = f(X)
    assert not torch.equal(a, c)                                        # train: dropout active


def test_flat_adam_matches_torch_adam():
    from ptranking_amd.scorer import FlatAdam
    torch.manual_seed(3)
    p1 = torch.nn.Parameter(torch.randn(34001, device="cuda"))
    p2 = torch.nn.Parameter(p1.detach().clone())
    o1 = FlatAdam([p1], lr=1e-3, weight_decay=1e-3)
    o2 = torch.optim.Adam([p2], lr=1e-3, weight_decay=1e-3)
    s1 = torch.optim.lr_scheduler.StepLR(o1, step_size=2, gamma=0.5)
    s2 = torch.optim.lr_scheduler.StepLR(o2, step_size=2, gamma=0.5)
    for it in range(6):
        g = torch.randn(34001, device="cuda") * (10.0 ** (it - 3))
        p1.grad, p2.grad = g.clone(), g.clone()
        o1.step(); o2.step(); s1.step(); s2.step()
        assert torch.allclose(p1, p2, rtol=1e-5, atol=1e-7), it
    assert o1.param_groups[0]["lr"] == o2.param_groups[0]["lr"]


def test_state_dict_is_interchangeable_with_the_reference_layout(tmp_path):
    fused, ref = make_pair(136, 3)
    sd = fused.state_dict()
    assert list(sd) == ["ff_2.weight", "ff_2.bias", "ff_3.weight", "ff_3.bias", "ff_4.weight", "ff_4.bias", "ff_5.weight", "ff_5.bias"]
    assert sd["ff_2.weight"].shape == (100, 136) and sd["ff_5.weight"].shape == (1, 100)
    torch.save(ref.state_dict(), tmp_path / "net.pkl")          # a checkpoint written by the torch / reference module
    from ptranking_amd.scorer import FusedPointScorer
    other = FusedPointScorer(136, 3).cuda()
    other.load_state_dict(torch.load(tmp_path / "net.pkl", map_location="cuda:0"))
    assert torch.equal(other.flat, fused.flat)
    assert sum(p.numel() for p in other.parameters()) == 34001
    w = sd["ff_3.weight"]                                       # xavier-normal: std = sqrt(2 / (fan_in + fan_out))
    assert abs(w.std().item() - (2.0 / 200) ** 0.5) < 0.01


def test_ranker_uses_the_fused_scorer_and_the_fused_stack_for_other_configs():
    import ptranking_amd as pa
    from ptranking_amd.scorer import FusedPointScorer, FlatAdam
    sf = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3,
          "pointsf": dict(num_features=136, num_layers=3, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None, bn_affine=False)}
    r = pa.LambdaRank(sf_para_dict=sf, model_para_dict={"sigma": 1.0}, gpu=True, device="cuda:0")
    r.init()
    assert isinstance(r.point_sf, FusedPointScorer) and isinstance(r.optimizer, FlatAdam)
    X = torch.randn(8, 32, 136, device="cuda")
    Y = torch.sort(torch.randint(0, 5, (8, 32), device="cuda").float(), dim=1, descending=True)[0]
    Y[:, 0] = 2.0
    r.train_mode()
    before = r.point_sf.flat.detach().clone()
    loss, stop = r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
    assert torch.isfinite(loss) and not torch.equal(before, r.point_sf.flat)
    sf2 = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3,
           "pointsf": dict(num_features=136, num_layers=3, AF="GE", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None, bn_affine=False)}
    r2 = pa.LambdaRank(sf_para_dict=sf2, model_para_dict={"sigma": 1.0}, gpu=True, device="cuda:0")
    r2.init()
    from ptranking_amd.linear import FusedStack
    from ptranking_amd.scorer import FlatViewAdam
    assert isinstance(r2.point_sf, FusedStack) and isinstance(r2.optimizer, FlatViewAdam)   # GELU: the layer-wise fused stack, flat parameters
    loss2, _ = r2.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
    assert torch.isfinite(loss2)


def test_dropout_generator_independence_within_and_across_rows():
    """Keep decisions of the counter-based generator: the four elements of a feature group, neighbouring feature groups of one row
    (they share the row key), neighbouring rows of one feature group and the same element at neighbouring sites are pairwise
    independent (joint keep rate = 0.81 at p = 0.1), and no column / row is biased."""
    from ptranking_amd.scorer import FusedPointScorer
    f = FusedPointScorer(136, 3, dropout=0.1).cuda()
    for seed in (1, 2 ** 40 + 12345, 987654321):
        m = f.dropout_mask(16384, 0, seed)                       # [R, 136] 1 / 0
        h = f.dropout_mask(16384, 1, seed)                       # [R, 100]
        assert abs(m.mean().item() - 0.9) < 1.5e-3
        tol = 4e-3
        for a, b, what in ((m[:, 0::4], m[:, 1::4], "elements 0/1"), (m[:, 1::4], m[:, 2::4], "elements 1/2"),
                           (m[:, 2::4], m[:, 3::4], "elements 2/3"), (m[:, 0::4], m[:, 3::4], "elements 0/3"),
                           (m[:, :-4], m[:, 4:], "neighbouring feature groups"), (m[:, :-8], m[:, 8:], "feature groups two apart"),
                           (m[:-1], m[1:], "neighbouring rows"), (m[:-64], m[64:], "rows 64 apart"),
                           (m[:, :100], h, "sites 0 / 1")):
            joint = (a * b).mean().item()
            assert abs(joint - 0.81) < tol, f"{what}: joint keep rate {joint:.4f} (seed {seed})"
        assert (m.mean(dim=0) - 0.9).abs().max() < 0.012 and (h.mean(dim=0) - 0.9).abs().max() < 0.012
        # runs along a row: P(two neighbours dropped) = 0.01
        dd = ((1 - m[:, :-1]) * (1 - m[:, 1:])).mean().item()
        assert abs(dd - 0.01) < 1.5e-3, dd


@pytest.mark.parametrize("opt", ["Adagrad", "RMS"])
def test_flat_adagrad_and_rmsprop_match_torch(opt):
    """The other two optimisers the reference configures (ptranking/base/ranker.py:518-521) as one kernel on the flat parameter tensor;
    the ranker picks them for opt = 'Adagrad' / 'RMS' (FusedPointScorer and, as flat views, the layer-wise stack)."""
    import ptranking_amd as pa
    from ptranking_amd.scorer import FLAT_OPTIMIZERS, FLAT_VIEW_OPTIMIZERS
    torch.manual_seed(3)
    p1 = torch.nn.Parameter(torch.randn(34001, device="cuda"))
    p2 = torch.nn.Parameter(p1.detach().clone())
    o1 = FLAT_OPTIMIZERS[opt]([p1], lr=1e-2, weight_decay=1e-3)
    o2 = (torch.optim.Adagrad if opt == "Adagrad" else torch.optim.RMSprop)([p2], lr=1e-2, weight_decay=1e-3)
    s1 = torch.optim.lr_scheduler.StepLR(o1, step_size=2, gamma=0.5)
    s2 = torch.optim.lr_scheduler.StepLR(o2, step_size=2, gamma=0.5)
    for it in range(6):
        g = torch.randn(34001, device="cuda") * (10.0 ** (it - 3))
        p1.grad, p2.grad = g.clone(), g.clone()
        o1.step(); o2.step(); s1.step(); s2.step()
        if opt == "Adagrad":
            assert torch.allclose(p1, p2, rtol=1e-5, atol=1e-7), it
        else:
            # RMSprop's first steps are lr * g / (0.1 |g| + eps): a +-10 lr sign function of g that flips within |g| ~ 1e-7, so the few
            # coordinates whose g = grad + wd * p cancels to ~0 amplify the rounding of that sum by 1e6 — on both sides alike
            d = (p1 - p2).abs()
            assert float(d.max()) < 2e-3 and float((d > 1e-6).float().mean()) < 1e-3, (it, float(d.max()))
            with torch.no_grad():
                p2.copy_(p1); o2.state[p2]["square_avg"].copy_(o1.state[p1]["square_avg"])       # re-synchronise the ill-conditioned coordinates
    for sfd, table in ((dict(num_features=136, num_layers=3, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None, bn_affine=False), FLAT_OPTIMIZERS),
                       (dict(num_features=136, num_layers=2, AF="GE", TL_AF="S", apply_tl_af=False, BN=True, bn_type="BN", bn_affine=True), FLAT_VIEW_OPTIMIZERS)):
        r = pa.LambdaRank(sf_para_dict={"sf_id": "pointsf", "opt": opt, "lr": 1e-3, "pointsf": sfd}, model_para_dict={"sigma": 1.0}, gpu=True,
                          device="cuda:0")
        r.init(); r.train_mode()
        assert type(r.optimizer) is table[opt]
        X = torch.randn(8, 32, 136, device="cuda")
        Y = torch.sort(torch.randint(0, 5, (8, 32), device="cuda").float(), dim=1, descending=True)[0]
        Y[:, 0] = 2.0
        before = [q.detach().clone() for q in r.get_parameters()]
        loss, _ = r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
        assert torch.isfinite(loss) and any(not torch.equal(a, b) for a, b in zip(before, r.get_parameters()))


# ---- r4: the layer-wise backward in two passes (the fused kernel's TAIL form + the first-layer dW kernel; NL = 2 / 3) against the NL + 2 kernels
@pytest.mark.parametrize("F,NL,R", [(24, 3, 1), (24, 3, 31), (24, 2, 33), (8, 3, 100), (46, 3, 515), (144, 3, 1111), (128, 2, 900), (180, 2, 300), (700, 3, 97),
                                    (136, 3, 4097), (200, 3, 2048 + 5)])
@pytest.mark.parametrize("p", [0.1, 0.0])
def test_two_pass_backward_equals_the_layer_wise_kernels(F, NL, R, p, monkeypatch):
    """Every parameter gradient through `ptr_mlp_backward` with PTR_BWD_TAIL=1 (default) and =0 on identical stored activations (PTR_BWD_FUSED=0 so
    that F = 136 takes this path too): equal to fp32 rounding (summation order differs), every entry written, no NaN from the rows past R
    of the last 32-document slab."""
    import ctypes as C
    from ptranking_amd import _lib
    from ptranking_amd.scorer import FusedPointScorer
    monkeypatch.setenv("PTR_BWD_FUSED", "0")
    torch.manual_seed(F + R)
    fused = FusedPointScorer(F, num_layers=NL, dropout=p).cuda()
    X = torch.randn(R, F, device="cuda"); w = torch.randn(R, device="cuda")
    preds = torch.empty(R, device="cuda"); acts = _scorer.alloc_acts(R, NL, "cuda")
    st = _lib.current_stream(X.device)
    seed = 77 + R
    _lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(fused.flat.data), R, F, NL, 1, C.c_float(p), C.c_uint64(seed), _lib.ptr(preds), _lib.ptr(acts), st)
    ws = torch.empty(_lib.query("ptr_mlp_backward_ws_floats", F, NL), device="cuda")
    dz = torch.empty(NL * R * 112, device="cuda")
    grads = {}
    for tail in ("1", "0"):
        monkeypatch.setenv("PTR_BWD_TAIL", tail)
        g = torch.full_like(fused.flat.data, float("nan"))
        ws.fill_(float("nan")); dz.fill_(float("nan"))
        _lib.call("ptr_mlp_backward", _lib.ptr(X), _lib.ptr(fused.flat.data), _lib.ptr(acts), _lib.ptr(w), R, F, NL, C.c_float(p), C.c_uint64(seed),
                  _lib.ptr(dz), _lib.ptr(ws), _lib.ptr(g), st)
        torch.cuda.synchronize()
        assert torch.isfinite(g).all(), tail
        grads[tail] = g.double().cpu()
    scale = max(1.0, float(grads["0"].abs().max()))
    assert float((grads["1"] - grads["0"]).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("kind", ["Adam", "Adagrad", "RMS"])
def test_opt_step_loss_equals_the_separate_step_and_sum(kind):
    """ABI v4 `ptr_opt_step_loss`: the optimiser step + loss-slot sum launch of the data-parallel direct step gives the bits of the separate
    ptr_adam_step / ptr_adagrad_step / ptr_rmsprop_step and ptr_sum_f32 calls (it runs the arithmetic of ptr_mlp_backward_step's reduction tail)."""
    import ctypes as C
    from ptranking_amd import _lib
    from ptranking_amd.scorer import FLAT_OPTIMIZERS
    torch.manual_seed(5)
    n, nq = 34001, 1000
    p0 = torch.randn(n, device="cuda"); g = torch.randn(n, device="cuda"); lq = torch.rand(nq, device="cuda")
    pa_, pb_ = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone())
    kw = dict(lr=1e-2, weight_decay=1e-3)
    oa, ob = FLAT_OPTIMIZERS[kind]([pa_], **kw), FLAT_OPTIMIZERS[kind]([pb_], **kw)
    st = _lib.current_stream(p0.device)
    for it in range(3):
        pa_.grad = g * (it + 1); pb_.grad = g * (it + 1)
        oa.step()
        want = torch.empty(1, device="cuda")
        _lib.call("ptr_sum_f32", _lib.ptr(lq), nq, C.c_float(1.0), _lib.ptr(want), st)
        k, lr, h1, h2, eps, wd, step, s1, s2 = ob.fused_step_args(pb_)
        got = torch.empty(1, device="cuda")
        _lib.call("ptr_opt_step_loss", _lib.ptr(pb_), _lib.ptr(pb_.grad), C.c_int64(n), k, C.c_float(lr), C.c_float(h1), C.c_float(h2), C.c_float(eps),
                  C.c_float(wd), step, _lib.ptr(s1), _lib.ptr(s2), _lib.ptr(lq), nq, _lib.ptr(got), st)
        assert torch.equal(pa_.detach(), pb_.detach()), (kind, it)
        assert torch.equal(got, want)
        assert torch.equal(pb_.grad, g * (it + 1))                  # the gradient is read, not rewritten
