"""GPU: the plugin surface end to end — ranker classes (scorer + fused loss + optimiser step) and the device Evaluator —
against the torch-CPU oracle restatement of the reference's train step / metric loop."""
import copy

import numpy as np
import pytest
import torch

import golden_util as G

pytestmark = pytest.mark.gpu

SF = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3,
      "pointsf": dict(num_features=24, num_layers=3, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None,
                      bn_affine=False, dropout=0.0)}


def make_data(seed, B, L, F):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((B, L, F)).astype(np.float32)
    Y = rng.choice(5, size=(B, L), p=[0.5147, 0.3250, 0.1339, 0.0183, 0.0081]).astype(np.float32)
    Y[:, 0] = np.maximum(Y[:, 0], 1)
    Y = -np.sort(-Y, axis=1)
    return torch.from_numpy(X), torch.from_numpy(Y)


def _make(name, paras):
    import ptranking_amd as pa
    cls = getattr(pa, name)
    if name == "ListNet":
        return cls(sf_para_dict=copy.deepcopy(SF), gpu=True, device="cuda:0")
    return cls(sf_para_dict=copy.deepcopy(SF), model_para_dict=paras, gpu=True, device="cuda:0")


CASES = [
    ("RankNet", dict(sigma=1.0), "ranknet_loss", dict(sigma=1.0)),
    ("LambdaRank", dict(sigma=1.0), "lambdarank_loss", dict(sigma=1.0)),
    ("LambdaLoss", dict(k=5, sigma=1.0, mu=5.0, loss_type="NDCG_Loss2"), "lambdaloss_loss", dict(k=5, sigma=1.0, mu=5.0, loss_type=1)),
    ("ApproxNDCG", dict(alpha=10.0), "approxndcg_loss", dict(alpha=10.0)),
    ("ListNet", None, "listnet_loss", {}),
]


@pytest.mark.parametrize("name,paras,oracle_fn,okw", CASES)
def test_train_op_matches_cpu_reference_step(name, paras, oracle_fn, okw):
    """3 optimiser steps: parameters and losses follow the CPU restatement of the reference's train_op."""
    from oracle import torch_ref as T
    import ptranking_amd as pa
    torch.manual_seed(137)
    ranker = _make(name, paras)
    ranker.init()
    ranker.train_mode()
    cpu_net = T.build_pointsf(24, dropout=0.0)                       # the torch-CPU restatement of the scorer ...
    cpu_net.load_state_dict({k: v.cpu() for k, v in ranker.point_sf.state_dict().items()})   # ... with the ranker's weights
    cpu_opt = torch.optim.Adam(cpu_net.parameters(), lr=1e-3, weight_decay=1e-3)
    X, Y = make_data(5, 6, 40, 24)
    for step in range(3):
        loss, stop = ranker.train_op(X.cuda(), Y.cuda(), epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
        ref = T.cpu_train_step(cpu_net, cpu_opt, X, Y, getattr(T, oracle_fn), **okw)
        assert stop is False
        G.assert_close(loss.item(), ref, f"{name} loss step {step}")
    for (n1, p1), (n2, p2) in zip(ranker.point_sf.state_dict().items(), cpu_net.named_parameters()):
        assert n1 == n2
        if n1 == "ff_5.bias":
            continue   # every in-scope loss is shift-invariant: this gradient is identically 0 and Adam turns its rounding
                       # noise into +-lr moves on both sides
        assert torch.allclose(p1.detach().cpu(), p2.detach(), rtol=1e-4, atol=2e-5), n1


def test_listmle_ranker_torch_tie_shuffle_matches_reference_stream():
    """With tie_shuffle='torch' the ranker consumes the same randperm stream as the reference's arg_shuffle_ties."""
    from oracle import torch_ref as T
    import ptranking_amd as pa
    ranker = _make("ListMLE", {})
    ranker.init()
    X, Y = make_data(6, 4, 30, 24)
    preds = ranker.forward(X.cuda()).detach()
    assert ranker.tie_shuffle == "device"                 # product default = what bench.py measures
    ranker.tie_shuffle = "torch"
    torch.manual_seed(99)
    perm_dev = ranker._shuffle_ties(Y.cuda())
    torch.manual_seed(99)
    # the CUDA generator stream differs from the CPU one, so compare semantics: valid tie-respecting order + loss parity
    pn = perm_dev.cpu()
    assert torch.equal(torch.gather(Y, 1, pn), Y)
    loss = pa.functional.listmle_loss(preds, perm_dev)
    ref, _ = T.loss_and_grad(T.listmle_loss, preds.cpu(), pn)
    G.assert_close(loss.item(), ref.item(), "listmle")
    ranker.tie_shuffle = "device"
    l2, _ = ranker.train_op(X.cuda(), Y.cuda(), epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
    assert torch.isfinite(l2)


def test_train_epoch_and_stop_check():
    import ptranking_amd as pa
    ranker = _make("LambdaRank", dict(sigma=1.0))
    ranker.init()
    X, Y = make_data(7, 8, 20, 24)
    loader = [(list(range(4)), X[:4], Y[:4]), (list(range(4, 8)), X[4:], Y[4:])]
    l1, stop = ranker.train(loader, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
    assert l1.shape == (1,) and l1.is_cuda and not stop
    l10, stop = ranker.train(loader, epoch_k=10, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)   # stop check epoch
    assert not stop
    with torch.no_grad():
        for p in ranker.point_sf.parameters():
            p.zero_()
    _, stop = ranker.train(loader, epoch_k=10, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
    assert stop is True   # 'All zero error.' (ranker.py:553-555)
    with pytest.raises(AssertionError):
        ranker.custom_loss_function(torch.zeros(1, 4, device="cuda"), torch.zeros(1, 4, device="cuda"), presort=False,
                                    label_type=pa.LABEL_TYPE.MultiLabel)


def test_device_evaluator_matches_cpu_metric_loop(tmp_path):
    from oracle import torch_ref as T
    import ptranking_amd as pa
    torch.manual_seed(1)
    ranker = _make("LambdaRank", dict(sigma=1.0))
    ranker.init()
    loaders = []
    for seed, (B, L) in enumerate([(5, 12), (7, 30), (3, 64), (4, 8)]):
        X, Y = make_data(20 + seed, B, L, 24)
        loaders.append((list(range(B)), X, Y))
    ks = [1, 3, 5, 10, 20, 50]
    ndcg, nerr, ap, p = ranker.adhoc_performance_at_ks(test_data=loaders, ks=ks, label_type=pa.LABEL_TYPE.MultiLabel,
                                                       presort=True, device="cpu")
    assert not ndcg.is_cuda and ndcg.shape == (len(ks),)
    ranker.eval_mode()
    sums = {m: torch.zeros(len(ks)) for m in ("ndcg", "nerr", "ap", "p")}
    nq = 0
    for ids, X, Y in loaders:
        preds = ranker.predict(X.cuda()).detach().cpu()
        out = T.evaluate_at_ks(preds, Y, ks, presort=True)
        for m in sums:
            sums[m] += out[m].sum(0)
        nq += len(ids)
    for got, m in ((ndcg, "ndcg"), (nerr, "nerr"), (ap, "ap"), (p, "p")):
        G.assert_close(got.numpy(), (sums[m] / nq).numpy(), m)
    # single-k forms skip lists shorter than k (ranker.py:41-42)
    k = 10
    got = ranker.ndcg_at_k(test_data=loaders, k=k, label_type=pa.LABEL_TYPE.MultiLabel, presort=True)
    s, nq = 0.0, 0
    for ids, X, Y in loaders:
        if Y.size(1) < k:
            continue
        preds = ranker.predict(X.cuda()).detach().cpu()
        s += T.evaluate_at_ks(preds, Y, [k], presort=True)["ndcg"].sum().item()
        nq += len(ids)
    assert got.shape == (1,)
    G.assert_close(got.item(), s / nq, "ndcg@10")
    assert ranker.validation(vali_data=loaders, vali_metric="nDCG", k=5, presort=True).shape == (1,)
    # checkpoint round trip (point_ranker.py:63-71)
    ranker.save(str(tmp_path) + "/", "net.pkl")
    other = _make("LambdaRank", dict(sigma=1.0))
    other.init()
    other.load(str(tmp_path) + "/net.pkl", device="cuda:0")
    for a, b in zip(ranker.point_sf.parameters(), other.point_sf.parameters()):
        assert torch.equal(a, b)
    # a checkpoint written by the fused scorer loads into the plain torch module (reference key names)
    plain = T.build_pointsf(24, dropout=0.0)
    plain.load_state_dict(torch.load(str(tmp_path) + "/net.pkl", map_location="cpu"))


def test_need_per_q_lists_hold_the_per_query_values():
    """`adhoc_performance_at_ks(need_per_q=True)` (ranker.py:202-263) returns, per metric, one [B, len(ks)] CPU tensor
    per batch — value-checked here against the oracle's metric loop, and against the averages of the same call."""
    from oracle import torch_ref as T
    import ptranking_amd as pa
    torch.manual_seed(2)
    ranker = _make("LambdaRank", dict(sigma=1.0))
    ranker.init()
    loaders = []
    for seed, (B, L) in enumerate([(6, 20), (3, 55), (5, 9)]):
        X, Y = make_data(40 + seed, B, L, 24)
        loaders.append((list(range(B)), X, Y))
    ks = [1, 3, 5, 10, 20]
    out = ranker.adhoc_performance_at_ks(test_data=loaders, ks=ks, label_type=pa.LABEL_TYPE.MultiLabel, presort=True, device="cpu",
                                         need_per_q=True)
    assert len(out) == 8
    avgs, lists = out[:4], out[4:]
    ranker.eval_mode()
    for mi, m in enumerate(("ndcg", "nerr", "ap", "p")):
        assert len(lists[mi]) == len(loaders)
        tot, nq = torch.zeros(len(ks)), 0
        for (ids, X, Y), got in zip(loaders, lists[mi]):
            assert not got.is_cuda and got.shape == (len(ids), len(ks))
            ref = T.evaluate_at_ks(ranker.predict(X.cuda()).detach().cpu(), Y, ks, presort=True)[m]
            G.assert_close(got.numpy(), ref.numpy(), f"{m} per query")
            tot += got.sum(0)
            nq += len(ids)
        G.assert_close(avgs[mi].numpy(), (tot / nq).numpy(), f"{m} average = mean of the per-query values")


@pytest.mark.parametrize("name,paras", [("RankNet", dict(sigma=1.0)), ("LambdaRank", dict(sigma=1.5)),
                                        ("LambdaLoss", dict(k=5, sigma=1.0, mu=5.0, loss_type="NDCG_Loss2++")), ("ListNet", None)])
@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_direct_train_step_is_bit_identical_to_the_autograd_path(name, paras, dropout):
    """rankers.FusedStepMixin._direct_train_op issues the same kernels with the same arguments in the same order as
    forward -> custom_loss_function (autograd): parameters, optimiser state and losses must agree bit for bit."""
    import ptranking_amd as pa
    sf = copy.deepcopy(SF)
    sf["pointsf"].update(num_features=136, dropout=dropout)

    def make(direct):
        torch.manual_seed(11)
        cls = getattr(pa, name)
        r = cls(sf_para_dict=copy.deepcopy(sf), gpu=True, device="cuda:0") if paras is None else \
            cls(sf_para_dict=copy.deepcopy(sf), model_para_dict=dict(paras), gpu=True, device="cuda:0")
        r.init()
        r.point_sf.dropout = dropout
        r.train_mode()
        r.use_direct_step = direct
        return r

    a, b = make(True), make(False)
    a3 = make(True)
    a3.single_call_step = False                                  # r6: `a` takes ONE C-ABI call per step (ptr_train_step), `a3` the three calls it chains
    assert a.single_call_step is True
    X, Y = make_data(5, 9, 70, 136)
    X, Y = X.cuda(), Y.cuda()
    lens = torch.tensor([70, 3, 70, 1, 55, 70, 64, 65, 2], dtype=torch.int32, device="cuda")
    for step, kw in enumerate([{}, {}, {"lens": lens}, {}]):
        out = []
        for r in (a, b, a3):
            torch.manual_seed(100 + step)                       # the dropout seed is drawn from torch's CPU generator
            out.append(r.train_op(X, Y, epoch_k=1 if step else 10, presort=True, label_type=pa.LABEL_TYPE.MultiLabel, **kw))
        (la, sa), (lb, sb), (l3, s3) = out
        assert sa == sb == s3 and torch.equal(la.reshape(()), lb.reshape(())) and torch.equal(la.reshape(()), l3.reshape(())), (step, la, lb, l3)
        assert torch.equal(a.point_sf.flat, b.point_sf.flat) and torch.equal(a.point_sf.flat, a3.point_sf.flat), step
    assert "desc" in next(iter(a._direct_buffers.values())) and "desc" not in next(iter(a3._direct_buffers.values()))
    sta, stb = a.optimizer.state[a.point_sf.flat], b.optimizer.state[b.point_sf.flat]
    assert sta["step"] == stb["step"] == 4 and torch.equal(sta["exp_avg_sq"], stb["exp_avg_sq"])
    assert "_direct_buffers" in a.__dict__ and "_direct_buffers" not in b.__dict__
    a.scheduler.step()                                           # no "scheduler before optimizer" warning path

    class Custom(type(a)):                                       # a plugin that overrides the loss keeps the autograd path
        def custom_loss_function(self, batch_preds, batch_std_labels, **kwargs):
            self.seen = True
            return super().custom_loss_function(batch_preds, batch_std_labels, **kwargs)

    c = Custom(sf_para_dict=copy.deepcopy(sf), gpu=True, device="cuda:0") if paras is None else \
        Custom(sf_para_dict=copy.deepcopy(sf), model_para_dict=dict(paras), gpu=True, device="cuda:0")
    c.init(); c.train_mode()
    c.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
    assert getattr(c, "seen", False) and "_direct_buffers" not in c.__dict__


@pytest.mark.parametrize("x6", ["0", "2"])
def test_single_call_step_matches_the_three_calls_under_either_forward(x6, monkeypatch):
    """ptr_train_step (ABI v5) with the fp32-MFMA forward (wimg = NULL) and with the bf16x6 forward (PTR_MLP_X6=2 forces it at any row
    count): bit-identical parameters, optimiser state and loss to the three separate entry points, over several steps with dropout."""
    import ptranking_amd as pa
    monkeypatch.setenv("PTR_MLP_X6", x6)
    sf = copy.deepcopy(SF)
    sf["pointsf"].update(num_features=136, dropout=0.1)

    def make(single):
        torch.manual_seed(12)
        r = pa.LambdaRank(sf_para_dict=copy.deepcopy(sf), model_para_dict=dict(sigma=1.0), gpu=True, device="cuda:0")
        r.init(); r.point_sf.dropout = 0.1; r.train_mode()
        r.single_call_step = single
        return r

    a, b = make(True), make(False)
    X, Y = make_data(6, 33, 128, 136)
    X, Y = X.cuda(), Y.cuda()
    for step in range(3):
        out = []
        for r in (a, b):
            torch.manual_seed(200 + step)
            out.append(r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel))
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(a.point_sf.flat, b.point_sf.flat), step
    d = next(iter(a._direct_buffers.values()))["desc"]
    assert (d.wimg is not None) == (x6 == "2")


def test_weight_image_hand_over_between_train_steps(monkeypatch):
    """r6: with the bf16x6 forward, ptr_train_step's optimiser launch rewrites the forward's weight image element by element and the next step skips
    the prep launch (`wimg_current`).  The image must be trusted exactly when nothing touched the parameters or the (shared) image in between:
    a run that keeps handing the image over, a run that always rebuilds it (`reuse_weight_image = False`) and the three-call path must produce the
    same bits through: plain steps, an evaluation forward in between, an in-place edit of the parameters by torch, a second ranker stepping through
    the same image buffer, and a step on the three-call path."""
    import ptranking_amd as pa
    monkeypatch.setenv("PTR_MLP_X6", "2")
    sf = copy.deepcopy(SF)
    sf["pointsf"].update(num_features=136, dropout=0.1)

    def make(seed, reuse=True, single=True):
        torch.manual_seed(seed)
        r = pa.LambdaRank(sf_para_dict=copy.deepcopy(sf), model_para_dict=dict(sigma=1.0), gpu=True, device="cuda:0")
        r.init(); r.point_sf.dropout = 0.1; r.train_mode()
        r.reuse_weight_image, r.single_call_step = reuse, single
        return r

    a, b, c = make(12), make(12, reuse=False), make(12, single=False)
    other = make(99)                                              # same (device, F, NL): shares the image buffer with a / b / c
    X, Y = make_data(6, 33, 128, 136)
    X, Y = X.cuda(), Y.cuda()
    kw = dict(epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
    seen = []

    def step(i, rankers=None):
        outs = []
        for r in (rankers or (a, b, c)):
            torch.manual_seed(300 + i)
            outs.append(r.train_op(X, Y, **kw)[0])
            if r is a:
                seen.append(int(next(iter(a._direct_buffers.values()))["desc"].wimg_current))
        return outs

    def same(i):
        assert torch.equal(a.point_sf.flat, b.point_sf.flat) and torch.equal(a.point_sf.flat, c.point_sf.flat), i

    for i in range(3):                                            # a alone hands its image over only when it steps back to back
        o = step(i, (a,)); step(i, (b,)); step(i, (c,)); same(i)
    # every ranker shares ONE image buffer, so b's / c's steps in between invalidate a's tag: a rebuilt the image each time so far
    assert seen == [0, 0, 0], seen
    for i in range(3, 6):                                         # back-to-back steps of a: the hand-over happens
        step(i, (a,))
    assert seen[3:] == [0, 1, 1], seen
    for i in range(3, 6):
        step(i, (b,)); step(i, (c,))
    same(6)
    a.eval_mode(); _ = a.predict(X); a.train_mode()               # an evaluation forward rebuilds the image through the other entry point
    for r in (b, c):
        r.eval_mode(); _ = r.predict(X); r.train_mode()
    n0 = len(seen); step(7, (a,)); step(8, (a,))
    assert seen[n0:] == [0, 1], seen
    step(7, (b,)); step(8, (b,)); step(7, (c,)); step(8, (c,)); same(8)
    with torch.no_grad():                                         # torch edits the parameters in place: version counter moves
        for r in (a, b, c):
            r.point_sf.flat.mul_(0.999)
    n0 = len(seen); step(9, (a,)); step(10, (a,))
    assert seen[n0:] == [0, 1], seen
    step(9, (b,)); step(10, (b,)); step(9, (c,)); step(10, (c,)); same(10)
    torch.manual_seed(5); other.train_op(X, Y, **kw)              # another ranker steps through the shared image
    n0 = len(seen); step(11, (a,))
    assert seen[n0:] == [0], seen
    a.single_call_step = False; step(12, (a,)); a.single_call_step = True      # a three-call step moves the parameters without refreshing the image
    n0 = len(seen); step(13, (a,)); step(14, (a,))
    assert seen[n0:] == [0, 1], seen
    for i in (11, 12, 13, 14):
        step(i, (b,)); step(i, (c,))
    same(14)
    sa, sb = a.optimizer.state[a.point_sf.flat], b.optimizer.state[b.point_sf.flat]
    assert sa["step"] == sb["step"] and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])


def test_train_step_descriptor_is_checked():
    """ptr_train_step refuses a descriptor of another size / an unknown loss before any launch."""
    import ctypes as C
    from ptranking_amd import _lib
    d = _lib.TrainStepDesc()
    d.struct_bytes = C.sizeof(_lib.TrainStepDesc) - 8
    with pytest.raises(RuntimeError, match="descriptor of"):
        _lib.call("ptr_train_step", C.addressof(d), None)
    d.struct_bytes = C.sizeof(_lib.TrainStepDesc)
    d.B, d.L, d.loss_kind = 1, 8, 9
    with pytest.raises(RuntimeError, match="unknown loss"):
        _lib.call("ptr_train_step", C.addressof(d), None)


@pytest.mark.parametrize("opt", ["Adagrad", "RMS"])
def test_weight_image_hand_over_with_the_other_flat_optimisers(opt, monkeypatch):
    """The optimiser launch refreshes the bf16x6 weight image behind an Adagrad / RMSprop step as it does behind Adam's (ptranking/base/ranker.py:516-521
    `opt` choices): four back-to-back steps of the one-call path (image handed over from the second step on) against the three-call path, bit for bit."""
    import ptranking_amd as pa
    monkeypatch.setenv("PTR_MLP_X6", "2")
    sf = copy.deepcopy(SF)
    sf["opt"] = opt
    sf["pointsf"].update(num_features=136, dropout=0.1)

    def make(single):
        torch.manual_seed(21)
        r = pa.LambdaRank(sf_para_dict=copy.deepcopy(sf), model_para_dict=dict(sigma=1.0), gpu=True, device="cuda:0")
        r.init(); r.point_sf.dropout = 0.1; r.train_mode()
        r.single_call_step = single
        return r

    X, Y = make_data(8, 17, 128, 136)
    X, Y = X.cuda(), Y.cuda()
    kw = dict(epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
    a = make(True)
    cur, la = [], []
    for i in range(4):                                            # a's steps back to back: nothing else touches the shared image in between
        torch.manual_seed(400 + i)
        la.append(a.train_op(X, Y, **kw)[0].clone())
        cur.append(int(next(iter(a._direct_buffers.values()))["desc"].wimg_current))
    assert cur == [0, 1, 1, 1], cur
    c = make(False)
    for i in range(4):
        torch.manual_seed(400 + i)
        lc = c.train_op(X, Y, **kw)[0]
        assert torch.equal(la[i], lc), (opt, i)
    assert type(a.optimizer).__name__ == {"Adagrad": "FlatAdagrad", "RMS": "FlatRMSprop"}[opt]
    assert torch.equal(a.point_sf.flat, c.point_sf.flat)


@pytest.mark.parametrize("name", sorted(G.STEP_CASES))
def test_train_steps_match_the_reference_itself(name):
    """r6: three whole train steps of the product path — fused scorer kernels, fused loss + gradient kernel, fused backward / Adam, enqueued by
    ptr_train_step — against the REFERENCE's own ranker objects run for three steps on the same weights and batch (tests/golden/step.npz, produced by
    tests/golden/make_golden_step.py: NeuralRanker.init + train_op, ranker.py:512-525,589-603; dropout 0).  Losses within the golden gates; the
    parameters after three Adam steps within 1e-4 relative / 2e-5 absolute (Adam divides by sqrt(v): rounding noise on near-zero gradients moves a
    coordinate by up to lr)."""
    import ptranking_amd as pa
    c = G.steps()[name]
    cls_name, paras, _, _ = G.STEP_CASES[name]
    X, Y = torch.from_numpy(c["X"]).cuda(), torch.from_numpy(c["Y"]).cuda()
    sf = copy.deepcopy(SF)
    sf["pointsf"].update(num_features=int(X.shape[2]), dropout=0.0)
    cls = getattr(pa, cls_name)
    r = cls(sf_para_dict=sf, gpu=True, device="cuda:0") if paras is None else cls(sf_para_dict=sf, model_para_dict=dict(paras), gpu=True, device="cuda:0")
    r.init()
    r.point_sf.load_state_dict({k[len("sd0/"):]: torch.from_numpy(v).cuda() for k, v in c.items() if k.startswith("sd0/")})
    r.train_mode()
    for step in range(3):
        loss, stop = r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
        assert stop is False
        G.assert_close(loss.item(), c["losses"][step], f"loss of step {step}")
    assert "desc" in next(iter(r._direct_buffers.values())), "the one-call step must be the path taken"
    for k, v in r.point_sf.state_dict().items():
        if k == "ff_5.bias":
            continue
        assert np.allclose(v.detach().cpu().numpy(), c[f"sd3/{k}"], rtol=1e-4, atol=2e-5), k
