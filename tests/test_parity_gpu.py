"""GPU parity tests: the HIP path (through the C ABI) against the golden fixtures produced by the reference, against
the oracle on seeded inputs, and through size-independent properties at BASELINE.json's full list lengths.

Tolerance (north_star: "losses/grads match the reference CPU path within 1e-5 fp32"): |got - ref| <= 1e-5 * max(1,
max|ref|) — see tests/golden_util.py.  Sort indices: bit-exact on tie-free inputs.
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


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def loss_and_grad(fn, preds_np, *args, **kw):
    p = dev(preds_np).requires_grad_(True)
    loss = fn(p, *args, **kw)
    loss.backward()
    return loss.detach().cpu().numpy(), p.grad.detach().cpu().numpy()


def _presort(c):
    return bool(int(c["presort"])) if "presort" in c else True


# ------------------------------------------------------------------------------------------ golden fixtures (reference outputs)
@pytest.mark.parametrize("name", G.case_ids("ranknet"))
def test_golden_ranknet(F, name):
    c = G.losses()["ranknet"][name]
    loss, grad = loss_and_grad(F.ranknet_loss, c["preds"], dev(c["labels"]), sigma=float(c["sigma"]))
    G.assert_close(loss, c["loss"], "loss")
    G.assert_close(grad, c["grad"], "grad")


@pytest.mark.parametrize("name", G.case_ids("lambdarank"))
def test_golden_lambdarank(F, name):
    c = G.losses()["lambdarank"][name]
    loss, grad = loss_and_grad(F.lambdarank_loss, c["preds"], dev(c["labels"]), sigma=float(c["sigma"]))
    G.assert_close(loss, c["loss"], "loss")
    G.assert_close(grad, c["grad"], "grad")
    vals, idx = F.sort_desc(dev(c["preds"]))
    assert np.array_equal(idx.cpu().numpy(), c["sort_idx"]), "sort indices must be bit-exact"


@pytest.mark.parametrize("name", G.case_ids("lambdaloss"))
def test_golden_lambdaloss(F, name):
    c = G.losses()["lambdaloss"][name]
    lt = {1: "NDCG_Loss2", 2: "NDCG_Loss2++"}[int(c["loss_type"])]
    loss, grad = loss_and_grad(F.lambdaloss_loss, c["preds"], dev(c["labels"]), k=int(c["k"]), sigma=float(c["sigma"]),
                               mu=float(c["mu"]), loss_type=lt, presort=_presort(c))
    G.assert_close(loss, c["loss"], "loss")
    G.assert_close(grad, c["grad"], "grad")


@pytest.mark.parametrize("name", G.case_ids("approxndcg"))
def test_golden_approxndcg(F, name):
    c = G.losses()["approxndcg"][name]
    loss, grad = loss_and_grad(F.approxndcg_loss, c["preds"], dev(c["labels"]), alpha=float(c["alpha"]), presort=_presort(c))
    G.assert_close(loss, c["loss"], "loss")
    G.assert_close(grad, c["grad"], "grad")


# The saturation band (tests/golden/losses_knife.npz) through the kernel forms the default dispatch does not pick for these list lengths: the
# LDS LambdaRank / ApproxNDCG kernels behind PTR_LAMBDARANK_RING=0 / PTR_APPROX_RING=0 (the switches are read per call), and the knife lists
# padded to L = 40 / 300 with `lens`, which moves RankNet off its two-queries-per-wavefront form and LambdaRank onto the LDS kernel (L > 256).
@pytest.mark.parametrize("name", [n for n in G.case_ids("lambdarank") if n.startswith("knife")])
def test_knife_lambdarank_other_kernel_forms(F, name, monkeypatch):
    c = G.losses()["lambdarank"][name]
    monkeypatch.setenv("PTR_LAMBDARANK_RING", "0")
    loss, grad = loss_and_grad(F.lambdarank_loss, c["preds"], dev(c["labels"]), sigma=float(c["sigma"]))
    G.assert_close(loss, c["loss"], "loss (LDS kernel)")
    G.assert_close(grad, c["grad"], "grad (LDS kernel)")
    monkeypatch.delenv("PTR_LAMBDARANK_RING")
    B, L = c["preds"].shape
    for Lp in (40, 300):
        if Lp <= L:
            continue
        P = np.zeros((B, Lp), np.float32); Y = np.zeros((B, Lp), np.float32)
        P[:, :L] = c["preds"]; Y[:, :L] = c["labels"]; P[:, L:] = 7.0          # padding scores must not matter
        lens = torch.full((B,), L, dtype=torch.int32, device="cuda")
        loss, grad = loss_and_grad(F.lambdarank_loss, P, dev(Y), sigma=float(c["sigma"]), lens=lens)
        G.assert_close(loss, c["loss"], f"loss (padded to {Lp})")
        G.assert_close(grad[:, :L], c["grad"], f"grad (padded to {Lp})")
        assert not grad[:, L:].any()


@pytest.mark.parametrize("name", [n for n in G.case_ids("ranknet") if n.startswith("knife")])
def test_knife_ranknet_unpacked_form(F, name):
    c = G.losses()["ranknet"][name]
    B, L = c["preds"].shape
    for Lp in (40, 300):
        if Lp <= L:
            continue
        P = np.zeros((B, Lp), np.float32); Y = np.zeros((B, Lp), np.float32)
        P[:, :L] = c["preds"]; Y[:, :L] = c["labels"]; P[:, L:] = -3.0
        lens = torch.full((B,), L, dtype=torch.int32, device="cuda")
        loss, grad = loss_and_grad(F.ranknet_loss, P, dev(Y), sigma=float(c["sigma"]), lens=lens)
        G.assert_close(loss, c["loss"], f"loss (padded to {Lp})")
        G.assert_close(grad[:, :L], c["grad"], f"grad (padded to {Lp})")


@pytest.mark.parametrize("name", [n for n in G.case_ids("approxndcg") if n.startswith("knife")])
def test_knife_approxndcg_lds_kernel(F, name, monkeypatch):
    c = G.losses()["approxndcg"][name]
    monkeypatch.setenv("PTR_APPROX_RING", "0")
    loss, grad = loss_and_grad(F.approxndcg_loss, c["preds"], dev(c["labels"]), alpha=float(c["alpha"]), presort=True)
    G.assert_close(loss, c["loss"], "loss (LDS kernel)")
    G.assert_close(grad, c["grad"], "grad (LDS kernel)")


@pytest.mark.parametrize("name", G.case_ids("listnet"))
def test_golden_listnet(F, name):
    c = G.losses()["listnet"][name]
    loss, grad = loss_and_grad(F.listnet_loss, c["preds"], dev(c["labels"]))
    G.assert_close(loss, c["loss"], "loss")
    G.assert_close(grad, c["grad"], "grad")


@pytest.mark.parametrize("name", G.case_ids("listmle"))
def test_golden_listmle(F, name):
    c = G.losses()["listmle"][name]
    loss, grad = loss_and_grad(F.listmle_loss, c["preds"], dev(c["perm"]))
    G.assert_close(loss, c["loss"], "loss")
    G.assert_close(grad, c["grad"], "grad")


@pytest.mark.parametrize("name", G.case_ids("rand", "metrics"))
def test_golden_metrics(F, name):
    c = G.metrics()["rand"][name]
    out = F.metrics_at_ks(dev(c["preds"]), dev(c["labels"]), [int(k) for k in c["ks"]], presort=bool(int(c["presort"])))
    for m in ("ndcg", "nerr", "ap", "p"):
        G.assert_close(out[m].cpu().numpy(), c[m], m)
    vals, idx = F.sort_desc(dev(c["preds"]))
    assert np.array_equal(idx.cpu().numpy(), c["sort_idx"])
    assert np.array_equal(vals.cpu().numpy(), c["sorted_vals"])
    # single cut-off forms (torch_*_at_k) equal the matching column
    k1 = int(c["k1"])
    one = F.metrics_at_ks(dev(c["preds"]), dev(c["labels"]), [k1], presort=bool(int(c["presort"])),
                          max_label=float(c["max_label"]))
    for m, key in (("ndcg", "ndcg_k"), ("nerr", "nerr_k"), ("ap", "ap_k"), ("p", "p_k")):
        G.assert_close(one[m].cpu().numpy(), c[key], key)


def test_golden_known_answer_vectors(F):
    """The reference's own testing/metric/testing_metric.py vectors, pushed through the device prologue."""
    checked = 0
    for name in G.case_ids("kat", "metrics"):
        c = G.metrics()["kat"][name]
        sys_sorted = c["sys_sorted"]
        if sorted(sys_sorted[0].tolist(), reverse=True) != c["ideal_sorted"][0].tolist():
            continue
        L = sys_sorted.shape[1]
        preds = -np.arange(L, dtype=np.float32)[None]
        out = F.metrics_at_ks(dev(preds), dev(sys_sorted), [int(k) for k in c["ks"]], presort=False)
        G.assert_close(out[str(c["kind"])].cpu().numpy(), c["expected"], name)
        assert np.allclose(out[str(c["kind"])].cpu().numpy()[0], c["commented"], atol=5e-5)
        checked += 1
    assert checked >= 3


# ------------------------------------------------------------------------------------------ oracle on seeded inputs (all tilings)
def synth(seed, B, L, lens=False, presort=True):
    rng = np.random.default_rng(seed)
    preds = rng.standard_normal((B, L)).astype(np.float32)
    labels = rng.choice(5, size=(B, L), p=[0.5147, 0.3250, 0.1339, 0.0183, 0.0081]).astype(np.float32)
    labels[:, 0] = np.maximum(labels[:, 0], 1.0)
    ln = None
    if lens:
        ln = rng.integers(1, L + 1, size=B).astype(np.int32)
        ln[0] = L
        for b in range(B):
            labels[b, ln[b]:] = 0
            if labels[b, :ln[b]].max() < 1:
                labels[b, 0] = 1
    if presort:
        for b in range(B):
            n = L if ln is None else ln[b]
            labels[b, :n] = -np.sort(-labels[b, :n])
    return preds, labels, ln


SHAPES = [(37, 1), (16, 2), (9, 64), (33, 65), (16, 128), (7, 129), (12, 256), (5, 300), (6, 512), (3, 1000), (3, 1024),
          (2, 1500), (2, 2048), (1, 4096)]


@pytest.mark.parametrize("B,L", SHAPES)
@pytest.mark.parametrize("use_lens", [False, True])
def test_oracle_pairwise(F, B, L, use_lens):
    from oracle import c_oracle as CO
    preds, labels, ln = synth(1000 + L, B, L, lens=use_lens)
    lens_t = None if ln is None else dev(ln)
    for name, fn, orc in (("lambdarank", F.lambdarank_loss, CO.lambdarank), ("ranknet", F.ranknet_loss, CO.ranknet)):
        loss, grad = loss_and_grad(fn, preds, dev(labels), sigma=1.0, lens=lens_t)
        lq, g = orc(preds, labels, 1.0, lens=ln)
        G.assert_close(loss, lq.astype(np.float64).sum(), f"{name} loss")
        G.assert_close(grad, g, f"{name} grad")


@pytest.mark.parametrize("L", [40, 128, 200, 256, 300, 512])
@pytest.mark.parametrize("case", ["huge", "tiny", "all_tied", "some_tied", "mixed_scale"])
def test_ring_rank_count_fallback(F, L, case):
    """The ring kernel counts ranks with one packed fma-with-clamp per two compares (exact unless BIG*s overflows or a score
    difference is below 2^-100) and recounts with compares when the sums are not a permutation: scores that overflow the
    scaled form, differences below its resolution, and ties must all come out as the oracle's (index-order tie-break)."""
    from oracle import c_oracle as CO
    B = 6
    preds, labels, ln = synth(4000 + L, B, L, lens=True)
    if case == "huge":
        preds = preds * np.float32(1e27)                  # BIG*s = inf -> inf - inf = NaN -> clamps to 0 (padding sentinel: -1e30)
        sigma = 1e-27
    elif case == "tiny":
        preds = preds * np.float32(1e-36)                 # differences below 2^-100: fractional counts
        sigma = 1.0
    elif case == "all_tied":
        preds = np.zeros_like(preds)
        sigma = 1.0
    elif case == "some_tied":
        preds = np.round(preds * 2) / 2                   # many equal scores
        sigma = 1.0
    else:
        preds[:, ::3] *= np.float32(1e-34)                # a third of the list far below the resolution, the rest ordinary
        sigma = 1.0
    preds = preds.astype(np.float32)
    loss, grad = loss_and_grad(F.lambdarank_loss, preds, dev(labels), sigma=sigma, lens=dev(ln))
    lq, g = CO.lambdarank(preds, labels, sigma, lens=ln)
    G.assert_close(loss, lq.astype(np.float64).sum(), f"lambdarank loss {case}")
    G.assert_close(grad, g, f"lambdarank grad {case}")


@pytest.mark.parametrize("L", [128, 256, 384, 512])
def test_ring_kernel_equal_label_slot_skipping(F, L):
    """The ring kernel skips the (own slot, travelling slot) blocks among the trailing 64-document slots whose documents all carry ONE label
    (r3; r6: lists of up to 512 documents, Z rounded down to an even slot count there): every head length — from one relevant document to
    none irrelevant — must give the oracle's loss and gradients, with and without padding."""
    from oracle import c_oracle as CO
    rng = np.random.default_rng(L)
    heads = sorted(set([1, 2, 63, 64, 65, 127, 128, 129, L // 2, L - 65, L - 64, L - 1, L]) & set(range(1, L + 1)))
    B = len(heads)
    labels = np.zeros((B, L), np.float32)
    for b, h in enumerate(heads):
        labels[b, :h] = np.sort(rng.integers(1, 5, size=h))[::-1]
    preds = rng.standard_normal((B, L)).astype(np.float32)
    for ln in (None, np.array([max(1, L - 7 * (b % 5)) for b in range(B)], np.int32)):
        loss, grad = loss_and_grad(F.lambdarank_loss, preds, dev(labels), sigma=1.0, lens=None if ln is None else dev(ln))
        lq, g = CO.lambdarank(preds, labels, 1.0, lens=ln)
        G.assert_close(loss, lq.astype(np.float64).sum(), "lambdarank loss")
        G.assert_close(grad, g, "lambdarank grad")
    # one grade everywhere (all slots pure): loss and gradients exactly 0
    same = np.full((3, L), 2.0, np.float32)
    loss, grad = loss_and_grad(F.lambdarank_loss, preds[:3], dev(same), sigma=1.0)
    assert float(loss) == 0.0 and not np.any(grad)


@pytest.mark.parametrize("B,L", [(5, 7), (16, 128), (9, 256), (4, 512), (2, 1030)])
@pytest.mark.parametrize("use_lens", [False, True])
def test_oracle_lambdaloss_approx(F, B, L, use_lens):
    from oracle import c_oracle as CO
    for presort in (True, False):
        preds, labels, ln = synth(2000 + L, B, L, lens=use_lens, presort=presort)
        lens_t = None if ln is None else dev(ln)
        for lt, code in (("NDCG_Loss2", 1), ("NDCG_Loss2++", 2)):
            for k in (5, L):
                loss, grad = loss_and_grad(F.lambdaloss_loss, preds, dev(labels), k=k, sigma=1.0, mu=5.0, loss_type=lt,
                                           presort=presort, lens=lens_t)
                lq, g = CO.lambdaloss(preds, labels, k, 1.0, 5.0, code, presort, lens=ln)
                G.assert_close(loss, lq.astype(np.float64).sum(), f"lambdaloss {lt} k={k}")
                G.assert_close(grad, g, f"lambdaloss {lt} k={k} grad")
        for couple in (True, False):
            loss, grad = loss_and_grad(F.approxndcg_loss, preds, dev(labels), alpha=10.0, presort=presort,
                                       couple_batch=couple, lens=lens_t)
            ol, dcg, inv, g = CO.approxndcg(preds, labels, 10.0, presort, couple, lens=ln)
            G.assert_close(loss, ol, f"approxndcg couple={couple}")
            G.assert_close(grad, g, f"approxndcg couple={couple} grad")


@pytest.mark.parametrize("B,L", [(37, 8), (64, 64), (33, 128), (40, 256), (9, 512), (5, 1024)])
@pytest.mark.parametrize("use_lens", [False, True])
def test_lambdaloss_small_cutoff_kernel(F, B, L, use_lens):
    """r5 `lambdaloss_topk_kernel` (presorted labels, k <= 11, L % 4 == 0): wavefront arg-max selection + one pair per lane, against the C
    oracle for every cut-off 1..11, both loss types and tied scores among the best documents (index tie-break)."""
    from oracle import c_oracle as CO
    preds, labels, ln = synth(4000 + L, B, L, lens=use_lens, presort=True)
    preds[::3, : min(L, 6)] = np.float32(0.25)                       # ties among the best-scored documents of every third query
    preds[1::3, 0] = np.float32(9.0)
    lens_t = None if ln is None else dev(ln)
    for lt, code in (("NDCG_Loss2", 1), ("NDCG_Loss2++", 2)):
        for k in (1, 2, 3, 5, 8, 11):
            loss, grad = loss_and_grad(F.lambdaloss_loss, preds, dev(labels), k=k, sigma=1.0, mu=5.0, loss_type=lt, presort=True, lens=lens_t)
            lq, g = CO.lambdaloss(preds, labels, k, 1.0, 5.0, code, True, lens=ln)
            G.assert_close(loss, lq.astype(np.float64).sum(), f"lambdaloss top-k {lt} k={k}")
            G.assert_close(grad, g, f"lambdaloss top-k {lt} k={k} grad")
    # sigma != 1 and the saturated / clamped branches through the same kernel
    loss, grad = loss_and_grad(F.lambdaloss_loss, preds * 30.0, dev(labels), k=5, sigma=2.5, mu=5.0, loss_type="NDCG_Loss2", presort=True, lens=lens_t)
    lq, g = CO.lambdaloss((preds * 30.0).astype(np.float32), labels, 5, 2.5, 5.0, 1, True, lens=ln)
    G.assert_close(loss, lq.astype(np.float64).sum(), "lambdaloss top-k saturated")
    G.assert_close(grad, g, "lambdaloss top-k saturated grad")


# (L = 64 / 128 / 256 / 512 / 700: the register kernels' lane groupings G = 16 / 32 / 64 and 1 / 2 / 4 float4 per lane; 3, 30 (L % 4 != 0)
# and 4096 take the LDS kernels; `scale` = 10 / 12 spreads the scores past the range of 24 behind which ListMLE switches to its libm form; far beyond that the reference
# itself returns log(0) = -inf)
@pytest.mark.parametrize("B,L,scale", [(5, 3, 1.0), (7, 30, 1.0), (40, 64, 1.0), (21, 128, 1.0), (33, 256, 1.0), (9, 512, 1.0), (8, 700, 1.0),
                                       (3, 4096, 1.0), (6, 256, 12.0), (5, 64, 10.0)])
@pytest.mark.parametrize("use_lens", [False, True])
def test_oracle_listwise(F, B, L, scale, use_lens):
    from oracle import c_oracle as CO
    preds, labels, ln = synth(3000 + L, B, L, lens=use_lens)
    preds = (preds * np.float32(scale)).astype(np.float32)
    lens_t = None if ln is None else dev(ln)
    loss, grad = loss_and_grad(F.listnet_loss, preds, dev(labels), lens=lens_t)
    lq, g = CO.listnet(preds, labels, lens=ln)
    G.assert_close(loss, lq.astype(np.float64).sum(), "listnet loss")
    G.assert_close(grad, g, "listnet grad")
    # ListMLE with the DEVICE tie shuffle: check the permutation is a valid tie-respecting order, then parity
    perm = F.shuffle_ties_order(dev(labels), seed=11, lens=lens_t)
    pn = perm.cpu().numpy()
    for b in range(B):
        n = L if ln is None else ln[b]
        assert sorted(pn[b, :n].tolist()) == list(range(n))
        assert np.all(np.diff(labels[b, pn[b, :n]]) <= 0)
    loss, grad = loss_and_grad(F.listmle_loss, preds, perm, lens=lens_t)
    lq, g = CO.listmle(preds, pn, lens=ln)
    G.assert_close(loss, lq.astype(np.float64).sum(), "listmle loss")
    G.assert_close(grad, g, "listmle grad")


def test_tie_shuffle_is_uniform_enough(F):
    """Device RNG path (not the torch.randperm stream): every position inside a tie group is about equally likely."""
    L, B = 8, 4096
    labels = np.tile(np.array([[2, 1, 1, 1, 1, 0, 0, 0]], np.float32), (B, 1))
    perm = F.shuffle_ties_order(dev(labels), seed=5).cpu().numpy()
    assert np.all(perm[:, 0] == 0)
    counts = np.stack([(perm[:, 1:5] == d).sum(axis=0) for d in range(1, 5)])   # doc d at slot s
    assert np.all(np.abs(counts / B - 0.25) < 0.03)
    perm2 = F.shuffle_ties_order(dev(labels), seed=6).cpu().numpy()
    assert not np.array_equal(perm, perm2)


@pytest.mark.parametrize("L", [256, 1024])
def test_tie_shuffle_has_no_index_order_bias_in_long_tie_groups(F, L):
    """two documents of one grade that draw the same random field were ordered by INDEX (r5 left 16 field bits at 513..1024
    documents: 6-8 such pairs per query in a tie group of 1000).  Such pairs sit next to each other in ascending index order, so they show
    as an excess of ASCENTS (perm[k] < perm[k+1]) inside the group: a uniform permutation of m items has (m - 1) / 2 ascents with variance
    (m + 1) / 12.  The kernel now re-draws a list's fields until none collide; the mean ascent count over 512 lists must sit within 4
    standard errors of (m - 1) / 2 (the 16-bit form was 9 standard errors off at L = 1024)."""
    B, m = 512, L - 24
    labels = np.zeros((B, L), np.float32)
    labels[:, :24] = np.sort(np.random.default_rng(L).integers(1, 5, size=(B, 24)), axis=1)[:, ::-1]
    perm = F.shuffle_ties_order(dev(labels), seed=77).cpu().numpy()
    for b in range(0, B, 37):
        assert sorted(perm[b].tolist()) == list(range(L)) and np.all(np.diff(labels[b, perm[b]]) <= 0)
    tail = perm[:, 24:]                                            # the grade-0 group: documents 24 .. L-1 in random order
    assert np.all(tail >= 24)
    ascents = (np.diff(tail, axis=1) > 0).sum(axis=1).astype(np.float64)
    stderr = np.sqrt((m + 1) / 12.0 / B)
    assert abs(ascents.mean() - (m - 1) / 2.0) < 4.0 * stderr, (ascents.mean(), (m - 1) / 2.0, stderr)
    # first-position uniformity over the group (chi-square-free: 16 equal bins of the index range, 512 draws)
    bins = np.bincount((tail[:, 0] - 24) * 16 // m, minlength=16)
    assert bins.min() >= 10 and bins.max() <= 60, bins


@pytest.mark.parametrize("B,L", [(6, 5), (32, 128), (9, 333), (4, 1024), (2, 4096)])
@pytest.mark.parametrize("use_lens", [False, True])
def test_oracle_metrics_and_sort(F, B, L, use_lens):
    from oracle import c_oracle as CO
    ks = [1, 3, 5, 10, 20, 50]
    for presort in (True, False):
        preds, labels, ln = synth(4000 + L, B, L, lens=use_lens, presort=presort)
        lens_t = None if ln is None else dev(ln)
        out = F.metrics_at_ks(dev(preds), dev(labels), ks, presort=presort, lens=lens_t)
        ref = CO.metrics_at_ks(preds, labels, ks, presort, lens=ln)
        for m in ("ndcg", "nerr", "ap", "p"):
            G.assert_close(out[m].cpu().numpy(), ref[m], m)
    vals, idx = F.sort_desc(dev(preds), lens=lens_t)
    rv, ri = CO.sort_desc(preds, lens=ln)
    assert np.array_equal(idx.cpu().numpy(), ri)
    assert np.array_equal(vals.cpu().numpy(), rv)
    if not use_lens:   # bit-exact against torch.sort itself on tie-free rows (torch.sort is not stable)
        tv, ti = torch.sort(torch.from_numpy(preds), dim=1, descending=True)
        tie_free = np.array([len(np.unique(row)) == L for row in preds])
        assert np.array_equal(idx.cpu().numpy()[tie_free], ti.numpy()[tie_free])
        assert np.array_equal(vals.cpu().numpy(), tv.numpy())


# ------------------------------------------------------------------------------------------ properties at BASELINE.json's full sizes
@pytest.mark.parametrize("L", [128, 256, 512])
def test_properties_full_size(F, L):
    B = 1024
    preds, labels, _ = synth(77, B, L)
    p, y = dev(preds), dev(labels)
    # (1) run-to-run bit stability (no cross-wave float atomics)
    l1, g1 = loss_and_grad(F.lambdarank_loss, preds, y)
    l2, g2 = loss_and_grad(F.lambdarank_loss, preds, y)
    assert l1 == l2 and np.array_equal(g1, g2)
    # (2) every in-scope loss is invariant to a per-query score shift => gradients sum to zero per query
    for fn, kw in ((F.lambdarank_loss, {}), (F.ranknet_loss, {}), (F.approxndcg_loss, {}), (F.listnet_loss, {}),
                   (F.lambdaloss_loss, dict(k=L))):
        _, g = loss_and_grad(fn, preds, y, **kw)
        scale = max(1.0, np.abs(g).max())
        assert np.abs(g.sum(axis=1)).max() <= 2e-4 * scale, fn.__name__
    # (3) queries are independent: a batch equals the concatenation of its halves (sum of losses, same grads)
    la, ga = loss_and_grad(F.lambdarank_loss, preds[: B // 2], dev(labels[: B // 2]))
    lb, gb = loss_and_grad(F.lambdarank_loss, preds[B // 2:], dev(labels[B // 2:]))
    assert np.array_equal(np.concatenate([ga, gb]), g1)
    assert abs((la + lb) - l1) <= 1e-5 * abs(l1)
    # (4) the sort kernel returns a permutation, values non-increasing, idempotent on its own output
    vals, idx = F.sort_desc(p)
    v = vals.cpu().numpy()
    assert np.all(np.diff(v, axis=1) <= 0)
    assert np.array_equal(np.sort(idx.cpu().numpy(), axis=1), np.tile(np.arange(L), (B, 1)))
    v2, i2 = F.sort_desc(vals)
    assert torch.equal(v2, vals) and np.array_equal(i2.cpu().numpy(), np.tile(np.arange(L), (B, 1)))
    # (5) a perfect ranker has nDCG = 1 at every cut-off; padding a batch does not change real rows
    out = F.metrics_at_ks(dev(-np.tile(np.arange(L, dtype=np.float32), (B, 1))), y, [1, 5, 10, 50], presort=True)
    assert torch.allclose(out["ndcg"], torch.ones_like(out["ndcg"]), atol=1e-6)


def test_padded_batch_equals_per_length_batches(F):
    """The reference only ever batches equal-length lists (data_utils.py:683-742): a padded batch must equal them."""
    preds, labels, ln = synth(99, 24, 96, lens=True)
    lens_t = dev(ln)
    _, g = loss_and_grad(F.lambdarank_loss, preds, dev(labels), lens=lens_t)
    lq_total = 0.0
    for b in range(24):
        n = int(ln[b])
        lb, gb = loss_and_grad(F.lambdarank_loss, preds[b:b + 1, :n].copy(), dev(labels[b:b + 1, :n].copy()))
        assert np.allclose(g[b, :n], gb[0], rtol=1e-6, atol=1e-7)
        assert np.all(g[b, n:] == 0)
        lq_total += float(lb)
    l_all, _ = loss_and_grad(F.lambdarank_loss, preds, dev(labels), lens=lens_t)
    assert abs(l_all - lq_total) <= 1e-5 * max(1.0, abs(lq_total))


def test_edge_cases_do_not_hang(F):
    # a query without any relevant document: LambdaRank / ApproxNDCG -> NaN like the reference, LambdaLoss -> 0
    preds = np.random.default_rng(0).standard_normal((2, 16)).astype(np.float32)
    labels = np.zeros((2, 16), np.float32)
    labels[1, 0] = 1
    l, g = loss_and_grad(F.lambdarank_loss, preds, dev(labels))
    assert np.isnan(l) and np.all(np.isfinite(g[1]))
    l, g = loss_and_grad(F.lambdaloss_loss, preds[:1], dev(labels[:1]), k=5)
    assert l == 0 and np.all(g == 0)
    # empty batch and zero-length queries
    l, g = loss_and_grad(F.lambdarank_loss, np.zeros((0, 8), np.float32), dev(np.zeros((0, 8), np.float32)))
    assert l == 0 and g.shape == (0, 8)
    lens = dev(np.array([0, 16], np.int32))
    l, g = loss_and_grad(F.listnet_loss, preds, dev(labels), lens=lens)
    assert np.isfinite(l) and np.all(g[0] == 0)
    # NaN scores must not hang or crash
    bad = preds.copy()
    bad[0, 3] = np.nan
    loss_and_grad(F.lambdarank_loss, bad, dev(labels + 1))
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------ raw C ABI (no wrapper logic)
def test_c_abi_direct_calls_and_errors():
    import ctypes as C
    from ptranking_amd import _lib
    lib = _lib.load()
    x = torch.arange(1000, dtype=torch.float32, device="cuda")
    out = torch.zeros(1, device="cuda")
    rc = lib.ptr_sum_f32(C.c_void_p(x.data_ptr()), 1000, C.c_float(0.5), C.c_void_p(out.data_ptr()),
                         C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    assert out.item() == 0.5 * 999 * 1000 / 2
    # errors: list too long -> PTR_ERR_UNSUPPORTED with a message; NULL pointer -> PTR_ERR_INVALID_ARG
    rc = lib.ptr_lambdarank_fwd_bwd(C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), None, 1, 5000, C.c_float(1.0), None,
                                    C.c_void_p(out.data_ptr()), C.c_void_p(x.data_ptr()), None)
    assert rc == 1002 and b"PTR_MAX_LIST_LEN" in lib.ptr_last_error()
    rc = lib.ptr_lambdarank_fwd_bwd(None, None, None, 1, 8, C.c_float(1.0), None, None, None, None)
    assert rc == 1001
    with pytest.raises(RuntimeError, match="sigma"):
        from ptranking_amd import functional
        functional.lambdarank_loss(torch.zeros(1, 4, device="cuda"), torch.zeros(1, 4, device="cuda"), sigma=-1.0)


@pytest.mark.parametrize("L", [100, 128, 256, 333, 512, 1024])
def test_sort_and_metrics_with_scores_that_collide_in_the_packed_keys(F, L):
    """r5: lists of 65 .. 1024 documents are ordered by ONE register sort of packed keys (the score's top 32 - log2(64 DPT) bits | the
    index).  Scores that agree in those bits reach the repair round (isolated pairs, inside a lane and across a lane boundary) or the
    fallback (runs of three and more); exact ties, signed zeros and infinities must keep the (score descending, index ascending) order of
    torch.sort(stable=True).  Bit-exact values and indices; the metric kernel on the same rows against the C oracle."""
    from oracle import c_oracle as CO
    rng = np.random.default_rng(L)
    B = 12
    P = np.empty((B, L), np.float32)
    base = np.linspace(3.0, -3.0, L).astype(np.float32)
    P[0] = base[rng.permutation(L)]
    for a, b in ((10, 11), (63, 64), (3, 4), (L - 2, L - 1), (40, 90)):      # pairs one ulp apart, the larger score at the larger index
        x = np.float32(0.5 + 0.01 * a)
        P[0, a], P[0, b] = x, np.nextafter(x, np.float32(4.0))
    P[1] = (np.float32(1.0) + rng.permutation(L).astype(np.float32) * np.float32(2.0 ** -23))      # one long run in the truncated keys
    P[2] = rng.choice(np.array([0.0, -0.0, 1.0, -1.0], np.float32), size=L)
    P[3] = np.float32(0.75)
    P[4] = base[rng.permutation(L)]
    P[4, [1, 5, L - 1]] = np.inf
    P[4, [0, 7, L - 3]] = -np.inf
    P[5] = np.round(rng.standard_normal(L) * 8).astype(np.float32) / np.float32(8)                 # many exact ties
    P[6] = (np.float32(-2.0) - rng.permutation(L).astype(np.float32) * np.float32(2.0 ** -22))     # negative scores, runs
    P[7:] = rng.standard_normal((B - 7, L)).astype(np.float32) * np.float32(1e-3) + np.float32(0.5)   # dense: several colliding pairs
    vals, idx = F.sort_desc(dev(P))
    tv, ti = torch.sort(torch.from_numpy(P), dim=1, descending=True, stable=True)
    assert np.array_equal(idx.cpu().numpy(), ti.numpy())
    assert np.array_equal(vals.cpu().numpy().view(np.int32), tv.numpy().view(np.int32))
    ln = np.array([L, L - 1, L // 2 + 1, 70, L, 66, L - 3, L, 65, L, L - 2, L], np.int32)
    vals, idx = F.sort_desc(dev(P), lens=dev(ln))
    rv, ri = CO.sort_desc(P, lens=ln)
    assert np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(vals.cpu().numpy().view(np.int32), rv.view(np.int32))
    labels = rng.integers(0, 5, size=(B, L)).astype(np.float32)
    ks = [1, 3, 5, 10, 20, 50, 100]
    for lens in (None, ln):
        out = F.metrics_at_ks(dev(P), dev(labels), ks, presort=False, lens=None if lens is None else dev(lens))
        ref = CO.metrics_at_ks(P, labels, ks, False, lens=lens)
        for m in ("ndcg", "nerr", "ap", "p"):
            G.assert_close(out[m].cpu().numpy(), ref[m], m)


def test_tie_shuffle_key_paths(F):
    """r5: integer grades up to 63 take the packed-key register sort (grade | random field | index in 32 bits: grades >= 32 set the key's top
    bit); other labels take the exact comparison.  Every path returns a permutation that orders the labels descending, and different seeds
    give different orders inside the tie groups."""
    rng = np.random.default_rng(5)
    for L in (64, 96, 256, 500, 1024):
        for hi, frac in ((5, False), (64, False), (200, False), (5, True)):
            y = rng.integers(0, hi, size=(9, L)).astype(np.float32)
            if frac:
                y = y + np.float32(0.5)
            ln = rng.integers(1, L + 1, size=9).astype(np.int32)
            for lens in (None, ln):
                p1 = F.shuffle_ties_order(dev(y), seed=3, lens=None if lens is None else dev(lens)).cpu().numpy()
                p2 = F.shuffle_ties_order(dev(y), seed=4, lens=None if lens is None else dev(lens)).cpu().numpy()
                for b in range(9):
                    n = L if lens is None else int(lens[b])
                    assert sorted(p1[b, :n].tolist()) == list(range(n)) and np.array_equal(p1[b, n:], np.arange(n, L))
                    assert np.all(np.diff(y[b, p1[b, :n]]) <= 0)
                if L >= 256:
                    assert not np.array_equal(p1, p2)


@pytest.mark.parametrize("L", [64, 256, 1024])
def test_sort_family_on_rows_that_are_not_16_byte_aligned(F, L):
    """The one-wavefront sort / metric / tie-shuffle paths use 16-byte loads and stores only when the tensors' bases allow it: a view that
    starts one float into its buffer gives the same results as an aligned copy."""
    torch.manual_seed(L)
    B = 7
    buf_p = torch.randn(B * L + 1, device="cuda"); buf_y = torch.randint(0, 5, (B * L + 1,), device="cuda").float()
    p_un, y_un = buf_p[1:].view(B, L), buf_y[1:].view(B, L)
    assert p_un.data_ptr() % 16 != 0 and y_un.data_ptr() % 16 != 0
    p_al, y_al = p_un.clone(), y_un.clone()
    ks = [1, 5, 10, 50]
    for a, b in zip(F.sort_desc(p_un), F.sort_desc(p_al)):
        assert torch.equal(a, b)
    ma, mb = F.metrics_at_ks(p_un, y_un, ks, presort=False), F.metrics_at_ks(p_al, y_al, ks, presort=False)
    for m in ("ndcg", "nerr", "ap", "p"):
        assert torch.equal(ma[m], mb[m]), m
    assert torch.equal(F.shuffle_ties_order(y_un, seed=3), F.shuffle_ties_order(y_al, seed=3))
