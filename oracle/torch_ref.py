"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement, in torch fp32 ops + autograd, of the op sequences wildltr/ptranking executes
for the ltr_adhoc loss / metric hot path.  The reference's arithmetic lives in PyTorch ATen CPU
kernels, so the closest possible CPU "port" is the same ATen ops in the same
order: sigmoid rounding, binary_cross_entropy's -100 log clamp and 1e-12 backward epsilon,
pow-with-tensor-exponent, cumsum order etc. are inherited from torch rather than re-derived.

Parity pin: tests/test_oracle_golden.py checks every function here against tests/golden/*.npz,
which tests/golden/make_golden.py produced by running the reference itself (incl. the five
known-answer vectors of the reference's testing/metric/testing_metric.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Each function cites the reference file:line it follows.
"""
import math

import torch
import torch.nn.functional as F

EPS_LAMBDALOSS = 1e-8  # ptranking/ltr_global.py:10


# --------------------------------------------------------------------------- helpers
def _gain(labels):
    """2^l - 1 (MultiLabel gains), ptranking/metric/adhoc/adhoc_metric.py:208-209."""
    return torch.pow(2.0, labels) - 1.0


def dcg_full(rankings):
    """DCG over the whole list -> [B,1]; ptranking/metric/adhoc/adhoc_metric.py:197-217 (cutoff=None)."""
    L = rankings.size(1)
    disc = torch.log2(torch.arange(L, dtype=torch.float32) + 2.0)
    return torch.sum(_gain(rankings) / disc, dim=1, keepdim=True)


def _pairwise(preds, labels, sigma):
    """s_ij, p_ij, target p̄_ij; ptranking/ltr_adhoc/util/lambda_utils.py:5-23."""
    s_ij = preds.unsqueeze(2) - preds.unsqueeze(1)
    p_ij = torch.sigmoid(sigma * s_ij)
    S_ij = torch.clamp(labels.unsqueeze(2) - labels.unsqueeze(1), min=-1.0, max=1.0)
    return p_ij, 0.5 * (1.0 + S_ij)


def _delta_ndcg(ideal, ranked):
    """|ΔnDCG| of swapping two predicted positions; ptranking/metric/metric_utils.py:19-45."""
    G = _gain(ranked) / dcg_full(ideal)
    dG = G.unsqueeze(2) - G.unsqueeze(1)
    D = 1.0 / torch.log2(torch.arange(ranked.size(1), dtype=torch.float32) + 2.0)
    dD = D.view(1, -1, 1) - D.view(1, 1, -1)
    return torch.abs(dG) * torch.abs(dD)


# --------------------------------------------------------------------------- losses (autograd)
def ranknet_loss(preds, labels, sigma=1.0):
    """ptranking/ltr_adhoc/pairwise/ranknet.py:32-36 — unweighted BCE over the upper triangle, input order."""
    p, t = _pairwise(preds, labels, sigma)
    l = F.binary_cross_entropy(torch.triu(p, 1), torch.triu(t, 1), reduction="none")
    return l.sum()


def lambdarank_loss(preds, labels, sigma=1.0):
    """ptranking/ltr_adhoc/listwise/lambdarank.py:39-56 — labels must be in ideal (descending) order."""
    sp, idx = torch.sort(preds, dim=1, descending=True)
    ranked = torch.gather(labels, 1, idx)
    p, t = _pairwise(sp, ranked, sigma)
    w = _delta_ndcg(labels, ranked)
    l = F.binary_cross_entropy(torch.triu(p, 1), torch.triu(t, 1), weight=torch.triu(w, 1), reduction="none")
    return l.sum()


def lambdaloss_loss(preds, labels, k=5, sigma=1.0, mu=5.0, loss_type=1, presort=True):
    """ptranking/ltr_adhoc/listwise/lambdaloss.py:83-132.  loss_type 0 = NDCG_Loss1 (:33-34), 1 = NDCG_Loss2 (:36-45),
    2 = NDCG_Loss2++ (:47-58).
    The inverted discount table (`pow(1/log2(r+2), -1)`) is reproduced as the reference computes it (SURVEY §7 iii)."""
    if presort:
        tp, ideal = preds, labels
    else:
        ideal, li = torch.sort(labels, dim=1, descending=True)
        tp = torch.gather(preds, 1, li)
    sp, idx = torch.sort(tp, dim=1, descending=True)
    ranked = torch.gather(ideal, 1, idx)
    L = tp.size(1)
    disc = 1.0 / torch.log2(torch.arange(L, dtype=torch.float32) + 2.0)
    G = _gain(ranked) / dcg_full(ideal)
    r = torch.arange(L)
    dist = (r[:, None] - r[None, :]).abs()
    inv = torch.pow(disc, -1.0)
    delta = torch.abs(inv[dist - 1] - inv[dist])       # index -1 wraps on the diagonal, then zeroed (:41-42)
    delta.diagonal().zero_()
    absG = torch.abs(G[:, :, None] - G[:, None, :])
    if loss_type == 0:
        # NDCG_Loss1 (:33-34): weights [B, L] = G / discounts.  The reference broadcasts them against [B, L, L] (only legal at
        # B == 1, where they index the COLUMN j); restated per query, w[b, i, j] = G[b, j] / disc[j].
        w = (G / disc[None])[:, None, :].expand(-1, L, -1)
    elif loss_type == 1:
        w = delta[None] * absG
    elif loss_type == 2:
        rho = torch.abs(inv[:, None] - inv[None, :])
        w = (rho + mu * delta)[None] * absG
    else:
        raise NotImplementedError(loss_type)
    d = (sp.unsqueeze(2) - sp.unsqueeze(1)).clamp(min=-1e8, max=1e8)
    d = torch.where(torch.isnan(d), torch.zeros_like(d), d)
    wp = (torch.sigmoid(sigma * d).clamp(min=EPS_LAMBDALOSS) ** w).clamp(min=EPS_LAMBDALOSS)
    lw = torch.log2(wp)
    trunc = torch.zeros(L, L, dtype=torch.bool)
    trunc[:k, :k] = True
    if loss_type == 0:
        return -torch.sum(lw[trunc[None].expand_as(lw)])      # :130, no label mask
    mask = ((ranked.unsqueeze(2) - ranked.unsqueeze(1)) > 0) & trunc
    return -torch.sum(lw[mask])


def _robust_sigmoid(x_in, sigma):
    """Forward of Robust_Sigmoid (ptranking/base/utils.py:57-81); autograd through these ops gives the same
    derivative sigma*y*(1-y) the reference stores by hand, because the inactive where-branch gets zero gradient."""
    x = x_in if sigma == 1.0 else sigma * x_in
    pos = torch.where(x_in > 0, 1.0 / (1.0 + torch.exp(-x)), torch.full_like(x, 0.5))
    ex = torch.exp(x)
    return torch.where(x_in < 0, ex / (1.0 + ex), pos)


class _RobustSigmoid(torch.autograd.Function):
    """Same hand-written backward as the reference (saved grad tensor), ptranking/base/utils.py:78-93."""

    @staticmethod
    def forward(ctx, inp, sigma):
        with torch.no_grad():
            y = _robust_sigmoid(inp, sigma)
            g = y * (1.0 - y) if sigma == 1.0 else sigma * y * (1.0 - y)
        ctx.save_for_backward(g)
        return y

    @staticmethod
    def backward(ctx, go):
        return go * ctx.saved_tensors[0], None


def approxndcg_loss(preds, labels, alpha=10.0, presort=True, couple_batch=True):
    """ptranking/ltr_adhoc/listwise/approxNDCG.py:19-27,45-62.  couple_batch=True reproduces the reference's
    [B]/[B,1] broadcast (loss = -(Σ_b DCG_b)(Σ_a 1/IDCG_a), SURVEY §7 vi); False is the per-query form."""
    if presort:
        tp, ideal = preds, labels
    else:
        ideal, li = torch.sort(labels, dim=1, descending=True)
        tp = torch.gather(preds, 1, li)
    diffs = tp.unsqueeze(2) - tp.unsqueeze(1)
    ind = _RobustSigmoid.apply(diffs.transpose(1, 2), alpha)
    hat_pi = ind.sum(dim=2) + 0.5
    idcg = dcg_full(ideal)                                      # [B,1]
    dcg = torch.sum(_gain(ideal) / torch.log2(hat_pi + 1.0), dim=1)  # [B]
    if couple_batch:
        return -torch.sum(dcg / idcg)                            # [B]/[B,1] -> [B,B]
    return -torch.sum(dcg / idcg.view(-1))


def listnet_loss(preds, labels):
    """ptranking/ltr_adhoc/listwise/listnet.py:39."""
    return torch.sum(-torch.sum(F.softmax(labels, dim=1) * F.log_softmax(preds, dim=1), dim=1))


def gumbel_from_uniform(unif):
    """ptranking/ltr_adhoc/listwise/st_listnet.py:18,43 (EPS = 1e-20)."""
    return -torch.log(-torch.log(unif + 1e-20) + 1e-20)


def softrank_loss(preds, labels, delta=2.0, top_k=None):
    """ptranking/ltr_adhoc/listwise/softrank.py:47-69 (labels in ideal order; the reference asserts presort)."""
    dlt = torch.tensor([delta], dtype=torch.float32)
    sub = preds.unsqueeze(2) - preds.unsqueeze(1)
    var = 2 * dlt ** 2
    phi0 = 0.5 * torch.erfc(sub / torch.sqrt(2 * var))
    phi0 = torch.triu(phi0, diagonal=1) + torch.tril(phi0, diagonal=-1)
    ranks = torch.sum(phi0, dim=2) + 1.0
    dists = 1.0 / torch.log2(ranks + 1.0)
    gains = _gain(labels)
    idcg = dcg_full(labels)
    k = labels.size(1) if top_k is None else min(top_k, labels.size(1))
    return -torch.sum(torch.sum(dists[:, :k] * gains[:, :k] / idcg, dim=1))


def stlistnet_loss(preds, labels, unif, temperature=1.0):
    """ptranking/ltr_adhoc/listwise/st_listnet.py:41-49 with the uniform draws supplied by the caller."""
    z = (preds + gumbel_from_uniform(unif)) / temperature
    return torch.sum(-torch.sum(F.softmax(labels, dim=1) * F.log_softmax(z, dim=1), dim=1))


def rankmse_loss(preds, labels):
    """ptranking/ltr_adhoc/pointwise/rank_mse.py:13-22."""
    return torch.mean(torch.sum(F.mse_loss(preds, labels, reduction="none"), dim=1))


def rankcosine_loss(preds, labels):
    """ptranking/ltr_adhoc/listwise/rank_cosine.py:15,32."""
    return torch.sum((1.0 - torch.nn.CosineSimilarity(dim=1)(preds, labels)) / 0.5)


def mdprank_loss(preds, labels, perm, top_k=10, gamma=1.0):
    """ptranking/ltr_adhoc/listwise/mdprank.py:46-75 on a given sampled ranking `perm` (int64 [B, L]); `preds` = action scores by
    original document index.  Restated per query (the reference asserts batch size 1)."""
    B, L = preds.shape
    top = L if not top_k else min(int(top_k), L)
    u = torch.gather(preds, 1, perm)
    stds = torch.gather(labels, 1, perm)
    gains = torch.pow(2.0, stds) - 1.0
    disc = torch.log2(2.0 + torch.arange(top, dtype=torch.float32)).view(1, -1)
    rewards = gains[:, :top] / disc
    G = torch.flip(torch.cumsum(torch.flip(rewards, dims=[1]), dim=1), dims=[1])
    if gamma != 1.0:
        G = G * torch.cumprod(torch.ones(top).view(1, -1) * gamma, dim=1)
    m, _ = torch.max(u, dim=1, keepdim=True)
    y = torch.exp(u - m)
    lcse = torch.log(torch.flip(torch.cumsum(torch.flip(y, dims=[1]), dim=1), dims=[1])) + m
    return torch.sum(torch.sum((lcse[:, :top] - u[:, :top]) * G[:, :top], dim=1))


def arg_shuffle_ties(labels, generator=None):
    """Random tie-broken descending order; ptranking/ltr_adhoc/util/sampling_utils.py:13-28."""
    B, L = labels.shape
    rperm = torch.stack([torch.randperm(L, generator=generator) for _ in range(B)], dim=0)
    shuffled = torch.gather(labels, 1, rperm)
    desc = torch.argsort(shuffled, descending=True)
    return torch.gather(rperm, 1, desc)


def listmle_loss(preds, perm):
    """ptranking/ltr_adhoc/listwise/listmle.py:81-97 with the permutation supplied by the caller."""
    u = torch.gather(preds, 1, perm)
    m, _ = torch.max(u, dim=1, keepdim=True)
    y = torch.exp(u - m)
    tail = torch.flip(torch.cumsum(torch.flip(y, dims=[1]), dim=1), dims=[1])
    return torch.sum(torch.log(tail) + m - u)


def loss_and_grad(fn, preds, *args, **kw):
    """Evaluate `fn` on a leaf copy of preds -> (loss float32 0-d tensor, dL/dpreds)."""
    p = preds.detach().clone().requires_grad_(True)
    loss = fn(p, *args, **kw)
    loss.backward()
    return loss.detach(), p.grad.detach()


# --------------------------------------------------------------------------- metrics
def sort_desc(preds):
    """torch.sort(descending=True) as ptranking/base/ranker.py:50 calls it (values, int64 indices)."""
    return torch.sort(preds, dim=1, descending=True)


def _pad_ks(vals_at, ks, L, B):
    used = [k for k in ks if k <= L]
    out = torch.zeros(B, len(ks))
    if used:
        out[:, :len(used)] = vals_at(used)
    return out


def ndcg_at_ks(sys_sorted, ideal_sorted, ks, permutation_labels=False):
    """ptranking/metric/adhoc/adhoc_metric.py:219-260 (cumsum DCG, zero-fill cut-offs > L at the END of the row);
    LABEL_TYPE.Permutation: the label is the gain (:225-230)."""
    B, L = sys_sorted.shape
    gain = (lambda t: t) if permutation_labels else _gain

    def at(used):
        m = max(used)
        disc = torch.log2(torch.arange(m, dtype=torch.float32) + 2.0)
        s = torch.cumsum(gain(sys_sorted[:, :m]) / disc, dim=1)
        i = torch.cumsum(gain(ideal_sorted[:, :m]) / disc, dim=1)
        ix = torch.tensor(used) - 1
        return s[:, ix] / i[:, ix]
    return _pad_ks(at, ks, L, B)


def ap_at_ks(sys_sorted, ideal_sorted, ks):
    """ptranking/metric/adhoc/adhoc_metric.py:91-123; denominator = cumsum of GRADED ideal labels (bug-compatible)."""
    B, L = sys_sorted.shape

    def at(used):
        m = max(used)
        rel = torch.clamp(sys_sorted[:, :m], min=0, max=1)
        prec = torch.cumsum(rel, dim=1) / (torch.arange(m, dtype=torch.float32) + 1.0)
        num = torch.cumsum(prec * rel, dim=1)
        den = torch.cumsum(ideal_sorted, dim=1)[:, :m]
        ix = torch.tensor(used) - 1
        return (num / den)[:, ix]
    return _pad_ks(at, ks, L, B)


def precision_at_ks(sys_sorted, ks):
    """ptranking/metric/adhoc/adhoc_metric.py:36-62."""
    B, L = sys_sorted.shape

    def at(used):
        m = max(used)
        rel = torch.clamp(sys_sorted[:, :m], min=0, max=1)
        prec = torch.cumsum(rel, dim=1) / (torch.arange(m, dtype=torch.float32) + 1.0)
        return prec[:, torch.tensor(used) - 1]
    return _pad_ks(at, ks, L, B)


def _rankwise_err(rankings, max_label, m):
    """ptranking/metric/adhoc/adhoc_metric.py:127-150 (point=False)."""
    lab = rankings[:, :m]
    sat = (torch.pow(2.0, lab) - 1.0) / math.pow(2.0, float(max_label))
    unsat = torch.cumprod(1.0 - sat, dim=1)
    casc = torch.ones_like(lab)
    casc[:, 1:m] = unsat[:, 0:m - 1]
    return torch.cumsum((1.0 / (torch.arange(m, dtype=torch.float32) + 1.0)) * sat * casc, dim=1)


def nerr_at_ks(sys_sorted, ideal_sorted, ks, max_label=None):
    """ptranking/metric/adhoc/adhoc_metric.py:164-193; max_label defaults to the BATCH maximum (:174-175)."""
    B, L = sys_sorted.shape
    if max_label is None:
        max_label = float(torch.max(ideal_sorted))

    def at(used):
        m = max(used)
        ix = torch.tensor(used) - 1
        return (_rankwise_err(sys_sorted, max_label, m) / _rankwise_err(ideal_sorted, max_label, m))[:, ix]
    return _pad_ks(at, ks, L, B)


def evaluate_at_ks(preds, labels, ks, presort, max_label=None):
    """Evaluator prologue + four metrics for one batch, ptranking/base/ranker.py:220-243 -> dict of [B,len(ks)]."""
    _, idx = sort_desc(preds)
    sys_sorted = torch.gather(labels, 1, idx)
    ideal = labels if presort else torch.sort(labels, dim=1, descending=True)[0]
    return dict(ndcg=ndcg_at_ks(sys_sorted, ideal, ks), nerr=nerr_at_ks(sys_sorted, ideal, ks, max_label),
                ap=ap_at_ks(sys_sorted, ideal, ks), p=precision_at_ks(sys_sorted, ks), sort_idx=idx)


# --------------------------------------------------------------------------- whole train step (cpu_baseline leg)
def build_pointsf(num_features, h_dim=100, num_layers=3, dropout=0.1, seed=137):
    """Pointwise MLP scorer as ptranking/base/point_ranker.py:30-42 + base/utils.py:288-356 build it with
    AF='R', BN=False, apply_tl_af=False: (Dropout -> Linear(xavier_normal) -> ReLU) x num_layers -> Linear."""
    g = torch.Generator().manual_seed(seed)
    dims = [num_features] + [h_dim] * num_layers + [1]
    net = torch.nn.Sequential()
    for i in range(1, len(dims) - 1):
        net.add_module(f"dr_{i}", torch.nn.Dropout(dropout))
        lin = torch.nn.Linear(dims[i - 1], dims[i])
        torch.nn.init.xavier_normal_(lin.weight, generator=g)
        net.add_module(f"ff_{i + 1}", lin)
        net.add_module(f"act_{i + 1}", torch.nn.ReLU())
    last = torch.nn.Linear(dims[-2], dims[-1])
    torch.nn.init.xavier_normal_(last.weight, generator=g)
    net.add_module(f"ff_{len(dims)}", last)
    return net


def cpu_train_step(net, opt, X, Y, loss_fn, **loss_kw):
    """One reference-shaped train step on CPU: forward (ptranking/base/point_ranker.py:45-55) -> loss ->
    zero_grad/backward/step (e.g. lambdarank.py:58-60) -> .item() (ptranking/base/ranker.py:584)."""
    preds = net(X).view(-1, X.size(1))
    loss = loss_fn(preds, Y, **loss_kw)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss.item()


# ================================================================================ listwise scorer (listsf) restatement
def layer_norm_ref(x, a_2, b_2, eps=1e-6):
    """ptranking/base/list_ranker.py:170-174 — torch.std is the UNBIASED estimator, eps is added to the std."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return a_2 * (x - mean) / (std + eps) + b_2


def mhsa_core_ref(Q, K, V, n_heads, keep_mask=None, p_drop=0.0, lens=None):
    """ptranking/base/list_ranker.py:213-247: split heads, Q K^T / sqrt(d_h), softmax, dropout (as an explicit keep mask
    [B, H, L, L] scaled by 1/(1-p), what nn.Dropout does), times V, merge heads.  lens: padded keys are excluded (not in the
    reference, which has no padding)."""
    B, L, Fd = Q.shape
    dh = Fd // n_heads
    q = Q.view(B, L, n_heads, dh).permute(0, 2, 1, 3)
    k = K.view(B, L, n_heads, dh).permute(0, 2, 1, 3)
    v = V.view(B, L, n_heads, dh).permute(0, 2, 1, 3)
    scale = torch.sqrt(torch.tensor([dh], dtype=torch.float32))
    att = torch.matmul(q, k.permute(0, 1, 3, 2)) / scale
    if lens is not None:
        key_ok = torch.arange(L)[None, :] < torch.as_tensor(lens)[:, None].long()          # [B, L]
        att = att.masked_fill(~key_ok[:, None, None, :], float("-inf"))
    att = torch.softmax(att, dim=-1)
    if keep_mask is not None and p_drop > 0.0:
        att = att * keep_mask / (1.0 - p_drop)
    x = torch.matmul(att, v)
    return x.permute(0, 2, 1, 3).contiguous().view(B, L, Fd)


def mhsa_ref(x, sd, n_heads, keep_mask=None, p_drop=0.0, lens=None, prefix=""):
    """MultiheadAttention.forward (list_ranker.py:204-254) from a state_dict of tensors."""
    lin = lambda name, t: F.linear(t, sd[f"{prefix}{name}.weight"], sd[f"{prefix}{name}.bias"])  # noqa: E731
    core = mhsa_core_ref(lin("w_q", x), lin("w_k", x), lin("w_v", x), n_heads, keep_mask, p_drop, lens)
    return lin("fc", core)


def _ffnet_ref(x, sd, prefix, n_layers, act_last):
    """get_stacked_FFNet in eval mode with AF='R', BN=False (ptranking/base/utils.py:288-356): Linear+ReLU per hidden layer,
    last Linear followed by the activation only when apply_tl_af."""
    for i in range(2, n_layers + 3):                       # n_layers hidden Linear+ReLU (ff_2 ..), then the last Linear
        x = F.linear(x, sd[f"{prefix}ff_{i}.weight"], sd[f"{prefix}ff_{i}.bias"])
        if i < n_layers + 2 or act_last:
            x = F.relu(x)
    return x


def listsf_ref(x, sd, encoder_type, n_heads, encoder_layers, n_ff):
    """ListNeuralRanker.forward in eval mode (list_ranker.py:352-378 with :46-150) from a flat state_dict whose keys are
    '<part>/<module key>' for part in head_ffnns / encoder / tail_ffnns.  n_ff = len(ff_dims)."""
    head = {k.split("/", 1)[1]: v for k, v in sd.items() if k.startswith("head_ffnns/")}
    enc = {k.split("/", 1)[1]: v for k, v in sd.items() if k.startswith("encoder/")}
    tail = {k.split("/", 1)[1]: v for k, v in sd.items() if k.startswith("tail_ffnns/")}
    fc_map = _ffnet_ref(x, head, "", n_ff, True)            # head: apply_tl_af=True with TL_AF=AF (:312-313)

    def encoder(t):
        for l in range(encoder_layers):
            pre = f"layers.{l}."
            att = lambda u: mhsa_ref(u, enc, n_heads, prefix=pre + "mhsa.")  # noqa: E731
            if encoder_type == "AllRank":
                n0 = lambda u: layer_norm_ref(u, enc[pre + "sublayer_cont.0.norm.a_2"], enc[pre + "sublayer_cont.0.norm.b_2"])  # noqa: E731
                n1 = lambda u: layer_norm_ref(u, enc[pre + "sublayer_cont.1.norm.a_2"], enc[pre + "sublayer_cont.1.norm.b_2"])  # noqa: E731
                t = t + att(n0(t))
                h = F.relu(F.linear(n1(t), enc[pre + "fc.w1.weight"], enc[pre + "fc.w1.bias"]))
                t = t + F.linear(h, enc[pre + "fc.w2.weight"], enc[pre + "fc.w2.bias"])
            else:
                nrm = lambda u: layer_norm_ref(u, enc[pre + "sublayer_cont.norm.a_2"], enc[pre + "sublayer_cont.norm.b_2"])  # noqa: E731
                t = nrm(att(t)) if encoder_type == "DASALC" else nrm(t + att(t))
        if encoder_type == "AllRank":
            t = layer_norm_ref(t, enc["norm.a_2"], enc["norm.b_2"])
        return t

    if encoder_type == "AllRank":
        z = encoder(fc_map)
    elif encoder_type == "DASALC":
        z = (encoder(x) + 1.0) * fc_map
    elif encoder_type == "AttnDIN":
        z = encoder(fc_map) + x
    else:
        raise NotImplementedError(encoder_type)
    return _ffnet_ref(z, tail, "", n_ff, False).squeeze(2)
