"""r5: sort-family entry points (metrics / sort_desc / tie shuffle / LambdaLoss k=5) at 65 536 x L: time per call and a torch cross-check"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import functional as F
B = int(os.environ.get("B", 65536))
KS = [1, 3, 5, 10, 20, 50]
def tm(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
probs = torch.tensor([0.5147, 0.3250, 0.1339, 0.0183, 0.0081], device="cuda")
for L in (64, 256, 200, 1024):
    torch.manual_seed(L)
    Bq = B if L <= 256 else B // 4
    p = torch.randn(Bq, L, device="cuda")
    y = torch.multinomial(probs.expand(Bq, -1), L, replacement=True).float()
    ys = torch.sort(y, dim=1, descending=True)[0].contiguous()
    # checks
    vals, idx = F.sort_desc(p)
    tv, ti = torch.sort(p, dim=1, descending=True, stable=True)
    ok_sort = bool(torch.equal(vals, tv) and torch.equal(idx, ti))
    perm = F.shuffle_ties_order(ys, seed=3)
    ok_perm = bool(torch.equal(torch.sort(perm, dim=1)[0], torch.arange(L, device="cuda").expand(Bq, L)) and
                   bool((torch.gather(ys, 1, perm)[:, 1:] <= torch.gather(ys, 1, perm)[:, :-1]).all()))
    perm_u = F.shuffle_ties_order(y, seed=4)
    yg = torch.gather(y, 1, perm_u)
    ok_perm2 = bool((yg[:, 1:] <= yg[:, :-1]).all()) and bool(torch.equal(torch.sort(perm_u, dim=1)[0], torch.arange(L, device="cuda").expand(Bq, L)))
    # uniformity inside the first tie group of row 0's pattern: position of doc 0 among label-max docs should vary
    out = F.metrics_at_ks(p, ys, KS, presort=True)
    out2 = F.metrics_at_ks(p, y, KS, presort=False)
    # torch reference nDCG@k
    ysys = torch.gather(ys, 1, ti)
    disc = 1.0 / torch.log2(torch.arange(L, device="cuda").double() + 2.0)
    dcg = torch.cumsum((2.0 ** ysys.double() - 1) * disc, 1); idcg = torch.cumsum((2.0 ** ys.double() - 1) * disc, 1)
    kk = [k for k in KS if k <= L]
    ref = torch.stack([dcg[:, k - 1] / idcg[:, k - 1] for k in kk], 1)
    err = float((out["ndcg"][:, :len(kk)].double() - ref).abs().max())
    ysys2 = torch.gather(y, 1, ti); dcg2 = torch.cumsum((2.0 ** ysys2.double() - 1) * disc, 1)
    ref2 = torch.stack([dcg2[:, k - 1] / idcg[:, k - 1] for k in kk], 1)
    err2 = float((out2["ndcg"][:, :len(kk)].double() - ref2).abs().max())
    t_m = tm(lambda: F.metrics_at_ks(p, ys, KS, presort=True)); t_m1 = tm(lambda: F.metrics_at_ks(p, ys, KS, presort=True, which=("ndcg",)) if False else F.metrics_at_ks(p, y, KS, presort=False))
    t_s = tm(lambda: F.sort_desc(p)); t_h = tm(lambda: F.shuffle_ties_order(ys, seed=5))
    pl = p.detach().requires_grad_(True)
    t_l = tm(lambda: F.lambdaloss_loss(pl, ys, k=5, sigma=1.0, mu=5.0, loss_type="NDCG_Loss2", presort=True)) if L % 4 == 0 else float("nan")
    print(f"B={Bq} L={L}: metrics(presort) {t_m:6.1f} us  metrics(sort) {t_m1:6.1f}  sort_desc {t_s:6.1f}  shuffle {t_h:6.1f}  lambdaloss k=5 {t_l:6.1f} | sort ok {ok_sort} perm ok {ok_perm} {ok_perm2} ndcg err {err:.2e} {err2:.2e}", flush=True)
# uniformity of the packed-key shuffle
lab = torch.tensor([[2, 1, 1, 1, 1, 0, 0, 0]], device="cuda").float().expand(8192, 8).contiguous()
pm = F.shuffle_ties_order(lab, seed=9)
print("uniformity (doc 1..4 at slot 1):", [(pm[:, 1] == d).float().mean().item() for d in range(1, 5)])
