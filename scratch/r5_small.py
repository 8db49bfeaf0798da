"""r5: sort / metrics / tie shuffle / LambdaLoss entry points at 65 536 x 256 (HIP events over 20 calls each)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import functional as F
B, L = 65536, 256
torch.manual_seed(1)
probs = torch.tensor([0.5147, 0.3250, 0.1339, 0.0183, 0.0081], device="cuda")
p = torch.randn(B, L, device="cuda")
y = torch.multinomial(probs.expand(B, -1), L, replacement=True).float(); y[:, 0].clamp_(min=1.0); y = y.sort(dim=1, descending=True)[0].contiguous()
KS = [1, 3, 5, 10, 20, 50]
def lg(fn, *a, **k):
    fn(p.detach().requires_grad_(True), *a, **k)
cases = [("metrics ndcg", lambda: F.metrics_at_ks(p, y, KS, presort=True, which=("ndcg",))),
         ("metrics all4", lambda: F.metrics_at_ks(p, y, KS, presort=True)),
         ("sort_desc", lambda: F.sort_desc(p)),
         ("shuffle_ties", lambda: F.shuffle_ties_order(y, seed=11)),
         ("lambdaloss k=5", lambda: lg(F.lambdaloss_loss, y, k=5, sigma=1.0, mu=5.0, loss_type="NDCG_Loss2", presort=True)),
         ("lambdarank", lambda: lg(F.lambdarank_loss, y, sigma=1.0))]
for name, fn in cases:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:18s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per entry call", flush=True)
