import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ptranking_amd as pa
from ptranking_amd import _lib
F = pa.functional
torch.manual_seed(0)
probs = torch.tensor([0.5147, 0.3250, 0.1339, 0.0183, 0.0081], device="cuda")
for L in (128, 256):
    B = 4096
    preds = torch.randn(B, L, device="cuda")
    Y = torch.multinomial(probs.expand(B, -1), L, replacement=True).float(); Y[:, 0].clamp_(min=1.0)
    Y, _ = torch.sort(Y, dim=1, descending=True)
    p = preds.clone().requires_grad_(True)
    for _ in range(5):
        F.lambdarank_loss(p, Y, sigma=1.0)
    torch.cuda.synchronize()
    _lib.TIMING = {}
    for _ in range(int(os.environ.get("ITERS", "50"))):
        F.lambdarank_loss(p, Y, sigma=1.0)
    torch.cuda.synchronize()
    t = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in _lib.TIMING.items()}
    _lib.TIMING = None
    ms = t["ptr_lambdarank_fwd_bwd"]
    print(os.environ.get("PTR_LIB", "default")[-12:], f"L={L}: {ms*1e3:.1f} us  {B*L*(L-1)/2/(ms*1e-3):.3e} pairs/s  sum {t.get('ptr_sum_f32', 0)*1e3:.1f} us", flush=True)
