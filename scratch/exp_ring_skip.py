"""Ring kernel: time at B = 4096 on the MSLR label mix and on skip-free labels (all grades distinct runs shorter than a slot), + block statistics."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ptranking_amd as pa
from ptranking_amd import _lib
F = pa.functional
torch.manual_seed(0)
probs = torch.tensor([0.5147, 0.3250, 0.1339, 0.0183, 0.0081], device="cuda")
B = 4096
for L in (128, 256):
    preds = torch.randn(B, L, device="cuda")
    Y = torch.multinomial(probs.expand(B, -1), L, replacement=True).float(); Y[:, 0].clamp_(min=1.0)
    Y, _ = torch.sort(Y, dim=1, descending=True)
    Yd = (torch.arange(L, device="cuda").flip(0) % 5).float().expand(B, L).contiguous()     # no pure slot anywhere (unsorted on purpose)
    # block statistics of the MSLR batch
    S = L // 64
    Ys = Y.view(B, S, 64)
    pure = (Ys.max(dim=2)[0] == Ys.min(dim=2)[0])
    z = torch.zeros(B, device="cuda")
    run = torch.ones(B, dtype=torch.bool, device="cuda")
    lastlab = Ys[:, -1, 0]
    for k in range(S - 1, -1, -1):
        run = run & pure[:, k] & (Ys[:, k, 0] == lastlab)
        z += run.float()
    skipped = (z * z).mean().item() / (S * S)
    eq = (Y[:, :, None] == Y[:, None, :]).float().mean().item()
    print(f"L={L}: mean trailing pure slots {z.mean().item():.2f} of {S}; blocks skipped {skipped*100:.1f} %; zero-weight pairs {eq*100:.1f} %")
    for name, yy in (("mslr", Y), ("dense", Yd)):
        p = preds.clone().requires_grad_(True)
        for _ in range(3):
            F.lambdarank_loss(p, yy, sigma=1.0)
        torch.cuda.synchronize()
        _lib.TIMING = {}
        for _ in range(50):
            F.lambdarank_loss(p, yy, sigma=1.0)
        torch.cuda.synchronize()
        t = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in _lib.TIMING.items()}
        _lib.TIMING = None
        print(f"   {name}: {t['ptr_lambdarank_fwd_bwd']*1e3:.1f} us", flush=True)
