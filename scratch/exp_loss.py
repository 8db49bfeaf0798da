"""LambdaRank loss kernel: ring (register / DPP) kernel vs the LDS kernel — agreement and pairs/s at L = 64/128/256."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ptranking_amd as pa
from ptranking_amd import _lib
F = pa.functional
torch.manual_seed(0)
probs = torch.tensor([0.5147, 0.3250, 0.1339, 0.0183, 0.0081], device="cuda")
for L in (64, 100, 128, 200, 256):
    B = 4096
    preds = torch.randn(B, L, device="cuda")
    Y = torch.multinomial(probs.expand(B, -1), L, replacement=True).float(); Y[:, 0].clamp_(min=1.0)
    Y, _ = torch.sort(Y, dim=1, descending=True)
    lens = torch.randint(max(1, L // 2), L + 1, (B,), device="cuda", dtype=torch.int32)
    out = {}
    for ring in (0, 1):
        os.environ["PTR_LAMBDARANK_RING"] = str(ring)
        for use_lens in (False, True):
            p = preds.clone().requires_grad_(True)
            loss = F.lambdarank_loss(p, Y, sigma=1.0, lens=lens if use_lens else None)
            loss.backward()
            out[(ring, use_lens)] = (loss.item(), p.grad.clone())
        p = preds.clone().requires_grad_(True)
        for _ in range(3):
            F.lambdarank_loss(p, Y, sigma=1.0)
        torch.cuda.synchronize()
        _lib.TIMING = {}
        for _ in range(20):
            F.lambdarank_loss(p, Y, sigma=1.0)
        torch.cuda.synchronize()
        t = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in _lib.TIMING.items()}
        _lib.TIMING = None
        ms = t["ptr_lambdarank_fwd_bwd"]
        print(f"L={L} ring={ring}: {ms*1e3:.1f} us  {B*L*(L-1)/2/(ms*1e-3):.3e} pairs/s", flush=True)
    for ul in (False, True):
        (l0, g0), (l1, g1) = out[(0, ul)], out[(1, ul)]
        print(f"   lens={ul}: loss {l0:.6f} vs {l1:.6f} rel {abs(l0-l1)/abs(l0):.2e}; grad max|d| {(g0-g1).abs().max().item():.3e} / max {g0.abs().max().item():.3e}", flush=True)
