"""r6: LambdaRank loss + gradient entry point at L = 128 .. 1024 on the MSLR label mix: ring kernel (default) against the LDS kernel (PTR_LAMBDARANK_RING=0)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import functional as F
torch.manual_seed(137)
probs = torch.tensor([0.5147, 0.3250, 0.1339, 0.0183, 0.0081], device="cuda")
for L, B in ((128, 4096), (256, 4096), (512, 4096), (512, 32768), (1024, 2048), (1024, 16384)):
    p = torch.randn(B, L, device="cuda")
    y = torch.multinomial(probs.expand(B, -1), L, replacement=True).float(); y[:, 0].clamp_(min=1.0); y = y.sort(dim=1, descending=True)[0].contiguous()
    out = {}
    for ring in ("1", "0"):
        os.environ["PTR_LAMBDARANK_RING"] = ring
        def run():
            q = p.detach().requires_grad_(True)
            return F.lambdarank_loss(q, y, sigma=1.0), q
        l, q = run(); l.backward(); g = q.grad.clone()
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        out[ring] = (e0.elapsed_time(e1) / 10 * 1e3, float(l), g)
    pairs = B * L * (L - 1) / 2
    d = float((out["1"][2] - out["0"][2]).abs().max())
    print(f"L={L} B={B}: ring {out['1'][0]:9.1f} us ({pairs / out['1'][0] / 1e6:.2f} T pairs/s)  LDS {out['0'][0]:9.1f} us ({pairs / out['0'][0] / 1e6:.2f} T pairs/s)  "
          f"loss {out['1'][1]:.4f} / {out['0'][1]:.4f}  max|dgrad| {d:.2e}", flush=True)
