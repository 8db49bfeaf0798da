"""r6: the LambdaRank ring kernel against waves per workgroup (PTR_RING_WAVES) — how many waves share a SIMD decides the issue rate (scratch/valu_rate)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import functional as F
torch.manual_seed(137)
probs = torch.tensor([0.5147, 0.3250, 0.1339, 0.0183, 0.0081], device="cuda")
for L, B in ((128, 4096), (128, 65536), (256, 4096), (256, 65536)):
    p = torch.randn(B, L, device="cuda")
    y = torch.multinomial(probs.expand(B, -1), L, replacement=True).float(); y[:, 0].clamp_(min=1.0); y = y.sort(dim=1, descending=True)[0].contiguous()
    row = []
    for w in ("0", "2", "4", "8", "16"):
        os.environ["PTR_RING_WAVES"] = w
        def run():
            q = p.detach().requires_grad_(True)
            return F.lambdarank_loss(q, y, sigma=1.0)
        for _ in range(3): l = run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        row.append(f"waves/wg {w:>2s}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us (loss {float(l):.4f})")
    print(f"L={L} B={B}: " + "  ".join(row), flush=True)
