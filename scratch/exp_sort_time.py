"""Time of the sort / metric kernels at 65 536 x L (torch events around 10 launches)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import functional as F
B = 65536
for L in (64, 128, 256, 512, 1024):
    Bq = B if L <= 256 else B * 256 // L
    g = torch.Generator(device="cuda").manual_seed(1)
    p = torch.randn(Bq, L, device="cuda", generator=g)
    y = torch.randint(0, 5, (Bq, L), device="cuda", generator=g).float()
    ys = torch.sort(y, dim=1, descending=True)[0]
    for name, fn in (("metrics presort", lambda: F.metrics_at_ks(p, ys, [1, 3, 5, 10, 20, 50], presort=True)),
                     ("metrics ideal-sort", lambda: F.metrics_at_ks(p, y, [1, 3, 5, 10, 20, 50], presort=False)),
                     ("sort_desc", lambda: F.sort_desc(p))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"L={L:5d} B={Bq:6d} {name:20s} {e0.elapsed_time(e1) * 100:8.1f} us", flush=True)
