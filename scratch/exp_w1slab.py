"""Forward of the fused scorer at F = 700 (config 4 shape): W1 slab staging vs per-wave streaming."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from ptranking_amd.scorer import FusedPointScorer

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for F, R in [(700, 1024 * 512), (256, 4096 * 128), (700, 128 * 512)]:
    f = FusedPointScorer(F, num_layers=3, dropout=0.1).cuda()
    X = torch.randn(R, F, device="cuda")
    for mode in ("0", "1"):
        os.environ["PTR_FWD_W1_STREAM"] = mode
        f.train()
        tr = t(lambda: f(X))
        f.eval()
        with torch.no_grad():
            ev = t(lambda: f(X))
        print(f"F={F} R={R} stream={mode}: train fwd {tr:.0f} us, eval fwd {ev:.0f} us", flush=True)
