"""Fused single-pass scorer backward vs the layer-wise kernels: agreement on odd row counts, then timing at the bench shape."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptranking_amd import _lib
from ptranking_amd.scorer import FusedPointScorer

def grads(f, X, w, fused):
    os.environ["PTR_BWD_FUSED"] = "1" if fused else "0"
    f.flat.grad = None
    torch.manual_seed(5)
    out = f(X); (out * w).sum().backward()
    return f.flat.grad.detach().clone()

torch.manual_seed(0)
F = 136
for p in (0.1, 0.0):
    for R in (1, 31, 32, 33, 64 * 128 + 5, 300 * 32 * 3 + 17):
        f = FusedPointScorer(F, 3, dropout=p).cuda(); f.train()
        X = torch.randn(R, F, device="cuda"); w = torch.randn(R, 1, device="cuda")
        a, b = grads(f, X, w, True), grads(f, X, w, False)
        a2 = grads(f, X, w, True)
        err = (a - b).abs().max().item(); sc = b.abs().max().item()
        rel = ((a - b).abs() / (b.abs() + 1e-6 * sc)).max().item()
        print(f"p={p} R={R}: max|d|={err:.3e} scale={sc:.3e} maxrel={rel:.3e} bitstable={torch.equal(a, a2)} finite={torch.isfinite(a).all().item()}", flush=True)
        assert err <= 2e-5 * max(1.0, sc), "MISMATCH"
R = 4096 * 128
X = torch.randn(R, F, device="cuda"); w = torch.randn(R, 1, device="cuda")
for fused in (0, 1):
    os.environ["PTR_BWD_FUSED"] = str(fused)
    f = FusedPointScorer(F, 3, dropout=0.1).cuda(); f.train()
    for it in range(3):
        out = f(X); (out * w).sum().backward()
    torch.cuda.synchronize()
    _lib.TIMING = {}
    for it in range(10):
        out = f(X); (out * w).sum().backward()
    torch.cuda.synchronize()
    t = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in _lib.TIMING.items()}
    _lib.TIMING = None
    print("fused", fused, {k: round(v, 4) for k, v in t.items()}, flush=True)
