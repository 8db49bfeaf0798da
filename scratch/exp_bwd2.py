"""Timing of forward / fused backward at the bench shape (+ agreement with the layer-wise path)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptranking_amd import _lib
from ptranking_amd.scorer import FusedPointScorer
torch.manual_seed(0)
F = 136
R = 4096 * 128
X = torch.randn(R, F, device="cuda"); w = torch.randn(R, 1, device="cuda")
res = {}
for fused in (0, 1):
    os.environ["PTR_BWD_FUSED"] = str(fused)
    torch.manual_seed(1)
    f = FusedPointScorer(F, 3, dropout=0.1).cuda(); f.train()
    for it in range(3):
        torch.manual_seed(7); f.flat.grad = None
        out = f(X); (out * w).sum().backward()
    res[fused] = f.flat.grad.clone()
    torch.cuda.synchronize()
    _lib.TIMING = {}
    for it in range(10):
        out = f(X); (out * w).sum().backward()
    torch.cuda.synchronize()
    t = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in _lib.TIMING.items()}
    _lib.TIMING = None
    print(os.environ.get("PTR_LIB", "default"), "fused", fused, {k: round(v, 4) for k, v in t.items()}, flush=True)
d = (res[0] - res[1]).abs().max().item(); sc = res[0].abs().max().item()
print("agreement max|d| / scale:", d, sc, d / sc)
