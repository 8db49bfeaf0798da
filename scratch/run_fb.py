"""A few scorer forward+backward calls at the bench shape (profiling target)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptranking_amd.scorer import FusedPointScorer
torch.manual_seed(0)
R, F = 4096 * 128, 136
X = torch.randn(R, F, device="cuda"); w = torch.randn(R, 1, device="cuda")
f = FusedPointScorer(F, 3, dropout=0.1).cuda(); f.train()
for it in range(int(os.environ.get("ITERS", "4"))):
    out = f(X); (out * w).sum().backward()
torch.cuda.synchronize()
