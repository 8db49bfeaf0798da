import sys, os, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ptranking_amd as pa
from ptranking_amd import _lib
from ptranking_amd.scorer import FusedPointScorer
torch.manual_seed(0)
R, F = 4096*128, 136
X = torch.randn(R, F, device="cuda")
w = torch.randn(R, 1, device="cuda")
for p in (0.1, 0.0):
    f = FusedPointScorer(F, 3, dropout=p).cuda(); f.train()
    for it in range(3):
        out = f(X); (out*w).sum().backward()
    torch.cuda.synchronize()
    _lib.TIMING = {}
    for it in range(10):
        out = f(X); (out*w).sum().backward()
    torch.cuda.synchronize()
    t = {k: sum(a.elapsed_time(b) for a,b in v)/len(v) for k,v in _lib.TIMING.items()}
    _lib.TIMING = None
    print("p", p, {k: round(v,3) for k,v in t.items()})
