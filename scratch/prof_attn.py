import sys, torch
sys.path.insert(0, ".")
from ptranking_amd import listsf as LS
dev = "cuda:0"; B, L, F, H = 1024, 256, 136, 2
torch.manual_seed(0)
q, k, v, g = (torch.randn(B, L, F, device=dev) for _ in range(4))
qd, kd, vd = (t.clone().requires_grad_(True) for t in (q, k, v))
for _ in range(3):
    o = LS.mhsa_core(qd, kd, vd, H, p_drop=0.1, seed=7, site=0); o.backward(g)
torch.cuda.synchronize()
