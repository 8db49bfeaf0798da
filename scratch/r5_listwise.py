"""r5: ListNet / ListMLE entry points at 65 536 x L, register kernels vs the LDS kernels (PTR_LISTNET_VEC / PTR_LISTMLE_VEC = 0 in a fresh process)"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import functional as F
B = 65536
for L in (64, 128, 256, 512):
    torch.manual_seed(L)
    p = torch.randn(B, L, device="cuda"); y = torch.randint(0, 5, (B, L), device="cuda").float().sort(dim=1, descending=True)[0]
    perm = F.shuffle_ties_order(y, seed=3)
    for name, fn, nbytes in (("listnet", lambda: F.listnet_loss(p.requires_grad_(True), y), B * (12 * L + 4)),
                             ("listmle", lambda: F.listmle_loss(p.requires_grad_(True), perm), B * (16 * L + 4))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{name} B={B} L={L}: {us:7.1f} us (forward entry incl. slot sum)  {nbytes / us / 1e6:6.2f} TB/s  vec={os.environ.get('PTR_LISTNET_VEC', '1')}", flush=True)
