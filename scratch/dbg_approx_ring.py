"""Ring vs LDS ApproxNDCG kernel through the raw entry point: per-query DCG, 1/IDCG and gradients."""
import os, sys, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import _lib
def run(p, y, ring, presort=1, couple=0):
    os.environ["PTR_APPROX_RING"] = "1" if ring else "0"
    B, L = p.shape
    out = torch.empty(1, device="cuda"); dcg = torch.empty(B, device="cuda"); inv = torch.empty(B, device="cuda")
    scale = torch.empty(2, device="cuda"); grad = torch.empty(B, L, device="cuda")
    _lib.call("ptr_approxndcg_fwd_bwd", _lib.ptr(p), _lib.ptr(y), None, B, L, C.c_float(10.0), presort, couple, C.c_float(0.0), _lib.ptr(out),
              _lib.ptr(dcg), _lib.ptr(inv), _lib.ptr(scale), _lib.ptr(grad), _lib.current_stream(p.device))
    torch.cuda.synchronize()
    return out.item(), dcg.cpu().numpy(), inv.cpu().numpy(), grad.cpu().numpy()
for L in (5, 40, 64, 100, 128, 512):
    torch.manual_seed(L)
    B = 3
    p = torch.randn(B, L, device="cuda")
    y = torch.sort(torch.randint(0, 5, (B, L), device="cuda").float(), dim=1, descending=True)[0].contiguous(); y[:, 0].clamp_(min=1)
    a = run(p, y, True); b = run(p, y, False)
    print("L", L, "loss", a[0], b[0], "dcg", a[1], b[1], "inv", a[2], b[2])
    d = np.abs(a[3] - b[3]); print("   grad max diff", np.nanmax(d) if np.isfinite(d).any() else "all nan", "nan count", int(np.isnan(a[3]).sum()), "where", np.argwhere(np.isnan(a[3]))[:8].tolist())
