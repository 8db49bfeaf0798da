"""Attention core forward / forward+backward time at config 5's shape (1024 x 256 x 136, 2 heads, dropout 0.1); env: PTR_ATTN_RT1, PTR_ATTN_WAVES, PTR_LIB."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptranking_amd import listsf as LS
B, L, F, H = 1024, 256, 136, 2
torch.manual_seed(0)
q, k, v, g = (torch.randn(B, L, F, device="cuda") for _ in range(4))
qd, kd, vd = (t.clone().requires_grad_(True) for t in (q, k, v))
def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
f = lambda: LS.mhsa_core(qd, kd, vd, H, p_drop=0.1, seed=7, site=0)
def fb():
    f().backward(g)
tf, tfb = timeit(f), timeit(fb)
print(f"{os.environ.get('PTR_LIB', 'default')[-10:]} RT1={os.environ.get('PTR_ATTN_RT1', '0')} W={os.environ.get('PTR_ATTN_WAVES', '4')}: fwd {tf*1e3:.0f} us  fwd+bwd {tfb*1e3:.0f} us  bwd {1e3*(tfb-tf):.0f} us")
