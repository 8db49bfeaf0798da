"""Linear-layer kernels at config 5's shapes with COLD operands: NSET distinct operand sets (> 256 MB MALL in total) used in rotation, as inside
the train step where every kernel reads what the previous one wrote to HBM.  exp_linear.py re-uses one set (L2 / MALL-hot)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptranking_amd import _lib
from ptranking_amd import linear as LN
R = 1024 * 256
torch.manual_seed(0)
shapes = [(136, 136), (136, 128), (128, 256), (136, 408), (512, 136), (100, 100)]
for K, N in shapes:
    nset = max(3, int(1.2e9 / (R * (K + 2 * N) * 4)) + 1)
    xs = [torch.randn(R, K, device="cuda") for _ in range(nset)]
    dys = [torch.randn(R, N, device="cuda") for _ in range(nset)]
    gates = [torch.randn(R, K, device="cuda") for _ in range(nset)]
    w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    def run(i):
        LN._fwd(xs[i % nset], K, w, b); LN._bwd_input(dys[i % nset], w, gates[i % nset], 0.0) if hasattr(LN, "_bwd_input") else None
    for i in range(nset): run(i)
    torch.cuda.synchronize()
    _lib.TIMING = {}
    for i in range(2 * nset): run(i)
    torch.cuda.synchronize()
    t = {k: sum(a.elapsed_time(b_) for a, b_ in v) / len(v) for k, v in _lib.TIMING.items()}
    _lib.TIMING = None
    fl = 2.0 * R * K * N / 1e9
    f, bi = t["ptr_linear_forward"], t["ptr_linear_backward_input"]
    gb_f = R * (K + N) * 4 / 1e9; gb_b = R * (2 * K + N) * 4 / 1e9
    print(f"K={K:4d} N={N:4d} ({nset} sets): fwd {f*1e3:7.1f} us {fl/f:6.1f} TF/s {gb_f/f:5.2f} TB/s | bwd-input+gate {bi*1e3:7.1f} us {fl/bi:6.1f} TF/s {gb_b/bi:5.2f} TB/s", flush=True)
    del xs, dys, gates
