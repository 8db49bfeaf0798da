"""Linear-layer kernels at config 5's shapes (R = 1024 x 256 documents): time and TFLOP/s of forward, backward-input, backward-weight."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptranking_amd import _lib
from ptranking_amd import linear as LN
R = 1024 * 256
torch.manual_seed(0)
shapes = [(136, 128), (128, 256), (256, 512), (512, 136), (136, 408), (136, 136), (512, 1)]
tot = {"f": 0.0, "bi": 0.0, "bw": 0.0}
for K, N in shapes:
    x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    dy = torch.randn(R, N, device="cuda")
    for _ in range(2):
        LN._fwd(x, K, w, b); LN._bwd_input(dy, w); LN._bwd_weight(x, K, dy, True)
    torch.cuda.synchronize()
    _lib.TIMING = {}
    for _ in range(5):
        LN._fwd(x, K, w, b); LN._bwd_input(dy, w); LN._bwd_weight(x, K, dy, True)
    torch.cuda.synchronize()
    t = {k: sum(a.elapsed_time(b_) for a, b_ in v) / len(v) for k, v in _lib.TIMING.items()}
    _lib.TIMING = None
    fl = 2.0 * R * K * N / 1e9
    f, bi, bw = t["ptr_linear_forward"], t["ptr_linear_backward_input"], t["ptr_linear_backward_weight"]
    tot["f"] += f; tot["bi"] += bi; tot["bw"] += bw
    print(f"K={K:4d} N={N:4d}: fwd {f*1e3:7.1f} us {fl/f:6.1f} TF/s | bwd-input {bi*1e3:7.1f} us {fl/bi:6.1f} | bwd-weight {bw*1e3:7.1f} us {fl/bw:6.1f}", flush=True)
print(os.environ.get("PTR_LIB", "default")[-16:], "totals ms:", {k: round(v, 3) for k, v in tot.items()})
