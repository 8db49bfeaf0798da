import os, sys, torch, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptranking_amd import _lib
torch.manual_seed(0)
F, NL = 136, 3
R = 4096 * 128
X = torch.randn(R, F, device="cuda")
NP = _lib.query("ptr_mlp_num_params", F, NL)
P = torch.randn(NP, device="cuda") * 0.1
preds = torch.zeros(R + 4096, device="cuda"); acts = torch.empty((NL, R, 112), device="cuda")
st = _lib.current_stream(X.device)
for train in (1, 0):
    for _ in range(3):
        _lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(P), R, F, NL, train, C.c_float(0.1), C.c_uint64(5), _lib.ptr(preds), _lib.ptr(acts), st)
    torch.cuda.synchronize()
    _lib.TIMING = {}
    for _ in range(10):
        _lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(P), R, F, NL, train, C.c_float(0.1), C.c_uint64(5), _lib.ptr(preds), _lib.ptr(acts), st)
    torch.cuda.synchronize()
    t = sum(a.elapsed_time(b) for a, b in _lib.TIMING["ptr_mlp_forward"]) / 10
    _lib.TIMING = None
    print(os.environ.get("PTR_LIB", "default")[-14:], "train" if train else "eval ", f"{t*1e3:.1f} us")
