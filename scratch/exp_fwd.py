import sys, torch, ctypes as C
sys.path.insert(0, ".")
from ptranking_amd import _lib
dev = "cuda:0"; R, F, NL = 4096 * 128, 136, 3
torch.manual_seed(0)
X = torch.randn(R, F, device=dev)
npar = _lib.query("ptr_mlp_num_params", F, NL)
P = torch.randn(npar, device=dev) * 0.1
preds = torch.empty(R, device=dev); acts = torch.empty(NL, R, 112, device=dev)
def run(train, p):
    _lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(P), R, F, NL, train, C.c_float(p), C.c_uint64(5), _lib.ptr(preds), _lib.ptr(acts), _lib.current_stream(X.device))
def t(train, p, n=20):
    for _ in range(3): run(train, p)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run(train, p)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
print("eval (no stores, no dropout)", t(0, 0.0))
print("train p=0 (stores, no dropout hash)", t(1, 0.0))
print("train p=0.1", t(1, 0.1))
