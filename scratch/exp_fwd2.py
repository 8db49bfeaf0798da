import sys, torch, ctypes as C
sys.path.insert(0, ".")
from ptranking_amd import _lib
dev = "cuda:0"; F, NL = 136, 3
torch.manual_seed(0)
npar = _lib.query("ptr_mlp_num_params", F, NL)
P = torch.randn(npar, device=dev) * 0.1
for R in (256 * 8 * 32, 256 * 8 * 32 * 2, 256 * 8 * 32 * 4, 4096 * 128):
    X = torch.randn(R, F, device=dev)
    preds = torch.empty(R, device=dev); acts = torch.empty(NL, R, 112, device=dev)
    def run(train, p):
        _lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(P), R, F, NL, train, C.c_float(p), C.c_uint64(5), _lib.ptr(preds), _lib.ptr(acts), _lib.current_stream(X.device))
    def t(train, p, n=30):
        for _ in range(3): run(train, p)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): run(train, p)
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
    print(f"R={R} ({R // (256 * 8 * 32)} tiles of 32 rows per wave): eval {t(0, 0.0) * 1e3:.1f} us, train {t(1, 0.1) * 1e3:.1f} us")
