"""bf16x6 forward vs fp32-MFMA forward through the C ABI (no autograd), HIP events: python scratch/exp_x6_time.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import _lib
from ptranking_amd.scorer import FusedPointScorer, x6_workspace
NL = 3
for F, R in ((136, 4096 * 128), (136, 1024 * 128), (136, 256 * 128), (136, 64 * 128), (700, 1024 * 512), (256, 2048 * 128)):
    torch.manual_seed(0)
    fused = FusedPointScorer(F, num_layers=NL, dropout=0.1).cuda()
    nb = 4 if R * F * 4 > 2e8 else 8
    Xs = [torch.randn(R, F, device="cuda") for _ in range(nb)]
    preds = torch.empty(R, device="cuda"); acts = torch.empty(NL, R, 112, device="cuda")
    ws = x6_workspace(Xs[0].device, F, NL)
    st = _lib.current_stream(Xs[0].device)

    def fwd(i, train, x6):
        X = Xs[i % nb]
        if x6:
            _lib.call("ptr_mlp_forward_x6", _lib.ptr(X), _lib.ptr(fused.flat.data), R, F, NL, train, C.c_float(0.1), C.c_uint64(7 + i), _lib.ptr(preds),
                      _lib.ptr(acts), _lib.ptr(ws), st)
        else:
            _lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(fused.flat.data), R, F, NL, train, C.c_float(0.1), C.c_uint64(7 + i), _lib.ptr(preds),
                      _lib.ptr(acts), st)
    for train in (0, 1):
        for x6 in (0, 1):
            for i in range(3):
                fwd(i, train, x6)
            torch.cuda.synchronize()
            n = 30
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                fwd(i, train, x6)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            flop = 2.0 * (100 * F + (NL - 1) * 100 * 100 + 100) * R
            print(f"F={F:4d} R={R:7d} train={train} x6={x6}: {ms * 1e3:8.1f} us   {flop / ms / 1e9:7.1f} TFLOP/s (effective fp32)", flush=True)
    del Xs, acts
