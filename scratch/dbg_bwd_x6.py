"""bf16x6 fused backward (csrc/scorer_bwd_x6.hip, PTR_BWD_X6=2) vs the fp32-MFMA fused backward (PTR_BWD_X6=0) on the same stored activations:
gradient agreement and time."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import _lib
from ptranking_amd.scorer import FusedPointScorer, x6_workspace
NL = 3
for F, R in ((136, 2085), (136, 32), (132, 777), (140, 4096 + 5), (136, 131072), (136, 524288)):
    torch.manual_seed(R)
    fused = FusedPointScorer(F, num_layers=NL, dropout=0.1).cuda()
    X = torch.randn(R, F, device="cuda"); dp = torch.randn(R, device="cuda")
    preds = torch.empty(R, device="cuda"); acts = torch.empty(NL * ((R + 15) // 16) * 16 * 112, device="cuda")
    st = _lib.current_stream(X.device)
    _lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(fused.flat.data), R, F, NL, 1, C.c_float(0.1), C.c_uint64(77), _lib.ptr(preds), _lib.ptr(acts), st)
    ws = torch.empty(_lib.query("ptr_mlp_backward_ws_floats", F, NL), device="cuda")
    g = {}
    for mode in ("0", "2"):
        os.environ["PTR_BWD_X6"] = "1" if mode == "2" else "0"
        grad = torch.full_like(fused.flat.data, float("nan"))
        def bwd():
            _lib.call("ptr_mlp_backward", _lib.ptr(X), _lib.ptr(fused.flat.data), _lib.ptr(acts), _lib.ptr(dp), R, F, NL, C.c_float(0.1), C.c_uint64(77), None,
                      _lib.ptr(ws), _lib.ptr(grad), st)
        bwd(); torch.cuda.synchronize()
        g[mode] = grad.clone()
        for _ in range(3): bwd()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): bwd()
        e1.record(); torch.cuda.synchronize()
        g[mode + "t"] = e0.elapsed_time(e1) / 10 * 1e3
    d = (g["0"] - g["2"]).abs()
    n1 = 100 * F
    print(f"F={F} R={R}: max|grad| {float(g['0'].abs().max()):.3e}; max diff {float(d.max()):.3e} (nan: {int(torch.isnan(g['2']).sum())}); W1 {float(d[:n1].max()):.2e} b1 {float(d[n1:n1+100].max()):.2e} "
          f"W2 {float(d[n1+100:n1+10100].max()):.2e} b2 {float(d[n1+10100:n1+10200].max()):.2e} W3 {float(d[n1+10200:n1+20200].max()):.2e} b3 {float(d[n1+20200:n1+20300].max()):.2e} "
          f"wo {float(d[n1+20300:n1+20400].max()):.2e} bo {float(d[-1]):.2e} | fp32 {g['0t']:.1f} us, x6 {g['2t']:.1f} us", flush=True)
