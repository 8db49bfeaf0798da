"""r6: A/B of bf16x6 fused-backward builds (PTR_LIB = a variant library): time at 524 288 / 131 072 documents x 136 features and agreement with the
fp32-MFMA fused backward (PTR_BWD_X6=0) on the same stored activations."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import _lib
from ptranking_amd.scorer import FusedPointScorer, alloc_acts
NL, F = 3, 136
tag = os.path.basename(os.environ.get("PTR_LIB", "product"))
res = []
for R in (524288, 131072):
    torch.manual_seed(R)
    fused = FusedPointScorer(F, num_layers=NL, dropout=0.1).cuda()
    X = torch.randn(R, F, device="cuda"); dp = torch.randn(R, device="cuda")
    preds = torch.empty(R, device="cuda"); acts = alloc_acts(R, NL, "cuda")
    st = _lib.current_stream(X.device)
    os.environ["PTR_MLP_X6"] = "0"
    _lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(fused.flat.data), R, F, NL, 1, C.c_float(0.1), C.c_uint64(77), _lib.ptr(preds), _lib.ptr(acts), st)
    ws = torch.empty(_lib.query("ptr_mlp_backward_ws_floats", F, NL), device="cuda")
    g = {}
    for mode in ("0", "1"):
        os.environ["PTR_BWD_X6"] = mode
        grad = torch.full_like(fused.flat.data, float("nan"))
        def bwd():
            _lib.call("ptr_mlp_backward", _lib.ptr(X), _lib.ptr(fused.flat.data), _lib.ptr(acts), _lib.ptr(dp), R, F, NL, C.c_float(0.1), C.c_uint64(77), None,
                      _lib.ptr(ws), _lib.ptr(grad), st)
        bwd(); torch.cuda.synchronize()
        g[mode] = grad.clone()
        if mode == "0" and R != 131072:
            continue
        for _ in range(3): bwd()
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): bwd()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        g[mode + "t"] = min(ts)
    d = (g["0"] - g["1"]).abs()
    res.append(f"R={R}: x6 {g['1t']:.1f} us" + (f" (fp32 {g['0t']:.1f})" if "0t" in g else "") + f" maxdiff {float(d.max()):.2e}/{float(g['0'].abs().max()):.1e} nan {int(torch.isnan(g['1']).sum())}")
print(f"{tag:44s} " + " | ".join(res), flush=True)
