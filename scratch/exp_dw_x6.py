"""A/B of the layer-wise backward at wide inputs: PTR_DW_X6=0 (fp32-MFMA first-layer dW) vs 2 (bf16x6), torch events around ptr_mlp_backward."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import _lib
from ptranking_amd.scorer import FusedPointScorer
for F, NL, R in ((700, 3, 524288), (256, 3, 524288), (400, 3, 262144)):
    torch.manual_seed(0)
    fused = FusedPointScorer(F, num_layers=NL, dropout=0.1).cuda()
    X = torch.randn(R, F, device="cuda"); w = torch.randn(R, device="cuda")
    preds = torch.empty(R, device="cuda"); acts = torch.empty(NL, R, 112, device="cuda")
    st = _lib.current_stream(X.device)
    _lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(fused.flat.data), R, F, NL, 1, C.c_float(0.1), C.c_uint64(5), _lib.ptr(preds), _lib.ptr(acts), st)
    ws = torch.empty(_lib.query("ptr_mlp_backward_ws_floats", F, NL), device="cuda")
    dz = torch.empty(max(1, _lib.query("ptr_mlp_backward_dz_floats", R, F, NL)), device="cuda")
    g = torch.empty_like(fused.flat.data)
    for mode, tail in (("0", "0"), ("2", "0"), ("2", "1"), ("0", "0"), ("2", "0"), ("2", "1")):
        os.environ["PTR_DW_X6"] = mode
        os.environ["PTR_BWD_TAIL"] = tail
        def run():
            _lib.call("ptr_mlp_backward", _lib.ptr(X), _lib.ptr(fused.flat.data), _lib.ptr(acts), _lib.ptr(w), R, F, NL, C.c_float(0.1), C.c_uint64(5),
                      _lib.ptr(dz), _lib.ptr(ws), _lib.ptr(g), st)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        print(f"F={F} NL={NL} R={R} PTR_DW_X6={mode} PTR_BWD_TAIL={tail}: backward {e0.elapsed_time(e1) * 100:8.1f} us", flush=True)
