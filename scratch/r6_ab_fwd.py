"""r6: A/B of bf16x6 forward builds (PTR_LIB): eval / training time at 524 288 and 131 072 documents x 136 features (four rotating inputs: nothing cache resident)
and agreement with the fp32-MFMA forward (scores + stored activations)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import _lib
from ptranking_amd.scorer import FusedPointScorer, x6_workspace, alloc_acts, acts_rowmajor
NL, F = 3, 136
tag = os.path.basename(os.environ.get("PTR_LIB", "product"))
res = []
for R in (524288, 131072):
    torch.manual_seed(0)
    fused = FusedPointScorer(F, num_layers=NL, dropout=0.1).cuda()
    Xs = [torch.randn(R, F, device="cuda") for _ in range(4)]
    preds = torch.empty(R, device="cuda"); acts = alloc_acts(R, NL, "cuda")
    p2 = torch.empty(R, device="cuda"); a2 = alloc_acts(R, NL, "cuda")
    ws = x6_workspace(Xs[0].device, F, NL)
    st = _lib.current_stream(Xs[0].device)
    _lib.call("ptr_mlp_forward", _lib.ptr(Xs[0]), _lib.ptr(fused.flat.data), R, F, NL, 1, C.c_float(0.1), C.c_uint64(7), _lib.ptr(p2), _lib.ptr(a2), st)
    _lib.call("ptr_mlp_forward_x6", _lib.ptr(Xs[0]), _lib.ptr(fused.flat.data), R, F, NL, 1, C.c_float(0.1), C.c_uint64(7), _lib.ptr(preds), _lib.ptr(acts), _lib.ptr(ws), st)
    torch.cuda.synchronize()
    dpred = float((preds - p2).abs().max()); dact = float((acts_rowmajor(acts, R, NL)[:, :, :101] - acts_rowmajor(a2, R, NL)[:, :, :101]).abs().max())
    out = []
    for train in (0, 1):
        def fwd(i):
            _lib.call("ptr_mlp_forward_x6", _lib.ptr(Xs[i % 4]), _lib.ptr(fused.flat.data), R, F, NL, train, C.c_float(0.1), C.c_uint64(7 + i), _lib.ptr(preds),
                      _lib.ptr(acts), _lib.ptr(ws), st)
        for i in range(3): fwd(i)
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(12): fwd(i)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 12 * 1e3)
        out.append(min(ts))
    res.append(f"R={R}: eval {out[0]:.1f} train {out[1]:.1f} us dpred {dpred:.1e} dact {dact:.1e}")
print(f"{tag:44s} " + " | ".join(res), flush=True)
