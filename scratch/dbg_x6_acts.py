import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import _lib
from ptranking_amd.scorer import FusedPointScorer, x6_workspace
F, NL = 136, 3
for R in (65573, 131072, 262144):
    torch.manual_seed(3)
    fused = FusedPointScorer(F, num_layers=NL, dropout=0.1).cuda()
    X = torch.randn(R, F, device="cuda")
    ws = x6_workspace(X.device, F, NL); st = _lib.current_stream(X.device)
    pa, pb = torch.empty(R, device="cuda"), torch.empty(R, device="cuda")
    aa = torch.full((NL, R, 112), float("nan"), device="cuda"); ab = torch.full((NL, R, 112), float("nan"), device="cuda")
    _lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(fused.flat.data), R, F, NL, 1, C.c_float(0.1), C.c_uint64(99), _lib.ptr(pa), _lib.ptr(aa), st)
    _lib.call("ptr_mlp_forward_x6", _lib.ptr(X), _lib.ptr(fused.flat.data), R, F, NL, 1, C.c_float(0.1), C.c_uint64(99), _lib.ptr(pb), _lib.ptr(ab), _lib.ptr(ws), st)
    torch.cuda.synchronize()
    print(f"R={R}: preds max diff {float((pa - pb).abs().max()):.2e}  nan in x6 acts: {int(torch.isnan(ab).sum())}  nan in old acts: {int(torch.isnan(aa).sum())}")
    for l in range(NL):
        d = (aa[l] - ab[l]).abs(); d[torch.isnan(d)] = 1e9
        bad = (d.max(dim=1).values > 1e-4).nonzero().flatten()
        print(f"   layer {l}: max diff {float(d.max()):.2e}; rows off by > 1e-4: {bad.numel()}  first {bad[:12].tolist()}  cols of first bad row: {(d[bad[0]] > 1e-4).nonzero().flatten().tolist() if bad.numel() else []}")
    for l in range(NL):
        ga, gb = aa[l, :, :100] > 0, ab[l, :, :100] > 0
        mism = (ga != gb)
        print(f"   layer {l}: gate mismatches old vs x6: {int(mism.sum())} of {ga.numel()};  |value| at mismatches (old, x6): {aa[l, :, :100][mism][:6].tolist()} {ab[l, :, :100][mism][:6].tolist()}")
