import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import _lib, scorer as S
F, NL, R = 136, 3, 2085
torch.manual_seed(0)
fused = S.FusedPointScorer(F, num_layers=NL, dropout=0.1).cuda()
X = torch.randn(R, F, device="cuda")
st = _lib.current_stream(X.device)
out = {}
for name in ("x6", "fp32"):
    preds = torch.empty(R, device="cuda"); acts = torch.full((S.acts_floats(R, NL),), float("nan"), device="cuda")
    if name == "x6":
        ws = S.x6_workspace(X.device, F, NL)
        _lib.call("ptr_mlp_forward_x6", _lib.ptr(X), _lib.ptr(fused.flat.data), R, F, NL, 1, C.c_float(0.1), C.c_uint64(7), _lib.ptr(preds), _lib.ptr(acts), _lib.ptr(ws), st)
    else:
        _lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(fused.flat.data), R, F, NL, 1, C.c_float(0.1), C.c_uint64(7), _lib.ptr(preds), _lib.ptr(acts), st)
    torch.cuda.synchronize()
    out[name] = (preds.cpu(), acts.cpu())
print("preds diff", (out["x6"][0] - out["fp32"][0]).abs().max().item())
a, b = S.acts_rowmajor(out["x6"][1], R, NL), S.acts_rowmajor(out["fp32"][1], R, NL)
print("nan x6", torch.isnan(out["x6"][1]).sum().item(), "nan fp32", torch.isnan(out["fp32"][1]).sum().item())
d = (a - b).abs()
for l in range(NL):
    bad = (d[l] > 1e-3).nonzero()
    print("layer", l, "bad", bad.shape[0], "rows", bad[:, 0].unique()[:20].tolist(), "cols", bad[:, 1].unique()[:40].tolist())
