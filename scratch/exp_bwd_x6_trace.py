"""per-phase shader-clock stamps of the bf16x6 fused backward (PTR_LIB = a -DPTR_B6_TRACE build, PTR_BWD_X6=2): 10 stamps per slab:
   0 slab start | 1 staging done -> B1 | 2 B1 released | 3 chain3+dW3 done | 4 B2 released | 5 chain2+dW2+X done | 6 B3 released | 7 dW1 done | 8 DMA landed | next slab"""
import ctypes as C, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PTR_BWD_X6"] = "1"
from ptranking_amd import _lib
from ptranking_amd.scorer import FusedPointScorer
NL, F, R = 3, 136, 524288
torch.manual_seed(0)
fused = FusedPointScorer(F, num_layers=NL, dropout=0.1).cuda()
X = torch.randn(R, F, device="cuda"); dp = torch.randn(R, device="cuda")
preds = torch.empty(R, device="cuda"); acts = torch.empty(NL * ((R + 15) // 16) * 16 * 112, device="cuda")
st = _lib.current_stream(X.device)
os.environ["PTR_MLP_X6"] = "0"
_lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(fused.flat.data), R, F, NL, 1, C.c_float(0.1), C.c_uint64(77), _lib.ptr(preds), _lib.ptr(acts), st)
ws = torch.zeros(_lib.query("ptr_mlp_backward_ws_floats", F, NL), device="cuda")
grad = torch.empty_like(fused.flat.data)
for _ in range(2):
    _lib.call("ptr_mlp_backward", _lib.ptr(X), _lib.ptr(fused.flat.data), _lib.ptr(acts), _lib.ptr(dp), R, F, NL, C.c_float(0.1), C.c_uint64(77), None, _lib.ptr(ws), _lib.ptr(grad), st)
torch.cuda.synchronize()
NP = _lib.query("ptr_mlp_num_params", F, NL)
tr = ws[256 * NP:256 * NP + 8 * 512].cpu().numpy().view(np.uint64).reshape(8, 256).astype(np.int64)
names = ["chain3+dW3", "B2 wait", "chain2+dW2+X", "B3 wait", "dW1+staging(next)", "B4 wait", "prefetch issue"]
if os.environ.get("TRACE2"):      # -DPTR_B6_TRACE2 builds: 12 stamps per slab
    names = ["c3 chain MFMAs", "c3 epilogue", "c3 dW3 (+prefetch issue)", "B2 wait", "c2 chain MFMAs", "c2 epilogue", "c2 dW2+X stage", "B3 wait", "staging(next)", "dW1", "B4 wait", "loop"]
NS = len(names)
for w in (0, 3, 4, 7):
    t = tr[w]
    n = int((t > 0).sum()) // NS
    d = np.diff(t[:NS * n + 1] if t[NS * n] > 0 else t[:NS * n]).reshape(-1, NS)[1:6] if n > 6 else None
    if d is None: print("wave", w, "too few stamps", n); continue
    print(f"wave {w}: per slab (mean of slabs 1..5): " + ", ".join(f"{nm} {int(v)}" for nm, v in zip(names, d.mean(axis=0))) + f" | total {int(d.sum(axis=1).mean())}")
