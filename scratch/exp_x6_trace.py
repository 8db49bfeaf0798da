"""per-step shader-clock stamps of the bf16x6 forward (PTR_LIB = a -DPTR_X6_TRACE build): arrival at / release from every SYNC of workgroup 0"""
import ctypes as C, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import _lib
from ptranking_amd.scorer import FusedPointScorer, x6_workspace
NL, F, R = 3, 136, int(os.environ.get("R", 524288))
train = int(os.environ.get("TRAIN", 0))
torch.manual_seed(0)
fused = FusedPointScorer(F, num_layers=NL, dropout=0.1).cuda()
X = torch.randn(R, F, device="cuda")
preds = torch.empty(R, device="cuda"); acts = torch.empty(NL, R, 112, device="cuda")
ws = x6_workspace(X.device, F, NL)
st = _lib.current_stream(X.device)
for i in range(3):
    ws[-16384:].zero_()
    _lib.call("ptr_mlp_forward_x6", _lib.ptr(X), _lib.ptr(fused.flat.data), R, F, NL, train, C.c_float(0.1), C.c_uint64(7), _lib.ptr(preds), _lib.ptr(acts), _lib.ptr(ws), st)
torch.cuda.synchronize()
tr = ws[-16384:].cpu().numpy().view(np.uint64).reshape(8, 256).astype(np.int64)
ns = 13
for w in (0, 4, 7):
    t = tr[w]; n = min(int((t[:252] > 0).sum()) // 2, 126)
    arr, rel = t[0:2 * n:2], t[1:2 * n:2]
    print(f"wave {w}: {n} syncs; step = release(i) -> arrival(i+1) [compute], wait = arrival -> release [barrier]")
    comp = arr[1:] - rel[:-1]; wait = rel - arr
    for p in range(min(3, (n - 1) // ns)):
        print(f"  pass {p}: compute per step {comp[p * ns:(p + 1) * ns].tolist()}  sum {int(comp[p * ns:(p + 1) * ns].sum())}")
        print(f"          barrier wait    {wait[p * ns:(p + 1) * ns].tolist()}  sum {int(wait[p * ns:(p + 1) * ns].sum())}")
    n = min(n, 126)
    dt_us = (int(t[255]) - int(t[253])) / 100.0
    print(f"  total first->last stamp {int(t[2 * n - 1] - t[0])} cycles for {n} steps in {dt_us:.1f} us (100 MHz counter) = {int(t[2 * n - 1] - t[0]) / max(dt_us, 1e-9) / 1e3:.2f} GHz")
