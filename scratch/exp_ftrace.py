"""Per-tile phase stamps of the scorer forward (PTR_LIB=...ftrace.so): layer 1 / hidden layers / epilogue, training vs eval."""
import os, sys, torch, ctypes as C
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptranking_amd import _lib
torch.manual_seed(0)
F, NL = 136, 3
R = 4096 * 128
X = torch.randn(R, F, device="cuda")
NP = _lib.query("ptr_mlp_num_params", F, NL)
P = torch.randn(NP, device="cuda") * 0.1
preds = torch.zeros(R + 8 + 8 * 16 * 8 * 2 + 64, device="cuda"); acts = torch.empty((NL, R, 112), device="cuda")
st = _lib.current_stream(X.device)
for train in (1, 0):
    for _ in range(2):
        _lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(P), R, F, NL, train, C.c_float(0.1), C.c_uint64(5), _lib.ptr(preds), _lib.ptr(acts), st)
    torch.cuda.synchronize()
    tr = preds[R + 4: R + 4 + 8 * 16 * 8 * 2].cpu().numpy().view(np.uint64).reshape(8, 16, 8).astype(np.int64)
    print("train" if train else "eval")
    for t in range(1, 5):
        d = np.diff(tr[t, :8, :4], axis=1)
        tot = tr[t, :8, 3] - tr[t, :8, 0]
        gap = tr[t, :8, 0] - tr[t - 1, :8, 3]
        mhz = (tr[t, :8, 3] - tr[t, :8, 0]) / np.maximum(tr[t, :8, 5] - tr[t, :8, 4], 1) * 100.0
        print(f"  tile {t}: shader clock {mhz.mean():6.0f} MHz  layer1 {d[:,0].mean():7.0f}  hidden {d[:,1].mean():7.0f}  epilogue {d[:,2].mean():6.0f}  total {tot.mean():7.0f}  gap-to-prev {gap.mean():6.0f}   per-wave totals {tot}")
