"""Dump the phase stamps of the fused backward (PTR_LIB=...trace.so): per slab and wave, cycles spent per phase."""
import os, sys, torch, ctypes as C
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptranking_amd import _lib
torch.manual_seed(0)
F, NL = 136, 3
R = 4096 * 128
X = torch.randn(R, F, device="cuda"); dp = torch.randn(R, device="cuda")
NP = _lib.query("ptr_mlp_num_params", F, NL)
P = torch.randn(NP, device="cuda") * 0.1
preds = torch.empty(R, device="cuda"); acts = torch.empty(NL * ((R + 15) // 16) * 16 * 112, device="cuda")
ws = torch.zeros(_lib.query("ptr_mlp_backward_ws_floats", F, NL), device="cuda"); grad = torch.empty(NP, device="cuda")
st = _lib.current_stream(X.device)
_lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(P), R, F, NL, 1, C.c_float(0.1), C.c_uint64(5), _lib.ptr(preds), _lib.ptr(acts), st)
for _ in range(2):
    _lib.call("ptr_mlp_backward", _lib.ptr(X), _lib.ptr(P), _lib.ptr(acts), _lib.ptr(dp), R, F, NL, C.c_float(0.1), C.c_uint64(5), None, _lib.ptr(ws), _lib.ptr(grad), st)
torch.cuda.synchronize()
tr = ws[256 * NP:256 * NP + 8 * 8 * 16 * 2].cpu().numpy().view(np.uint64).reshape(8, 8, 16).astype(np.int64)
names = ["chainA", "waitA", "chainB", "waitB", "finX", "P3", "top", "waitC"]
t0 = tr[0, :, 0].min()
for s_ in range(2, 5):
    print(f"slab {s_}: start {tr[s_, :, 0] - t0}")
    d = np.diff(tr[s_, :, :9], axis=1)
    for w in range(8):
        print("   w%d " % w + " ".join(f"{names[i]}={d[w, i]:5d}" for i in range(8)) + f"  total={tr[s_, w, 8] - tr[s_, w, 0]}")
# sustained shader clock: shader-clock stamps against the 100 MHz real-time counter (slot 15) between the starts of slabs 1 and 7
mhz = (tr[7, :, 0] - tr[1, :, 0]) / np.maximum(tr[7, :, 15] - tr[1, :, 15], 1) * 100.0
print("shader clock inside mlp_bwd_fused_kernel: %.0f MHz (per wave %s)" % (mhz.mean(), np.round(mhz).astype(int)))
