"""r6: prologue / epilogue of mlp_bwd_x6_kernel<9> (PTR_LIB = a -DPTR_B6_TRACE -DPTR_B6_TRACE_EDGE build): shader-clock stamps of workgroup 0
   E0 entry | E1 LDS zeroed | E2 W^T fragments split | E3 first slab landed | (staging slab 0 + slab loop) | E4 loop done | E5 partial stored"""
import ctypes as C, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PTR_BWD_X6"] = "1"
from ptranking_amd import _lib
from ptranking_amd.scorer import FusedPointScorer, alloc_acts
NL, F = 3, 136
for R in (8192, 131072, 524288):
    torch.manual_seed(0)
    fused = FusedPointScorer(F, num_layers=NL, dropout=0.1).cuda()
    X = torch.randn(R, F, device="cuda"); dp = torch.randn(R, device="cuda")
    preds = torch.empty(R, device="cuda"); acts = alloc_acts(R, NL, "cuda")
    st = _lib.current_stream(X.device)
    os.environ["PTR_MLP_X6"] = "0"
    _lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(fused.flat.data), R, F, NL, 1, C.c_float(0.1), C.c_uint64(77), _lib.ptr(preds), _lib.ptr(acts), st)
    ws = torch.zeros(_lib.query("ptr_mlp_backward_ws_floats", F, NL), device="cuda")
    grad = torch.empty_like(fused.flat.data)
    for _ in range(3):
        _lib.call("ptr_mlp_backward", _lib.ptr(X), _lib.ptr(fused.flat.data), _lib.ptr(acts), _lib.ptr(dp), R, F, NL, C.c_float(0.1), C.c_uint64(77), None, _lib.ptr(ws), _lib.ptr(grad), st)
    torch.cuda.synchronize()
    NP = _lib.query("ptr_mlp_num_params", F, NL)
    base = 256 * NP
    ed = ws[base + 2 * 8 * 256: base + 2 * 8 * 256 + 2 * 8 * 16].cpu().numpy().view(np.uint64).reshape(8, 16).astype(np.int64)
    names = ["zero LDS", "W^T load + split", "first DMA + wait", "staging(0) + slab loop", "partial store"]
    for w in (0, 4, 7):
        d = np.diff(ed[w][:6])
        print(f"R={R} wave {w}: " + ", ".join(f"{n} {int(v)}" for n, v in zip(names, d)) + f" | total {int(ed[w][5] - ed[w][0])} cycles", flush=True)
