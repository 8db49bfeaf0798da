"""time of the bf16x6 forward for the library named by PTR_LIB (ablation builds: scratch/ab_x6.sh): F = 136, 524288 rows"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import _lib
from ptranking_amd import scorer as _scorer
from ptranking_amd.scorer import FusedPointScorer, x6_workspace
NL, F, R = 3, int(os.environ.get("F", 136)), int(os.environ.get("R", 524288))
torch.manual_seed(0)
fused = FusedPointScorer(F, num_layers=NL, dropout=0.1).cuda()
Xs = [torch.randn(R, F, device="cuda") for _ in range(4)]
preds = torch.empty(R, device="cuda"); acts = _scorer.alloc_acts(R, NL, "cuda")
ws = x6_workspace(Xs[0].device, F, NL)
st = _lib.current_stream(Xs[0].device)
out = []
for train in (0, 1):
    def fwd(i):
        _lib.call("ptr_mlp_forward_x6", _lib.ptr(Xs[i % 4]), _lib.ptr(fused.flat.data), R, F, NL, train, C.c_float(0.1), C.c_uint64(7 + i), _lib.ptr(preds),
                  _lib.ptr(acts), _lib.ptr(ws), st)
    for i in range(3): fwd(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(30): fwd(i)
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 30 * 1e3)
print(f"{os.path.basename(os.environ.get('PTR_LIB', 'product')):40s} eval {out[0]:7.1f} us   train {out[1]:7.1f} us", flush=True)
