"""r6: where a small-batch LambdaRank train step spends its time: B queries x 128 x 136.
   (a) train_op with the single C-ABI call  (b) train_op with the three calls  (c) the bare ptr_train_step call in a loop (no Python logic at all)
   (d) host time of one train_op (no GPU wait: time to ENQUEUE)"""
import ctypes as C, os, sys, time, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptranking_amd as pa
from ptranking_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
L, F = 128, 136
SF = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3, "pointsf": dict(num_features=F, num_layers=3, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None, bn_affine=False, dropout=0.1)}
torch.manual_seed(137)
r = pa.LambdaRank(sf_para_dict=copy.deepcopy(SF), model_para_dict=dict(sigma=1.0), gpu=True, device="cuda:0"); r.init(); r.train_mode()
X = torch.randn(B, L, F, device="cuda"); Y = torch.randint(0, 5, (B, L), device="cuda").float().sort(dim=1, descending=True)[0].contiguous()
kw = dict(epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
def loop(n):
    acc = torch.zeros((), device="cuda")
    for _ in range(n):
        loss, _ = r.train_op(X, Y, **kw)
        acc += loss.detach()
    return acc
def timed(n):
    loop(20); torch.cuda.synchronize()
    t0 = time.perf_counter(); loop(n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6
r.single_call_step = True;  a = timed(N)
r.single_call_step = False; b = timed(N)
r.single_call_step = True
d = next(iter(r._direct_buffers.values()))["desc"]
fn = _lib.load().ptr_train_step
st = _lib.current_stream(X.device)
def bare(n):
    for _ in range(n):
        d.step += 1
        fn(C.addressof(d), st)
bare(20); torch.cuda.synchronize()
t0 = time.perf_counter(); bare(N); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"B={B}: single-call step {a[0]:.1f} us (host enqueue {a[1]:.1f}) | three-call step {b[0]:.1f} us (host {b[1]:.1f}) | bare ptr_train_step loop {(t2 - t0) / N * 1e6:.1f} us (host {(t1 - t0) / N * 1e6:.1f})", flush=True)
