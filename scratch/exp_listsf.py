"""Timing of the fused listsf pieces at BASELINE config 5 (L=256, F=136, 2 heads) vs the eager composition the reference runs."""
import sys, time, copy
import torch
sys.path.insert(0, ".")
import ptranking_amd as pa
from ptranking_amd import listsf as LS, _lib

dev = "cuda:0"
B, L, F, H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 256, 136, 2
torch.manual_seed(0)
q, k, v, g = (torch.randn(B, L, F, device=dev) for _ in range(4))


def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def eager(qq, kk, vv, p):
    dh = F // H
    Q = qq.view(B, L, H, dh).permute(0, 2, 1, 3); K = kk.view(B, L, H, dh).permute(0, 2, 1, 3); V = vv.view(B, L, H, dh).permute(0, 2, 1, 3)
    att = torch.matmul(Q, K.permute(0, 1, 3, 2)) / (dh ** 0.5)
    att = torch.nn.functional.dropout(torch.softmax(att, dim=-1), p, True)
    return torch.matmul(att, V).permute(0, 2, 1, 3).contiguous().view(B, L, F)


for p in (0.0, 0.1):
    qd, kd, vd = (t.clone().requires_grad_(True) for t in (q, k, v))
    f_fwd = lambda: LS.mhsa_core(qd, kd, vd, H, p_drop=p, seed=7, site=0)
    t_f = timeit(f_fwd)
    def fb():
        o = LS.mhsa_core(qd, kd, vd, H, p_drop=p, seed=7, site=0); o.backward(g)
    t_fb = timeit(fb)
    e_f = timeit(lambda: eager(qd, kd, vd, p))
    def efb():
        o = eager(qd, kd, vd, p); o.backward(g)
    e_fb = timeit(efb)
    flop_f = 4.0 * B * H * L * L * (F // H)
    print(f"p={p}: fused fwd {t_f:.3f} ms ({flop_f / t_f / 1e9:.1f} TFLOP/s useful), fwd+bwd {t_fb:.3f} ms | eager fwd {e_f:.3f} fwd+bwd {e_fb:.3f} ms")

# per-kernel timing through the TIMING hook
_lib.TIMING = {}
qd, kd, vd = (t.clone().requires_grad_(True) for t in (q, k, v))
for _ in range(5):
    o = LS.mhsa_core(qd, kd, vd, H, p_drop=0.1, seed=7, site=0); o.backward(g)
torch.cuda.synchronize()
for name, evs in _lib.TIMING.items():
    ts = [a.elapsed_time(b) for a, b in evs][2:]
    print(f"  {name}: {sum(ts) / len(ts):.3f} ms")
_lib.TIMING = None

x = torch.randn(B * L, F, device=dev, requires_grad=True)
ln = LS.LayerNorm(F).to(dev)
def lnfb():
    y = ln(x); y.backward(g.view(-1, F))
def lneager():   # the eager composition the reference runs (list_ranker.py:170-174)
    mean = x.mean(-1, keepdim=True); std = x.std(-1, keepdim=True)
    y = ln.a_2 * (x - mean) / (std + 1e-6) + ln.b_2; y.backward(g.view(-1, F))
print(f"LayerNorm fwd+bwd fused {timeit(lnfb):.3f} ms, eager {timeit(lneager):.3f} ms  ({B * L * F * 4 / 1e6:.0f} MB tensor)")

# whole C5 step
listsf = dict(num_features=F, ff_dims=[128, 256, 512], AF='R', TL_AF='GE', apply_tl_af=False, BN=False, bn_type='BN2',
              bn_affine=False, n_heads=H, encoder_layers=6, encoder_type='DASALC')
sf = dict(sf_id='listsf', opt='Adagrad', lr=0.001, listsf=listsf)
r = pa.LambdaLoss(sf_para_dict=copy.deepcopy(sf), model_para_dict=dict(pa.DEFAULT_PARAS["LambdaLoss"]), gpu=True, device=dev)
r.init(); r.train_mode()
X = torch.randn(B, L, F, device=dev)
Y = torch.sort(torch.randint(0, 5, (B, L), device=dev).float(), dim=1, descending=True)[0].contiguous()
step = lambda: r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
t = timeit(step, n=5, warm=2)
print(f"C5 train step (listsf DASALC 6 layers + LambdaLoss): {t:.2f} ms -> {B / t * 1e3:.0f} queries/s")
