"""Debug driver: where does the fused scorer path diverge from the CPU modules at bench scale?  (acts per layer, fused vs layer-wise backward)"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd import _lib
from ptranking_amd.scorer import FusedPointScorer
from ptranking_amd.host import build_pointsf

def run(R, F=136, NL=3, p=0.1):
    torch.manual_seed(1)
    fused = FusedPointScorer(F, NL, dropout=p).cuda()
    ref = build_pointsf(num_features=F, num_layers=NL, AF="R", BN=False, apply_tl_af=False, dropout=0.0)
    ref.load_state_dict({k: v.cpu() for k, v in fused.state_dict().items()})
    ref = ref.double()
    lin = [m for m in ref if isinstance(m, torch.nn.Linear)]
    X = torch.randn(R, F, device="cuda")
    seed = 424242 + R
    dev = X.device
    preds = torch.empty(R, device=dev); acts = torch.zeros((NL, R, 112), device=dev)
    st = _lib.current_stream(dev)
    _lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(fused.flat), R, F, NL, 1, C.c_float(p), C.c_uint64(seed), _lib.ptr(preds), _lib.ptr(acts), st)
    masks = [fused.dropout_mask(R, s, seed).cpu().double() for s in range(NL)]
    a = X.cpu().double() * masks[0] / (1 - p)
    a.requires_grad_(False)
    exp_acts = []
    h_in = a
    for l in range(NL):
        h = torch.relu(lin[l](h_in))
        h_in = h * masks[l + 1] / (1 - p) if l < NL - 1 else h
        exp_acts.append(h_in)
    exp = lin[NL](h_in).reshape(-1)
    print(f"R={R}: preds err {float((preds.cpu().double()-exp).abs().max()):.2e}")
    for l in range(NL):
        d = (acts[l, :, :100].cpu().double() - exp_acts[l].detach()).abs()
        bad = (d.max(dim=1)[0] > 1e-4).nonzero().flatten()
        print(f"   acts[{l}] max err {float(d.max()):.2e}; bad rows {bad.numel()} {bad[:12].tolist()}  ones col min/max {float(acts[l,:,100].min()):.1f} {float(acts[l,:,100].max()):.1f}")
    w = torch.randn(R, device=dev)
    (exp * w.cpu().double()).sum().backward()
    for mode in ("1", "0"):
        os.environ["PTR_BWD_FUSED"] = mode
        ndz = _lib.query("ptr_mlp_backward_dz_floats", R, F, NL)
        dz = torch.empty(ndz, device=dev) if ndz else None
        ws = torch.empty(_lib.query("ptr_mlp_backward_ws_floats", F, NL), device=dev)
        grad = torch.empty_like(fused.flat)
        _lib.call("ptr_mlp_backward", _lib.ptr(X), _lib.ptr(fused.flat), _lib.ptr(acts), _lib.ptr(w), R, F, NL, C.c_float(p), C.c_uint64(seed), _lib.ptr(dz), _lib.ptr(ws), _lib.ptr(grad), st)
        torch.cuda.synchronize()
        out = []
        for (k, o, s), (n, prm) in zip(fused.layout(), ref.named_parameters()):
            import math
            g = grad[o:o + math.prod(s)].view(s).cpu().double()
            e = (g - prm.grad).abs()
            out.append(f"{k}:{float(e.max()):.1e}/{float(prm.grad.abs().max()):.1e}")
        print(f"   bwd fused={mode}: " + " ".join(out))
        if mode == "1":
            g = grad[:100 * F].view(100, F).cpu().double(); e = (g - lin[0].weight.grad).abs()
            print("      dW0 err by out-feature tile:", [f"{float(e[16*m:16*m+16].max()):.1e}" for m in range(7)])
            print("      dW0 err by in-feature tile:", [f"{float(e[:,16*n:16*n+16].max()):.1e}" for n in range(9)])
    ref.zero_grad()

for R in [int(a) for a in sys.argv[1:]] or [2085, 8192, 8192 + 32, 16384, 65536]:
    run(R)
