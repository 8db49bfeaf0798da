"""gradient error of (fp32-MFMA | bf16x6) forward + fused backward against float64 CPU modules with the kernel's masks; rows with a
pre-activation within 1e-5 of a ReLU kink are given zero upstream gradient (tests/test_regime_gpu.py's screening) so that gate flips do not count"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_regime_gpu as TR
from ptranking_amd.scorer import FusedPointScorer
F, NL, p, R = 136, 3, 0.1, 131072
torch.manual_seed(11)
fused = FusedPointScorer(F, NL, dropout=p).cuda(); fused.train()
ref = TR._cpu_modules(fused, F, NL, torch.float64)
seed = 424242
masks = TR._masks(fused, R, seed, p, NL)
Xc = torch.randn(R, F)
TR._screen_relu_kinks(Xc, ref, masks, p, NL)
X = Xc.cuda()
w = torch.randn(R, 1)
exp = TR._masked_forward(ref, fused, Xc, seed, p, NL, torch.float64, masks=masks)
(exp * w.double()).sum().backward()
gref = torch.cat([q.grad.reshape(-1) for q in ref.parameters()])
for mode in ("0", "2"):
    os.environ["PTR_MLP_X6"] = mode
    fused.flat.grad = None
    orig = torch.randint; torch.randint = lambda *a, **k: torch.tensor([seed])
    out = fused(X); torch.randint = orig
    (out * w.cuda()).sum().backward()
    g = fused.flat.grad.detach().cpu().double()
    e = (g - gref).abs()
    print(f"x6={mode}: preds max err {float((out.detach().cpu().double() - exp).abs().max()):.2e}; grad: max err {float(e.max()):.3e}, rms err {float(e.pow(2).mean().sqrt()):.3e}, max |g| {float(gref.abs().max()):.3e}, rms |g| {float(gref.pow(2).mean().sqrt()):.3e}; W1 rms err {float(e[:13600].pow(2).mean().sqrt()):.3e}")
