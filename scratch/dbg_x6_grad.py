import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd.scorer import FusedPointScorer
for R in (131072, 131072 + 37, 4096):
    torch.manual_seed(3)
    fused = FusedPointScorer(136, num_layers=3, dropout=0.1).cuda(); fused.train()
    X = torch.randn(R, 136, device="cuda"); w = torch.randn(R, 1, device="cuda")
    g = {}
    for mode in ("0", "2"):
        os.environ["PTR_MLP_X6"] = mode
        fused.flat.grad = None
        orig = torch.randint; torch.randint = lambda *a, **k: torch.tensor([4242])
        out = fused(X); torch.randint = orig
        (out * w).sum().backward()
        g[mode] = fused.flat.grad.clone(); o = out.detach().clone() if mode == "0" else o
    d = (g["0"] - g["2"]).abs()
    print(f"R={R}: preds max diff {float((out.detach() - o).abs().max()):.2e}; grad max diff {float(d.max()):.3e} (max |grad| {float(g['0'].abs().max()):.3e}); W1 part {float(d[:13600].max()):.3e} b1 {float(d[13600:13700].max()):.3e} W2 {float(d[13700:23700].max()):.3e}")
