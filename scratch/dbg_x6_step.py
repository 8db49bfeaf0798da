"""ListNet direct train step (3 Adam steps, 512 x 256 x 136, dropout 0.1) with the fp32-MFMA / bf16x6 forward against CPU modules in float32 and
float64: how far do the parameters drift from each reference?  (what tests/test_regime_gpu.py's 3e-4 bound was calibrated on)"""
import copy, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_regime_gpu as TR
from oracle import torch_ref as T
import ptranking_amd as pa

def run(x6, dtype):
    os.environ["PTR_MLP_X6"] = x6
    F, NL, p, B, L = 136, 3, 0.1, 512, 256
    sf = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3, "pointsf": dict(num_features=F, num_layers=NL, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None, bn_affine=False, dropout=p)}
    torch.manual_seed(137)
    ranker = pa.ListNet(sf_para_dict=copy.deepcopy(sf), gpu=True, device="cuda:0"); ranker.init(); ranker.train_mode()
    fused = ranker.point_sf
    cpu_net = TR._cpu_modules(fused, F, NL, dtype)
    cpu_opt = torch.optim.Adam(cpu_net.parameters(), lr=1e-3, weight_decay=1e-3)
    rng = np.random.default_rng(5)
    X = torch.from_numpy(rng.standard_normal((B, L, F)).astype(np.float32))
    Y = rng.choice(5, size=(B, L), p=[0.5147, 0.3250, 0.1339, 0.0183, 0.0081]).astype(np.float32); Y[:, 0] = np.maximum(Y[:, 0], 1)
    Y = torch.from_numpy(-np.sort(-Y, axis=1))
    Xd, Yd = X.cuda(), Y.cuda()
    for step in range(3):
        torch.manual_seed(1000 + step); seed = int(torch.randint(0, 2 ** 62, (1,)).item()); torch.manual_seed(1000 + step)
        loss, stop = ranker.train_op(Xd, Yd, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
        preds = TR._masked_forward(cpu_net, fused, X.reshape(-1, F), seed, p, NL, dtype).view(B, L)
        ref_loss = T.listnet_loss(preds, Y.to(dtype)); cpu_opt.zero_grad(); ref_loss.backward(); cpu_opt.step()
    out = []
    for (n1, p1), (n2, p2) in zip(fused.state_dict().items(), cpu_net.named_parameters()):
        d = (p1.detach().cpu().double() - p2.detach().double()).abs()
        out.append(f"{n1}: max {float(d.max()):.2e} frac>2e-5 {float((d > 2e-5 + 1e-4 * p2.detach().abs()).double().mean()):.1e}")
    print(f"x6={x6} reference {dtype}: loss {loss.item():.6f} vs {ref_loss.item():.6f} | " + " | ".join(out[:2] + out[2:4:2]), flush=True)

for dtype in (torch.float32, torch.float64):
    for x6 in ("0", "2"):
        run(x6, dtype)
