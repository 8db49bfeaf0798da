import copy, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptranking_amd as pa
def run(x6, nsteps):
    os.environ["PTR_MLP_X6"] = x6
    F, NL, p, B, L = 136, 3, 0.1, 512, 256
    sf = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3, "pointsf": dict(num_features=F, num_layers=NL, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None, bn_affine=False, dropout=p)}
    torch.manual_seed(137)
    ranker = pa.ListNet(sf_para_dict=copy.deepcopy(sf), gpu=True, device="cuda:0"); ranker.init(); ranker.train_mode()
    rng = np.random.default_rng(5)
    X = torch.from_numpy(rng.standard_normal((B, L, F)).astype(np.float32)).cuda()
    Y = rng.choice(5, size=(B, L), p=[0.5147, 0.3250, 0.1339, 0.0183, 0.0081]).astype(np.float32); Y[:, 0] = np.maximum(Y[:, 0], 1)
    Y = torch.from_numpy(-np.sort(-Y, axis=1)).cuda()
    out = []
    for step in range(nsteps):
        torch.manual_seed(1000 + step)
        loss, _ = ranker.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
        out.append((loss.item(), ranker.point_sf.flat.detach().clone(), ranker.point_sf.flat.grad.detach().clone()))
    return out
a, b = run("0", 3), run("2", 3)
for s in range(3):
    dp = (a[s][1] - b[s][1]).abs(); dg = (a[s][2] - b[s][2]).abs()
    w1 = dp[:13600].view(100, 136)
    rows = (w1.max(dim=1).values > 1e-4).nonzero().flatten().tolist()
    print(f"step {s}: loss {a[s][0]:.6f} / {b[s][0]:.6f}; param max diff {float(dp.max()):.2e}; grad max diff {float(dg.max()):.2e} (max |g| {float(a[s][2].abs().max()):.2e}); W1 rows off > 1e-4: {rows}; coords off in those rows: {[int((w1[r] > 1e-4).sum()) for r in rows]}")
    g1 = a[s][2][:13600].view(100, 136)
    if rows: print("      |grad| of the old run in the first such row (median, max):", float(g1[rows[0]].abs().median()), float(g1[rows[0]].abs().max()), " vs all rows median", float(g1.abs().median()))
