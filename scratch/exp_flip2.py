import os, sys, torch, torch.nn as nn
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptranking_amd.host import build_stacked_ffnet
dims, R = [136, 128, 256, 512, 136], 777
torch.manual_seed(sum(dims))
net = build_stacked_ffnet(dims, AF='R', TL_AF='R', apply_tl_af=True, dropout=0.0, BN=False)
lins = [m for m in net if isinstance(m, nn.Linear)]
for attempt in range(6):
    a = torch.randn(R, dims[0]).double()
    mins = []
    for l in lins:
        z = a @ l.weight.detach().double().t() + l.bias.detach().double()
        mins.append(float(z.abs().min())); a = torch.relu(z)
    print(attempt, ["%.1e" % m for m in mins], "units", [z.shape[1] for z in [l.weight for l in lins]])
