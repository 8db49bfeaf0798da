"""Is the ReLU-stack test's mismatch a gate flip?  Hidden pre-activations of the CPU reference (fp32 vs fp64) with the kernel's masks."""
import os, sys, ctypes as C, torch, torch.nn as nn
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ptranking_amd.host import build_stacked_ffnet
from ptranking_amd import _lib
dims, p, R = [136, 128, 256, 512, 136], 0.1, 777
torch.manual_seed(sum(dims))
net = build_stacked_ffnet(dims, AF='R', TL_AF='R', apply_tl_af=True, dropout=p, BN=False).cuda()
net.train()
x = torch.randn(3, R // 3, dims[0], device="cuda", requires_grad=True)
out = net(x); seed = net.last_seed
def mask(site, width):
    ones = torch.ones(R, width, device="cuda"); m = torch.empty_like(ones)
    _lib.call("ptr_dropout_apply", _lib.ptr(ones), width, R, width, C.c_float(p), C.c_uint64(seed), site, _lib.ptr(m), width, _lib.current_stream(ones.device))
    return (m > 0).cpu()
n = len(dims) - 1
masks = [mask(0, dims[0])] + [mask(i + 1, dims[i + 1]) for i in range(n - 2)]
lins = [m for m in net if isinstance(m, nn.Linear)]
for dt in (torch.float32, torch.float64):
    a = x.detach().cpu().reshape(R, dims[0]).to(dt) * masks[0].to(dt) / (1 - p)
    zs = []
    for i, l in enumerate(lins):
        z = a @ l.weight.detach().cpu().to(dt).t() + l.bias.detach().cpu().to(dt)
        zs.append(z)
        a = torch.relu(z)
        if i + 1 < len(masks): a = a * masks[i + 1].to(dt) / (1 - p)
    print(dt, "min |z| per layer:", [f"{float(z.abs().min()):.2e}" for z in zs], " out diff vs GPU:", float((zs[-1].relu().float() - out.detach().cpu().reshape(R, -1)).abs().max()))
    if dt == torch.float32: z32 = zs
    else:
        print("gate differences fp32 vs fp64 per layer:", [int(((a > 0) != (b > 0)).sum()) for a, b in zip(z32, zs)])
