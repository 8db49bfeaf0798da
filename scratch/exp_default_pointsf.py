"""Train-step time of the reference's DEFAULT pointsf (5 x [Linear -> BN -> GELU], Sigmoid tail) + LambdaRank: fused stack vs the same
modules executed by torch (library GEMMs + eager elementwise kernels)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ptranking_amd as pa
from ptranking_amd.linear import FusedStack
B, L, F = 1024, 128, 136
sf = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-4, "pointsf": dict(num_features=F, num_layers=5, AF='GE', TL_AF='S', apply_tl_af=True, BN=True, bn_type='BN', bn_affine=True)}
torch.manual_seed(0)
X = torch.randn(B, L, F, device="cuda")
Y = torch.sort(torch.randint(0, 5, (B, L), device="cuda").float(), dim=1, descending=True)[0]; Y[:, 0] = 2.0
for mode in ("fused stack", "torch modules"):
    r = pa.LambdaRank(sf_para_dict=sf, model_para_dict={"sigma": 1.0}, gpu=True, device="cuda:0")
    r.init(); r.train_mode()
    if mode == "torch modules":
        r.point_sf._plan = False          # module-by-module: torch elementwise + (hand-written) FusedLinear GEMMs
        for m in r.point_sf:
            if isinstance(m, torch.nn.Linear):
                m.forward = (lambda mm: (lambda x: torch.nn.functional.linear(x, mm.weight, mm.bias)))(m)   # library GEMM
    for _ in range(5):
        r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"default pointsf, {B} x {L} x {F}: {mode}: {dt*1e3:.3f} ms/step = {B/dt:.0f} q/s", flush=True)
