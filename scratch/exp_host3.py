"""Host enqueue time per train step vs GPU time per step (is the step GPU-bound?)."""
import sys, time, torch
sys.path.insert(0, ".")
import ptranking_amd as pa
dev = "cuda:0"; B, L, F = 4096, 128, 136
sf = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3, "pointsf": dict(num_features=F, num_layers=3, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None, bn_affine=False)}
r = pa.LambdaRank(sf_para_dict=sf, model_para_dict={"sigma": 1.0}, gpu=True, device=dev); r.init(); r.train_mode()
X = torch.randn(B, L, F, device=dev); Y = torch.sort(torch.randint(0, 5, (B, L), device=dev).float(), dim=1, descending=True)[0].contiguous()
step = lambda: r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
for _ in range(50): step()
torch.cuda.synchronize()
for B2 in (4096, 1024, 256):
    Xs, Ys = X[:B2].contiguous(), Y[:B2].contiguous()
    st = lambda: r.train_op(Xs, Ys, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
    for _ in range(20): st()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n): st()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"B={B2}: host enqueue {1e3 * t_enq / n:.3f} ms/step, wall incl. GPU {1e3 * t_all / n:.3f} ms/step")
