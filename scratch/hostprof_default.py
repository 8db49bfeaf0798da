"""cProfile of the host side of the default-pointsf train step (where do the ~0.45 ms of launch gaps go?)."""
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ptranking_amd as pa
B, L, F = 1024, 128, 136
sf = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-4, "pointsf": dict(num_features=F, num_layers=5, AF='GE', TL_AF='S', apply_tl_af=True, BN=True, bn_type='BN', bn_affine=True)}
torch.manual_seed(0)
X = torch.randn(B, L, F, device="cuda")
Y = torch.sort(torch.randint(0, 5, (B, L), device="cuda").float(), dim=1, descending=True)[0]; Y[:, 0] = 2.0
r = pa.LambdaRank(sf_para_dict=sf, model_para_dict={"sigma": 1.0}, gpu=True, device="cuda:0")
r.init(); r.train_mode()
for _ in range(5):
    r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
