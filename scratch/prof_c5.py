import sys, copy, torch
sys.path.insert(0, ".")
import ptranking_amd as pa
dev = "cuda:0"; B, L, F, H = 1024, 256, 136, 2
listsf = dict(num_features=F, ff_dims=[128, 256, 512], AF='R', TL_AF='GE', apply_tl_af=False, BN=False, bn_type='BN2',
              bn_affine=False, n_heads=H, encoder_layers=6, encoder_type='DASALC')
sf = dict(sf_id='listsf', opt='Adagrad', lr=0.001, listsf=listsf)
r = pa.LambdaLoss(sf_para_dict=copy.deepcopy(sf), model_para_dict=dict(pa.DEFAULT_PARAS["LambdaLoss"]), gpu=True, device=dev)
r.init(); r.train_mode()
X = torch.randn(B, L, F, device=dev)
Y = torch.sort(torch.randint(0, 5, (B, L), device=dev).float(), dim=1, descending=True)[0].contiguous()
for _ in range(6):
    r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
torch.cuda.synchronize()
