#!/usr/bin/env python3
"""Golden fixtures for the listwise scorer (listsf), produced by RUNNING THE REFERENCE's modules on CPU (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_listsf.py

Reference entry points exercised (ptranking/base/list_ranker.py):
  :152-174  LayerNorm                     (unbiased std, eps added to the std)
  :176-254  MultiheadAttention.forward    (eval mode: the Dropout on the attention probabilities is the identity)
  :284-378  ListNeuralRanker.ini_listsf / forward for encoder_type in DASALC / AllRank / AttnDIN (eval mode)
For every case the fixture holds the module's state_dict, the input, the output and the gradients of  sum(output * R)
(R a fixed random tensor) with respect to the input and to every parameter — autograd of the reference itself.
Training-mode dropout cannot be pinned by a fixture (different generators); the GPU tests pin it against the oracle with the
kernel's own mask instead.
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference"
if not os.path.isdir(REF):
    raise SystemExit("the reference tree is only mounted in the build container")
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 137


def put_module(store, tag, module, x, R):
    x = x.clone().requires_grad_(True)
    y = module(x)
    (y * R).sum().backward()
    store[f"{tag}/x"] = x.detach().numpy()
    store[f"{tag}/R"] = R.numpy()
    store[f"{tag}/y"] = y.detach().numpy()
    store[f"{tag}/dx"] = x.grad.numpy()
    for k, v in module.state_dict().items():
        store[f"{tag}/sd/{k}"] = v.numpy()
    for k, p in module.named_parameters():
        store[f"{tag}/grad/{k}"] = p.grad.numpy()


def main():
    from ptranking.base.list_ranker import LayerNorm, MultiheadAttention, ListNeuralRanker

    torch.manual_seed(SEED)
    store = {}
    # ---- LayerNorm
    for ci, (shape, F) in enumerate([((3, 7, 24), 24), ((2, 33, 136), 136), ((5, 2), 2)]):
        ln = LayerNorm(F)
        with torch.no_grad():
            ln.a_2.copy_(torch.randn(F)); ln.b_2.copy_(torch.randn(F))
        put_module(store, f"layernorm/c{ci}", ln, torch.randn(*shape) * 3 + 1, torch.randn(*shape))
    # ---- MultiheadAttention (eval)
    for ci, (B, L, F, H) in enumerate([(2, 10, 24, 2), (2, 37, 136, 2), (1, 70, 68, 4), (3, 16, 40, 5), (2, 130, 32, 1)]):
        m = MultiheadAttention(hid_dim=F, n_heads=H, dropout=0.1, device="cpu")
        m.eval()
        put_module(store, f"mhsa/c{ci}_h{H}", m, torch.randn(B, L, F), torch.randn(B, L, F))
        store[f"mhsa/c{ci}_h{H}/n_heads"] = np.int32(H)
    # ---- whole listsf scorer (eval)
    for enc in ("DASALC", "AllRank", "AttnDIN"):
        F = 24
        listsf = dict(num_features=F, ff_dims=[16, 32], AF='R', TL_AF='GE', apply_tl_af=False, BN=False, bn_type='BN2',
                      bn_affine=False, n_heads=2, encoder_layers=2, encoder_type=enc)
        sf = dict(sf_id='listsf', opt='Adagrad', lr=0.001, listsf=listsf)
        r = ListNeuralRanker(sf_para_dict=sf, gpu=False, device="cpu")
        r.init()
        r.eval_mode()
        x = torch.randn(2, 9, F)
        R = torch.randn(2, 9)
        preds = r.forward(x)
        (preds * R).sum().backward()
        tag = f"listsf/{enc}"
        store[f"{tag}/x"] = x.numpy(); store[f"{tag}/R"] = R.numpy(); store[f"{tag}/preds"] = preds.detach().numpy()
        for part in ("head_ffnns", "encoder", "tail_ffnns"):
            for k, v in r.list_sf[part].state_dict().items():
                store[f"{tag}/sd/{part}/{k}"] = v.numpy()
            for k, p in r.list_sf[part].named_parameters():
                store[f"{tag}/grad/{part}/{k}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    np.savez_compressed(os.path.join(HERE, "listsf.npz"), **store)
    print(f"listsf.npz: {len(store)} arrays, torch {torch.__version__}")


if __name__ == "__main__":
    main()
