#!/usr/bin/env python3
"""Golden fixtures for the WHOLE train step, produced by RUNNING THE REFERENCE's own rankers on CPU (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_step.py

r6.  Reference entry points exercised, unmodified and end to end: `NeuralRanker.init` (scorer construction `point_ranker.py:20-42` ->
`get_stacked_FFNet`, `config_optimizer` `ranker.py:512-525`: `optim.Adam(lr, weight_decay=1e-3)`), then three calls of
`NeuralRanker.train_op` (`ranker.py:589-603`) -> `custom_loss_function` (`lambdarank.py:27-62`, `ranknet.py:25-42`, `listnet.py:22-45`,
`lambdaloss.py:73-138`: loss, zero_grad, backward, optimizer.step).  Dropout is 0 (`sf_para_dict['pointsf']['dropout'] = 0.0`: different
generators cannot be pinned by a fixture; the GPU tests pin dropout against the CPU op sequence with the kernel's own masks).
Every case holds the initial state_dict, the batch, the loss the reference returned at each step and the state_dict after the third step.
The product's rankers (scorer kernels + fused loss kernel + fused backward / Adam, through ptr_train_step) must reproduce them
(tests/test_ranker_gpu.py::test_train_steps_match_the_reference_itself); both oracles' restatements are pinned to the same file on the CPU.
"""
import copy
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

from ptranking.data.data_utils import LABEL_TYPE  # noqa: E402
from ptranking.ltr_adhoc.pairwise.ranknet import RankNet  # noqa: E402
from ptranking.ltr_adhoc.listwise.lambdarank import LambdaRank  # noqa: E402
from ptranking.ltr_adhoc.listwise.lambdaloss import LambdaLoss  # noqa: E402
from ptranking.ltr_adhoc.listwise.listnet import ListNet  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 137
MSLR_P = [0.5147, 0.3250, 0.1339, 0.0183, 0.0081]
CASES = {                                   # name: (class, model_para_dict, B, L, F)
    "lambdarank_6x40": (LambdaRank, dict(model_id="LambdaRank", sigma=1.0), 6, 40, 136),
    "lambdarank_4x128": (LambdaRank, dict(model_id="LambdaRank", sigma=1.0), 4, 128, 136),
    "ranknet_8x32": (RankNet, dict(model_id="RankNet", sigma=1.0), 8, 32, 136),
    "listnet_4x256": (ListNet, None, 4, 256, 136),
    "lambdaloss_4x64": (LambdaLoss, dict(model_id="LambdaLoss", k=5, sigma=1.0, mu=5.0, loss_type="NDCG_Loss2"), 4, 64, 136),
}


def main():
    store = {}
    rng = np.random.default_rng(SEED + 31)
    for tag, (cls, mpd, B, L, F) in CASES.items():
        torch.manual_seed(SEED + len(tag))
        sf = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3,
              "pointsf": dict(num_features=F, h_dim=100, out_dim=1, num_layers=3, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None,
                              bn_affine=False, dropout=0.0)}
        ranker = cls(sf_para_dict=copy.deepcopy(sf), gpu=False, device="cpu") if mpd is None else \
            cls(sf_para_dict=copy.deepcopy(sf), model_para_dict=dict(mpd), gpu=False, device="cpu")
        ranker.init()
        ranker.train_mode()
        X = torch.from_numpy(rng.standard_normal((B, L, F)).astype(np.float32))
        Yn = rng.choice(5, size=(B, L), p=MSLR_P).astype(np.float32)
        Yn[:, 0] = np.maximum(Yn[:, 0], 1)
        Y = torch.from_numpy(-np.sort(-Yn, axis=1).copy())
        for k, v in ranker.point_sf.state_dict().items():
            store[f"{tag}/sd0/{k}"] = v.numpy().copy()
        losses = []
        for step in range(3):
            loss, stop = ranker.train_op(X, Y, epoch_k=1, presort=True, label_type=LABEL_TYPE.MultiLabel)
            assert stop is False
            losses.append(float(loss.item()))
        store[f"{tag}/X"] = X.numpy()
        store[f"{tag}/Y"] = Y.numpy()
        store[f"{tag}/losses"] = np.asarray(losses, np.float64)
        for k, v in ranker.point_sf.state_dict().items():
            store[f"{tag}/sd3/{k}"] = v.detach().numpy().copy()
        print(tag, losses)
    np.savez_compressed(os.path.join(HERE, "step.npz"), **store)
    print(f"step.npz: {len(store)} arrays; torch {torch.__version__}")


if __name__ == "__main__":
    main()
