"""GPU: padded device-resident batches (ids, X, Y, lens) give the same training signal and metrics as the reference-style
equal-length batches — the reference itself never pads (data_utils.py:683-742)."""
import copy

import numpy as np
import pytest
import torch

import golden_util as G

pytestmark = pytest.mark.gpu

SF = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3,
      "pointsf": dict(num_features=24, num_layers=3, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None,
                      bn_affine=False, dropout=0.0)}
LENGTHS = [5, 17, 30, 31, 64, 100, 12, 33]


def _queries(seed=0, F=24):
    rng = np.random.default_rng(seed)
    out = []
    for i, n in enumerate(LENGTHS):
        y = rng.integers(0, 5, n).astype(np.float32)
        y[rng.integers(n)] = max(1.0, y.max())
        y = -np.sort(-y)
        out.append((i, rng.standard_normal((n, F)).astype(np.float32), y))
    return out


def _ranker(name, paras):
    import ptranking_amd as pa
    torch.manual_seed(3)
    cls = getattr(pa, name)
    r = cls(sf_para_dict=copy.deepcopy(SF), gpu=True, device="cuda:0") if name == "ListNet" else \
        cls(sf_para_dict=copy.deepcopy(SF), model_para_dict=paras, gpu=True, device="cuda:0")
    r.init()
    r.train_mode()
    return r


@pytest.mark.parametrize("name,paras", [("LambdaRank", {"sigma": 1.0}), ("RankNet", {"sigma": 1.0}), ("ListNet", None),
                                         ("LambdaLoss", dict(k=5, sigma=1.0, mu=5.0, loss_type="NDCG_Loss2"))])
def test_padded_epoch_equals_per_query_epoch(name, paras):
    import ptranking_amd as pa
    qs = _queries()
    padded = pa.PaddedQueryBatches(qs, "cuda:0", rough_batch_size=10 ** 6, pad_to=32)
    per_query = [([qid], torch.from_numpy(x)[None], torch.from_numpy(y)[None]) for qid, x, y in qs]   # reference-style batches
    grads, losses = [], []
    for loader in (padded, per_query):
        r = _ranker(name, paras)
        r.optimizer.step = lambda *a, **k: None                          # accumulate gradients over the epoch, no update
        r.optimizer.zero_grad = lambda *a, **k: None
        loss, stop = r.train(loader, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
        grads.append(r.point_sf.flat.grad.detach().cpu().numpy().copy())
        losses.append(float(loss.item()))
    G.assert_close(losses[0], losses[1], "epoch loss")
    G.assert_close(grads[0], grads[1], "accumulated parameter gradient")


def test_padded_evaluation_equals_per_query_evaluation():
    import ptranking_amd as pa
    qs = _queries(seed=1)
    padded = pa.PaddedQueryBatches(qs, "cuda:0", rough_batch_size=64 * 3, pad_to=16)
    per_query = [([qid], torch.from_numpy(x)[None], torch.from_numpy(y)[None]) for qid, x, y in qs]
    r = _ranker("LambdaRank", {"sigma": 1.0})
    ks = [1, 3, 5, 10, 20, 50]
    a = r.adhoc_performance_at_ks(test_data=padded, ks=ks, label_type=pa.LABEL_TYPE.MultiLabel, presort=True, max_label=4.0)
    b = r.adhoc_performance_at_ks(test_data=per_query, ks=ks, label_type=pa.LABEL_TYPE.MultiLabel, presort=True, max_label=4.0)
    for x, y, m in zip(a, b, ("ndcg", "nerr", "ap", "p")):
        G.assert_close(x.numpy(), y.numpy(), m)
    # single cut-off: queries shorter than k are skipped per query (ranker.py:41-42)
    for k in (10, 20):
        x = r.ndcg_at_k(test_data=padded, k=k, label_type=pa.LABEL_TYPE.MultiLabel, presort=True)
        y = r.ndcg_at_k(test_data=per_query, k=k, label_type=pa.LABEL_TYPE.MultiLabel, presort=True)
        G.assert_close(x.numpy(), y.numpy(), f"ndcg@{k}")
    assert 0.0 < padded.padded_fraction < 0.5
