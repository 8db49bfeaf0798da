"""CPU, world_size 2, gloo: the data-parallel exchange step of the hot path (ptranking_amd.dp + FusedStepMixin).

The HIP loss kernels cannot run here, so `ptranking_amd.functional` is monkeypatched INSIDE THE TEST with the oracle's
torch-CPU restatement (test infrastructure); what is under test is the sharding + single flattened all-reduce +
ApproxNDCG batch-coupling algebra, which must reproduce the single-process full-batch step."""
import copy
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

SF = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-2,
      "pointsf": dict(num_features=10, num_layers=3, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None,
                      bn_affine=False, dropout=0.0)}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(B=8, L=12, F=10):
    rng = np.random.default_rng(3)
    X = torch.from_numpy(rng.standard_normal((B, L, F)).astype(np.float32))
    Y = rng.choice(5, size=(B, L), p=[0.5, 0.3, 0.15, 0.03, 0.02]).astype(np.float32)
    Y[:, 0] = np.maximum(Y[:, 0], 1)
    return X, torch.from_numpy(-np.sort(-Y, axis=1).copy())


def _patch_functional_with_oracle():
    """Route the two losses used below to the oracle's CPU autograd restatement (TEST ONLY)."""
    import ptranking_amd.functional as F_
    from oracle import torch_ref as T

    def lambdarank_loss(preds, labels, sigma=1.0, lens=None):
        return T.lambdarank_loss(preds, labels, sigma=sigma)

    def approxndcg_loss(preds, labels, alpha=10.0, presort=True, couple_batch=True, lens=None, grad_scale_override=0.0,
                        return_parts=False):
        ideal = labels if presort else torch.sort(labels, dim=1, descending=True)[0]
        inv = (1.0 / T.dcg_full(ideal)).view(-1)
        if couple_batch and grad_scale_override > 0:
            # scale-1 form: -(sum DCG) * override
            hat = T.approxndcg_loss(preds, labels, alpha=alpha, presort=presort, couple_batch=True)
            loss = hat / inv.sum() * grad_scale_override
        else:
            loss = T.approxndcg_loss(preds, labels, alpha=alpha, presort=presort, couple_batch=couple_batch)
        parts = dict(scale=torch.stack([inv.sum() if grad_scale_override <= 0 else torch.tensor(grad_scale_override), inv.sum()]))
        return (loss, parts) if return_parts else loss

    def rankmse_loss(preds, labels, lens=None):
        return T.rankmse_loss(preds, labels)

    F_.lambdarank_loss = lambdarank_loss
    F_.approxndcg_loss = approxndcg_loss
    F_.rankmse_loss = rankmse_loss


def _make(name):
    import ptranking_amd as pa
    torch.manual_seed(11)
    paras = {"LambdaRank": {"sigma": 1.0}, "ApproxNDCG": {"alpha": 10.0}, "RankMSE": None}[name]
    if paras is None:
        r = getattr(pa, name)(sf_para_dict=copy.deepcopy(SF), gpu=False, device="cpu")
    else:
        r = getattr(pa, name)(sf_para_dict=copy.deepcopy(SF), model_para_dict=paras, gpu=False, device="cpu")
    r.init()
    r.train_mode()
    return r


def _worker(rank, world, port, name, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import ptranking_amd as pa
    from ptranking_amd import dp
    _patch_functional_with_oracle()
    rk, ws, _ = dp.init_from_env(backend="gloo")
    assert (rk, ws) == (rank, world) and dp.is_distributed()
    X, Y = _data(B=9 if name == "RankMSE" else 8)          # RankMSE (a batch mean): unequal shards 5 + 4
    lo, hi = dp.shard_queries(X.size(0))
    r = _make(name)
    dp.broadcast_parameters(r.get_parameters())
    losses, grads1 = [], None
    for step in range(3):
        loss, _ = r.train_op(X[lo:hi], Y[lo:hi], epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
        losses.append(float(loss.detach()))
        if step == 0:
            grads1 = [p.grad.detach().clone() for p in r.get_parameters()]    # the all-reduced gradient of step 1
    torch.save({"params": [p.detach().clone() for p in r.get_parameters()], "losses": losses, "shard": (lo, hi), "grads1": grads1},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["LambdaRank", "ApproxNDCG", "RankMSE"])
def test_two_rank_step_equals_single_process_full_batch(name, tmp_path):
    import ptranking_amd as pa
    import ptranking_amd.functional as F_
    saved = (F_.lambdarank_loss, F_.approxndcg_loss, F_.rankmse_loss)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, name, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{i}.pt") for i in range(2))
    assert (r0["shard"], r1["shard"]) == (((0, 5), (5, 9)) if name == "RankMSE" else ((0, 4), (4, 8)))
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b), "replicas diverged"
    # single-process reference run on the full batch
    try:
        _patch_functional_with_oracle()
        X, Y = _data(B=9 if name == "RankMSE" else 8)
        r = _make(name)
        ref_losses, ref_grads1 = [], None
        for step in range(3):
            loss, _ = r.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
            ref_losses.append(float(loss.detach()))
            if step == 0:
                ref_grads1 = [p.grad.detach().clone() for p in r.get_parameters()]
    finally:
        F_.lambdarank_loss, F_.approxndcg_loss, F_.rankmse_loss = saved
    names = [n for n, _ in r.point_sf.named_parameters()]
    # the exchanged gradient equals the full-batch gradient (sum over queries is what the reference's losses compute)
    gmax = max(float(g.abs().max()) for g in ref_grads1)
    for n, a, b in zip(names, r0["grads1"], ref_grads1):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, gmax), n
    # parameters after 3 Adam steps: Adam normalises every coordinate, so near-zero gradients amplify rounding noise —
    # compare loosely (a small fraction of the 3*lr a coordinate can move)
    for n, a, b in zip(names, r0["params"], r.get_parameters()):
        assert float((a - b.detach()).abs().max()) <= 0.1 * 3 * SF["lr"], n
    if name == "LambdaRank":     # per-rank loss = loss of its shard; the shards sum to the full-batch loss
        for step in range(3):
            assert abs(r0["losses"][step] + r1["losses"][step] - ref_losses[step]) <= 1e-4 * abs(ref_losses[step])
    else:                        # ApproxNDCG / RankMSE under DP return the GLOBAL (coupled / batch-mean) loss on every rank
        for step in range(3):
            assert abs(r0["losses"][step] - ref_losses[step]) <= 1e-4 * abs(ref_losses[step])
            assert abs(r1["losses"][step] - ref_losses[step]) <= 1e-4 * abs(ref_losses[step])


def test_flat_bucket_views_and_sharding():
    from ptranking_amd import dp
    lin = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 1))
    b = dp.FlatGradBucket(list(lin.parameters()), extra=2)
    assert b.flat.numel() == 3 * 4 + 4 + 4 + 1 + 2 and b.extras.numel() == 2
    lin(torch.ones(2, 3)).sum().backward()
    off = 0
    for p in lin.parameters():
        assert p.grad.data_ptr() == b.flat[off:].data_ptr()        # gradients accumulate straight into the bucket
        off += p.numel()
    assert b.flat[:-2].abs().sum() > 0
    b.zero()
    assert b.flat.abs().sum() == 0 and all(p.grad is not None for p in lin.parameters())
    assert [dp.shard_queries(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert dp.world_size() == 1 and dp.rank() == 0 and not dp.is_distributed()


def test_row_offset_folding_and_view_bucket():
    """dp.fold_row_offset shifts the row index the dropout generator sees (csrc/ptr_dropout.h: key = row * ROW_KEY + ... + seed_lo);
    dp.ViewGradBucket is FlatGradBucket's interface over a buffer someone else owns (no re-pointing of .grad)."""
    from ptranking_amd import dp
    seed = (123456789 << 32) | 0xFFFFFF00
    for row0 in (0, 1, 4096 * 128, 2 ** 31 + 5):
        s2 = dp.fold_row_offset(seed, row0)
        assert s2 >> 32 == seed >> 32 and 0 <= s2 < 2 ** 62
        for row in (0, 7, 1000):                          # generator key of local row `row` under s2 == key of global row row0 + row under seed
            k_local = (row * dp.ROW_KEY + (s2 & 0xFFFFFFFF)) & 0xFFFFFFFF
            k_global = ((row0 + row) * dp.ROW_KEY + (seed & 0xFFFFFFFF)) & 0xFFFFFFFF
            assert k_local == k_global
    assert dp.local_dropout_seed(seed, 100) == seed        # single process: unchanged
    gbuf = torch.arange(14, dtype=torch.float32)
    zeroed = []
    b = dp.ViewGradBucket(gbuf, 10, 2, lambda: (gbuf[:10].zero_(), zeroed.append(1)))
    assert b.flat.numel() == 12 and b.extras.data_ptr() == gbuf[10:].data_ptr()
    b.extras[0] = 5.0
    b.zero()
    assert zeroed == [1] and float(gbuf[:12].abs().sum()) == 0 and float(gbuf[12]) == 12.0
    with pytest.raises(ValueError):
        dp.ViewGradBucket(gbuf, 13, 2, lambda: None)


def _uneven_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    from ptranking_amd import dp
    dp.init_from_env(backend="gloo")
    seed, L = (77 << 32) | 12345, 16
    lo, hi = dp.shard_queries(5)                      # 5 queries over 2 ranks: (0, 3) and (3, 5)
    got = dp.local_dropout_seed(seed, (hi - lo) * L)                 # pointwise scorer: rows = documents
    got_heads = dp.local_dropout_seed(seed, (hi - lo) * 2 * L)       # attention rows: queries x heads x documents
    # a batch with a DIFFERENT local query count whose rows merely divide by the recorded slice's is not placed by that slice
    other_q = 2 * (hi - lo)
    stale = dp.local_dropout_seed(seed, other_q * L, local_queries=other_q)
    assert stale == dp.fold_row_offset(seed, rank * other_q * L), (stale, rank)
    assert dp.local_dropout_seed(seed, (hi - lo) * L, local_queries=hi - lo) == got
    dp.end_step()                                                      # the record describes one step
    assert dp.QUERY_SHARD is None and dp.local_dropout_seed(seed, (hi - lo) * L) == dp.fold_row_offset(seed, rank * (hi - lo) * L)
    dp.shard_queries(5)
    torch.save({"lo": lo, "hi": hi, "seed": got, "want": dp.fold_row_offset(seed, lo * L), "seed_heads": got_heads,
                "want_heads": dp.fold_row_offset(seed, lo * 2 * L), "naive": dp.fold_row_offset(seed, rank * (hi - lo) * L)},
               os.path.join(out_dir, f"u{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_uneven_shards_get_disjoint_mask_windows(tmp_path):
    """with B % world != 0 shard_queries gives the remainder to the first ranks; `rank * local_rows` would start rank 1's dropout
    rows at 2 * 16 = 32 while rank 0 owns rows 0..47 — overlapping mask windows.  The slice recorded by shard_queries() places every replica
    at its true global row (queries in front x rows per query), for any rows-per-query unit, without a collective."""
    port = _free_port()
    mp.spawn(_uneven_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"u{r}.pt")) for r in range(2))
    assert (r0["lo"], r0["hi"], r1["lo"], r1["hi"]) == (0, 3, 3, 5)
    for r in (r0, r1):
        assert r["seed"] == r["want"] and r["seed_heads"] == r["want_heads"]
    assert r1["seed"] != r1["naive"]                  # rank 1: 3 * 16 = 48 rows in front of it, not 1 * 32
