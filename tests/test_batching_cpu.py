"""CPU: packing logic of PaddedQueryBatches (the device-resident replacement of LTRDataset + LETORSampler + DataLoader)."""
import numpy as np
import torch

from ptranking_amd.batching import PaddedQueryBatches, unpack_batch


def _queries(lengths, F=6, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for i, n in enumerate(lengths):
        y = rng.integers(0, 5, n).astype(np.float32)
        y[rng.integers(n)] = max(1.0, y.max())
        out.append((f"q{i}", rng.standard_normal((n, F)).astype(np.float32), y))
    return out


def test_buckets_lens_presort_and_padding():
    lengths = [5, 17, 30, 31, 64, 100, 1, 32]
    qs = _queries(lengths)
    pb = PaddedQueryBatches(qs, "cpu", rough_batch_size=10 ** 6, pad_to=32, presort=True)
    assert pb.num_queries == len(lengths) and pb.num_features == 6
    seen = {}
    for ids, X, Y, lens in pb:
        Lp = X.shape[1]
        assert Lp % 32 == 0 and Y.shape == X.shape[:2] and lens.dtype == torch.int32
        for i, qid in enumerate(ids):
            n = int(lens[i])
            seen[qid] = (Lp, n)
            assert Lp - 32 < n <= Lp
            assert torch.all(Y[i, :n][:-1] >= Y[i, :n][1:])            # presort: labels descending
            assert torch.all(X[i, n:] == 0) and torch.all(Y[i, n:] == 0)  # padding
            src = dict((q, (x, y)) for q, x, y in qs)[qid]
            order = np.argsort(-src[1], kind="stable")
            assert np.array_equal(X[i, :n].numpy(), src[0][order]) and np.array_equal(Y[i, :n].numpy(), src[1][order])
    assert {q: v[1] for q, v in seen.items()} == {f"q{i}": n for i, n in enumerate(lengths)}
    assert len(pb) == 3                                                  # buckets 32, 64, 128
    real, slots = sum(lengths), 32 * 6 + 64 + 128
    assert abs(pb.padded_fraction - (1 - real / slots)) < 1e-9


def test_rough_batch_size_counts_padded_documents_and_shuffle_is_seeded():
    qs = _queries([16] * 10)
    pb = PaddedQueryBatches(qs, "cpu", rough_batch_size=64, pad_to=16, shuffle=True, seed=3)
    sizes = sorted(len(ids) for ids, _, _, _ in pb)
    assert sizes == [2, 4, 4]                                            # 64 // 16 = 4 queries per batch
    o1 = [tuple(ids) for ids, _, _, _ in pb]
    o2 = [tuple(ids) for ids, _, _, _ in pb]
    assert sorted(o1) == sorted(o2) and len(o1) == 3
    assert unpack_batch(("ids", 1, 2)) == ("ids", 1, 2, None) and unpack_batch((1, 2, 3, 4)) == (1, 2, 3, 4)
