"""Device-resident, padded query batches — the MI355X replacement for the reference's loader stack
LTRDataset + LETORSampler + DataLoader(num_workers=0) (ptranking/data/data_utils.py:553-742, ltr_adhoc/eval/ltr.py:125-154).

The reference can only batch queries with the SAME number of documents (LETORSampler groups by length; there is no padding or
mask anywhere in it) and copies every batch host->device synchronously in the train loop (ptranking/base/ranker.py:577).  On
real collections (MSLR-WEB30K: 1..1251 documents per query) that makes the loader, not the GPU, the bottleneck.  Here all
queries of a split are packed ONCE into a few length buckets of padded `[B, Lp, F]` / `[B, Lp]` device tensors plus an int32
`lens` vector; every HIP kernel of the path takes `lens` (padded documents are excluded from every sum and get gradient 0),
so an epoch touches host memory only for the batch ids.

`PaddedQueryBatches` is iterable like the reference's DataLoader but yields 4-tuples
    (batch_ids, batch_q_doc_vectors [B, Lp, F], batch_std_labels [B, Lp], lens int32 [B])
which `DeviceTrainLoop.train` and `DeviceEvaluator` accept next to the reference's 3-tuples.
"""
import math

import numpy as np
import torch


class PaddedQueryBatches:
    def __init__(self, queries, device, rough_batch_size=4096, pad_to=16, presort=True, shuffle=False, seed=137):
        """queries: iterable of (qid, features [n, F], labels [n]) with array-likes / tensors (what LTRDataset.__getitem__
        yields, data_utils.py:663-679).  rough_batch_size: documents (incl. padding) per batch, the meaning of the reference's
        `train_rough_batch_size` (data_utils.py:683-718).  pad_to: bucket granularity of the padded list length.
        presort: sort every query's documents by label, descending (what `train_presort` does, data_utils.py:500-516)."""
        self.device = torch.device(device)
        self.shuffle, self._rng = shuffle, np.random.default_rng(seed)
        buckets = {}
        self.num_queries = 0
        self.num_features = None
        for qid, feats, labels in queries:
            x = np.asarray(feats.cpu() if isinstance(feats, torch.Tensor) else feats, dtype=np.float32)
            y = np.asarray(labels.cpu() if isinstance(labels, torch.Tensor) else labels, dtype=np.float32).reshape(-1)
            n = y.shape[0]
            if n == 0:
                continue
            if x.shape != (n, x.shape[-1]):
                raise ValueError(f"query {qid}: features {x.shape} do not match {n} labels")
            if self.num_features is None:
                self.num_features = x.shape[1]
            if presort:
                order = np.argsort(-y, kind="stable")
                x, y = x[order], y[order]
            Lp = int(math.ceil(n / pad_to) * pad_to)
            buckets.setdefault(Lp, []).append((qid, x, y))
            self.num_queries += 1
        self._batches = []       # (ids, X, Y, lens) views into per-bucket device tensors
        for Lp in sorted(buckets):
            items = buckets[Lp]
            nq, F = len(items), self.num_features
            X = np.zeros((nq, Lp, F), np.float32)
            Y = np.zeros((nq, Lp), np.float32)
            lens = np.empty(nq, np.int32)
            ids = []
            for i, (qid, x, y) in enumerate(items):
                n = y.shape[0]
                X[i, :n], Y[i, :n], lens[i] = x, y, n
                ids.append(qid)
            Xd = torch.from_numpy(X).to(self.device)
            Yd = torch.from_numpy(Y).to(self.device)
            Ld = torch.from_numpy(lens).to(self.device)
            per = max(1, rough_batch_size // Lp)
            for lo in range(0, nq, per):
                hi = min(nq, lo + per)
                self._batches.append((ids[lo:hi], Xd[lo:hi], Yd[lo:hi], Ld[lo:hi]))

    @classmethod
    def from_letor_file(cls, path, device, rough_batch_size=4096, pad_to=16, presort=True, shuffle=False, seed=137,
                        **query_kwargs):
        """LETOR text file -> padded device batches.  The file is tokenised by the native parser (csrc/letor.cpp) and
        grouped / scaled / filtered by letor.load_letor_queries (**query_kwargs: min_docs, min_rele, binary_rele,
        unknown_as_zero, scaler_id, one_indexed, ...) — the job of LTRDataset.__init__ + iter_queries
        (ptranking/data/data_utils.py:420-549, :553-660)."""
        from .letor import load_letor_queries
        return cls(load_letor_queries(path, **query_kwargs), device, rough_batch_size=rough_batch_size, pad_to=pad_to,
                   presort=presort, shuffle=shuffle, seed=seed)

    def __len__(self):
        return len(self._batches)

    def __iter__(self):
        order = np.arange(len(self._batches))
        if self.shuffle:
            self._rng.shuffle(order)
        for i in order:
            yield self._batches[i]

    @property
    def padded_fraction(self):
        """Share of document slots that are padding."""
        tot = sum(x.shape[0] * x.shape[1] for _, x, _, _ in self._batches)
        real = sum(int(l.sum().item()) for _, _, _, l in self._batches)
        return 1.0 - real / max(1, tot)


def unpack_batch(batch):
    """(ids, X, Y) from the reference's loaders, or (ids, X, Y, lens) from PaddedQueryBatches -> always a 4-tuple."""
    if len(batch) == 4:
        return batch
    ids, X, Y = batch
    return ids, X, Y, None
