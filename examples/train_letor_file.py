#!/usr/bin/env python3
"""End to end on one MI355X: LETOR text file -> device-resident padded batches -> LambdaRank training -> nDCG / AP evaluation.

    python examples/train_letor_file.py path/to/train.txt [path/to/test.txt] [--epochs 5] [--model LambdaRank] [--scaler StandardScaler]

What the reference does for the same job (ptranking/ltr_adhoc/eval/ltr.py:125-171, 300-360): parse the file token by token in
Python, pickle per-query tensors, batch equal-length queries only, copy every batch host->device inside the train loop, and
run the loss as ~15 ATen kernels over [B, L, L] tensors.  Here: the native parser (csrc/letor.cpp), one packing pass into a few
padded [B, Lp, F] device tensors (`PaddedQueryBatches`), the fused scorer / loss / optimiser kernels, and an evaluator that never
leaves the device until the final average.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a source checkout
import ptranking_amd as pa  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("train")
    ap.add_argument("test", nargs="?")
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--model", default="LambdaRank", choices=list(pa.RANKER_NAMES))
    ap.add_argument("--scaler", default=None, choices=[None, "StandardScaler", "MinMaxScaler", "RobustScaler", "SLog1P"])
    ap.add_argument("--rough-batch-size", type=int, default=262144, help="documents (incl. padding) per batch")
    ap.add_argument("--min-docs", type=int, default=10)
    ap.add_argument("--seed", type=int, default=137, help="torch seed: initial weights and dropout streams (ltr_global.py:7)")
    args = ap.parse_args()
    torch.manual_seed(args.seed)
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X: ptranking_amd has no CPU fallback")
    dev = "cuda:0"

    t0 = time.perf_counter()
    kw = dict(min_docs=args.min_docs, min_rele=1, scaler_id=args.scaler)
    train = pa.PaddedQueryBatches.from_letor_file(args.train, dev, rough_batch_size=args.rough_batch_size, presort=True, shuffle=True, **kw)
    test = pa.PaddedQueryBatches.from_letor_file(args.test or args.train, dev, rough_batch_size=args.rough_batch_size, presort=True, **kw)
    print(f"loaded {train.num_queries} / {test.num_queries} queries, {train.num_features} features, {len(train)} train batches "
          f"({100 * train.padded_fraction:.1f} % padding) in {time.perf_counter() - t0:.2f} s")

    sf = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3,
          "pointsf": dict(num_features=train.num_features, num_layers=3, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None,
                          bn_affine=False)}
    cls = getattr(pa, args.model)
    paras = pa.DEFAULT_PARAS[args.model]
    ranker = cls(sf_para_dict=sf, gpu=True, device=dev) if args.model in ("ListNet", "RankCosine", "RankMSE") else \
        cls(sf_para_dict=sf, model_para_dict=dict(paras), gpu=True, device=dev)
    ranker.init()
    ks = [1, 3, 5, 10]
    for epoch in range(1, args.epochs + 1):
        t0 = time.perf_counter()
        loss, stop = ranker.train(train, epoch_k=epoch, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ndcg = ranker.ndcg_at_ks(test_data=test, ks=ks, label_type=pa.LABEL_TYPE.MultiLabel, presort=True)
        print(f"epoch {epoch}: loss/query {float(loss):.4f}  {train.num_queries / dt:,.0f} queries/s  "
              + "  ".join(f"nDCG@{k} {v:.4f}" for k, v in zip(ks, ndcg.tolist())))
        if stop:
            break
    perf = ranker.adhoc_performance_at_ks(test_data=test, ks=ks, label_type=pa.LABEL_TYPE.MultiLabel, max_label=4.0, presort=True)
    names = ("nDCG", "nERR", "AP", "P")
    for name, vals in zip(names, perf[:4]):
        print(f"{name:5s}@{ks}: " + ", ".join(f"{v:.4f}" for v in vals.tolist()))


if __name__ == "__main__":
    main()
