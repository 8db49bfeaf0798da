#!/usr/bin/env python3
"""Golden fixtures for the LETOR text parser, produced by RUNNING THE REFERENCE's own parser (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_letor.py

Writes letor_sample.txt (a seeded synthetic LETOR file: sparse one-indexed features, exponent / negative / integer value
spellings, a query whose rows are split in two runs, queries that the min_docs / min_rele filters drop), its
zero-indexed Yahoo!-style twin letor_sample_zero.txt, and letor.npz holding what
  ptranking/data/data_utils.py:339-387  parse_letor      (dense float64 matrix, labels, qid strings)
  ptranking/data/data_utils.py:420-549  iter_queries     (grouping, query-level scaling, clipping, filtering)
return for them.
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

HERE = os.path.dirname(os.path.abspath(__file__))


def write_sample(path, one_indexed, seed=20260925):
    rng = np.random.default_rng(seed)
    F = 23
    plan = [(101, 7), (7, 12), (350, 1), (12, 9), (7, 5), (99, 6), (1000, 15), (64, 4)]   # qid 7 appears twice (split runs)
    lines = []
    for qi, (qid, n) in enumerate(plan):
        for d in range(n):
            if qid == 99:
                lab = 0                                   # no relevant document: dropped when min_rele >= 1
            else:
                lab = int(rng.integers(0, 5))
            toks = [str(lab), f"qid:{qid}"]
            present = np.flatnonzero(rng.random(F) < 0.7)
            if qi == 0 and d == 0:
                present = np.arange(F)                    # make sure the widest row defines F
            for f in present:
                v = rng.standard_normal() * 10.0 ** int(rng.integers(-3, 4))
                style = int(rng.integers(0, 5))
                if style == 0:
                    txt = f"{v:.6f}"
                elif style == 1:
                    txt = f"{v:.9e}"
                elif style == 2:
                    txt = repr(float(v))
                elif style == 3:
                    txt = str(int(v))
                else:
                    txt = f"{abs(v):.3f}"
                toks.append(f"{f + (1 if one_indexed else 0)}:{txt}")
            sep = "  " if d % 3 == 0 else " "
            lines.append(sep.join(toks))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def main():
    from ptranking.data.data_utils import parse_letor, iter_queries

    store = {}
    one = os.path.join(HERE, "letor_sample.txt")
    zero = os.path.join(HERE, "letor_sample_zero.txt")
    write_sample(one, True)
    write_sample(zero, False)
    for tag, path, one_indexed in (("one", one, True), ("zero", zero, False)):
        with open(path, encoding="iso-8859-1") as fo:
            X, y, qids = parse_letor(fo.readlines(), has_comment=False, one_indexed=one_indexed)
        store[f"{tag}/X"] = np.asarray(X, np.float64)
        store[f"{tag}/y"] = np.asarray(y, np.float64)
        store[f"{tag}/qids"] = np.asarray([int(q) for q in qids], np.int64)

    def run_iter(tag, data_id, scaler_id, min_docs, min_rele, binary_rele, unknown_as_zero, path):
        dd = dict(data_id=data_id, min_docs=min_docs, min_rele=min_rele, unknown_as_zero=unknown_as_zero,
                  binary_rele=binary_rele, has_comment=False)
        Qs = iter_queries(in_file=path, presort=False, data_dict=dd, scale_data=scaler_id is not None, scaler_id=scaler_id,
                          perquery_file=os.path.join(HERE, "_does_not_exist.np"), buffer=False)
        store[f"{tag}/n"] = np.int64(len(Qs))
        for i, (qid, fm, lv) in enumerate(Qs):
            store[f"{tag}/q{i}/qid"] = np.int64(int(qid))
            store[f"{tag}/q{i}/X"] = np.asarray(fm, np.float64)
            store[f"{tag}/q{i}/y"] = np.asarray(lv, np.float64)

    run_iter("iter_plain", "MQ2008_Super", None, 1, 1, False, False, one)
    run_iter("iter_filter", "MQ2008_Super", None, 5, 1, True, False, one)
    run_iter("iter_std", "MSLRWEB10K", "StandardScaler", 1, 1, False, False, one)
    run_iter("iter_minmax", "MSLRWEB10K", "MinMaxScaler", 2, 1, False, True, one)
    run_iter("iter_robust", "MSLRWEB10K", "RobustScaler", 1, 1, False, False, one)
    run_iter("iter_yahoo", "Set1", None, 1, 1, False, False, zero)
    np.savez_compressed(os.path.join(HERE, "letor.npz"), **store)
    print(f"letor.npz: {len(store)} arrays")


if __name__ == "__main__":
    main()
