"""LETOR text files -> per-query arrays -> device-resident padded batches, through the native parser.

Host-side mirror of the reference's file loading for the path (ptranking/data/data_utils.py):
  parse_letor_file      <- iter_lines / parse_letor (:276-387), done by the multi-threaded C++ parser behind the C ABI
                           (ptr_letor_scan / ptr_letor_load, csrc/letor.cpp) instead of a per-token Python loop;
  load_letor_queries    <- iter_queries + clip_query_data (:389-549): group rows by qid (first-seen order, also when a
                           qid's rows are not contiguous), optional query-level scaling, binary / unknown-as-zero label
                           clipping, min_docs / min_rele filtering, presort by label;
  PaddedQueryBatches.from_letor_file (batching.py) packs the result for the GPU.
The per-query buffering to pickle files (:432, :541-547) is not reproduced: parsing MSLR-WEB30K takes seconds here.
"""
import ctypes as C

import numpy as np

from . import _lib

SCALER_ID = ("MinMaxScaler", "RobustScaler", "StandardScaler", "SLog1P")   # data_utils.py:36
ISTELLA_MAX = 1000000.0                                                     # data_utils.py:38 (clip before scaling)


def parse_letor_file(path, one_indexed=True, missing=0.0, dtype=np.float32):
    """-> (X [n_docs, F] dtype, y float32 [n_docs], qids int64 [n_runs], qoff int64 [n_runs+1]).
    Runs = maximal blocks of consecutive rows with the same qid, in file order.  dtype float32 gives the matrix the
    reference ends up with after its FloatTensor cast; float64 gives parse_letor's own matrix (needed before scaling)."""
    lib = _lib.load()
    bpath = str(path).encode()
    n_docs, n_feat, n_q = C.c_int64(), C.c_int32(), C.c_int64()
    rc = lib.ptr_letor_scan(bpath, int(one_indexed), C.addressof(n_docs), C.addressof(n_feat), C.addressof(n_q))
    if rc != 0:
        raise ValueError(lib.ptr_last_error().decode())
    dtype = np.dtype(dtype)
    if dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise ValueError("dtype must be float32 or float64")
    X = np.empty((n_docs.value, n_feat.value), dtype)
    y = np.empty(n_docs.value, np.float32)
    qids = np.empty(n_q.value, np.int64)
    qoff = np.empty(n_q.value + 1, np.int64)
    rc = lib.ptr_letor_load(bpath, int(one_indexed), float(missing), n_docs.value, n_feat.value, n_q.value,
                            X.ctypes.data, int(dtype == np.float64), y.ctypes.data, qids.ctypes.data, qoff.ctypes.data)
    if rc != 0:
        raise ValueError(lib.ptr_last_error().decode())
    return X, y, qids, qoff


def _scale_query(x, scaler_id):
    """Query-level feature scaling in float64, the arithmetic of the sklearn scalers the reference instantiates
    (data_utils.py:176-188) with their default arguments; constant columns are left centred, not divided by 0."""
    if scaler_id == "StandardScaler":
        mu = x.mean(axis=0)
        sd = np.sqrt(((x - mu) ** 2).mean(axis=0))
        sd[sd < 10 * np.finfo(np.float64).eps] = 1.0
        return (x - mu) / sd
    if scaler_id == "MinMaxScaler":
        lo, hi = x.min(axis=0), x.max(axis=0)
        rng = hi - lo
        rng[rng < 10 * np.finfo(np.float64).eps] = 1.0
        return (x - lo) / rng
    if scaler_id == "RobustScaler":
        q25, med, q75 = np.percentile(x, [25.0, 50.0, 75.0], axis=0)
        iqr = q75 - q25
        iqr[iqr < 10 * np.finfo(np.float64).eps] = 1.0
        return (x - med) / iqr
    if scaler_id == "SLog1P":
        return np.sign(x) * np.log1p(np.abs(x))
    raise ValueError(f"unknown scaler {scaler_id!r}; expected one of {SCALER_ID}")


def load_letor_queries(path, min_docs=None, min_rele=None, binary_rele=False, unknown_as_zero=False, presort=False,
                       scaler_id=None, one_indexed=True, clip_features_max=None, rank_position_labels=False):
    """-> list of (qid int, features float32 [n, F], labels float32 [n]) — what iter_queries returns, minus the pickling.
    presort uses a stable sort by label, one of the orders np_arg_shuffle_ties (data_utils.py:228-249) samples from.
    clip_features_max: the ISTELLA guard (np.clip(..., a_max=ISTELLA_MAX) before scaling, :529-531).
    rank_position_labels: MSLETOR_LIST's conversion of rank positions into grades, label = n - r (:520-523)."""
    X, y, run_qids, qoff = parse_letor_file(path, one_indexed=one_indexed,
                                            dtype=np.float64 if scaler_id is not None else np.float32)
    # rows of one qid, first-seen order (the reference collects them in a dict, so split runs are merged)
    uniq, first, inv = np.unique(run_qids, return_index=True, return_inverse=True)
    if uniq.shape[0] != run_qids.shape[0]:
        run_len = np.diff(qoff)
        doc_rank = np.repeat(np.argsort(np.argsort(first))[inv], run_len)      # first-seen rank of every row's qid
        order = np.argsort(doc_rank, kind="stable")
        X, y = X[order], y[order]
        seen_order = np.argsort(first)
        qids = uniq[seen_order]
        counts = np.bincount(np.repeat(inv, run_len), minlength=uniq.shape[0])[seen_order]
        qoff = np.concatenate([[0], np.cumsum(counts)])
    else:
        qids = run_qids
    clip_query = bool(min_rele is not None and min_rele > 0) or bool(min_docs is not None and min_docs > 0)
    out = []
    for q in range(qids.shape[0]):
        lo, hi = int(qoff[q]), int(qoff[q + 1])
        x, lab = X[lo:hi], y[lo:hi].astype(np.float32, copy=True)
        if rank_position_labels:
            lab = (hi - lo) - lab
        if scaler_id is not None:
            if clip_features_max is not None:
                x = np.minimum(x, clip_features_max)
            x = _scale_query(x, scaler_id)
        if binary_rele:
            lab = np.clip(lab, -10, 1)
        if unknown_as_zero:
            lab = np.clip(lab, 0, 10)
        if clip_query:
            if min_docs is not None and (hi - lo) < min_docs:
                continue
            if min_rele is not None and int((lab > 0).sum()) < min_rele:
                continue
        x = np.ascontiguousarray(x, dtype=np.float32)
        if presort:
            o = np.argsort(-lab, kind="stable")
            x, lab = x[o], lab[o]
        out.append((int(qids[q]), x, lab))
    return out
