"""ORACLE (test infrastructure): ctypes binding of oracle/liboracle.so (the plain-C restatement, ltr_oracle.c).

numpy in, numpy out, CPU only.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
`build()` compiles the library with gcc via oracle/Makefile when it is missing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "liboracle.so")
_lib = None

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


def build(force=False):
    src = os.path.join(_DIR, "ltr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _DIR, "-B", "liboracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=_f32p):
    return None if a is None else a.ctypes.data_as(t)


def _lens(lens):
    return None if lens is None else np.ascontiguousarray(lens, dtype=np.int32)


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with code {rc}")


def sort_desc(preds, lens=None):
    preds = _f(preds); B, L = preds.shape; lens = _lens(lens)
    vals = np.empty((B, L), np.float32); idx = np.empty((B, L), np.int64)
    _chk(lib().orc_sort_desc(_p(preds), _p(lens, _i32p), B, L, _p(vals), _p(idx, _i64p)), "sort_desc")
    return vals, idx


def _pair_loss(fn, preds, labels, lens, *params):
    preds, labels = _f(preds), _f(labels); B, L = preds.shape; lens = _lens(lens)
    loss_q = np.empty(B, np.float32); grad = np.empty((B, L), np.float32)
    _chk(fn(_p(preds), _p(labels), _p(lens, _i32p), B, L, *params, _p(loss_q), _p(grad)), fn.__name__)
    return loss_q, grad


def ranknet(preds, labels, sigma=1.0, lens=None):
    return _pair_loss(lib().orc_ranknet, preds, labels, lens, C.c_float(sigma))


def lambdarank(preds, labels, sigma=1.0, lens=None):
    return _pair_loss(lib().orc_lambdarank, preds, labels, lens, C.c_float(sigma))


def lambdaloss(preds, labels, k=5, sigma=1.0, mu=5.0, loss_type=1, presort=True, lens=None):
    return _pair_loss(lib().orc_lambdaloss, preds, labels, lens, C.c_int(k), C.c_float(sigma), C.c_float(mu),
                      C.c_int(loss_type), C.c_int(int(presort)))


def approxndcg(preds, labels, alpha=10.0, presort=True, couple_batch=True, lens=None):
    """-> (loss_total float32, dcg_q[B], inv_idcg_q[B], grad[B,L])"""
    preds, labels = _f(preds), _f(labels); B, L = preds.shape; lens = _lens(lens)
    loss = np.empty(1, np.float32); dcg = np.empty(B, np.float32); inv = np.empty(B, np.float32)
    grad = np.empty((B, L), np.float32)
    _chk(lib().orc_approxndcg(_p(preds), _p(labels), _p(lens, _i32p), B, L, C.c_float(alpha), C.c_int(int(presort)),
                              C.c_int(int(couple_batch)), _p(loss), _p(dcg), _p(inv), _p(grad)), "approxndcg")
    return loss[0], dcg, inv, grad


def softrank(preds, labels, delta=2.0, top_k=None, lens=None):
    return _pair_loss(lib().orc_softrank, preds, labels, lens, C.c_float(delta), C.c_int(int(top_k) if top_k else 0))


def listnet(preds, labels, lens=None):
    return _pair_loss(lib().orc_listnet, preds, labels, lens)


def stlistnet(preds, labels, unif, temperature=1.0, lens=None):
    preds, labels, unif = _f(preds), _f(labels), _f(unif); B, L = preds.shape; lens = _lens(lens)
    loss_q = np.empty(B, np.float32); grad = np.empty((B, L), np.float32)
    _chk(lib().orc_stlistnet(_p(preds), _p(labels), _p(unif), _p(lens, _i32p), B, L, C.c_float(temperature), _p(loss_q), _p(grad)),
         "stlistnet")
    return loss_q, grad


def rankmse(preds, labels, lens=None):
    """-> (per-query sums of squared errors [B] — the batch loss is their mean —, grad incl. the 1/B)"""
    return _pair_loss(lib().orc_rankmse, preds, labels, lens)


def rankcosine(preds, labels, lens=None):
    return _pair_loss(lib().orc_rankcosine, preds, labels, lens)


def listmle(preds, perm, lens=None):
    preds = _f(preds); B, L = preds.shape; lens = _lens(lens)
    perm = np.ascontiguousarray(perm, dtype=np.int64)
    loss_q = np.empty(B, np.float32); grad = np.empty((B, L), np.float32)
    _chk(lib().orc_listmle(_p(preds), _p(perm, _i64p), _p(lens, _i32p), B, L, _p(loss_q), _p(grad)), "listmle")
    return loss_q, grad


def mdprank(preds, labels, perm, top_k=10, gamma=1.0, lens=None):
    preds, labels = _f(preds), _f(labels); B, L = preds.shape; lens = _lens(lens)
    perm = np.ascontiguousarray(perm, dtype=np.int64)
    loss_q = np.empty(B, np.float32); grad = np.empty((B, L), np.float32)
    _chk(lib().orc_mdprank(_p(preds), _p(labels), _p(perm, _i64p), _p(lens, _i32p), B, L, C.c_int(int(top_k) if top_k else 0),
                           C.c_float(gamma), _p(loss_q), _p(grad)), "mdprank")
    return loss_q, grad


def metrics_at_ks(preds, labels, ks, presort, max_label=None, lens=None, permutation_labels=False):
    """-> dict(ndcg, nerr, ap, p) of [B, len(ks)] float32."""
    preds, labels = _f(preds), _f(labels); B, L = preds.shape; lens = _lens(lens)
    ks = np.ascontiguousarray(ks, dtype=np.int32); nk = len(ks)
    out = {m: np.empty((B, nk), np.float32) for m in ("ndcg", "nerr", "ap", "p")}
    ml = -1.0 if max_label is None else float(max_label)
    _chk(lib().orc_metrics_at_ks(_p(preds), _p(labels), _p(lens, _i32p), B, L, _p(ks, _i32p), nk, C.c_int(int(presort)),
                                 C.c_int(int(permutation_labels)), C.c_float(ml), _p(out["ndcg"]), _p(out["nerr"]), _p(out["ap"]), _p(out["p"])), "metrics")
    return out
