#!/usr/bin/env python3
"""The reference ITSELF (wildltr/ptranking imported read-only from /root/reference) timed beside the oracle's torch-CPU port of its
train step, same process, same synthetic MSLR-shaped batches, same thread count — run in the BUILD container (the reference cannot
travel to the GPU box).  Shows that bench.py's `cpu_baseline` (kind "port") is timing-faithful to the reference's train_op
(ptranking/base/ranker.py:589-603 followed by loss.item(), :579-584).  Writes profiles/r06_reference_vs_port_cpu.json.

    PYTHONDONTWRITEBYTECODE=1 python profiles/ref_vs_port_cpu.py
"""
import json
import os
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

from ptranking.data.data_utils import LABEL_TYPE
from ptranking.ltr_adhoc.listwise.lambdarank import LambdaRank
from oracle import torch_ref as T

SEED, L, F = 137, 128, 136
MSLR_P = [0.5147, 0.3250, 0.1339, 0.0183, 0.0081]


def synth(B):
    rng = np.random.default_rng(SEED)
    X = torch.from_numpy(rng.standard_normal((B, L, F)).astype(np.float32))
    Y = rng.choice(5, size=(B, L), p=MSLR_P).astype(np.float32)
    Y[:, 0] = np.maximum(Y[:, 0], 1)
    return X, torch.from_numpy(-np.sort(-Y, axis=1).copy())


def timed(fn, units, budget):
    for _ in range(3):
        fn()
    t0, it = time.perf_counter(), 0
    while True:
        fn()
        it += 1
        el = time.perf_counter() - t0
        if (el >= budget and it >= 10) or it >= 2000:
            return units * it / el, it, el


def main():
    threads = torch.get_num_threads()
    sf = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3,
          "pointsf": dict(num_features=F, num_layers=3, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None, bn_affine=False)}
    torch.manual_seed(SEED)
    ref = LambdaRank(sf_para_dict=sf, model_para_dict={"sigma": 1.0}, gpu=False, device="cpu")
    ref.init()
    ref.train_mode()
    net = T.build_pointsf(F, seed=SEED)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-3)
    rows = {}
    for B in (1, 64, 256):
        X, Y = synth(B)

        def ref_step():
            loss, _ = ref.train_op(X, Y, epoch_k=1, presort=True, label_type=LABEL_TYPE.MultiLabel)
            return loss.item()

        def port_step():
            return T.cpu_train_step(net, opt, X, Y, T.lambdarank_loss, sigma=1.0)

        r_q, r_it, r_el = timed(ref_step, B, 6.0)
        p_q, p_it, p_el = timed(port_step, B, 6.0)
        rows[f"B{B}"] = {"reference_queries_per_s": r_q, "reference_steps": r_it, "port_queries_per_s": p_q, "port_steps": p_it,
                         "port_over_reference": p_q / r_q}
        print(B, rows[f"B{B}"], flush=True)
    doc = {"note": "wildltr/ptranking's own LambdaRank.train_op + loss.item() (imported from /root/reference) vs oracle/torch_ref.cpu_train_step "
                   "(the `cpu_baseline` port of bench.py), same batches (MSLR-shaped synthetic, 128 docs x 136 feats), same process, build container",
           "host": {"threads": threads, "nproc": os.cpu_count(), "torch": torch.__version__},
           "rows": rows}
    with open(os.path.join(ROOT, "profiles", "r06_reference_vs_port_cpu.json"), "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
