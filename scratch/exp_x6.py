"""bf16x6 scorer forward (csrc/scorer_x6.hip) vs the fp32-MFMA forward: agreement with float64 CPU modules (same dropout masks) and time.
   python scratch/exp_x6.py [--time-only]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptranking_amd.scorer import FusedPointScorer  # noqa: E402
from ptranking_amd.host import build_pointsf  # noqa: E402


def ref64(fused, F, NL, X, seed, p, train):
    ref = build_pointsf(num_features=F, num_layers=NL, AF="R", BN=False, apply_tl_af=False, dropout=0.0).double()
    ref.load_state_dict({k: v.cpu().double() for k, v in fused.state_dict().items()})
    lin = [m for m in ref if isinstance(m, torch.nn.Linear)]
    R = X.shape[0]
    a = X.cpu().double()
    acts = []
    if train:
        a = a * fused.dropout_mask(R, 0, seed).cpu().double() / (1 - p)
    for l in range(NL):
        h = torch.relu(lin[l](a))
        a = h * fused.dropout_mask(R, l + 1, seed).cpu().double() / (1 - p) if (train and l < NL - 1) else h
        acts.append(a)
    return lin[NL](a), acts


def run(fused, X, seed, train, x6):
    os.environ["PTR_MLP_X6"] = "2" if x6 else "0"
    fused.train(train)
    orig = torch.randint
    torch.randint = lambda *a, **k: torch.tensor([seed])
    try:
        with torch.enable_grad() if train else torch.no_grad():
            out = fused(X)
    finally:
        torch.randint = orig
    return out


def check():
    worst = 0.0
    for (F, NL, R) in [(136, 3, 2085), (136, 3, 32), (136, 3, 31), (700, 3, 1111), (24, 2, 100), (256, 3, 640), (200, 4, 500), (132, 3, 777), (140, 5, 300),
                       (136, 3, 65536 + 37), (4, 2, 70), (32, 2, 64)]:
        torch.manual_seed(R)
        fused = FusedPointScorer(F, num_layers=NL, dropout=0.1).cuda()
        X = torch.randn(R, F, device="cuda")
        for train in (False, True):
            seed = 1234567 + R
            exp, eacts = ref64(fused, F, NL, X, seed, 0.1, train)
            res = {}
            for x6 in (False, True):
                try:
                    out = run(fused, X, seed, train, x6)
                except RuntimeError as e:
                    if "outside the fused scorer" not in str(e):
                        raise
                    res[x6] = float("nan")
                    continue
                err = float((out.detach().double().cpu().reshape(-1) - exp.reshape(-1)).abs().max())
                res[x6] = err
            sc = max(1.0, float(exp.abs().max()))
            worst = max(worst, res[True] / sc)
            print(f"F={F:4d} NL={NL} R={R:6d} train={int(train)}  max|err| fp32-mfma {res[False]:.3e}  x6 {res[True]:.3e}  (scale {sc:.2f})", flush=True)
        # stored activations of the x6 training forward against float64
        from ptranking_amd import _lib
        import ctypes as C
        from ptranking_amd.scorer import x6_workspace
        seed = 999 + R
        preds = torch.empty(R, device="cuda")
        acts = torch.full((NL, R, 112), float("nan"), device="cuda")
        ws = x6_workspace(X.device, F, NL)
        _lib.call("ptr_mlp_forward_x6", _lib.ptr(X), _lib.ptr(fused.flat.data), R, F, NL, 1, C.c_float(0.1), C.c_uint64(seed), _lib.ptr(preds),
                  _lib.ptr(acts), _lib.ptr(ws), _lib.current_stream(X.device))
        acts_old = torch.full((NL, R, 112), float("nan"), device="cuda")
        preds_old = torch.empty(R, device="cuda")
        try:
            _lib.call("ptr_mlp_forward", _lib.ptr(X), _lib.ptr(fused.flat.data), R, F, NL, 1, C.c_float(0.1), C.c_uint64(seed), _lib.ptr(preds_old),
                      _lib.ptr(acts_old), _lib.current_stream(X.device))
        except RuntimeError:
            acts_old = acts.clone()
        exp, eacts = ref64(fused, F, NL, X, seed, 0.1, True)
        for l in range(NL):
            e = float((acts[l, :, :100].double().cpu() - eacts[l]).abs().max())
            pad_same = torch.equal(acts[l, :, 100:], acts_old[l, :, 100:])
            print(f"    acts[{l}] max|err| {e:.3e}  padding columns identical to fp32-mfma kernel: {pad_same}", flush=True)
            worst = max(worst, e / max(1.0, float(eacts[l].abs().max())))
    print("WORST relative error", worst, "OK" if worst < 2e-5 else "FAIL")


def timeit():
    F, NL = 136, 3
    for R in (4096 * 128, 1024 * 128, 256 * 128):
        torch.manual_seed(0)
        fused = FusedPointScorer(F, num_layers=NL, dropout=0.1).cuda()
        Xs = [torch.randn(R, F, device="cuda") for _ in range(4 if R > 200000 else 8)]
        for train in (False, True):
            for x6 in (False, True):
                for i in range(3):
                    run(fused, Xs[i % len(Xs)], 5, train, x6)
                torch.cuda.synchronize()
                n = 20
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(n):
                    run(fused, Xs[i % len(Xs)], 5, train, x6)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / n
                flop = 2.0 * (100 * F + (NL - 1) * 100 * 100 + 100) * R
                print(f"R={R:7d} train={int(train)} x6={int(x6)}: {ms * 1e3:8.1f} us   {flop / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    t0 = time.time()
    if "--time-only" not in sys.argv:
        check()
    if "--check-only" not in sys.argv:
        timeit()
    print("done in", time.time() - t0, "s")
