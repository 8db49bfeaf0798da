#!/usr/bin/env python3
"""Benchmark of the hot path: queries/sec of one LambdaRank train step (scorer forward + fused delta-NDCG loss/grad kernel +
scorer backward + optimiser step) on MSLR-WEB30K-shaped synthetic batches, list_len=128, 136 features (BASELINE.json
configs[1]).  One process per GPU:

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      the dominant kernel of the step (the fused fp32-MFMA scorer forward): algorithmic flops per launch / its
                average launch duration measured with HIP events on the launch stream during the timed region, vs the
                157.3 TFLOP/s fp32 MFMA peak; `kernels.lambdarank_loss_grad` carries the same for the north-star loss kernel
                against the HBM roofline (12*L+4 bytes per query)
  cpu_baseline  the oracle's torch-CPU restatement of the reference train step, timed on this box's host cores on a
                bounded sample of the same workload (rank 0, N=1 only)
Inputs are resident in HBM before the timed region; a step does no host synchronisation.
"""
import argparse
import gc
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3      # dense fp32-input MFMA peak = fp32 vector peak (same guide)
MSLR_P = [0.5147, 0.3250, 0.1339, 0.0183, 0.0081]   # label histogram of MSLR-WEB30K
SEED = 137                        # ptranking/ltr_global.py:7


def synth_batch(gen, B, L, F, device):
    """MSLR-shaped synthetic batch: X ~ N(0,1), graded labels sorted descending (presort=True), >= 1 relevant doc."""
    X = torch.randn((B, L, F), generator=gen, device=device, dtype=torch.float32)
    probs = torch.tensor(MSLR_P, device=device)
    Y = torch.multinomial(probs.expand(B, -1), L, replacement=True, generator=gen).float()
    Y[:, 0] = torch.clamp(Y[:, 0], min=1.0)
    Y, _ = torch.sort(Y, dim=1, descending=True)
    return X, Y.contiguous()


def sf_para_dict(F, lr=1e-3):
    return {"sf_id": "pointsf", "opt": "Adam", "lr": lr,
            "pointsf": dict(num_features=F, num_layers=3, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None,
                            bn_affine=False)}   # h_dim=100, dropout=0.1 are hard-wired (point_ranker.py:30-31)


def cpu_baseline(L, F, budget_s):
    """Torch-CPU restatement of the reference's LambdaRank train step (oracle/torch_ref.py), all host cores.  Timed at the batch
    sizes SURVEY.md 8(d) names: 1 (the reference's own default for lists of >= 100 documents, data_utils.py:713-716), 64 and 256;
    `value` is the best of them.  Also a loss-only timing (leaf preds -> loss -> backward), kernel against kernel."""
    from oracle import torch_ref as T
    torch.manual_seed(SEED)
    gen = torch.Generator().manual_seed(SEED)
    Xf, Yf = synth_batch(gen, 256, L, F, "cpu")
    net = T.build_pointsf(F, seed=SEED)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-3)

    def timed(fn, units, budget, max_iters):
        for _ in range(2):
            fn()
        t0 = time.perf_counter()
        it = 0
        while True:
            fn()
            it += 1
            el = time.perf_counter() - t0
            if el >= budget or it >= max_iters:
                return units * it / el, it, el

    by_batch = {}
    total = 0.0
    for B, share, cap in ((1, 0.15, 2000), (64, 0.2, 400), (256, 0.5, 200)):
        X, Y = Xf[:B].contiguous(), Yf[:B].contiguous()
        qps, it, el = timed(lambda: T.cpu_train_step(net, opt, X, Y, T.lambdarank_loss, sigma=1.0), B, share * budget_s, cap)
        by_batch[f"B{B}"] = {"queries_per_s": qps, "steps": it, "seconds": el}
        total += el
    preds = torch.randn(256, L, generator=gen)

    def loss_only():
        p = preds.clone().requires_grad_(True)
        T.lambdarank_loss(p, Yf, sigma=1.0).backward()

    lq, lit, lel = timed(loss_only, 256, 0.15 * budget_s, 400)
    best = max(by_batch.values(), key=lambda d: d["queries_per_s"])
    return {"value": best["queries_per_s"], "unit": "queries/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"torch-CPU restatement of the reference train_op (pointsf scorer + LambdaRank + Adam) on {L} docs x {F} feats: "
                      f"batch sizes 1 / 64 / 256 for {total:.1f} s in total, value = best; plus {lel:.1f} s loss-only",
            "by_batch": by_batch, "loss_only_queries_per_s": lq}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="queries per GPU per step (weak scaling)")
    ap.add_argument("--list-len", type=int, default=128)
    ap.add_argument("--features", type=int, default=136)
    ap.add_argument("--nbatches", type=int, default=4, help="distinct HBM-resident batches cycled through (> L3 capacity)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--loss", default="LambdaRank", choices=["RankNet", "LambdaRank", "LambdaLoss", "ApproxNDCG", "ListNet", "ListMLE"],
                    help="ranker of the train step (the headline metric is LambdaRank; the others cover BASELINE.json configs 3-5)")
    ap.add_argument("--scorer", default="pointsf", choices=["pointsf", "listsf"],
                    help="listsf = BASELINE.json config 5: 2-head / 6-layer DASALC encoder (fused MFMA attention), use with --loss LambdaLoss "
                         "--list-len 256 --batch 1024")
    args = ap.parse_args()

    import ptranking_amd as pa
    from ptranking_amd import _lib, dp

    rank, world, local = dp.init_from_env()
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    B, L, F = args.batch, args.list_len, args.features

    torch.manual_seed(SEED)                       # identical initial weights on every rank
    cls = getattr(pa, args.loss)
    if args.scorer == "listsf":      # ptranking/ltr_adhoc/eval/parameter.py:152-166 defaults
        sfd = {"sf_id": "listsf", "opt": "Adagrad", "lr": 1e-3,
               "listsf": dict(num_features=F, ff_dims=[128, 256, 512], AF="R", TL_AF="GE", apply_tl_af=False, BN=False, bn_type="BN2",
                              bn_affine=False, n_heads=2, encoder_layers=6, encoder_type="DASALC")}
    else:
        sfd = sf_para_dict(F)
    if args.loss == "ListNet":
        ranker = cls(sf_para_dict=sfd, gpu=True, device=device)
    else:
        ranker = cls(sf_para_dict=sfd, model_para_dict=dict(pa.DEFAULT_PARAS[args.loss]), gpu=True, device=device)
    if args.loss == "ListMLE":
        ranker.tie_shuffle = "device"             # the reference's B host-side randperm calls per step would dominate
    ranker.init()
    ranker.train_mode()                           # dropout 0.1 active, exactly like the reference's train()
    gen = torch.Generator(device=device).manual_seed(SEED + 1000 * rank)   # every rank owns different queries
    batches = [synth_batch(gen, B, L, F, device) for _ in range(max(1, args.nbatches))]

    def step(i):
        X, Y = batches[i % len(batches)]
        loss, _ = ranker.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
        return loss

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(n):
        """n train steps, accumulating the loss on the device exactly like DeviceTrainLoop.train does (no host sync)."""
        acc = torch.zeros((), device=device)
        for i in range(n):
            acc += step(i).detach()
        return acc

    # Untimed pre-warm with EXACTLY the code of the timed region (event hook included): the first launch of every kernel
    # lazily loads its code object (tens of ms each), and a fresh box needs a few hundred ms before clocks / allocator settle.
    _lib.TIMING = {}
    for _ in range(5):                # a FIXED count: every rank must issue the same number of all-reduces
        float(run_steps(20).item())
    _lib.TIMING = None
    gc.collect()
    gc.disable()                      # no collector pauses inside the timed region
    run_steps(args.warmup)
    sync()
    _lib.TIMING = {}
    t0 = time.perf_counter()
    epoch_loss = run_steps(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    timing, _lib.TIMING = _lib.TIMING, None
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(epoch_loss.item())
    if not np.isfinite(final_loss):
        raise SystemExit(f"non-finite loss {final_loss}")

    if rank == 0:
        def avg_ms(name):
            ev = timing.get(name, [])
            return float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else None

        R = B * L
        NL = 3
        fwd_flop = 2.0 * (100 * F + (NL - 1) * 100 * 100 + 100) * R          # algorithmic: 2*(100F + 2*100*100 + 100) per document
        step_ms = 1e3 * elapsed / args.steps
        loss_entry = {"RankNet": "ptr_ranknet_fwd_bwd", "LambdaRank": "ptr_lambdarank_fwd_bwd", "LambdaLoss": "ptr_lambdaloss_fwd_bwd",
                      "ApproxNDCG": "ptr_approxndcg_fwd_bwd", "ListNet": "ptr_listnet_fwd_bwd", "ListMLE": "ptr_listmle_fwd_bwd"}[args.loss]
        t_fwd, t_loss, t_bwd, t_adam = (avg_ms(n) for n in ("ptr_mlp_forward", loss_entry, "ptr_mlp_backward", "ptr_adam_step"))
        pmc = {}
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                j = json.load(f)
            if j["config"] == {"queries_per_gpu_per_step": B, "list_len": L, "features": F}:
                pmc = {k: v["hbm_bytes_per_launch"] for k, v in j["kernels"].items()}
        except (OSError, KeyError, ValueError):
            pass

        def pmc_bytes(prefix):
            for k, v in pmc.items():
                if k.startswith(prefix):
                    return v
            return None

        loss_bytes = B * (12 * L + 4)
        kernels = {}
        if t_loss:
            gbps = loss_bytes / (t_loss * 1e-3) / 1e9
            kernels["lambdarank_loss_grad" if args.loss == "LambdaRank" else "loss_grad"] = {
                "kernel": ("pairwise_bce_kernel<64,2,WEIGHTED> (fused LambdaRank dNDCG loss + gradient)" if args.loss == "LambdaRank"
                           else f"{loss_entry} (fused {args.loss} loss + gradient)"), "bound": "hbm",
                "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                "traffic": pmc_bytes("ptr::pairwise_bce_kernel"), "avg_launch_ms": t_loss, "algorithmic_bytes_per_launch": loss_bytes,
                "pairs_per_s": B * (L * (L - 1) // 2) / (t_loss * 1e-3),
                "note": "O(L^2) pair work per 12L+4 bytes: VALU/transcendental-bound by construction (DESIGN.md 3.1)"}
        if t_bwd:
            kernels["scorer_backward"] = {"kernels": "mlp_bwd_dz + 3 x mlp_bwd_dw + reduce_partials", "avg_call_ms": t_bwd,
                                          "algorithmic_flop_per_call": 2.0 * fwd_flop - 2.0 * 100 * F * R,
                                          "achieved_TFLOPs": (2.0 * fwd_flop - 2.0 * 100 * F * R) / (t_bwd * 1e-3) / 1e12}
        if t_adam:
            kernels["adam"] = {"avg_launch_ms": t_adam}
        if t_fwd:
            tf = fwd_flop / (t_fwd * 1e-3) / 1e12
            roofline = {"kernel": "mlp_fwd_kernel<2,TRAIN,VEC> (fused pointsf scorer forward, fp32 MFMA 16x16x4, dropout in-kernel)",
                        "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                        "traffic": pmc_bytes("ptr::mlp_fwd_kernel"), "avg_launch_ms": t_fwd, "algorithmic_flop_per_launch": fwd_flop,
                        "algorithmic_bytes_per_launch": R * (4 * F + 4) + NL * R * 448,
                        "note": "dominant kernel of the step by time; traffic = PMC FETCH_SIZE(x2 on gfx950)+WRITE_SIZE from profiles/r01_pmc_traffic.json"}
        elif args.scorer == "listsf" and avg_ms("ptr_mhsa_forward"):
            t_af, t_ab = avg_ms("ptr_mhsa_forward"), avg_ms("ptr_mhsa_backward")
            att_flop = 4.0 * B * L * L * F                         # QK^T and PV, 2 flop per MAC, all heads (H * d_h = F)
            tf = att_flop / (t_af * 1e-3) / 1e12
            c5_traffic = None
            try:
                with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_c5.json")) as f:
                    j5 = json.load(f)
                if j5["config"] == {"queries_per_gpu_per_step": B, "list_len": L, "features": F}:
                    c5_traffic = next((v["hbm_bytes_per_launch"] for k, v in j5["kernels"].items() if k.startswith("ptr::mhsa_fwd_kernel")), None)
            except (OSError, KeyError, ValueError):
                pass
            roofline = {"kernel": "mhsa_fwd_kernel (fused attention core: QK^T/sqrt(d) -> online softmax -> dropout -> PV, fp32 MFMA 16x16x4)",
                        "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                        "traffic": c5_traffic, "avg_launch_ms": t_af, "algorithmic_flop_per_launch": att_flop,
                        "algorithmic_bytes_per_launch": 4 * B * L * F * 4 + B * 2 * L * 4,
                        "note": "one launch per encoder layer; the linear / feed-forward GEMMs of listsf are library calls"}
            kernels["attention_backward"] = {"kernels": "attn_rowdot + mhsa_bwd_dq + mhsa_bwd_dkv", "avg_call_ms": t_ab,
                                             "achieved_TFLOPs": 2.5 * att_flop / (t_ab * 1e-3) / 1e12}
            for nm in ("ptr_layernorm_forward", "ptr_layernorm_backward"):
                if avg_ms(nm):
                    kernels[nm] = {"avg_launch_ms": avg_ms(nm), "hbm_GBps": (2 if "forward" in nm else 3) * B * L * F * 4 / (avg_ms(nm) * 1e-3) / 1e9}
        else:   # scorer configuration not fusable: the north-star loss kernel is the only kernel of ours in the step
            roofline = dict(kernels.get("lambdarank_loss_grad", kernels.get("loss_grad", {})))
        qps = world * B * args.steps / elapsed
        out = {
            "metric": ("queries/sec fwd+bwd LambdaRank, MSLR-WEB30K-shaped list_len=128"
                       if (args.loss, L, F, args.scorer) == ("LambdaRank", 128, 136, "pointsf")
                       else f"queries/sec fwd+bwd {args.loss}, synthetic list_len={L}, {F} feats"),
            "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{args.loss} train step (pointsf 3x100 ReLU scorer, dropout 0.1, Adam), " if args.scorer == "pointsf" else
                                    f"{args.loss} train step (listsf: 2-head 6-layer DASALC encoder + 128/256/512 feed-forward stacks, "
                                    f"dropout 0.1, Adagrad), ") + f"MSLR-WEB30K-shaped synthetic, {F} feats, list_len={L}",
                       "queries_per_gpu_per_step": B, "global_batch": world * B, "list_len": L, "features": F,
                       "parallelism": f"dp{world}", "resident_batches": len(batches)},
            "roofline": roofline,
            "kernels": kernels,
            "final_epoch_loss": final_loss,
        }
        if world == 1 and not args.no_cpu_baseline and args.scorer == "pointsf":
            out["cpu_baseline"] = cpu_baseline(L, F, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
