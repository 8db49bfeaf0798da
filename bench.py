#!/usr/bin/env python3
"""Benchmark of the hot path: queries/sec of one LambdaRank train step (scorer forward + fused delta-NDCG loss/grad kernel +
scorer backward + optimiser step) on MSLR-WEB30K-shaped synthetic batches, list_len=128, 136 features (BASELINE.json
configs[1]).  One process per GPU:

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement).  `value` is measured at --batch queries per GPU per step (default
4096; weak scaling — or --global-batch G for strong scaling); `by_batch` carries SURVEY.md 8(d)'s sweep 64 / 256 / 1024 / 4096
queries per GPU (1024 is the survey's headline batch) measured the same way with fewer steps.  Extra objects:
  roofline      the dominant kernel of the step by time (the fused single-pass scorer backward, fp32 MFMA): algorithmic flops per
                launch / its average launch duration measured with HIP events on the launch stream during the timed region (every 4th step), vs
                the 157.3 TFLOP/s fp32 MFMA peak; `traffic` = PMC HBM bytes (profiles/r06_pmc_traffic.json, trusted only when its kernel-source hash matches the built sources), `algorithmic_bytes_*`
                = SURVEY 8(d)'s definition (features + scores), `design_bytes_*` = what the design additionally moves (stored
                activations, partial gradients)
  kernels       the other kernels of the step: scorer forward (MFMA roofline), the north-star LambdaRank loss kernel against the
                HBM roofline (12L+4 bytes / query) AND against its VALU-issue bound (`valu_roofline`), the same kernel at
                list_len=256, Adam; under data parallelism the gradient all-reduce (`allreduce_ms`)
  cpu_baseline  the oracle's torch-CPU restatement of the reference train step, timed on this box's host cores on a bounded
                sample of the same workload with a thread sweep (rank 0, N=1 only); the reference ITSELF timed beside the port on
                the build container is committed as profiles/r02_reference_vs_port_cpu.json
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

from ptranking_amd.peaks import (HBM_PEAK_GBPS, MFMA_F32_PEAK_TFLOPS, BF16X6_PEAK_TFLOPS, NUM_SIMD, PEAK_CLOCK_HZ,   # noqa: E402 — ONE set of
                                 VALU_CYCLES_PER_INSTR, TRANS_CYCLES_PER_INSTR, RING_MIN_FMA_OPS_PER_PAIR, RING_TRANS_PER_PAIR,
                                 RING_MIN_ISSUE_CYCLES_PER_PAIR, RING_PAIR_PEAK_PER_S)    # hardware constants, shared with profiles/prof_kernels.py
MSLR_P = [0.5147, 0.3250, 0.1339, 0.0183, 0.0081]   # label histogram of MSLR-WEB30K
SEED = 137                        # ptranking/ltr_global.py:7
EVENT_EVERY = 4                   # per-kernel HIP-event brackets on every 4th step of the timed region
# pair-loop VALU instructions per pair evaluation of lambdarank_ring_kernel<DPT> read off the gfx950 ISA, 3 of them
# transcendental (v_exp/v_rcp/v_log).  Issue cycles come from ptranking_amd/peaks.py (the guide's constants: 2 cycles per plain wave64
# VALU instruction, 8 per transcendental) — r5 priced a plain instruction at 4 here and at 2 in profiles/prof_kernels.py.
# The kernel's measured mix takes SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.6 cycles per instruction (profiles/r02_sq_c2.txt: packed fp32
# and DPP forms issue slower than the 2-cycle constant), which is why the fraction against this bound is low.
RING_INSTR_PER_PAIR = {1: 25.0, 2: 16.75, 4: 15.5625}


def synth_batch(gen, B, L, F, device):
    """MSLR-shaped synthetic batch: X ~ N(0,1), graded labels sorted descending (presort=True), >= 1 relevant doc."""
    X = torch.randn((B, L, F), generator=gen, device=device, dtype=torch.float32)
    probs = torch.tensor(MSLR_P, device=device)
    Y = torch.multinomial(probs.expand(B, -1), L, replacement=True, generator=gen).float()
    Y[:, 0] = torch.clamp(Y[:, 0], min=1.0)
    Y, _ = torch.sort(Y, dim=1, descending=True)
    return X, Y.contiguous()


def padded_lens(gen, B, L, device):
    """Documents per query of a padded batch: MSLR-WEB30K-like (1 .. 1251 documents, mean ~120, ptranking/data/data_utils.py:118-123) clipped to
    the padded length L — a Gamma(2.2) with mean 120, clipped to [1, L] (int32, on the device)."""
    g = torch._standard_gamma(torch.full((B,), 2.2, device=device), generator=gen) * (120.0 / 2.2)
    return g.round().clamp_(1, L).to(torch.int32)


def pad_batch(gen, X, Y, L):
    """(X, Y) of full lists -> the padded form PaddedQueryBatches produces: rows past lens[q] are zero features / zero labels, the real
    documents keep their label order (sorted descending), at least one relevant document per query."""
    lens = padded_lens(gen, X.shape[0], L, X.device)
    real = torch.arange(L, device=X.device)[None, :] < lens[:, None]
    Y = torch.where(real, Y, torch.zeros_like(Y))
    Y[:, 0] = torch.clamp(Y[:, 0], min=1.0)
    X = X * real[:, :, None]
    return X.contiguous(), Y.contiguous(), lens


def sf_para_dict(F, lr=1e-3):
    return {"sf_id": "pointsf", "opt": "Adam", "lr": lr,
            "pointsf": dict(num_features=F, num_layers=3, AF="R", TL_AF="S", apply_tl_af=False, BN=False, bn_type=None,
                            bn_affine=False)}   # h_dim=100, dropout=0.1 are hard-wired (point_ranker.py:30-31)


def cpu_baseline(L, F, budget_s):
    """Torch-CPU restatement of the reference's LambdaRank train step (oracle/torch_ref.py).  Timed at the batch sizes SURVEY.md
    8(d) names — 1 (the reference's own default for lists of >= 100 documents, data_utils.py:713-716), 64 and 256 — and, at 256, with
    torch.set_num_threads in {8, 16, 32, 64, all}: more threads than the small GEMMs can use is slower, so `value` is the best
    (batch, threads) found and `cores` the thread count that produced it.  Also a loss-only timing, kernel against kernel."""
    from oracle import torch_ref as T
    torch.manual_seed(SEED)
    gen = torch.Generator().manual_seed(SEED)
    Xf, Yf = synth_batch(gen, 256, L, F, "cpu")
    net = T.build_pointsf(F, seed=SEED)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-3)
    all_threads = torch.get_num_threads()

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

    sweep, total = [], 0.0
    cands = sorted({t for t in (8, 16, 32, 64, all_threads) if t <= all_threads})
    X, Y = Xf, Yf
    for nt in cands:
        torch.set_num_threads(nt)
        qps, it, el = timed(lambda: T.cpu_train_step(net, opt, X, Y, T.lambdarank_loss, sigma=1.0), 256, 0.45 * budget_s / len(cands), 200)
        sweep.append({"threads": nt, "batch": 256, "queries_per_s": qps, "steps": it, "seconds": el})
        total += el
    best_t = max(sweep, key=lambda d: d["queries_per_s"])["threads"]
    torch.set_num_threads(best_t)
    by_batch = {"B256": max(sweep, key=lambda d: d["queries_per_s"])}
    for B, share, cap in ((1, 0.15, 2000), (64, 0.2, 400)):
        X, Y = Xf[:B].contiguous(), Yf[:B].contiguous()
        qps, it, el = timed(lambda: T.cpu_train_step(net, opt, X, Y, T.lambdarank_loss, sigma=1.0), B, share * budget_s, cap)
        by_batch[f"B{B}"] = {"threads": best_t, "batch": B, "queries_per_s": qps, "steps": it, "seconds": el}
        total += el
    preds = torch.randn(256, L, generator=gen)

    def loss_only():
        p = preds.clone().requires_grad_(True)
        T.lambdarank_loss(p, Yf, sigma=1.0).backward()

    lq, lit, lel = timed(loss_only, 256, 0.15 * budget_s, 400)
    torch.set_num_threads(all_threads)
    best = max(by_batch.values(), key=lambda d: d["queries_per_s"])
    return {"value": best["queries_per_s"], "unit": "queries/s", "cores": best["threads"], "host_threads_available": all_threads,
            "kind": "port", "reference_on_box": False,      # the reference tree exists only in the build container: the GPU box times the port
            "sample": f"torch-CPU restatement of the reference train_op (pointsf scorer + LambdaRank + Adam) on {L} docs x {F} feats: "
                      f"thread sweep {cands} at batch 256, then batch 1 / 64 at the best thread count, {total:.1f} s in total, "
                      f"value = best; plus {lel:.1f} s loss-only.  The reference itself vs this port on the build container: "
                      f"profiles/r02_reference_vs_port_cpu.json",
            "thread_sweep": sweep, "by_batch": by_batch, "loss_only_queries_per_s": lq}


def metric_path(ranker, B, L, F, device, rank, ks=(1, 3, 5, 10, 20, 50), cpu_seconds=3.0):
    """Evaluator.ndcg_at_ks + Evaluator.ap_at_k over a synthetic loader (8 batches of B queries, device-resident, presort=True): the
    device path (DeviceEvaluator: predict -> ptr_metrics_at_ks) against the port of the reference's CPU loop (ptranking/base/ranker.py:67-95,
    130-160: predict -> sort -> gather -> torch_ndcg_at_ks / torch_ap_at_k), timed on a bounded sample.  Also the metric kernel alone
    against its 8L + 4|ks| bytes per query."""
    import ptranking_amd as pa
    from oracle import torch_ref as T
    ks = list(ks)
    gen = torch.Generator(device=device).manual_seed(SEED + 31 + rank)
    base = [synth_batch(gen, B, L, F, device) for _ in range(4)]
    loader = [(range(B), X, Y) for X, Y in base] * 2
    nq = B * len(loader)
    ranker.eval_mode()
    ranker.ndcg_at_ks(test_data=loader[:2], ks=ks, label_type=pa.LABEL_TYPE.MultiLabel, presort=True)
    ranker.ap_at_k(test_data=loader[:2], k=10, presort=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nd = ranker.ndcg_at_ks(test_data=loader, ks=ks, label_type=pa.LABEL_TYPE.MultiLabel, presort=True)
    apk = ranker.ap_at_k(test_data=loader, k=10, presort=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # the metric kernel alone (nDCG at the six cut-offs), 20 launches inside one HIP-event pair on the launch stream
    preds = torch.randn((B, L), generator=gen, device=device)
    Y = base[0][1]
    for _ in range(3):
        pa.functional.metrics_at_ks(preds, Y, ks, presort=True, which=("ndcg",))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        pa.functional.metrics_at_ks(preds, Y, ks, presort=True, which=("ndcg",))
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / 20
    k_bytes = B * (8 * L + 4 * len(ks))
    ranker.train_mode()
    # CPU: the port's loop on a bounded sample (batches of 64 queries, the reference's own evaluation batching is per query length)
    net = T.build_pointsf(F, seed=SEED)
    net.eval()
    Xc, Yc = base[0][0][:64].cpu(), base[0][1][:64].cpu()
    done, t0 = 0, time.perf_counter()
    with torch.no_grad():
        while True:
            p = net(Xc).view(64, L)
            _, idx = T.sort_desc(p)
            sys_sorted = torch.gather(Yc, 1, idx)
            T.ndcg_at_ks(sys_sorted, Yc, ks)
            T.ap_at_ks(sys_sorted, Yc, [10])
            done += 64
            cel = time.perf_counter() - t0
            if cel >= cpu_seconds:
                break
    del base, loader
    return {"value": nq / el, "unit": "queries/s", "queries": nq, "seconds": el, "ks": ks, "list_len": L,
            "what": "DeviceEvaluator.ndcg_at_ks(ks) + ap_at_k(10) over a loader of 8 x B device-resident batches (eval-mode scorer forward + ptr_metrics_at_ks per batch and call)",
            "ndcg_at_ks": [float(v) for v in nd], "ap_at_10": float(apk[0]),
            "metrics_kernel": {"bound": "hbm", "avg_launch_ms": k_ms, "algorithmic_bytes_per_launch": k_bytes, "achieved": k_bytes / (k_ms * 1e-3) / 1e9,
                               "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": k_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                               "note": "ptr_metrics_at_ks entry (nDCG only) at B x L, 8L + 4|ks| bytes per query"},
            "cpu_port": {"value": done / cel, "unit": "queries/s", "kind": "port", "reference_on_box": False, "threads": torch.get_num_threads(),
                         "sample": f"{done} queries in {cel:.1f} s: torch-CPU scorer forward + sort + gather + nDCG@ks + AP@10 on batches of 64 queries"}}


PMC_FILE = os.path.join("profiles", "r06_pmc_traffic.json")


def kernel_source_hash():
    """sha256 over the HIP sources + headers the .so is built from (ptranking_amd/build.py SOURCES / HEADERS): identifies the kernels a
    PMC traffic file was collected on."""
    import hashlib
    from ptranking_amd import build as _b
    h = hashlib.sha256()
    for name in sorted(_b.SOURCES) + sorted(_b.HEADERS):
        with open(os.path.join(_b.CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def ring_pair_statistics(Y):
    """What the ring kernel skips on a label batch [B, L] (L <= 256, sorted descending): slot k = documents 64k..64k+63; the trailing Z
    slots whose documents all carry one label are mutually weight-free, so Z^2 of the S^2 (own slot, travelling slot) blocks are not
    evaluated.  Also the share of pairs whose weight is exactly 0 (equal labels, metric_utils.py:43)."""
    B, L = Y.shape
    S = (L + 63) // 64
    Yp = torch.nn.functional.pad(Y, (0, S * 64 - L), value=float("nan")).view(B, S, 64)
    real = ~torch.isnan(Yp)
    hi = torch.where(real, Yp, torch.full_like(Yp, -1e9)).max(dim=2)[0]
    lo = torch.where(real, Yp, torch.full_like(Yp, 1e9)).min(dim=2)[0]
    pure = hi == lo
    z = torch.zeros(B, device=Y.device)
    run = torch.ones(B, dtype=torch.bool, device=Y.device)
    last = hi[:, -1]
    for k in range(S - 1, -1, -1):
        run = run & pure[:, k] & (hi[:, k] == last)
        z += run.float()
    eq = (Y[:, :, None] == Y[:, None, :]).float().sum(dim=(1, 2)) - L          # ordered pairs i != j with equal labels
    return {"slots": S, "mean_trailing_pure_slots": float(z.mean()), "blocks_skipped_frac": float((z * z).mean()) / (S * S),
            "zero_weight_pairs_frac": float(eq.mean()) / (L * (L - 1))}


def load_pmc(B, L, F):
    """HBM bytes per launch from the committed rocprofv3 PMC collection (profiles/pmc_traffic.py).  PMC passes cannot run inside the timed
    process, so the file is only trusted when it was collected on EXACTLY these kernels (source hash) at this shape; otherwise `traffic`
    is null rather than a stale number."""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            j = json.load(f)
        if j["config"] != {"queries_per_gpu_per_step": B, "list_len": L, "features": F}:
            return {}, "shape mismatch"
        if j.get("kernel_source_hash") != kernel_source_hash():
            return {}, f"stale: {PMC_FILE} was collected on sources {j.get('kernel_source_hash')}, built {kernel_source_hash()}"
        return {k: v["hbm_bytes_per_launch"] for k, v in j["kernels"].items()}, f"{PMC_FILE} (source hash {j['kernel_source_hash']})"
    except (OSError, KeyError, ValueError) as e:
        return {}, f"unavailable: {type(e).__name__}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="queries per GPU per step (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0, help="strong scaling: total queries per step, split evenly over the GPUs")
    ap.add_argument("--sweep", default="64,256,1024,4096", help="SURVEY 8(d) per-GPU batch sweep reported in by_batch ('' = off)")
    ap.add_argument("--sweep-steps", type=int, default=100)
    ap.add_argument("--list-len", type=int, default=128)
    ap.add_argument("--features", type=int, default=136)
    ap.add_argument("--nbatches", type=int, default=4, help="distinct HBM-resident batches cycled through (> L3 capacity)")
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps: the first is the contract's, all are reported")
    ap.add_argument("--cpu-seconds", type=float, default=14.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--loss", default="LambdaRank", choices=["RankNet", "LambdaRank", "LambdaLoss", "ApproxNDCG", "ListNet", "ListMLE"],
                    help="ranker of the train step (the headline metric is LambdaRank; the others cover BASELINE.json configs 3-5)")
    ap.add_argument("--scorer", default="pointsf", choices=["pointsf", "pointsf_default", "listsf"],
                    help="listsf = BASELINE.json config 5: 2-head / 6-layer DASALC encoder (fused MFMA attention), use with --loss LambdaLoss "
                         "--list-len 256 --batch 1024")
    ap.add_argument("--extras", default="auto", choices=["auto", "on", "off"],
                    help="SURVEY 8(d)'s further measurements in the same JSON line: the padded variant, the metric path (DeviceEvaluator vs the "
                         "port's CPU loop) and a short window of BASELINE configs 1 / 3 / 4 / 5 (auto: headline workload on one GPU)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="N = 1 only: initialise a one-rank RCCL process group and run the data-parallel step through it (backward -> "
                         "all-reduce -> optimiser step instead of the fused single-device step); what a 1-GPU box can execute of the RCCL path")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: become the launcher.  One rank per GPU over RCCL, exactly the command line the
    # module docstring gives; rank 0 of the child job prints the one JSON line.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {torch.cuda.device_count()} GPU(s); one rank per GPU is required")
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    # stdout carries exactly ONE line, the JSON: RCCL prints a version banner to stdout when a process group is torn down, so from here on
    # file descriptor 1 of every rank points at stderr and the JSON goes to a private duplicate of the original stdout
    json_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
    if args.force_collectives:
        if args.gpus != 1 or int(os.environ.get("WORLD_SIZE", "1")) != 1:
            raise SystemExit("--force-collectives is for single-GPU runs")
        os.environ["PTR_DP_INIT_SINGLE"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with __import__("socket").socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))

    import ptranking_amd as pa
    from ptranking_amd import _lib, dp

    rank, world, local = dp.init_from_env()
    if args.force_collectives:
        dp.SINGLE_RANK_COLLECTIVES = True
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    scaling = "weak"
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit(f"--global-batch {args.global_batch} is not a multiple of {world} ranks")
        args.batch, scaling = args.global_batch // world, "strong"
    B, L, F = args.batch, args.list_len, args.features
    headline = (args.loss, L, F, args.scorer) == ("LambdaRank", 128, 136, "pointsf")

    def build_ranker(loss=None, scorer=None, F=F):
        loss, scorer = loss or args.loss, scorer or args.scorer
        torch.manual_seed(SEED)                       # identical initial weights on every rank
        cls = getattr(pa, loss)
        if scorer == "listsf":      # ptranking/ltr_adhoc/eval/parameter.py:152-166 defaults
            sfd = {"sf_id": "listsf", "opt": "Adagrad", "lr": 1e-3,
                   "listsf": dict(num_features=F, ff_dims=[128, 256, 512], AF="R", TL_AF="GE", apply_tl_af=False, BN=False, bn_type="BN2",
                                  bn_affine=False, n_heads=2, encoder_layers=6, encoder_type="DASALC")}
        elif scorer == "pointsf_default":      # the driver's default scoring function, parameter.py:145-146
            sfd = {"sf_id": "pointsf", "opt": "Adam", "lr": 1e-3,
                   "pointsf": dict(num_features=F, num_layers=5, AF="GE", TL_AF="S", apply_tl_af=True, BN=True, bn_type="BN", bn_affine=True)}
        else:
            sfd = sf_para_dict(F)
        if loss == "ListNet":
            r = cls(sf_para_dict=sfd, gpu=True, device=device)
        else:
            r = cls(sf_para_dict=sfd, model_para_dict=dict(pa.DEFAULT_PARAS[loss]), gpu=True, device=device)
        if loss == "ListMLE":
            assert r.tie_shuffle == "device"     # the product default (the reference's B host-side randperm calls per step would dominate)
        r.init()
        r.train_mode()                           # dropout 0.1 active, exactly like the reference's train()
        dp.seed_replica(SEED)                    # CPU generator identical on every rank, CUDA generator per rank (dp.py)
        return r

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(ranker, Bq, steps, warmup, prewarm_rounds, L=L, F=F, nbatches=None, padded=False, prewarm_steps=20):
        """Times `steps` train steps at Bq queries per GPU.  Returns (seconds [max over ranks], per-entry-point event timings,
        all-reduce event timings, final accumulated loss).  padded: every query gets an MSLR-like number of real documents
        (padded_lens) and the step runs with `lens`, the way PaddedQueryBatches feeds it."""
        gen = torch.Generator(device=device).manual_seed(SEED + 1000 * rank + Bq)   # every rank owns different queries
        batches = [synth_batch(gen, Bq, L, F, device) + (None,) for _ in range(max(1, nbatches or args.nbatches))]
        if padded:
            batches = [pad_batch(gen, X, Y, L) for X, Y, _ in batches]
            measure.mean_len = float(torch.cat([b[2] for b in batches]).float().mean().item())

        def run_steps(n, hooks=None):
            """n train steps, accumulating the loss on the device exactly like DeviceTrainLoop.train does (no host sync).
            hooks = (entry-point timings, all-reduce timings): HIP-event brackets around every C-ABI call of every EVENT_EVERY-th
            step (two events per call are not free: at 64 queries per step they would be a fifth of the step)."""
            acc = torch.zeros((), device=device)
            for i in range(n):
                X, Y, ln = batches[i % len(batches)]
                on = hooks is not None and i % EVENT_EVERY == 0
                if on:
                    _lib.TIMING, dp.TIMING = hooks
                if ln is None:
                    loss, _ = ranker.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel)
                else:
                    loss, _ = ranker.train_op(X, Y, epoch_k=1, presort=True, label_type=pa.LABEL_TYPE.MultiLabel, lens=ln)
                if on:
                    _lib.TIMING, dp.TIMING = None, None
                acc += loss.detach()
            return acc

        # Untimed pre-warm with EXACTLY the code of the timed region (event hooks included): the first launch of every kernel
        # lazily loads its code object (tens of ms each), and a fresh box needs a few hundred ms before clocks / allocator settle.
        # The collector runs BEFORE the pre-warm: tens of ms of host work with an idle GPU right in front of the timed region let the
        # clocks fall back, and W = 5 warm-up steps (5 ms) do not bring them up again (7 % on a 20-step run).
        gc.collect()
        gc.disable()                      # no collector pauses inside the timed region
        for _ in range(prewarm_rounds):   # a FIXED count: every rank must issue the same number of all-reduces
            float(run_steps(prewarm_steps, ({}, [])).item())
        run_steps(warmup)
        sync()
        timing, ar_timing = {}, []
        t0 = time.perf_counter()
        epoch_loss = run_steps(steps, (timing, ar_timing))
        sync()
        elapsed = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            t = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        final_loss = float(epoch_loss.item())
        if not np.isfinite(final_loss):
            raise SystemExit(f"non-finite loss {final_loss}")
        del batches
        return elapsed, timing, ar_timing, final_loss

    ranker = build_ranker()
    elapsed, timing, ar_timing, final_loss = measure(ranker, B, args.steps, args.warmup, 5)
    single_call = bool(getattr(ranker, "single_call_step", False)) and "desc" in next(iter(ranker.__dict__.get("_direct_buffers", {"": {}}).values()))
    # The contract's window is the one above (`steps`, `ms_per_step`, `value`).  It is tens of ms long, so its spread is reported too:
    # args.windows - 1 further windows of the same K steps, each bracketed the same way
    # Every window gets the pre-warm of the first (with ONE warm-up step in front of freshly synthesised batches the
    # later windows ran 10 % slower than the contract's — a cold-start artefact, not noise).
    window_ms = [1e3 * elapsed / args.steps]
    for _ in range(max(0, args.windows - 1)):
        el, _, _, _ = measure(ranker, B, args.steps, args.warmup, 5)
        window_ms.append(1e3 * el / args.steps)

    by_batch = {}
    if args.sweep and world == 1 and args.scorer == "pointsf":
        for bq in [int(x) for x in args.sweep.split(",") if x]:
            if bq == B:
                by_batch[str(bq)] = {"queries_per_s_per_gpu": B * args.steps / elapsed, "ms_per_step": 1e3 * elapsed / args.steps,
                                     "steps": args.steps, "is_value": True}
                continue
            # r6: the sweep sizes get the headline's pre-warm (5 rounds of 20 steps).  r5 gave them one round: at 1024 queries that is 5 ms of GPU work behind
            # the host-side batch synthesis — the clocks had not come back up, and `value_at_1024` read 0.251 ms where a dedicated --batch 1024 run of the
            # same build measured 0.232 (profiles/r05_bench_B1024.json)
            el, _, _, _ = measure(ranker, bq, args.sweep_steps, args.warmup, 5)
            by_batch[str(bq)] = {"queries_per_s_per_gpu": bq * args.sweep_steps / el, "ms_per_step": 1e3 * el / args.sweep_steps,
                                 "steps": args.sweep_steps, "is_value": False}

    # ---- SURVEY 8(d)'s further measurements, in the same line
    extras = {}
    if world == 1 and args.scorer == "pointsf" and (args.extras == "on" or (args.extras == "auto" and headline and B >= 1024)):
        # (i) the padded variant: the same step on lists of MSLR-like lengths padded to L, through `lens` (what PaddedQueryBatches feeds)
        el, _, _, _ = measure(ranker, B, args.sweep_steps, args.warmup, 1, padded=True)      # (same kernels as the headline: no separate guard)
        extras["padded"] = {"value": B * args.sweep_steps / el, "unit": "queries/s", "ms_per_step": 1e3 * el / args.sweep_steps,
                            "mean_len": measure.mean_len, "padded_len": L, "documents_per_s": B * measure.mean_len * args.sweep_steps / el,
                            "steps": args.sweep_steps,
                            "note": "lens ~ Gamma(2.2) with mean 120 clipped to [1, list_len] (MSLR-WEB30K: 1..1251 documents, mean ~120); the scorer "
                                    "runs on all padded rows, the loss kernel skips padded documents"}
        # (ii) the metric path: Evaluator.ndcg_at_ks + ap_at_k (ptranking/base/ranker.py:67-95,130-160) over a synthetic loader, device-resident
        # (predict -> metrics kernel, nothing leaves the GPU but [len(ks)] numbers) vs the port's CPU loop (predict -> sort -> gather -> metric)
        try:
            extras["metric_path"] = metric_path(ranker, B, L, F, device, rank)
        except Exception as e:                     # an extra must never take the contract's line down with it
            extras["metric_path"] = {"error": f"{type(e).__name__}: {e}"}
        # (iii) BASELINE.json configs 1, 3, 4, 5: one short window each (10 steps) so that the driver's record carries them
        cfgs = [("C1_ranknet_L32", "RankNet", "pointsf", 32, 136, 4096), ("C3_listnet_L256", "ListNet", "pointsf", 256, 136, 4096),
                ("C3_listmle_L256", "ListMLE", "pointsf", 256, 136, 4096), ("C4_approxndcg_L512_F700", "ApproxNDCG", "pointsf", 512, 700, 1024),
                ("C5_listsf_lambdaloss_L256", "LambdaLoss", "listsf", 256, 136, 1024)]
        extras["configs"] = {}
        del ranker
        for tag, loss_c, scorer_c, Lc, Fc, Bc in cfgs:
            Bc = min(Bc, B)
            torch.cuda.empty_cache()
            try:
                rc = build_ranker(loss_c, scorer_c, F=Fc)
                el, _, _, _ = measure(rc, Bc, 10, 2, 1, L=Lc, F=Fc, nbatches=2, prewarm_steps=6)
                extras["configs"][tag] = {"value": Bc * 10 / el, "unit": "queries/s", "ms_per_step": 1e3 * el / 10, "steps": 10, "loss": loss_c,
                                          "scorer": scorer_c, "list_len": Lc, "features": Fc, "queries_per_step": Bc}
                del rc
            except Exception as e:
                gc.enable()
                extras["configs"][tag] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()

    # the loss kernel alone, at the headline list length and at the north-star's stated one (BASELINE.json: list_len=256), same
    # number of queries: 40 launches of the C entry back to back (no loss_out: the kernel only, no slot sum) inside ONE HIP-event
    # pair on the launch stream — the per-call event bracket of the timed step also holds the slot-sum kernel and two event gaps
    ring_alone, ring_stats = {}, {}
    if rank == 0 and args.loss == "LambdaRank" and args.scorer == "pointsf":
        import ctypes as C
        gen = torch.Generator(device=device).manual_seed(SEED + 7)
        for Lk in sorted({L, 256}):
            if Lk > 256:
                continue
            pk = torch.randn((B, Lk), generator=gen, device=device)
            _, Yk = synth_batch(gen, B, Lk, 1, device)
            ring_stats[Lk] = ring_pair_statistics(Yk)
            lq = torch.empty(B, device=device); gk = torch.empty_like(pk)
            st = _lib.current_stream(device)
            def go():
                _lib.call("ptr_lambdarank_fwd_bwd", _lib.ptr(pk), _lib.ptr(Yk), None, B, Lk, C.c_float(1.0), None, _lib.ptr(lq), _lib.ptr(gk), st)
            for _ in range(5):
                go()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(40):
                go()
            e1.record()
            torch.cuda.synchronize()
            ring_alone[Lk] = e0.elapsed_time(e1) / 40
    l256 = ring_alone.get(256)

    # facts of the parallel run, gathered from every rank: backend, device, and that one all-reduce per step ran on all of them
    facts = {"rank": rank, "device": torch.cuda.get_device_name(local), "device_index": local,
             "allreduce_calls": len(ar_timing), "dropout_seed_row_offset": rank * B * L}
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, facts)
        backend = dist.get_backend()
    else:
        gathered, backend = [facts], (dist.get_backend() if dist.is_initialized() else None)
    parallel_facts = {"world_size": world, "backend": backend, "rccl": backend == "nccl", "ranks": gathered,
                      "rccl_ranks": world if backend == "nccl" else 0,
                      "allreduce_ms": (float(np.mean([a.elapsed_time(b) for a, b in ar_timing])) if ar_timing else None),
                      "launch": "self-launched torch.distributed.run" if os.environ.get("TORCHELASTIC_RUN_ID") or world > 1 else "single process",
                      "single_rank_collectives": bool(args.force_collectives)}
    if world > 1:
        assert len({g["allreduce_calls"] for g in gathered}) == 1 and gathered[0]["allreduce_calls"] > 0, gathered
        if backend == "nccl":
            assert len({g["device_index"] for g in gathered}) == world, f"RCCL ranks must own distinct GPUs: {gathered}"
            assert parallel_facts["rccl_ranks"] == world
    if rank == 0:
        def avg_ms(name):
            ev = timing.get(name, [])
            return float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else None

        R = B * L
        NL = NL_ = 3
        timed_steps = (args.steps + EVENT_EVERY - 1) // EVENT_EVERY       # steps of the timed region that carried event brackets
        fwd_flop = 2.0 * (100 * F + (NL - 1) * 100 * 100 + 100) * R          # algorithmic: 2*(100F + 2*100*100 + 100) per document
        bwd_flop = 2.0 * (100 * F + (NL - 1) * 100 * 100) * R + 2.0 * (NL - 1) * 100 * 100 * R + 2.0 * 100 * R   # dW + dZ chain + top
        step_ms = 1e3 * elapsed / args.steps
        loss_entry = {"RankNet": "ptr_ranknet_fwd_bwd", "LambdaRank": "ptr_lambdarank_fwd_bwd", "LambdaLoss": "ptr_lambdaloss_fwd_bwd",
                      "ApproxNDCG": "ptr_approxndcg_fwd_bwd", "ListNet": "ptr_listnet_fwd_bwd", "ListMLE": "ptr_listmle_fwd_bwd"}[args.loss]
        t_fwd, t_loss, t_bwd, t_adam, t_sum = (avg_ms(n) for n in ("ptr_mlp_forward", loss_entry, "ptr_mlp_backward", "ptr_adam_step",
                                                                    "ptr_sum_f32"))
        fwd_x6 = t_fwd is None and avg_ms("ptr_mlp_forward_x6") is not None
        if fwd_x6:               # the bf16x6 forward (csrc/scorer_x6.hip): from scorer.X6_MIN_ROWS documents per step on
            t_fwd = avg_ms("ptr_mlp_forward_x6")
        bwd_fused_step = t_bwd is None and avg_ms("ptr_mlp_backward_step") is not None
        if bwd_fused_step:       # single device: backward + optimiser step + loss-slot sum in one entry point (rankers._direct_train_op)
            t_bwd = avg_ms("ptr_mlp_backward_step")
        pmc, pmc_source = load_pmc(B, L, F)

        def pmc_bytes(prefix):
            for k, v in pmc.items():
                if k.startswith(prefix):
                    return v
            return None

        def loss_kernel_entry(Lk, t_ms, traffic):
            bytes_ = B * (12 * Lk + 4)
            pairs = B * (Lk * (Lk - 1) // 2)
            gbps = bytes_ / (t_ms * 1e-3) / 1e9
            dpt = 1 if Lk <= 64 else 2 if Lk <= 128 else 4
            st = ring_stats.get(Lk, {})
            evaluated = pairs * (1.0 - st.get("blocks_skipped_frac", 0.0))
            cyc_own = (RING_INSTR_PER_PAIR[dpt] - RING_TRANS_PER_PAIR) * VALU_CYCLES_PER_INSTR + RING_TRANS_PER_PAIR * TRANS_CYCLES_PER_INSTR
            bound = RING_PAIR_PEAK_PER_S
            return {"kernel": f"lambdarank_ring_kernel<{dpt}> (fused LambdaRank dNDCG loss + gradient, one wavefront per query, register/DPP ring, "
                              "equal-label slot blocks skipped)",
                    "bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                    "traffic": traffic, "avg_launch_ms": t_ms, "algorithmic_bytes_per_launch": bytes_,
                    "pairs_per_s": pairs / (t_ms * 1e-3), "evaluated_pairs_per_s": evaluated / (t_ms * 1e-3),
                    "effective_pairs_per_s": pairs * (1.0 - st.get("zero_weight_pairs_frac", 0.0)) / (t_ms * 1e-3),
                    "pair_statistics": st,
                    "valu_roofline": {"bound": "valu-issue", "achieved": evaluated / (t_ms * 1e-3), "unit": "evaluated pairs/s", "peak": bound,
                                      "frac": evaluated / (t_ms * 1e-3) / bound, "frac_all_pairs": pairs / (t_ms * 1e-3) / bound,
                                      "min_fma_class_ops_per_pair": RING_MIN_FMA_OPS_PER_PAIR, "transcendentals_per_pair": RING_TRANS_PER_PAIR,
                                      "cycles_per_wave64_valu_instr": VALU_CYCLES_PER_INSTR, "cycles_per_transcendental": TRANS_CYCLES_PER_INSTR,
                                      "min_issue_cycles_per_pair": RING_MIN_ISSUE_CYCLES_PER_PAIR,
                                      "kernel_issue_cycles_per_pair": cyc_own, "kernel_instr_per_pair": RING_INSTR_PER_PAIR[dpt],
                                      "note": "peak = 1024 SIMDs x 2.4 GHz x 64 lanes / MINIMAL issue cycles per pair: 16 FMA-class ops packed two "
                                              "pairs per instruction (8 x 2 cycles) + exp2, rcp, log2 (3 x 8 cycles) = 40 — a count of the "
                                              "arithmetic, not of our ISA (our pair loop issues kernel_issue_cycles_per_pair); achieved counts "
                                              "the pairs the kernel EVALUATES (equal-label blocks are skipped, pair_statistics); the shader "
                                              "clock sustained under this kernel is ~2.1 GHz; avg_launch_ms = 40 back-to-back launches inside "
                                              "one HIP-event pair / 40"},
                    "note": "O(L^2) pair work per 12L+4 bytes: VALU-bound by construction (DESIGN.md 3.1); pairs_per_s counts all L(L-1)/2 "
                            "pairs per query, effective_pairs_per_s the pairs with a non-zero weight"}

        kernels = {}
        if t_loss:
            if args.loss == "LambdaRank" and L <= 256:
                kernels["lambdarank_loss_grad"] = loss_kernel_entry(L, ring_alone.get(L, t_loss), pmc_bytes("ptr::lambdarank_ring_kernel"))
                kernels["lambdarank_loss_grad"]["entry_ms_in_step"] = t_loss      # C entry inside the timed step: ring kernel + slot sum
            else:
                gbps = B * (12 * L + 4) / (t_loss * 1e-3) / 1e9
                kernels["loss_grad"] = {"kernel": f"{loss_entry} (fused {args.loss} loss + gradient)", "bound": "hbm", "achieved": gbps,
                                        "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, "traffic": None,
                                        "avg_launch_ms": t_loss, "algorithmic_bytes_per_launch": B * (12 * L + 4)}
        if l256:
            kernels["lambdarank_loss_grad_L256"] = loss_kernel_entry(256, l256, None)
        if t_fwd:
            tf = fwd_flop / (t_fwd * 1e-3) / 1e12
            fwd_peak = BF16X6_PEAK_TFLOPS if fwd_x6 else MFMA_F32_PEAK_TFLOPS
            kernels["scorer_forward"] = {"kernel": ("x6_prep_kernel + mlp_fwd_x6_kernel<TRAIN,STORE,2> (fused pointsf scorer forward, every fp32 product as six "
                                                    "v_mfma_f32_16x16x32_bf16 products with fp32 accumulation; weight planes streamed through an LDS ring by LDS-DMA; "
                                                    "8 waves x 32-document tiles)" if fwd_x6 else
                                                    "mlp_fwd_kernel<RT,TRAIN,VEC> (fused pointsf scorer forward, fp32 MFMA 16x16x4, dropout in-kernel; training: 16 waves x 16-row tiles)"),
                                         "formulation": "bf16x6 (fp32 results)" if fwd_x6 else "fp32 MFMA",
                                         "bound": "mfma", "achieved": tf, "peak": fwd_peak, "unit": "TFLOP/s (effective fp32)" if fwd_x6 else "TFLOP/s",
                                         "frac": tf / fwd_peak, "frac_of_fp32_mfma_peak": tf / MFMA_F32_PEAK_TFLOPS,
                                         "traffic": pmc_bytes("ptr::mlp_fwd_x6_kernel" if fwd_x6 else "ptr::mlp_fwd_kernel"), "avg_launch_ms": t_fwd,
                                         "algorithmic_flop_per_launch": fwd_flop, "algorithmic_bytes_per_launch": R * (4 * F + 4),
                                         "design_bytes_per_launch": NL * R * 448}
        if t_adam:
            kernels["adam"] = {"avg_launch_ms": t_adam}
        if t_sum:
            kernels["loss_slot_sum"] = {"avg_launch_ms": t_sum}
        if world > 1 and ar_timing:
            kernels["gradient_allreduce"] = {"ranks": world, "backend": dist.get_backend(), "rccl_ranks": world if dist.get_backend() == "nccl" else 0,
                                             "allreduce_ms": float(np.mean([a.elapsed_time(b) for a, b in ar_timing])),
                                             "bytes": 4 * (100 * F + 100 + (NL - 1) * 10100 + 101), "calls_per_step": len(ar_timing) / timed_steps}
        if t_bwd and args.scorer == "pointsf":
            tf = bwd_flop / (t_bwd * 1e-3) / 1e12
            fused = (NL == 3 and 129 <= F <= 143 and F % 4 == 0)
            bwd_x6 = fused and os.environ.get("PTR_BWD_X6", "1") != "0"       # r5: the bf16x6 single-pass backward is the default where it serves the shape
            bwd_peak = BF16X6_PEAK_TFLOPS if bwd_x6 else MFMA_F32_PEAK_TFLOPS
            roofline = {"kernel": ("mlp_bwd_x6_kernel<9> (single-pass scorer backward: dZ chain + all weight gradients, every fp32 product as six "
                                   "v_mfma_f32_16x16x32_bf16 products with fp32 accumulation; bf16 plane images in LDS, software-pipelined slabs) "
                                   "+ reduce_partials_kernel" if bwd_x6 else
                                   "mlp_bwd_fused_kernel<3,9> (single-pass scorer backward: dZ chain + all weight gradients, fp32 MFMA 16x16x4) "
                                   "+ reduce_partials_kernel" if fused else "mlp_bwd_dz + 3 x mlp_bwd_dw + reduce_partials (layer-wise backward)"),
                        "formulation": "bf16x6 (fp32 results)" if bwd_x6 else "fp32 MFMA",
                        "bound": "mfma", "achieved": tf, "peak": bwd_peak, "unit": "TFLOP/s (effective fp32)" if bwd_x6 else "TFLOP/s", "frac": tf / bwd_peak,
                        "frac_of_fp32_mfma_peak": tf / MFMA_F32_PEAK_TFLOPS,
                        "traffic": pmc_bytes("ptr::mlp_bwd_x6_kernel" if bwd_x6 else "ptr::mlp_bwd_fused_kernel"),
                        "traffic_measured_in_run": False,      # PMC passes cannot run inside the timed process: the committed collection of the same kernels (hash-guarded)
                        "avg_launch_ms": t_bwd, "algorithmic_flop_per_launch": bwd_flop,
                        "algorithmic_bytes_per_launch": R * (4 * F + 4),
                        "design_bytes_per_launch": NL * R * 448 + 256 * 4 * (100 * F + 100 + (NL - 1) * 10100 + 101),
                        "entry_point": "ptr_mlp_backward_step" if bwd_fused_step else "ptr_mlp_backward",
                        "note": "dominant kernel of the step by time; avg_launch_ms brackets the whole entry point (fused backward kernel + "
                                "the 136 KB partial reduction, which on one device also applies the Adam step and sums the loss slots); algorithmic bytes = SURVEY 8(d) (features read once more for dW1 + dLoss/dscore), "
                                "design bytes = the stored activations read back (3 x 448 B / document) + one partial gradient per workgroup; "
                                "traffic = PMC FETCH_SIZE(x2 on gfx950)+WRITE_SIZE from " + pmc_source}
        elif args.scorer == "listsf" and avg_ms("ptr_mhsa_forward"):
            t_af, t_ab = avg_ms("ptr_mhsa_forward"), avg_ms("ptr_mhsa_backward")
            att_flop = 4.0 * B * L * L * F                         # QK^T and PV, 2 flop per MAC, all heads (H * d_h = F)
            tf = att_flop / (t_af * 1e-3) / 1e12
            roofline = {"kernel": "mhsa_fwd_kernel (fused attention core: QK^T/sqrt(d) -> online softmax -> dropout -> PV, fp32 MFMA 16x16x4)",
                        "bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                        "traffic": None, "avg_launch_ms": t_af, "algorithmic_flop_per_launch": att_flop,
                        "algorithmic_bytes_per_launch": 4 * B * L * F * 4 + B * 2 * L * 4,
                        "note": "one launch per encoder layer"}
            kernels["attention_backward"] = {"kernels": "attn_rowdot + mhsa_bwd_dq + mhsa_bwd_dkv", "avg_call_ms": t_ab,
                                             "achieved_TFLOPs": 2.5 * att_flop / (t_ab * 1e-3) / 1e12}
            for nm in ("ptr_layernorm_forward", "ptr_layernorm_backward", "ptr_linear_forward", "ptr_linear_backward"):
                if avg_ms(nm):
                    kernels[nm] = {"avg_launch_ms": avg_ms(nm), "launches_per_step": len(timing[nm]) / timed_steps}
        elif args.scorer == "pointsf_default" and avg_ms("ptr_linear_forward"):
            # layer-wise stack: 6 Linear layers (F -> 100 x5 -> 1), each linear -> batch statistics -> normalise / GELU / dropout
            t_lf = avg_ms("ptr_linear_forward")
            hid_flop = 2.0 * R * 100 * 100
            roofline = {"kernel": "linear_fwd_kernel (hidden 100 -> 100 layers of the layer-wise stack, fp32 MFMA 16x16x4)", "bound": "hbm",
                        "achieved": R * 800 / (t_lf * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": R * 800 / (t_lf * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "traffic": None, "avg_launch_ms": t_lf, "algorithmic_bytes_per_launch": R * 800, "algorithmic_flop_per_launch": hid_flop,
                        "note": "average over the stack's linear launches (forward and backward-input share the entry point); a 100 -> 100 layer "
                                "moves 800 B per document for 20 kflop: HBM-bound as a separate kernel (ridge ~29 flop/B)"}
            for nm in ("ptr_linear_forward", "ptr_linear_backward_weight", "ptr_bn_stats", "ptr_bnact_forward", "ptr_bnact_backward"):
                if avg_ms(nm):
                    kernels[nm] = {"avg_launch_ms": avg_ms(nm), "launches_per_step": len(timing[nm]) / timed_steps}
        else:   # scorer configuration not fusable: the loss kernel is the only kernel of ours in the step
            roofline = dict(kernels.get("lambdarank_loss_grad", kernels.get("loss_grad", {})))
        qps = world * B * args.steps / elapsed
        ws = sorted(window_ms)
        windows = {"n": len(ws), "steps_each": args.steps, "ms_per_step": window_ms, "median_ms_per_step": float(np.median(ws)),
                   "min_ms_per_step": ws[0], "max_ms_per_step": ws[-1], "median_queries_per_s": world * B / (float(np.median(ws)) * 1e-3),
                   "note": "window 0 is the contract's timed region (`value`, `ms_per_step`); the others repeat it"}
        step_alg_bytes = B * (2 * 4 * L * F + 12 * L + 4)
        out = {
            "metric": ("queries/sec fwd+bwd LambdaRank, MSLR-WEB30K-shaped list_len=128" if headline
                       else f"queries/sec fwd+bwd {args.loss}, synthetic list_len={L}, {F} feats"),
            "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_ms, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "formulation": {"scorer_forward": "bf16x6: fp32 operands split exactly into three bf16 pieces, six bf16 MFMA products per fp32 product, fp32 "
                                              "accumulation (error vs float64 <= the fp32-MFMA path, tests/test_x6_gpu.py)" if fwd_x6 else "fp32 MFMA",
                            "scorer_backward": ("bf16x6 (the forward's arithmetic; error vs float64 within the fp32-MFMA backward's, tests/test_x6_gpu.py)"
                                                if (args.scorer == "pointsf" and NL_ == 3 and 129 <= F <= 143 and F % 4 == 0 and os.environ.get("PTR_BWD_X6", "1") != "0")
                                                else "fp32 MFMA"), "loss": "fp32 VALU"},
            "config": {"workload": (f"{args.loss} train step (pointsf 3x100 ReLU scorer, dropout 0.1, Adam), " if args.scorer == "pointsf" else
                                    f"{args.loss} train step (the reference's DEFAULT pointsf: 5 x [Linear 100 -> BatchNorm(affine) -> GELU] + "
                                    f"Linear -> BatchNorm -> Sigmoid, dropout 0.1, Adam), " if args.scorer == "pointsf_default" else
                                    f"{args.loss} train step (listsf: 2-head 6-layer DASALC encoder + 128/256/512 feed-forward stacks, "
                                    f"dropout 0.1, Adagrad), ") + f"MSLR-WEB30K-shaped synthetic, {F} feats, list_len={L}, "
                                    f"{B} queries per GPU per step (`value`; by_batch = SURVEY 8(d) sweep, 1024 = the survey's headline batch)",
                       "queries_per_gpu_per_step": B, "global_batch": world * B, "list_len": L, "features": F,
                       "parallelism": f"dp{world}", "resident_batches": max(1, args.nbatches)},
            "step_hbm_roofline": {"algorithmic_bytes_per_step": step_alg_bytes, "achieved_GBps": step_alg_bytes / (step_ms * 1e-3) / 1e9,
                                  "frac_of_hbm_peak": step_alg_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                  "note": "SURVEY 8(d) end-to-end definition: 2*4*L*F + 12L + 4 bytes per query"},
            "step_entry": ("ptr_train_step: ONE C-ABI call per step enqueues scorer forward -> fused loss + gradient kernel -> scorer backward + optimiser "
                           "step + loss-slot sum (ABI v5, csrc/train_step.hip); with the bf16x6 forward the optimiser launch also refreshes the forward's "
                           "weight image, so a step is FOUR kernel launches; the per-stage times below come from HIP events the call records between its "
                           "stages on every 4th step" if single_call
                           else "entry points chained from Python (autograd or data-parallel path)"),
            "roofline": roofline,
            "kernels": kernels,
            "by_batch": by_batch,
            **extras,
            "windows": windows,
            "final_epoch_loss": final_loss,
            "pmc_traffic_source": pmc_source,
            "parallel": parallel_facts,
        }
        if "1024" in by_batch:     # SURVEY 8(d)'s headline batch, next to `value` (measured at --batch queries per GPU)
            out["value_at_1024"] = by_batch["1024"]["queries_per_s_per_gpu"]
            out["ms_per_step_at_1024"] = by_batch["1024"]["ms_per_step"]
        if world == 1 and not args.no_cpu_baseline and args.scorer == "pointsf":
            out["cpu_baseline"] = cpu_baseline(L, F, args.cpu_seconds)
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
