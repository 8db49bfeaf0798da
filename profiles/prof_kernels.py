#!/usr/bin/env python3
"""Stand-alone launches of every non-headline kernel of the path, for rocprofv3 (missing item 4 / next-round item 6).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats -d OUT --output-format csv -- python profiles/prof_kernels.py run [B]
    python profiles/prof_kernels.py summarise OUT/**/_kernel_stats.csv profiles/r03_kernels_B65536.json [B]
    (PMC passes, each on its own:  rocprofv3 --pmc FETCH_SIZE -d OUT2 ... ;  rocprofv3 --pmc WRITE_SIZE -d OUT3 ... ;  rocprofv3 --pmc SQ_INSTS_VALU -d OUT4 ...)

`run` launches each entry point ITERS times at B queries (default 65 536; inputs far larger than the 256 MB Infinity Cache) on the MSLR label
mix.  `summarise` joins the rocprofv3 kernel averages with the ALGORITHMIC bytes of SURVEY.md 8(d) — 12L+4 per query for the fused loss
kernels (16L+4 for ListMLE with its int64 permutation), 8L+4*len(ks) for the metric kernel, 16L for the sort (4L in, 4L values + 8L int64
indices out), 12L for the tie shuffle (4L labels in, 8L int64 order out) — and prints achieved GB/s against the 8 TB/s HBM peak.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KS = [1, 3, 5, 10, 20, 50]
ITERS = 6
# (label, kernel-name substring, list length, algorithmic bytes per query)
CASES = [
    ("ranknet L=32", "pairwise_bce_kernel<32, 1, false>", 32, lambda L: 12 * L + 4),
    ("lambdarank L=128", "lambdarank_ring_kernel<2>", 128, lambda L: 12 * L + 4),
    ("lambdarank L=256", "lambdarank_ring_kernel<4>", 256, lambda L: 12 * L + 4),
    ("lambdarank L=512", "lambdarank_ring_kernel<8>", 512, lambda L: 12 * L + 4),            # r6: the ring form up to 512 documents
    ("lambdarank L=1024", "pairwise_bce_kernel<256, 4, true>", 1024, lambda L: 12 * L + 4),  # above: the LDS kernel (MSLR-WEB30K lists run to 1 251 documents)
    ("listnet L=256", "listnet_vec_kernel", 256, lambda L: 12 * L + 4),
    ("listmle L=256", "listmle_vec_kernel", 256, lambda L: 16 * L + 4),
    ("lambdaloss L=256 k=5", "lambdaloss_", 256, lambda L: 12 * L + 4),
    ("approxndcg L=512", "approxndcg", 512, lambda L: 12 * L + 4),
    ("metrics L=256", "metrics_kernel", 256, lambda L: 8 * L + 4 * len(KS) * 4),
    ("sort_desc L=256", "sort_desc_kernel", 256, lambda L: 16 * L),
    ("shuffle_ties L=256", "shuffle_ties", 256, lambda L: 12 * L),
]


def queries_at(B, L):
    """Queries per launch at list length L: B up to 256 documents, B / 2 up to 512, B / 4 above (the O(L^2) kernels)."""
    return B if L <= 256 else B // 2 if L <= 512 else B // 4


def run(B):
    import torch
    import ptranking_amd as pa
    F = pa.functional
    torch.manual_seed(137)
    probs = torch.tensor([0.5147, 0.3250, 0.1339, 0.0183, 0.0081], device="cuda")
    data = {}
    for L in (32, 128, 256, 512, 1024):
        Bq = queries_at(B, L)
        preds = torch.randn(Bq, L, device="cuda")
        Y = torch.multinomial(probs.expand(Bq, -1), L, replacement=True).float()
        Y[:, 0].clamp_(min=1.0)
        Y = torch.sort(Y, dim=1, descending=True)[0].contiguous()
        data[L] = (preds, Y)

    def lg(fn, preds, *a, **k):
        p = preds.detach().requires_grad_(True)
        fn(p, *a, **k)          # forward launches the fused loss + gradient kernel; no backward pass needed

    for it in range(ITERS + 2):
        p, y = data[32]; lg(F.ranknet_loss, p, y, sigma=1.0)
        p, y = data[128]; lg(F.lambdarank_loss, p, y, sigma=1.0)
        p, y = data[256]
        lg(F.lambdarank_loss, p, y, sigma=1.0)
        lg(F.listnet_loss, p, y)
        perm = F.shuffle_ties_order(y, seed=11 + it)
        lg(F.listmle_loss, p, perm)
        lg(F.lambdaloss_loss, p, y, k=5, sigma=1.0, mu=5.0, loss_type="NDCG_Loss2", presort=True)
        F.metrics_at_ks(p, y, KS, presort=True)
        F.sort_desc(p)
        p, y = data[512]; lg(F.approxndcg_loss, p, y, alpha=10.0, presort=True)
        lg(F.lambdarank_loss, p, y, sigma=1.0)
        p, y = data[1024]; lg(F.lambdarank_loss, p, y, sigma=1.0)
    torch.cuda.synchronize()


# VALU issue BOUND: 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles — the guide's per-instruction constant for a plain wave64 VALU instruction
# (MI355X_MICROARCH.md, "v_fma_f32 (wave64) 2 cyc").  r4 priced every instruction at 4 cycles (what packed-fp32 and fp32-FMA streams
# measure here) and two kernels came out ABOVE that "peak" (RankNet 1.18, ListNet 1.04: DPP moves, selects and integer
# ops issue faster) — a bound below the achieved rate is not a bound.  Against the 2-cycle constant no kernel can
# exceed 1; transcendental, packed and quarter-rate instructions cost more than 2 cycles, so a real mix saturates well below it.
from ptranking_amd.peaks import VALU_PEAK_GINST, HBM_PEAK_GBPS, VALU_CYCLES_PER_INSTR, TRANS_CYCLES_PER_INSTR, NUM_SIMD, PEAK_CLOCK_HZ, RING_PAIR_PEAK_PER_S, RING_MIN_ISSUE_CYCLES_PER_PAIR   # noqa: E402 — shared with bench.py


def summarise(stats_csv, out_json, B, fetch_csv=None, write_csv=None, valu_csv=None):
    import csv
    rows = list(csv.DictReader(open(stats_csv)))
    traffic = {}
    if fetch_csv and write_csv:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import pmc_traffic
        f, w = pmc_traffic.per_kernel(fetch_csv, "FETCH_SIZE"), pmc_traffic.per_kernel(write_csv, "WRITE_SIZE")
        for k in set(f) | set(w):
            traffic[k] = int(round((2.0 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024))
    valu = {}
    if valu_csv:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import pmc_traffic
        valu = pmc_traffic.per_kernel(valu_csv, "SQ_INSTS_VALU")
    out = {"note": "rocprofv3 --kernel-trace --stats averages of stand-alone launches (profiles/prof_kernels.py run), MSLR label mix, 1xMI355X; "
                   "achieved = SURVEY 8(d) algorithmic bytes / average kernel time; peak = 8000 GB/s (HBM3E spec); traffic = PMC "
                   "FETCH_SIZE (x2, gfx950) + WRITE_SIZE per launch when collected; valu_roofline (the pair / sort kernels are VALU-bound, not HBM-bound) = PMC "
                   "SQ_INSTS_VALU (wave-wide VALU instructions per launch) / average kernel time against 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles "
                   "= 1228.8 G wave-instructions/s (the guide's 2-cycle wave64 VALU constant: an upper bound for every instruction mix)",
           "queries": B, "kernels": {}}
    slot_sum = [r for r in rows if "sum_f32_kernel" in r["Name"]]
    slot_sum_us = float(slot_sum[0]["AverageNs"]) / 1e3 if slot_sum else None      # the loss-slot sum every *_fwd_bwd entry point ends with
    for label, sub, L, bytes_per_q in CASES:
        Bq = queries_at(B, L)
        match = [r for r in rows if sub in r["Name"] and "ptr::" in r["Name"]]
        if not match:
            continue
        r = max(match, key=lambda r: float(r["TotalDurationNs"]))
        avg_us = float(r["AverageNs"]) / 1e3
        bytes_ = Bq * bytes_per_q(L)
        gbps = bytes_ / (avg_us * 1e-6) / 1e9
        tr = next((v for k, v in traffic.items() if sub.split("<")[0] in k and (("<" not in sub) or sub in k)), None)
        out["kernels"][label] = {"kernel": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_us": avg_us, "queries": Bq, "list_len": L,
                                 "algorithmic_bytes": bytes_, "achieved_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS,
                                 "traffic_bytes": tr}
        if slot_sum_us is not None and not any(x in label for x in ("metrics", "sort", "shuffle", "approx")):
            e_us = avg_us + slot_sum_us                 # whole entry point = loss kernel + slot sum (ApproxNDCG: its own finish kernel instead)
            out["kernels"][label]["entry_point"] = {"avg_us": e_us, "achieved_GBps": bytes_ / (e_us * 1e-6) / 1e9, "frac_of_hbm_peak": bytes_ / (e_us * 1e-6) / 1e9 / HBM_PEAK_GBPS}
        if label.startswith("lambdarank"):          # the same minimal-op pair bound bench.py's `valu_roofline` uses (ptranking_amd/peaks.py), on ALL L(L-1)/2 pairs
            pps = Bq * (L * (L - 1) / 2.0) / (avg_us * 1e-6)
            out["kernels"][label]["pair_roofline"] = {"bound": "valu-issue", "pairs_per_s_all_pairs": pps, "peak": RING_PAIR_PEAK_PER_S, "frac_all_pairs": pps / RING_PAIR_PEAK_PER_S,
                                                      "min_issue_cycles_per_pair": RING_MIN_ISSUE_CYCLES_PER_PAIR, "cycles_per_wave64_valu_instr": VALU_CYCLES_PER_INSTR,
                                                      "cycles_per_transcendental": TRANS_CYCLES_PER_INSTR}
        vi = next((v for k, v in valu.items() if sub.split("<")[0] in k and (("<" not in sub) or sub in k)), None)
        if vi:
            ach = vi / (avg_us * 1e-6) / 1e9
            out["kernels"][label]["valu_roofline"] = {"bound": "valu", "wave_instructions_per_launch": int(vi), "achieved": ach, "peak": VALU_PEAK_GINST,
                                                      "unit": "G wave-instructions/s", "frac": ach / VALU_PEAK_GINST,
                                                      "per_pair": (vi * 64.0 / (Bq * L * (L - 1) / 2.0)) if "L=" in label and ("rank" in label or "approx" in label) else None}
        print(f"{label:24s} {avg_us:9.1f} us  {gbps:8.1f} GB/s  {gbps / 80:5.1f} % of HBM peak" + (f"  traffic {tr / 1e6:.1f} MB vs {bytes_ / 1e6:.1f} MB" if tr else "")
              + (f"  VALU {out['kernels'][label]['valu_roofline']['frac'] * 100:.0f} % of issue peak" if vi else ""))
    json.dump(out, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 65536)
    else:
        summarise(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 65536, *(sys.argv[5:8] if len(sys.argv) > 6 else []))
