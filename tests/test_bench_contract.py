"""bench.py keeps the driver's contract: flags, ONE JSON line on stdout, the required keys, `roofline` and `cpu_baseline` objects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config"]


def test_bench_cli_and_cpu_baseline_helpers():
    """CPU: the argument parser accepts the driver's flags; the cpu_baseline leg (the oracle's torch-CPU train step) runs."""
    sys.path.insert(0, ROOT)
    import bench
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--global-batch", "--sweep"):
        assert flag in out.stdout
    cb = bench.cpu_baseline(16, 24, 0.6)
    assert cb["kind"] == "port" and cb["unit"] == "queries/s" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb


@pytest.mark.gpu
def test_bench_prints_one_contract_json_line():
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--batch", "256",
                          "--sweep", "64", "--sweep-steps", "6", "--cpu-seconds", "0.5", "--extras", "on"], capture_output=True, text=True, timeout=900,
                         env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"]
    assert abs(d["value"] - 256 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic_measured_in_run"] is False
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert "64" in d["by_batch"] and d["by_batch"]["64"]["queries_per_s_per_gpu"] > 0
    assert cb["kind"] == "port" and cb["reference_on_box"] is False
    # SURVEY 8(d)'s further measurements ride in the same line: padded variant, metric path, BASELINE configs 1 / 3 / 4 / 5
    pd = d["padded"]
    assert pd["value"] > 0 and pd["ms_per_step"] > 0 and 1.0 <= pd["mean_len"] <= 128.0 and pd["padded_len"] == 128
    mp = d["metric_path"]
    assert mp["value"] > 0 and mp["cpu_port"]["value"] > 0 and mp["cpu_port"]["reference_on_box"] is False
    assert len(mp["ndcg_at_ks"]) == len(mp["ks"]) and all(0.0 <= v <= 1.0 for v in mp["ndcg_at_ks"]) and 0.0 <= mp["ap_at_10"] <= 1.0
    mk = mp["metrics_kernel"]
    assert mk["bound"] == "hbm" and 0.0 < mk["frac"] <= 1.0 and abs(mk["frac"] - mk["achieved"] / mk["peak"]) < 1e-9
    assert set(d["configs"]) == {"C1_ranknet_L32", "C3_listnet_L256", "C3_listmle_L256", "C4_approxndcg_L512_F700", "C5_listsf_lambdaloss_L256"}
    for c in d["configs"].values():
        assert c["value"] > 0 and c["ms_per_step"] > 0 and c["steps"] == 10

    def fracs(o):                         # every roofline fraction in the line is a fraction: a "bound" below the achieved rate is not a bound
        if isinstance(o, dict):
            for k, v in o.items():
                if k == "frac":
                    yield v
                else:
                    yield from fracs(v)
    assert all(0.0 <= f <= 1.0 for f in fracs(d)), list(fracs(d))


def test_one_set_of_issue_constants():
    """bench.py and profiles/prof_kernels.py price VALU issue with the SAME constants (ptranking_amd/peaks.py = the guide's 2 cycles per
    wave64 VALU instruction, 8 per transcendental) — r5 carried 4 cycles in one file and 2 in the other."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import bench
    import prof_kernels
    from ptranking_amd import peaks
    for name in ("VALU_CYCLES_PER_INSTR", "TRANS_CYCLES_PER_INSTR", "NUM_SIMD", "PEAK_CLOCK_HZ", "HBM_PEAK_GBPS", "RING_PAIR_PEAK_PER_S", "RING_MIN_ISSUE_CYCLES_PER_PAIR"):
        assert getattr(bench, name) is getattr(peaks, name) or getattr(bench, name) == getattr(peaks, name), name
        assert getattr(prof_kernels, name) == getattr(peaks, name), name
    assert peaks.VALU_CYCLES_PER_INSTR == 2.0 and peaks.TRANS_CYCLES_PER_INSTR == 8.0 and peaks.RING_MIN_ISSUE_CYCLES_PER_PAIR == 40.0
    assert prof_kernels.VALU_PEAK_GINST == peaks.NUM_SIMD * peaks.PEAK_CLOCK_HZ / peaks.VALU_CYCLES_PER_INSTR / 1e9
    for f in ("bench.py", os.path.join("profiles", "prof_kernels.py")):          # no second definition creeps back in
        src = open(os.path.join(ROOT, f)).read()
        assert "VALU_CYCLES_PER_INSTR =" not in src and "TRANS_CYCLES_PER_INSTR =" not in src, f


def test_pmc_traffic_is_refused_when_stale(tmp_path, monkeypatch):
    """bench.load_pmc only trusts a PMC traffic file collected on EXACTLY the kernel sources that are built."""
    sys.path.insert(0, ROOT)
    import bench
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    good = {"config": {"queries_per_gpu_per_step": 4096, "list_len": 128, "features": 136}, "kernel_source_hash": h,
            "kernels": {"ptr::mlp_bwd_fused_kernel<3, 9>": {"hbm_bytes_per_launch": 123}}}
    f = tmp_path / "pmc.json"
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "PMC_FILE", "pmc.json")
    f.write_text(json.dumps(good))
    got, src = bench.load_pmc(4096, 128, 136)
    assert got == {"ptr::mlp_bwd_fused_kernel<3, 9>": 123} and h in src
    assert bench.load_pmc(1024, 128, 136) == ({}, "shape mismatch")
    f.write_text(json.dumps(dict(good, kernel_source_hash="0" * 16)))
    got, src = bench.load_pmc(4096, 128, 136)
    assert got == {} and src.startswith("stale")
    f.unlink()
    assert bench.load_pmc(4096, 128, 136)[0] == {}


def test_ring_pair_statistics():
    """What the LambdaRank ring kernel skips, computed on the host for the bench line: trailing slots of 64 documents that hold one label."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    Y = torch.zeros(3, 128)
    Y[0, :10] = 1.0                       # slot 1 (documents 64..127) all zero -> 1 of 4 blocks skipped
    Y[1, :70] = 2.0                       # the label run crosses the slot boundary -> nothing skipped
    Y[2, :] = 1.0                         # all equal -> everything skipped
    st = bench.ring_pair_statistics(Y)
    assert st["slots"] == 2
    assert abs(st["mean_trailing_pure_slots"] - (1 + 0 + 2) / 3) < 1e-6
    assert abs(st["blocks_skipped_frac"] - ((1 + 0 + 4) / 3) / 4) < 1e-6
    eq = [(10 * 9 + 118 * 117) / (128 * 127), (70 * 69 + 58 * 57) / (128 * 127), 1.0]
    assert abs(st["zero_weight_pairs_frac"] - sum(eq) / 3) < 1e-6


def test_self_launch_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` with no rank environment launches its own N ranks (one per GPU over RCCL); on a node with fewer GPUs it
    must stop with a message instead of running one rank and reporting n_gpus = 1 (the round-3 behaviour the driver's scaling tier tripped on)."""
    import os, subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""                                   # a GPU box too: no device visible to this child
    env["CUDA_VISIBLE_DEVICES"] = ""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode != 0 and out.stdout.strip() == ""
    assert "one rank per GPU is required" in out.stderr


def test_committed_kernel_profiles_hold_no_fraction_above_one():
    """a roofline fraction above 1 means the "peak" was not a bound.  Every `frac` / `frac_of_hbm_peak` in the committed
    round-5 / round-6 stand-alone kernel profiles and bench lines is a fraction."""
    import glob

    def fracs(o, path=""):
        if isinstance(o, dict):
            for k, v in o.items():
                if k in ("frac", "frac_of_hbm_peak") and isinstance(v, (int, float)):
                    yield path + "/" + k, v
                else:
                    yield from fracs(v, path + "/" + k)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[56]_kernels_B65536.json")) + glob.glob(os.path.join(ROOT, "profiles", "r0[56]_bench_*.json")))
    assert files, "the round-5 / round-6 profiles are committed"
    for f in files:
        text = open(f).read().strip()
        try:
            d = json.loads(text)                                   # an indented summary file
        except ValueError:
            d = json.loads(text.splitlines()[-1])                  # a bench line (ONE line of JSON, possibly behind noise)
        bad = [(k, v) for k, v in fracs(d) if not (0.0 <= v <= 1.0)]
        assert not bad, (os.path.basename(f), bad)
