"""GPU: the end-to-end example (LETOR text -> native parser -> padded device batches -> fused train step -> device evaluator)
on a synthetic collection with a learnable signal: nDCG must rise clearly above its initial value."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_letor(path, n_q, rng, F=20):
    w = rng.standard_normal(F)
    with open(path, "w") as f:
        for q in range(n_q):
            n = int(rng.integers(12, 90))
            X = rng.standard_normal((n, F))
            s = X @ w + 0.3 * rng.standard_normal(n)
            lab = np.clip(np.floor((s - s.mean()) / (s.std() + 1e-9) + 1.5), 0, 4).astype(int)
            for i in range(n):
                f.write(f"{lab[i]} qid:{q + 1} " + " ".join(f"{k + 1}:{X[i, k]:.5f}" for k in range(F)) + "\n")


@pytest.fixture(scope="module")
def letor_file(tmp_path_factory):
    d = tmp_path_factory.mktemp("letor")
    _write_letor(d / "train.txt", 600, np.random.default_rng(7))
    return d / "train.txt"


@pytest.mark.parametrize("model", ["RankNet", "LambdaLoss", "ApproxNDCG", "ListNet", "ListMLE", "STListNet", "RankCosine", "RankMSE", "SoftRank"])
def test_every_loss_learns_to_rank(letor_file, model):
    """Sanity beyond parity: with each loss, a few epochs on a learnable collection lift nDCG@10 well above its start."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_letor_file.py"), str(letor_file), "--model", model,
                          "--epochs", "15", "--rough-batch-size", "8192"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("epoch ")]
    first = float(lines[0].split("nDCG@10 ")[1].split()[0])
    best = max(float(l.split("nDCG@10 ")[1].split()[0]) for l in lines)
    assert best > 0.75 and best > first, (model, first, best, out.stdout[-1500:])


def test_example_trains_from_a_letor_file(tmp_path):
    rng = np.random.default_rng(7)
    _write_letor(tmp_path / "train.txt", 600, rng)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_letor_file.py"), str(tmp_path / "train.txt"),
                          "--epochs", "12", "--rough-batch-size", "8192"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("epoch ")]
    assert len(lines) >= 2, out.stdout
    first = float(lines[0].split("nDCG@10 ")[1].split()[0])
    last = float(lines[-1].split("nDCG@10 ")[1].split()[0])
    assert last > first + 0.05 and last > 0.8, (first, last, out.stdout)
    assert "nERR" in out.stdout and "AP" in out.stdout
