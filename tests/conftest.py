"""pytest config: registers the `gpu` marker (tests that need a real MI355X) and puts the repo root on sys.path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def _deterministic_torch_seed():
    """Every test starts from the reference's global seed (ltr_global.py:7): model initialisation and the dropout seeds drawn
    from torch's CPU generator are the same on every run."""
    import torch
    torch.manual_seed(137)
    yield
