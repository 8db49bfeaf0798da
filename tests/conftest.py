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


@pytest.fixture(autouse=True)
def _tests_opt_out_of_strict_device(request):
    """The product default is strict: FusedLinear / FusedStack refuse CPU tensors (ptranking_amd/linear.py).  Only the `not gpu` suite opts
    out wholesale: it exercises the host-side plumbing on CPU tensors (state_dict round trips, the reference's own kfold loop on the
    installed classes); tests/test_host_cpu.py checks the strict default itself.  A `gpu` test runs under the product default — a module
    that accidentally runs on the CPU fails instead of falling through to torch's F.linear — unless its module sets
    `CPU_REFERENCE_MODULES = True`: those build their float64 / fp32 REFERENCE by running the very same module objects on CPU, where a
    FusedLinear is torch's nn.Linear and executes torch's F.linear, not a kernel of ours."""
    is_gpu = request.node.get_closest_marker("gpu") is not None
    if is_gpu and not getattr(request.module, "CPU_REFERENCE_MODULES", False):
        prev = os.environ.pop("PTR_STRICT_DEVICE", None)
        yield
        if prev is not None:
            os.environ["PTR_STRICT_DEVICE"] = prev
        return
    prev = os.environ.get("PTR_STRICT_DEVICE")        # (not monkeypatch: tests that call monkeypatch.undo() mid-way would drop it)
    os.environ["PTR_STRICT_DEVICE"] = "0"
    yield
    if prev is None:
        os.environ.pop("PTR_STRICT_DEVICE", None)
    else:
        os.environ["PTR_STRICT_DEVICE"] = prev
