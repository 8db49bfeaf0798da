"""GPU: the hand-written pointsf paths (FusedStack on linear.hip / bnact.hip; FusedPointScorer on scorer.hip / scorer_bwd.hip)
against the REFERENCE's own get_stacked_FFNet outputs and gradients (tests/golden/ffnet.npz, made by
tests/golden/make_golden_ffnet.py from ptranking/base/utils.py:288-356)."""
import numpy as np
import pytest
import torch

from test_ffnet_cpu import CASES, G, check_case, close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", sorted(CASES))
def test_fused_stack_reproduces_the_reference(tag):
    from ptranking_amd.linear import FusedStack
    from ptranking_amd.host import build_pointsf
    assert isinstance(build_pointsf(num_features=8, dropout=0.0, **CASES[tag]), FusedStack)
    check_case(tag, "cuda", tol=2e-5)


def test_fused_point_scorer_reproduces_the_reference():
    """The single-kernel 3 x ReLU scorer (the headline bench's) loaded from the reference's state_dict."""
    from ptranking_amd.scorer import FusedPointScorer
    tag = "relu3"
    F = int(G[f"{tag}/cfg"][0])
    sf = FusedPointScorer(F, num_layers=3, dropout=0.0).cuda()
    sf.load_state_dict({k[len(tag) + 4:]: torch.from_numpy(G[k]) for k in G.files if k.startswith(f"{tag}/sd/")})
    sf.train()
    x = torch.from_numpy(G[f"{tag}/x"]).cuda().requires_grad_(True)
    R = torch.from_numpy(G[f"{tag}/R"]).cuda()
    y = sf(x)
    (y.reshape(R.shape) * R).sum().backward()
    close(y.detach().cpu().numpy(), G[f"{tag}/y"], "output", rtol=2e-5)
    close(x.grad.cpu().numpy(), G[f"{tag}/dx"], "dX", rtol=2e-5) if x.grad is not None else None
    got = {k: v for k, v in sf.views(grad=True).items()}
    gscale = max(float(np.abs(G[f]).max()) for f in G.files if f.startswith(f"{tag}/grad/"))
    for k in [f[len(tag) + 6:] for f in G.files if f.startswith(f"{tag}/grad/")]:
        close(got[k].cpu().numpy(), G[f"{tag}/grad/{k}"], f"grad {k}", rtol=2e-5, floor=1e-1 * gscale)
