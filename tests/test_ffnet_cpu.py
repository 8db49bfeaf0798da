"""CPU: the pointsf modules this package builds (host.build_pointsf: the reference's module names, LTRBatchNorm / LTRBatchNorm2
mirrors) reproduce the REFERENCE's own get_stacked_FFNet — outputs and autograd gradients recorded by
tests/golden/make_golden_ffnet.py from /root/reference (ptranking/base/utils.py:288-356).  This pins the torch-module side that
tests/test_stack_gpu.py and tests/test_scorer_gpu.py compare the HIP kernels against."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ffnet.npz"))
CASES = {
    "default":  dict(num_layers=5, AF='GE', TL_AF='S', apply_tl_af=True, BN=True, bn_type='BN', bn_affine=True),
    "relu3":    dict(num_layers=3, AF='R', TL_AF='S', apply_tl_af=False, BN=False, bn_type=None, bn_affine=False),
    "bn2":      dict(num_layers=3, AF='GE', TL_AF='S', apply_tl_af=True, BN=True, bn_type='BN2', bn_affine=True),
    "tanh_bn":  dict(num_layers=2, AF='T', TL_AF='T', apply_tl_af=True, BN=True, bn_type='BN', bn_affine=False),
    "selu":     dict(num_layers=3, AF='SE', TL_AF='S', apply_tl_af=True, BN=False, bn_type=None, bn_affine=False),
}


def close(got, ref, what, rtol=1e-5, el_rtol=1e-3, floor=0.0):
    """max-norm gate + element-wise gate.  `floor`: the scale of the quantity's family (all parameter gradients of a case): a
    gradient that is mathematically zero — the bias of a Linear in front of a batch norm — is rounding noise in the reference too."""
    got = np.asarray(got, dtype=np.float64).reshape(np.asarray(ref).shape)
    ref = np.asarray(ref, dtype=np.float64)
    scale = max(np.abs(ref).max(), floor, 1e-30)
    d = np.abs(got - ref)
    assert d.max() <= 1e-5 + rtol * scale, f"{what}: max|diff| {d.max():.3e} (scale {scale:.3e})"
    assert np.all(d <= el_rtol * np.abs(ref) + 1e-5 * scale), f"{what}: element-wise gate"


def load_case(tag, device="cpu"):
    from ptranking_amd.host import build_pointsf
    F = int(G[f"{tag}/cfg"][0])
    net = build_pointsf(num_features=F, dropout=0.0, **CASES[tag])
    sd = {k[len(tag) + 4:]: torch.from_numpy(G[k]) for k in G.files if k.startswith(f"{tag}/sd/")}
    own = net.state_dict()
    assert set(sd) == set(own), f"{tag}: state_dict keys differ from the reference's: {sorted(set(sd) ^ set(own))}"
    net.load_state_dict(sd)
    for name, m in net.named_modules():                       # LTRBatchNorm2's moving statistics are plain attributes, not buffers
        if hasattr(m, "moving_mean"):
            m.moving_mean = torch.zeros_like(m.moving_mean)
            m.moving_var = torch.ones_like(m.moving_var)
    return net.to(device).train()


def check_case(tag, device, tol=1e-5):
    net = load_case(tag, device)
    x = torch.from_numpy(G[f"{tag}/x"]).to(device).requires_grad_(True)
    R = torch.from_numpy(G[f"{tag}/R"]).to(device)
    y = net(x)
    (y.reshape(R.shape) * R).sum().backward()
    close(y.detach().cpu().numpy(), G[f"{tag}/y"], f"{tag} output", rtol=tol)
    close(x.grad.cpu().numpy(), G[f"{tag}/dx"], f"{tag} dX", rtol=tol)
    gscale = max(float(np.abs(G[f]).max()) for f in G.files if f.startswith(f"{tag}/grad/"))
    for k, p in net.named_parameters():
        close(p.grad.cpu().numpy(), G[f"{tag}/grad/{k}"], f"{tag} grad {k}", rtol=tol, floor=1e-1 * gscale)
    if CASES[tag]["bn_type"] == 'BN2':
        for name, m in net.named_modules():
            if hasattr(m, "moving_mean"):
                close(m.moving_mean.cpu().numpy(), G[f"{tag}/moving_after/{name}/mean"], f"{tag} moving mean {name}", rtol=tol)
                close(m.moving_var.cpu().numpy(), G[f"{tag}/moving_after/{name}/var"], f"{tag} moving var {name}", rtol=tol)
        with torch.no_grad():
            close(net(x.detach()).cpu().numpy(), G[f"{tag}/y_nograd"], f"{tag} output under no_grad", rtol=tol)


@pytest.mark.parametrize("tag", sorted(CASES))
def test_pointsf_modules_reproduce_the_reference(tag):
    check_case(tag, "cpu")
