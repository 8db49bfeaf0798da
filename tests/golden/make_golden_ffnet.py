#!/usr/bin/env python3
"""Golden fixtures for the pointwise scoring function (pointsf), produced by RUNNING THE REFERENCE's own builder on CPU (build
container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ffnet.py

Reference entry points exercised: ptranking/base/utils.py:288-356 get_stacked_FFNet (with :101-143 get_AF, :200-223 LTRBatchNorm,
:227-286 LTRBatchNorm2 / ltr_batch_norm) through the dimension rule of ptranking/base/point_ranker.py:31-42 (ff_dims =
[num_features] + [h_dim] * num_layers + [out_dim]).  Cases: the driver's DEFAULT pointsf (parameter.py:145-146: 5 layers, GELU,
BN 'BN' affine, Sigmoid tail), the 3 x ReLU / no-BN scorer of the headline bench, a BN2 (per-query) stack in training mode and
under torch.no_grad() (moving statistics), and two more activations.  Dropout is 0 (different generators cannot be pinned by a
fixture; the GPU tests pin dropout against the modules with the kernel's own mask).  Every case holds the state_dict, the input
[B, L, F], the output, and the gradients of sum(output * R) w.r.t. the input and every parameter — the reference's own autograd.
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference"
if not os.path.isdir(REF):
    raise SystemExit("the reference tree is only mounted in the build container")
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 137

CASES = {
    "default":  dict(F=136, B=4, L=19, num_layers=5, AF='GE', TL_AF='S', apply_tl_af=True, BN=True, bn_type='BN', bn_affine=True),
    "relu3":    dict(F=136, B=3, L=33, num_layers=3, AF='R', TL_AF='S', apply_tl_af=False, BN=False, bn_type=None, bn_affine=False),
    "bn2":      dict(F=46, B=5, L=12, num_layers=3, AF='GE', TL_AF='S', apply_tl_af=True, BN=True, bn_type='BN2', bn_affine=True),
    "tanh_bn":  dict(F=24, B=2, L=40, num_layers=2, AF='T', TL_AF='T', apply_tl_af=True, BN=True, bn_type='BN', bn_affine=False),
    "selu":     dict(F=700, B=2, L=9, num_layers=3, AF='SE', TL_AF='S', apply_tl_af=True, BN=False, bn_type=None, bn_affine=False),
}


def main():
    from ptranking.base.utils import get_stacked_FFNet

    torch.manual_seed(SEED)
    store = {}
    for tag, c in CASES.items():
        ff_dims = [c["F"]] + [100] * c["num_layers"] + [1]
        net = get_stacked_FFNet(ff_dims=ff_dims, AF=c["AF"], TL_AF=c["TL_AF"], apply_tl_af=c["apply_tl_af"], dropout=0.0, BN=c["BN"],
                                bn_type=c["bn_type"], bn_affine=c["bn_affine"], device='cpu')
        with torch.no_grad():                     # non-trivial biases / affine parameters
            for n_, p in net.named_parameters():
                if p.dim() != 2:
                    p.add_(0.3 * torch.randn_like(p))
        net.train()
        x = (torch.randn(c["B"], c["L"], c["F"]) * 1.5 + 0.2).requires_grad_(True)
        R = torch.randn(c["B"], c["L"], 1)
        for k, v in net.state_dict().items():     # BEFORE the forward (BN2 replaces its moving statistics in forward)
            store[f"{tag}/sd/{k}"] = v.numpy().copy()
        y = net(x)
        (y * R).sum().backward()
        store[f"{tag}/x"] = x.detach().numpy()
        store[f"{tag}/R"] = R.numpy()
        store[f"{tag}/y"] = y.detach().numpy()
        store[f"{tag}/dx"] = x.grad.numpy()
        for k, p in net.named_parameters():
            store[f"{tag}/grad/{k}"] = p.grad.numpy()
        if c["bn_type"] == 'BN2':
            for name, m in net.named_modules():
                if hasattr(m, "moving_mean"):
                    store[f"{tag}/moving_after/{name}/mean"] = m.moving_mean.numpy().copy()
                    store[f"{tag}/moving_after/{name}/var"] = m.moving_var.numpy().copy()
            with torch.no_grad():                 # prediction mode of LTRBatchNorm2: the moving statistics just updated
                store[f"{tag}/y_nograd"] = net(x.detach()).numpy()
        store[f"{tag}/cfg"] = np.array([c["F"], c["B"], c["L"], c["num_layers"]])
    out = os.path.join(HERE, "ffnet.npz")
    np.savez_compressed(out, **store)
    print(f"wrote {out}: {len(store)} arrays, {os.path.getsize(out) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
