#!/usr/bin/env python3
"""Golden fixtures for batch-norm scorers on PADDED query batches, produced by RUNNING THE REFERENCE's own
get_stacked_FFNet (ptranking/base/utils.py:288-356 with :200-223 LTRBatchNorm, :227-286 LTRBatchNorm2) on the UNPADDED lists (build
container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ffnet_padded.py

The reference batches equal-length lists only (data_utils.py:683-742), so its batch norm never sees a padded row.  What a padded
batch of ragged queries has to reproduce:
  'BN'  (LTRBatchNorm = BatchNorm1d over batch x docs, no running statistics): documents are scored independently given the batch
        statistics, so the reference applied to ONE list holding all real documents of the batch, [1, sum(lens), F], is the same
        computation — outputs, dX and every parameter gradient of sum(y * R);
  'BN2' (LTRBatchNorm2, per-query statistics): the reference applied to every query on its own, [1, len_q, F]; parameter gradients add
        up over the queries; the moving statistics follow utils.py:242-245 — every query updates the SAME incoming statistics and the
        batch average of the updated values is kept — evaluated here query by query from the same initial statistics and averaged.
Dropout is 0.  Arrays per case: sd/* (state_dict before the forward), x [B, L, F] (padded rows hold junk, NOT zeros), lens, R [B, L, 1]
(0 at padded rows: the loss kernels emit exactly 0 there), y / dx [B, L, .] (0 at padded rows), grad/*, (BN2) moving_after/*.
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
    "bn_default": dict(F=136, L=24, lens=[24, 7, 1, 15, 24, 3], num_layers=5, AF='GE', TL_AF='S', apply_tl_af=True, BN=True, bn_type='BN',
                       bn_affine=True),
    "bn_tanh":    dict(F=24, L=40, lens=[40, 2, 33], num_layers=2, AF='T', TL_AF='T', apply_tl_af=False, BN=True, bn_type='BN', bn_affine=False),
    "bn2_gelu":   dict(F=46, L=16, lens=[16, 5, 1, 9, 12], num_layers=3, AF='GE', TL_AF='S', apply_tl_af=True, BN=True, bn_type='BN2',
                       bn_affine=True),
    "bn2_relu":   dict(F=136, L=64, lens=[64, 30, 2], num_layers=2, AF='R', TL_AF='S', apply_tl_af=False, BN=True, bn_type='BN2',
                       bn_affine=False),
}


def main():
    from ptranking.base.utils import get_stacked_FFNet

    torch.manual_seed(SEED)
    store = {}
    for tag, c in CASES.items():
        ff_dims = [c["F"]] + [100] * c["num_layers"] + [1]
        net = get_stacked_FFNet(ff_dims=ff_dims, AF=c["AF"], TL_AF=c["TL_AF"], apply_tl_af=c["apply_tl_af"], dropout=0.0, BN=c["BN"],
                                bn_type=c["bn_type"], bn_affine=c["bn_affine"], device='cpu')
        with torch.no_grad():
            for n_, p in net.named_parameters():
                if p.dim() != 2:
                    p.add_(0.3 * torch.randn_like(p))
        net.train()
        lens = c["lens"]
        B, L, F = len(lens), c["L"], c["F"]
        x = torch.randn(B, L, F) * 1.5 + 0.2
        R = torch.randn(B, L, 1)
        for b, n in enumerate(lens):
            x[b, n:] = 7.0 * torch.randn(L - n, F) + 3.0      # junk in the padding: must not reach any statistic
            R[b, n:] = 0.0
        for k, v in net.state_dict().items():
            store[f"{tag}/sd/{k}"] = v.numpy().copy()
        y = torch.zeros(B, L, 1)
        dx = torch.zeros(B, L, F)
        if c["bn_type"] == 'BN':
            xc = torch.cat([x[b, :n] for b, n in enumerate(lens)], dim=0).unsqueeze(0).requires_grad_(True)     # [1, sum(lens), F]
            Rc = torch.cat([R[b, :n] for b, n in enumerate(lens)], dim=0).unsqueeze(0)
            yc = net(xc)
            (yc * Rc).sum().backward()
            off = 0
            for b, n in enumerate(lens):
                y[b, :n] = yc[0, off:off + n].detach()
                dx[b, :n] = xc.grad[0, off:off + n]
                off += n
        else:
            mods = [(name, m) for name, m in net.named_modules() if hasattr(m, "moving_mean")]
            mm0 = {name: (m.moving_mean.clone(), m.moving_var.clone()) for name, m in mods}
            acc = {name: [torch.zeros_like(m.moving_mean), torch.zeros_like(m.moving_var)] for name, m in mods}
            for b, n in enumerate(lens):
                for name, m in mods:                              # every query starts from the incoming statistics
                    m.moving_mean, m.moving_var = mm0[name][0].clone(), mm0[name][1].clone()
                xq = x[b:b + 1, :n].clone().requires_grad_(True)
                yq = net(xq)
                (yq * R[b:b + 1, :n]).sum().backward()            # parameter gradients accumulate over the queries
                y[b, :n] = yq[0].detach()
                dx[b, :n] = xq.grad[0]
                for name, m in mods:
                    acc[name][0] += m.moving_mean / B
                    acc[name][1] += m.moving_var / B
            for name, _ in mods:
                store[f"{tag}/moving_after/{name}/mean"] = acc[name][0].numpy().copy()
                store[f"{tag}/moving_after/{name}/var"] = acc[name][1].numpy().copy()
        store[f"{tag}/x"] = x.numpy()
        store[f"{tag}/R"] = R.numpy()
        store[f"{tag}/lens"] = np.asarray(lens, dtype=np.int32)
        store[f"{tag}/y"] = y.numpy()
        store[f"{tag}/dx"] = dx.numpy()
        for k, p in net.named_parameters():
            store[f"{tag}/grad/{k}"] = p.grad.numpy()
        store[f"{tag}/cfg"] = np.array([F, B, L, c["num_layers"]])
    out = os.path.join(HERE, "ffnet_padded.npz")
    np.savez_compressed(out, **store)
    print(f"wrote {out}: {len(store)} arrays, {os.path.getsize(out) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
