"""nn.Linear layers and (Dropout -> Linear -> ReLU)* stacks of the scoring functions on the hand-written fp32-MFMA kernels of
csrc/linear.hip — no library GEMM behind any Linear of a scoring function on the GPU.

Reference: ptranking/base/utils.py:288-356 (get_stacked_FFNet), ptranking/base/list_ranker.py:176-254,303-350 (the listsf head /
tail stacks with ff_dims 128/256/512, the Q|K|V and fc projections).

  `linear(x, weight, bias)`   differentiable y = x W^T + b (forward / backward-input / backward-weight kernels)
  `FusedLinear`               nn.Linear (same parameters, initialisation, state_dict) on `linear`
  `ReluStack`                 nn.Sequential with the reference's module names (dr_i / ff_{i+1} / act_{i+1}) whose forward runs the
                              whole stack as one autograd node: ReLU and the NEXT layer's dropout live in the producing kernel's
                              epilogue (counter-based masks, recomputed in backward, never stored), the backward gates ride in
                              the backward-input kernel's epilogue.
CPU tensors take the plain torch ops (host-logic tests only; the product path is the GPU one).
"""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

NONE, RELU, RELU_DROPOUT = 0, 1, 2


def _rows(x):
    """[..., K] -> contiguous-row 2-D view [R, K] and its leading dimension."""
    x2 = x.reshape(-1, x.shape[-1])
    if x2.dtype != torch.float32:
        x2 = x2.float()
    if x2.stride(-1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < x2.shape[1]):
        x2 = x2.contiguous()
    return x2, (x2.stride(0) if x2.shape[0] > 1 else x2.shape[1])


def _fwd(x2, ldx, weight, bias, act=NONE, p=0.0, seed=0, site=0):
    R, K = x2.shape
    N = weight.shape[0]
    y = torch.empty((R, N), device=x2.device, dtype=torch.float32)
    with torch.cuda.device(x2.device):
        _lib.call("ptr_linear_forward", _lib.ptr(x2), ldx, _lib.ptr(weight), _lib.ptr(bias), R, K, N, act, C.c_float(p), C.c_uint64(seed), site,
                  _lib.ptr(y), N, _lib.current_stream(x2.device))
    return y


def _bwd_input(dy2, weight, gate=None, p=0.0):
    R, N = dy2.shape
    K = weight.shape[1]
    dx = torch.empty((R, K), device=dy2.device, dtype=torch.float32)
    with torch.cuda.device(dy2.device):
        _lib.call("ptr_linear_backward_input", _lib.ptr(dy2), N, _lib.ptr(weight), R, K, N, _lib.ptr(gate), K, C.c_float(p), _lib.ptr(dx), K,
                  _lib.current_stream(dy2.device))
    return dx


def _bwd_weight(x2, ldx, dy2, want_bias):
    R, K = x2.shape
    N = dy2.shape[1]
    dev = x2.device
    ws = torch.empty(_lib.query("ptr_linear_backward_weight_ws_floats", R, K, N), device=dev, dtype=torch.float32)
    dw = torch.empty((N, K), device=dev, dtype=torch.float32)
    db = torch.empty(N, device=dev, dtype=torch.float32) if want_bias else None
    with torch.cuda.device(dev):
        _lib.call("ptr_linear_backward_weight", _lib.ptr(x2), ldx, _lib.ptr(dy2), N, R, K, N, _lib.ptr(ws), _lib.ptr(dw), _lib.ptr(db),
                  _lib.current_stream(dev))
    return dw, db


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x2, ldx = _rows(x)
        w = weight.contiguous()
        y = _fwd(x2, ldx, w, bias)
        ctx.save_for_backward(x2, w)
        ctx.ldx, ctx.has_bias = ldx, bias is not None
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        dx = _bwd_input(dy2, w).view(*dy.shape[:-1], w.shape[1]) if ctx.needs_input_grad[0] else None
        dw, db = _bwd_weight(x2, ctx.ldx, dy2, ctx.has_bias)
        return dx, dw, db


def linear(x, weight, bias=None):
    if not x.is_cuda:
        return F.linear(x, weight, bias)
    return _LinearFn.apply(x, weight, bias)


class FusedLinear(nn.Linear):
    def forward(self, x):
        return linear(x, self.weight, self.bias)


class _ReluStackFn(torch.autograd.Function):
    """x -> dropout -> (Linear -> ReLU -> dropout)* -> Linear -> ReLU -> Linear [-> ReLU]   (utils.py:296-322 with AF='R', no BN)."""

    @staticmethod
    def forward(ctx, x, p, seed, tail_relu, *params):
        n = len(params) // 2                       # Linear layers; n - 1 of them hidden
        ws, bs = params[0::2], params[1::2]
        x2, ldx = _rows(x)
        dev = x2.device
        R, K0 = x2.shape
        if n > 1 and p > 0.0:
            if K0 % 4 or ldx % 4:
                x2, ldx = x2.contiguous(), K0
                if K0 % 4:
                    raise NotImplementedError("input width must be a multiple of 4 for the fused dropout")
            a = torch.empty((R, K0), device=dev, dtype=torch.float32)
            with torch.cuda.device(dev):
                _lib.call("ptr_dropout_apply", _lib.ptr(x2), ldx, R, K0, C.c_float(p), C.c_uint64(seed), 0, _lib.ptr(a), K0, _lib.current_stream(dev))
            acts, lda = [a], K0
        else:
            acts, lda = [x2], ldx
        for i in range(n - 1):                     # hidden layers: ReLU, and the dropout in front of the next hidden Linear
            last_hidden = i == n - 2
            a = _fwd(acts[-1], lda if i == 0 else acts[-1].shape[1], ws[i], bs[i], RELU if (last_hidden or p == 0.0) else RELU_DROPOUT, p, seed, i + 1)
            acts.append(a)
        out = _fwd(acts[-1], lda if n == 1 else acts[-1].shape[1], ws[-1], bs[-1], RELU if tail_relu else NONE)
        ctx.save_for_backward(out if tail_relu else None, *acts, *ws)
        ctx.meta = (n, p, seed, tail_relu, lda, [b is not None for b in bs])
        return out.view(*x.shape[:-1], ws[-1].shape[0])

    @staticmethod
    def backward(ctx, dout):
        n, p, seed, tail_relu, lda, has_b = ctx.meta
        saved = ctx.saved_tensors
        out, acts, ws = saved[0], saved[1:1 + n], saved[1 + n:]
        dev = dout.device
        dz = dout.reshape(-1, dout.shape[-1]).contiguous()
        if tail_relu:
            g = torch.empty_like(dz)
            with torch.cuda.device(dev):
                _lib.call("ptr_relu_gate", _lib.ptr(dz), _lib.ptr(out), C.c_int64(dz.numel()), _lib.ptr(g), _lib.current_stream(dev))
            dz = g
        grads = [None] * (2 * n)
        for i in range(n - 1, -1, -1):
            a_in = acts[i]
            dw, db = _bwd_weight(a_in, lda if i == 0 else a_in.shape[1], dz, has_b[i])
            grads[2 * i], grads[2 * i + 1] = dw, db
            if i > 0:                               # gate by the stored ReLU (+ dropout) output of the layer below
                dropped = p > 0.0 and i < n - 1     # acts[i] carries dropout unless it feeds the last Linear
                dz = _bwd_input(dz, ws[i], gate=a_in, p=p if dropped else 0.0)
            elif ctx.needs_input_grad[0]:
                dx = _bwd_input(dz, ws[0])
                if n > 1 and p > 0.0:
                    dxd = torch.empty_like(dx)
                    with torch.cuda.device(dev):
                        _lib.call("ptr_dropout_apply", _lib.ptr(dx), dx.shape[1], dx.shape[0], dx.shape[1], C.c_float(p), C.c_uint64(seed), 0,
                                  _lib.ptr(dxd), dx.shape[1], _lib.current_stream(dev))
                    dx = dxd
                return (dx.view(*dout.shape[:-1], ws[0].shape[1]), None, None, None, *grads)
        return (None, None, None, None, *grads)


class ReluStack(nn.Sequential):
    """The Sequential get_stacked_FFNet builds for AF='R', BN=False (module names and state_dict as the reference's), evaluated as
    one fused autograd node on the GPU.  `tail_relu`: apply_tl_af with TL_AF='R' (the listsf head stack, list_ranker.py:318)."""

    tail_relu = False

    def forward(self, x):
        if not x.is_cuda:
            return super().forward(x)
        lins = [m for m in self if isinstance(m, nn.Linear)]
        drops = [m for m in self if isinstance(m, nn.Dropout)]
        p = drops[0].p if (drops and self.training) else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0.0 else 0      # CPU generator: no device sync
        self.last_seed = seed
        params = []
        for m in lins:
            params += [m.weight, m.bias]
        return _ReluStackFn.apply(x, float(p), seed, bool(self.tail_relu), *params)
