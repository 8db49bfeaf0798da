"""The reference's listwise (permutation-equivariant) scorer `listsf` with its hot pieces on hand-written HIP kernels.

Mirror of ptranking/base/list_ranker.py (same class names, constructor arguments, parameter names — `state_dict`s are
interchangeable with the reference's head_ffnns / encoder / tail_ffnns): head feed-forward stack -> N encoder layers of
multi-head self-attention + LayerNorm (three published variants: DASALC, AllRank, AttnDIN) -> tail feed-forward stack.

What runs where:
  * the attention core  Q K^T / sqrt(d_h) -> softmax -> Dropout -> . V  (list_ranker.py:216-240) and its backward: the fused
    fp32-MFMA kernels behind ptr_mhsa_forward / ptr_mhsa_backward (csrc/listsf.hip) — no [B, H, L, L] tensor is ever written;
  * LayerNorm (list_ranker.py:152-174, unbiased std, eps added to the std): ptr_layernorm_forward / _backward;
  * the Linear projections and the feed-forward stacks: library GEMMs through torch (hipBLASLt).
Attention dropout uses the kernels' counter-based generator (seeded per call from torch's CPU generator, so
`torch.manual_seed` keeps runs reproducible): statistically, not bit-wise, the same masks as nn.Dropout.
Padded batches: `lens` (int32 [B]) excludes padded documents as attention KEYS; the reference has no padding at all.
There is no CPU / eager fallback: CPU tensors raise.
"""
import copy
import ctypes as C
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import dp
from .host import SplitKLinear, build_stacked_ffnet
from .linear import linear

MAX_HEAD_DIM = 128        # PTR_MHSA_MAX_HEAD_DIM
Encoder_Type = ['DASALC', 'AllRank', 'AttnDIN']   # list_ranker.py:13


def _need_gpu(t, who):
    if not t.is_cuda:
        raise _lib.NativeLibraryError(f"{who}: tensor on {t.device}; ptranking_amd has no CPU fallback — move it to the GPU")


def _voff(t, floats):
    return C.c_void_p(t.data_ptr() + 4 * floats)


DS_SPILL_MAX_BYTES = 8 << 30     # scaled-dS scratch of the attention backward (4 B H L^2 bytes): 0.54 GB at config 5 — 288 GB of HBM3E has room
DS_SPILL_FREE_FRACTION = 0.15    # ... but never more than this share of what the device has free right now
_DS_SPILL_ON = os.environ.get("PTR_ATTN_DS_SPILL", "1") != "0"      # A/B runs; read once at import


def _ds_scratch(B, H, L, dev):
    """[B*H*L*L] scratch that lets the dQ kernel be ONE GEMM unit (dS . K) instead of recomputing S and dP (ptr_mhsa_backward's ds_ws);
    None — the recomputing dQ kernel — for short lists (the recomputation is cheap there), beyond DS_SPILL_MAX_BYTES or
    DS_SPILL_FREE_FRACTION of the free device memory, when the allocation fails, or with PTR_ATTN_DS_SPILL=0."""
    n = B * H * L * L
    if L < 128 or 4 * n > DS_SPILL_MAX_BYTES or not _DS_SPILL_ON:
        return None
    try:
        free, _ = torch.cuda.mem_get_info(dev)
        cached = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)       # the caching allocator can serve from this too
        if 4 * n > DS_SPILL_FREE_FRACTION * (free + cached):
            return None
        return torch.empty(n, device=dev, dtype=torch.float32)
    except torch.cuda.OutOfMemoryError:
        return None


class _MhsaCoreFn(torch.autograd.Function):
    """Three separate [B, L, F] tensors (row stride F)."""

    @staticmethod
    def forward(ctx, Q, K, V, lens, n_heads, p, seed, site):
        B, L, Fdim = Q.shape
        dev = Q.device
        Q, K, V = Q.contiguous(), K.contiguous(), V.contiguous()
        O = torch.empty_like(Q)
        lse = torch.empty(B * n_heads * L, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.call("ptr_mhsa_forward", _lib.ptr(Q), _lib.ptr(K), _lib.ptr(V), Fdim, _lib.ptr(lens), B, L, Fdim, n_heads, C.c_float(p),
                      C.c_uint64(seed), site, _lib.ptr(O), _lib.ptr(lse), _lib.current_stream(dev))
        ctx.save_for_backward(Q, K, V, O, lse, lens)
        ctx.meta = (n_heads, p, seed, site)
        return O

    @staticmethod
    def backward(ctx, dO):
        Q, K, V, O, lse, lens = ctx.saved_tensors
        n_heads, p, seed, site = ctx.meta
        B, L, Fdim = Q.shape
        dev = Q.device
        dO = dO.contiguous()
        dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
        dvec = torch.empty_like(lse)
        ds_ws = _ds_scratch(B, n_heads, L, dev)
        with torch.cuda.device(dev):
            _lib.call("ptr_mhsa_backward", _lib.ptr(Q), _lib.ptr(K), _lib.ptr(V), Fdim, _lib.ptr(O), _lib.ptr(dO), _lib.ptr(lse),
                      _lib.ptr(lens), B, L, Fdim, n_heads, C.c_float(p), C.c_uint64(seed), site, _lib.ptr(dvec), _lib.ptr(dQ),
                      _lib.ptr(dK), _lib.ptr(dV), _lib.ptr(ds_ws), _lib.current_stream(dev))
        return dQ, dK, dV, None, None, None, None, None


class _MhsaPackedFn(torch.autograd.Function):
    """One packed [B, L, 3F] projection Q | K | V (row stride 3F): the kernels read the three column blocks in place and write
    dQ | dK | dV straight into the gradient of the projection."""

    @staticmethod
    def forward(ctx, qkv, lens, n_heads, p, seed, site):
        B, L, F3 = qkv.shape
        Fdim = F3 // 3
        dev = qkv.device
        qkv = qkv.contiguous()
        O = torch.empty((B, L, Fdim), device=dev, dtype=torch.float32)
        lse = torch.empty(B * n_heads * L, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.call("ptr_mhsa_forward", _voff(qkv, 0), _voff(qkv, Fdim), _voff(qkv, 2 * Fdim), F3, _lib.ptr(lens), B, L, Fdim, n_heads,
                      C.c_float(p), C.c_uint64(seed), site, _lib.ptr(O), _lib.ptr(lse), _lib.current_stream(dev))
        ctx.save_for_backward(qkv, O, lse, lens)
        ctx.meta = (n_heads, p, seed, site)
        return O

    @staticmethod
    def backward(ctx, dO):
        qkv, O, lse, lens = ctx.saved_tensors
        n_heads, p, seed, site = ctx.meta
        B, L, F3 = qkv.shape
        Fdim = F3 // 3
        dev = qkv.device
        dO = dO.contiguous()
        dqkv = torch.empty_like(qkv)
        dvec = torch.empty_like(lse)
        ds_ws = _ds_scratch(B, n_heads, L, dev)
        with torch.cuda.device(dev):
            _lib.call("ptr_mhsa_backward", _voff(qkv, 0), _voff(qkv, Fdim), _voff(qkv, 2 * Fdim), F3, _lib.ptr(O), _lib.ptr(dO),
                      _lib.ptr(lse), _lib.ptr(lens), B, L, Fdim, n_heads, C.c_float(p), C.c_uint64(seed), site, _lib.ptr(dvec),
                      _voff(dqkv, 0), _voff(dqkv, Fdim), _voff(dqkv, 2 * Fdim), _lib.ptr(ds_ws), _lib.current_stream(dev))
        return dqkv, None, None, None, None, None


def mhsa_core(Q, K, V, n_heads, p_drop=0.0, seed=0, site=0, lens=None):
    """softmax(Q_h K_h^T / sqrt(d_h)) [dropout] V_h for every head h = column block of width F / n_heads; [B, L, F] in and out."""
    for t in (Q, K, V):
        _need_gpu(t, "mhsa_core")
    if Q.dtype != torch.float32 or Q.dim() != 3 or Q.shape != K.shape or Q.shape != V.shape:
        raise ValueError("mhsa_core expects three fp32 [B, L, F] tensors of equal shape")
    if lens is not None:
        lens = lens.to(device=Q.device, dtype=torch.int32).contiguous()
    return _MhsaCoreFn.apply(Q, K, V, lens, int(n_heads), float(p_drop), int(seed), int(site))


def mhsa_core_packed(qkv, n_heads, p_drop=0.0, seed=0, site=0, lens=None):
    """Same, on the packed projection qkv = [Q | K | V] of shape [B, L, 3F]; returns O [B, L, F]."""
    _need_gpu(qkv, "mhsa_core_packed")
    if qkv.dtype != torch.float32 or qkv.dim() != 3 or qkv.shape[-1] % 3 != 0:
        raise ValueError("mhsa_core_packed expects an fp32 [B, L, 3F] tensor")
    if lens is not None:
        lens = lens.to(device=qkv.device, dtype=torch.int32).contiguous()
    return _MhsaPackedFn.apply(qkv, lens, int(n_heads), float(p_drop), int(seed), int(site))


def mhsa_dropout_mask(B, L, n_heads, p_drop, seed, site, device):
    """Keep-mask [B, H, L, L] of the attention dropout for a given call seed — test helper."""
    out = torch.empty((B, n_heads, L, L), device=device, dtype=torch.float32)
    with torch.cuda.device(out.device):
        _lib.call("ptr_mhsa_dropout_mask", B, L, n_heads, C.c_float(p_drop), C.c_uint64(seed), site, _lib.ptr(out),
                  _lib.current_stream(out.device))
    return out


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, a2, b2, eps):
        dev = x.device
        Fdim = x.shape[-1]
        x2 = x.contiguous().view(-1, Fdim)
        R = x2.shape[0]
        y = torch.empty_like(x2)
        stats = torch.empty((R, 3), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.call("ptr_layernorm_forward", _lib.ptr(x2), _lib.ptr(a2), _lib.ptr(b2), R, Fdim, C.c_float(eps), _lib.ptr(y),
                      _lib.ptr(stats), _lib.current_stream(dev))
        ctx.save_for_backward(x2, a2, stats)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, a2, stats = ctx.saved_tensors
        R, Fdim = x2.shape
        dev = x2.device
        dy2 = dy.contiguous().view(-1, Fdim)
        dx = torch.empty_like(x2)
        da2, db2 = torch.empty_like(a2), torch.empty_like(a2)
        ws = torch.empty(_lib.query("ptr_layernorm_backward_ws_floats", Fdim), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.call("ptr_layernorm_backward", _lib.ptr(x2), _lib.ptr(a2), _lib.ptr(dy2), _lib.ptr(stats), R, Fdim, _lib.ptr(ws),
                      _lib.ptr(dx), _lib.ptr(da2), _lib.ptr(db2), _lib.current_stream(dev))
        return dx.view(dy.shape), da2, db2, None


def layer_norm(x, a_2, b_2, eps=1e-6):
    """a_2 * (x - mean) / (std + eps) + b_2 over the last axis, std unbiased — ptranking/base/list_ranker.py:170-174."""
    _need_gpu(x, "layer_norm")
    if x.dtype != torch.float32:
        raise ValueError("layer_norm expects fp32")
    return _LayerNormFn.apply(x, a_2.contiguous(), b_2.contiguous(), float(eps))


# ------------------------------------------------------------------------------------------------ modules (list_ranker.py:46-281)
def make_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class LayerNorm(nn.Module):
    def __init__(self, hid_dim, eps=1e-6):
        super().__init__()
        self.a_2 = nn.Parameter(torch.ones(hid_dim))
        self.b_2 = nn.Parameter(torch.zeros(hid_dim))
        self.eps = eps

    def forward(self, x):
        return layer_norm(x, self.a_2, self.b_2, self.eps)


class MultiheadAttention(nn.Module):
    def __init__(self, hid_dim, n_heads, dropout=0.1, device=None):
        super().__init__()
        assert hid_dim % n_heads == 0
        if hid_dim // n_heads > MAX_HEAD_DIM:
            raise NotImplementedError(f"head dimension {hid_dim // n_heads} > {MAX_HEAD_DIM} is not covered by the fused kernels")
        self.hid_dim, self.n_heads = hid_dim, n_heads
        self.w_q = SplitKLinear(hid_dim, hid_dim)
        self.w_k = SplitKLinear(hid_dim, hid_dim)
        self.w_v = SplitKLinear(hid_dim, hid_dim)
        self.fc = SplitKLinear(hid_dim, hid_dim, bias=True)
        self.do_dropout = nn.Dropout(dropout)     # kept for its `p` and state_dict parity; the kernel applies the dropout
        self.site = 0                             # dropout stream id, set per encoder layer

    def forward(self, batch_rankings, lens=None):
        # the three projections (list_ranker.py:209-211) as ONE GEMM on the concatenated weights: a [B, L, 3F] tensor whose
        # column blocks the attention kernels read in place (the concatenation of three F x F matrices is negligible)
        w = torch.cat([self.w_q.weight, self.w_k.weight, self.w_v.weight], dim=0)
        b = torch.cat([self.w_q.bias, self.w_k.bias, self.w_v.bias], dim=0)
        qkv = linear(batch_rankings, w, b)
        p = self.do_dropout.p if self.training else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0.0 else 0      # CPU generator: no device sync
        if p > 0.0:     # data-parallel replicas draw the masks of THEIR (query, head, document) rows (dp.py): never shared masks
            seed = dp.local_dropout_seed(seed, batch_rankings.shape[0] * self.n_heads * batch_rankings.shape[1])
        x = mhsa_core_packed(qkv, self.n_heads, p_drop=p, seed=seed, site=self.site, lens=lens)
        self.last_seed = seed
        return self.fc(x)


class PositionwiseFeedForward(nn.Module):
    def __init__(self, num_features, hid_dim, dropout=0.1):
        super().__init__()
        self.w1 = SplitKLinear(num_features, hid_dim)
        self.w2 = SplitKLinear(hid_dim, num_features)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        return self.w2(self.dropout(F.relu(self.w1(x))))


class SublayerConnection(nn.Module):
    def __init__(self, hid_dim, encoder_type=None, dropout=None):
        super().__init__()
        self.encoder_type = encoder_type
        self.norm = LayerNorm(hid_dim=hid_dim)
        if 'AllRank' == encoder_type:
            self.dropout = nn.Dropout(dropout)

    def forward(self, x, sublayer):
        if 'AllRank' == self.encoder_type:
            return x + self.dropout(sublayer(self.norm(x)))
        elif 'DASALC' == self.encoder_type:
            return self.norm(sublayer(x))
        elif 'AttnDIN' == self.encoder_type:
            return self.norm(x + sublayer(x))
        raise NotImplementedError


class EncoderLayer(nn.Module):
    def __init__(self, hid_dim, mhsa, encoder_type=None, fc=None, dropout=None):
        super().__init__()
        self.mhsa, self.hid_dim, self.encoder_type = mhsa, hid_dim, encoder_type
        if 'AllRank' == encoder_type:
            self.fc = fc
            self.sublayer_cont = make_clones(SublayerConnection(hid_dim=hid_dim, encoder_type=encoder_type, dropout=dropout), 2)
        elif encoder_type in ['AttnDIN', 'DASALC']:
            self.sublayer_cont = SublayerConnection(hid_dim=hid_dim, encoder_type=encoder_type)

    def forward(self, x, lens=None):
        if 'AllRank' == self.encoder_type:
            x = self.sublayer_cont[0](x, lambda t: self.mhsa(t, lens))
            return self.sublayer_cont[1](x, self.fc)
        elif self.encoder_type in ['AttnDIN', 'DASALC']:
            return self.sublayer_cont(x, lambda t: self.mhsa(t, lens))
        raise NotImplementedError


class Encoder(nn.Module):
    def __init__(self, layer, num_layers, encoder_type=None):
        super().__init__()
        self.encoder_type = encoder_type
        self.layers = make_clones(layer, num_layers)
        for i, l in enumerate(self.layers):
            l.mhsa.site = i
        if 'AllRank' == encoder_type:
            self.norm = LayerNorm(layer.hid_dim)

    def forward(self, x, lens=None):
        for layer in self.layers:
            x = layer(x, lens)
        if 'AllRank' == self.encoder_type:
            return self.norm(x)
        elif self.encoder_type in ['AttnDIN', 'DASALC']:
            return x
        raise NotImplementedError


def build_listsf(num_features=None, ff_dims=[128, 256, 512], out_dim=1, AF='R', TL_AF='GE', apply_tl_af=False, BN=True, bn_type=None,
                 bn_affine=False, n_heads=2, encoder_layers=3, dropout=0.1, encoder_type=None, device=None):
    """ini_listsf, ptranking/base/list_ranker.py:303-350 -> {'head_ffnns', 'encoder', 'tail_ffnns'}."""
    head_ffnns = build_stacked_ffnet([num_features] + list(ff_dims) + [num_features], AF=AF, TL_AF=AF, apply_tl_af=True,
                                     dropout=dropout, BN=BN, bn_type=bn_type, bn_affine=bn_affine, device=device)
    mhsa = MultiheadAttention(hid_dim=num_features, n_heads=n_heads, dropout=dropout, device=device)
    if 'AllRank' == encoder_type:
        fc = PositionwiseFeedForward(num_features, hid_dim=num_features, dropout=dropout)
        layer = EncoderLayer(hid_dim=num_features, mhsa=copy.deepcopy(mhsa), encoder_type=encoder_type, fc=fc, dropout=dropout)
    elif encoder_type in ('DASALC', 'AttnDIN'):
        layer = EncoderLayer(hid_dim=num_features, mhsa=copy.deepcopy(mhsa), encoder_type=encoder_type)
    else:
        raise NotImplementedError(encoder_type)
    encoder = Encoder(layer=layer, num_layers=encoder_layers, encoder_type=encoder_type)
    tail_ffnns = build_stacked_ffnet([num_features] + list(ff_dims) + [out_dim], AF=AF, TL_AF=TL_AF, apply_tl_af=apply_tl_af,
                                     BN=BN, bn_type=bn_type, bn_affine=bn_affine, device=device)
    return {'head_ffnns': head_ffnns, 'encoder': encoder, 'tail_ffnns': tail_ffnns}


def listsf_forward(list_sf, encoder_type, batch_q_doc_vectors, lens=None):
    """ListNeuralRanker.forward, ptranking/base/list_ranker.py:352-378."""
    fc_map = list_sf['head_ffnns'](batch_q_doc_vectors)
    if 'AllRank' == encoder_type:
        out = list_sf['tail_ffnns'](list_sf['encoder'](fc_map, lens))
    elif 'DASALC' == encoder_type:
        enc = list_sf['encoder'](batch_q_doc_vectors, lens)
        out = list_sf['tail_ffnns']((enc + 1.0) * fc_map)
    elif 'AttnDIN' == encoder_type:
        enc = list_sf['encoder'](fc_map, lens)
        out = list_sf['tail_ffnns'](enc + batch_q_doc_vectors)
    else:
        raise NotImplementedError
    return torch.squeeze(out, dim=2)


class FusedListScorerMixin:
    """Makes a ranker (the stand-alone base or the reference's AdhocNeuralRanker) build the listsf scorer from the modules above
    and route `lens` of padded batches into the attention kernels."""

    def ini_listsf(self, **kw):
        list_sf = build_listsf(device=self.device, **kw)
        if self.gpu:
            list_sf = {k: m.to(self.device) for k, m in list_sf.items()}
        return list_sf

    def forward(self, batch_q_doc_vectors):
        if getattr(self, 'sf_id', 'pointsf') != 'listsf':
            return super().forward(batch_q_doc_vectors)
        return listsf_forward(self.list_sf, self.encoder_type, batch_q_doc_vectors, getattr(self, '_batch_lens', None))
