"""CPU model of the packed-key sort the one-wavefront sort / metric kernels use (csrc/ptr_device.h `sort_scores_packed`): the argument that
its result is EXACT whenever it does not fall back, checked on adversarial inputs without a GPU.

The kernel orders a list by ONE sort of 32-bit keys = order-preserving integer image of the fp32 score with its low log2(N) bits replaced
by N - 1 - index, repairs isolated out-of-order neighbours with one odd-even transposition round on the true (score, index) pairs, and
accepts the row only if every adjacent pair is ordered (else: the float sort + rank search).  This file restates those steps in numpy and
asserts, for every input, either "accepted and identical to the stable descending argsort" or "rejected"."""
import numpy as np
import pytest


def ordered_image(x):
    """uint32 whose unsigned order is the fp32 order (-0.0 keyed as +0.0, as in the kernel: x + 0.0f)."""
    x = (np.asarray(x, np.float32) + np.float32(0.0)).astype(np.float32)
    b = x.view(np.int32).astype(np.int64)
    mask = np.where(b < 0, 0xFFFFFFFF, 0x80000000)
    return ((b & 0xFFFFFFFF) ^ mask).astype(np.uint32)


def packed_sort(scores, n, N):
    """Returns (accepted, order) for the first n of N padded positions, N a power of two >= len(scores)."""
    s = np.full(N, -np.inf, np.float32)
    s[:len(scores)] = scores
    if np.isnan(s[:n]).any():
        return False, None
    idx = np.arange(N)
    key = np.where(idx < n, (ordered_image(s) & np.uint32(~(N - 1) & 0xFFFFFFFF)) | (N - 1 - idx).astype(np.uint32), 0).astype(np.uint32)
    ks = np.sort(key)[::-1]
    order = (N - 1 - (ks & np.uint32(N - 1))).astype(np.int64)
    sc = s[order]

    def wrong(a, b):                       # positions a, a + 1 = b out of (score descending, index ascending) order
        return sc[a] < sc[b] or (sc[a] == sc[b] and order[a] > order[b])
    coll = any(((int(ks[p]) ^ int(ks[p + 1])) < N) and p + 1 < n for p in range(N - 1))
    if coll:
        for parity in (0, 1):              # one odd-even transposition round
            for a in range(parity, N - 1, 2):
                if wrong(a, a + 1):
                    order[[a, a + 1]] = order[[a + 1, a]]
                    sc[[a, a + 1]] = sc[[a + 1, a]]
        if any(wrong(a, a + 1) for a in range(N - 1)):
            return False, None
    return True, order[:n]


def reference_order(scores, n):
    s = np.asarray(scores[:n], np.float32)
    return np.lexsort((np.arange(n), -s.astype(np.float64)))      # score descending, index ascending (torch.sort(stable=True))


def test_ordered_image_is_monotone():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(4000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 4000).astype(np.float32),
                        np.array([0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45, 3.4e38, -3.4e38], np.float32)])
    o = ordered_image(x)
    i = np.argsort(x, kind="stable")
    assert np.all(np.diff(o[i].astype(np.int64)) >= 0)
    assert ordered_image(np.float32(-0.0)) == ordered_image(np.float32(0.0))
    # equal images <=> equal scores
    xs, os_ = x[i], o[i]
    assert np.array_equal(np.diff(os_.astype(np.int64)) == 0, np.diff(xs) == 0)


@pytest.mark.parametrize("N", [128, 256, 1024])
def test_packed_sort_is_exact_or_rejects(N):
    rng = np.random.default_rng(N)
    accepted = rejected = 0
    for trial in range(60):
        n = int(rng.integers(N // 2 + 1, N + 1))
        kind = trial % 6
        if kind == 0:
            s = rng.standard_normal(n).astype(np.float32)
        elif kind == 1:                    # dense: many collisions in the truncated keys
            s = (np.float32(0.5) + rng.standard_normal(n).astype(np.float32) * np.float32(2e-4)).astype(np.float32)
        elif kind == 2:                    # heavy exact ties
            s = (np.round(rng.standard_normal(n) * 4) / 4).astype(np.float32)
        elif kind == 3:                    # a long run one ulp apart
            s = (np.float32(1.0) + rng.permutation(n).astype(np.float32) * np.float32(2.0 ** -23)).astype(np.float32)
        elif kind == 4:                    # signed zeros, infinities
            s = rng.choice(np.array([0.0, -0.0, np.inf, -np.inf, 1.0, -1.0], np.float32), size=n)
        else:                              # isolated pairs one ulp apart in reversed index order
            s = np.linspace(3, -3, n).astype(np.float32)[rng.permutation(n)]
            for a in rng.choice(n - 1, size=6, replace=False):
                x = np.float32(0.25 + 0.001 * a)
                s[a], s[a + 1] = x, np.nextafter(x, np.float32(2.0))
        ok, order = packed_sort(s, n, N)
        if ok:
            accepted += 1
            assert np.array_equal(order, reference_order(s, n)), (N, trial, kind)
        else:
            rejected += 1
    assert accepted >= 30                  # the fast path serves the ordinary inputs
    assert rejected >= 1                   # and the long runs are rejected, not mis-sorted


def test_nan_is_rejected():
    s = np.array([1.0, np.nan, 0.5, 2.0] * 32, np.float32)
    assert packed_sort(s, 128, 128) == (False, None)
