"""GPU: the hand-written linear-layer kernels (csrc/linear.hip) and the fused (Dropout -> Linear -> ReLU)* stack against plain
torch fp32 modules evaluated on the CPU (the reference's arithmetic), incl. odd shapes, strided inputs and the kernel's own dropout
masks fed to the torch side."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def close(a, b, tol=2e-5, what=""):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{what}: max|diff|={err:.3e} > {tol * scale:.3e}"


@pytest.mark.parametrize("R,K,N", [(1000, 136, 128), (777, 128, 256), (515, 256, 512), (300, 512, 136), (2049, 136, 408), (640, 136, 136),
                                   (333, 512, 1), (100, 10, 100), (65, 100, 1), (4096, 46, 100), (50, 7, 5), (1, 136, 128),
                                   # several trips per wave round the tile loop (256 CUs x 16 waves x 16 rows = 65 536 rows per trip), ragged tail
                                   (140001, 136, 136), (70003, 100, 100), (66000, 34, 200)])
@pytest.mark.parametrize("bias", [True, False])
def test_linear_forward_backward_match_torch_cpu(R, K, N, bias):
    from ptranking_amd.linear import linear
    torch.manual_seed(R + K + N)
    x = torch.randn(R, K)
    w = torch.randn(N, K) / K ** 0.5
    b = torch.randn(N) if bias else None
    g = torch.randn(R, N)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = torch.nn.functional.linear(xr, wr, br)
    yr.backward(g)
    xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True) if bias else None
    y = linear(xg, wg, bg)
    y.backward(g.cuda())
    close(y, yr, what="y")
    close(xg.grad, xr.grad, what="dx")
    close(wg.grad, wr.grad, tol=5e-5, what="dw")
    if bias:
        close(bg.grad, br.grad, tol=5e-5, what="db")


@pytest.mark.parametrize("wide", ["0", "1"])
def test_linear_tile_forms_agree_with_torch(wide, monkeypatch):
    """16 waves x 16-row tiles (default) and 8 waves x 32-row tiles (PTR_LIN_WIDE=0) of the forward / backward-input kernel: the switch is
    read once per process, so each form runs in its own interpreter."""
    import os, subprocess, sys
    code = (
        "import torch\n"
        "from ptranking_amd.linear import linear\n"
        "torch.manual_seed(5)\n"
        "for R, K, N in ((3000, 136, 136), (70001, 100, 100), (515, 256, 512), (50, 7, 5)):\n"
        "    x = torch.randn(R, K); w = torch.randn(N, K) / K ** 0.5; b = torch.randn(N); g = torch.randn(R, N)\n"
        "    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)\n"
        "    yr = torch.nn.functional.linear(xr, wr, b); yr.backward(g)\n"
        "    xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)\n"
        "    y = linear(xg, wg, b.cuda()); y.backward(g.cuda())\n"
        "    for a_, b_ in ((y, yr), (xg.grad, xr.grad)):\n"
        "        d = (a_.detach().cpu().double() - b_.detach().double()).abs().max().item()\n"
        "        assert d <= 2e-5 * max(1.0, b_.abs().max().item()), (R, K, N, d)\n"
        "print('forms ok')\n")
    env = dict(os.environ, PTR_LIN_WIDE=wide)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0 and "forms ok" in out.stdout, out.stderr[-2000:]


def test_linear_reads_strided_rows_in_place_and_is_bit_stable():
    from ptranking_amd.linear import linear
    torch.manual_seed(0)
    big = torch.randn(700, 408, device="cuda")
    x = big[:, 136:272]                              # a column block of a packed [R, 3F] buffer (row stride 408)
    w = torch.randn(64, 136, device="cuda", requires_grad=True)
    y = linear(x, w)
    close(y, x.contiguous().cpu() @ w.detach().cpu().t(), what="strided")
    y.sum().backward()
    g1 = w.grad.clone()
    w.grad = None
    linear(x, w).sum().backward()
    assert torch.equal(g1, w.grad)
    x3 = torch.randn(5, 40, 136, device="cuda", requires_grad=True)      # [B, L, K] inputs keep their leading shape
    y3 = linear(x3, w)
    assert y3.shape == (5, 40, 64)
    y3.sum().backward()
    assert x3.grad.shape == x3.shape


def _ref_stack(dims, tail_relu, masks, p, params):
    """torch-CPU restatement of the stack with explicit keep masks (what nn.Dropout does: x * mask / (1 - p))."""
    def f(x):
        n = len(dims) - 1
        a = x * masks[0] / (1 - p) if masks[0] is not None else x
        for i in range(n - 1):
            a = torch.relu(torch.nn.functional.linear(a, params[2 * i], params[2 * i + 1]))
            if i < n - 2 and masks[i + 1] is not None:
                a = a * masks[i + 1] / (1 - p)
        out = torch.nn.functional.linear(a, params[-2], params[-1])
        return torch.relu(out) if tail_relu else out
    return f


@pytest.mark.parametrize("dims,tail_relu", [([136, 128, 256, 512, 136], True), ([136, 128, 256, 512, 1], False), ([24, 100, 100, 100, 1], False),
                                            ([136, 64, 1], False), ([40, 8], False)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_relu_stack_matches_torch_cpu_with_the_kernels_own_masks(dims, tail_relu, p):
    from ptranking_amd.host import build_stacked_ffnet
    from ptranking_amd.linear import ReluStack
    from ptranking_amd import _lib
    import ctypes as C
    torch.manual_seed(sum(dims))
    net = build_stacked_ffnet(dims, AF='R', TL_AF='R', apply_tl_af=tail_relu, dropout=p, BN=False).cuda()
    assert isinstance(net, ReluStack)
    net.train()
    R = 777
    n = len(dims) - 1
    lins = [m_ for m_ in net if isinstance(m_, nn.Linear)]

    def mask(site, width, seed):                 # keep mask of dropout site `site` = ptr_dropout_apply on ones * (1 - p)
        if p == 0.0 or n == 1:
            return None
        ones = torch.ones(R, width, device="cuda")
        m = torch.empty_like(ones)
        if width % 4 == 0:
            _lib.call("ptr_dropout_apply", _lib.ptr(ones), width, R, width, C.c_float(p), C.c_uint64(seed), site, _lib.ptr(m), width,
                      _lib.current_stream(ones.device))
            return (m > 0).float().cpu()
        return None

    # A hidden pre-activation within rounding distance of 0 lands on either side of the ReLU depending on the summation order (the
    # CPU reference in fp32 and in fp64 already disagree on such a unit) and flips a whole gate of the gradient: draw the data
    # again until the float64 pre-activations keep a distance of 1e-6 from 0 (one draw in three to five does).
    for attempt in range(80):
        x = torch.randn(3, R // 3, dims[0], device="cuda", requires_grad=True)
        out = net(x)
        seed = net.last_seed
        masks = [mask(0, dims[0], seed)] + [mask(i + 1, dims[i + 1], seed) for i in range(n - 2)] + [None]
        a64 = x.detach().cpu().reshape(R, dims[0]).double()
        if masks[0] is not None:
            a64 = a64 * masks[0].double() / (1 - p)
        zmin = float("inf")
        for i, l in enumerate(lins):
            z64 = a64 @ l.weight.detach().cpu().double().t() + l.bias.detach().cpu().double()
            if i < n - 1 or tail_relu:
                zmin = min(zmin, float(z64.abs().min()))
            a64 = torch.relu(z64)
            if i < n - 2 and masks[i + 1] is not None:
                a64 = a64 * masks[i + 1].double() / (1 - p)
        if zmin >= 1e-6:
            break
        net.zero_grad()
    assert zmin >= 1e-6, "no draw without a pre-activation at rounding distance of 0"
    assert out.shape == (3, R // 3, dims[-1])
    g = torch.randn_like(out)
    out.backward(g)
    params = []
    for m_ in net:
        if isinstance(m_, nn.Linear):
            params += [m_.weight.detach().cpu().clone().requires_grad_(True), m_.bias.detach().cpu().clone().requires_grad_(True)]
    xr = x.detach().cpu().reshape(R, dims[0]).clone().requires_grad_(True)
    ref = _ref_stack(dims, tail_relu, masks, p, params)(xr)
    ref.backward(g.cpu().reshape(R, dims[-1]))
    close(out.reshape(R, -1), ref, what="out")
    close(x.grad.reshape(R, -1), xr.grad, tol=5e-5, what="dx")
    ptol = 5e-5
    for i, m_ in enumerate(lins):
        close(m_.weight.grad, params[2 * i].grad, tol=ptol, what=f"dW{i}")
        close(m_.bias.grad, params[2 * i + 1].grad, tol=ptol, what=f"db{i}")
    if p > 0 and n > 1:
        keep = masks[0].mean().item()
        assert abs(keep - (1 - p)) < 0.01
    # eval mode: no dropout, deterministic
    net.eval()
    with torch.no_grad():
        a, b = net(x), net(x)
    assert torch.equal(a, b)
    assert list(net.state_dict()) == [f"ff_{i + 2}.{k}" for i in range(n) for k in ("weight", "bias")]


@pytest.mark.parametrize("R,N,group", [(70000, 100, 0), (4097, 24, 0), (6 * 129, 100, 129), (512, 7, 0), (300, 136, 0)])
def test_bn_stats_single_pass_is_accurate_with_large_offsets(R, N, group):
    """Column mean / rstd in ONE pass (pivoted sums per 256-row chunk + parallel-variance combination) against float64 — on data
    whose mean is 1e3..1e4 times its standard deviation, where E[z^2] - mean^2 in fp32 would lose every digit."""
    from ptranking_amd.linear import _bn_stats, BN_EPS
    torch.manual_seed(R + N)
    base = torch.randn(1, N, dtype=torch.float64) * 300.0 + 1000.0
    sd = torch.rand(1, N, dtype=torch.float64) * 0.2 + 0.05
    z64 = base + sd * torch.randn(R, N, dtype=torch.float64)
    z64[0] += 5.0 * sd[0]                                         # an outlying pivot row
    z = z64.float().cuda().contiguous()
    mean, rstd = _bn_stats(z, group)
    zz = z.double().cpu()
    if group:
        zz = zz.view(R // group, group, N)
        m_ref = zz.mean(dim=1).reshape(-1)
        v_ref = zz.var(dim=1, unbiased=False).reshape(-1)
    else:
        m_ref = zz.mean(dim=0)
        v_ref = zz.var(dim=0, unbiased=False)
    r_ref = 1.0 / torch.sqrt(v_ref + BN_EPS)
    assert float(((mean.double().cpu() - m_ref).abs() / m_ref.abs()).max()) < 5e-7       # a few fp32 ulps of the mean itself
    assert float(((rstd.double().cpu() - r_ref).abs() / r_ref).max()) < 2e-4, float(((rstd.double().cpu() - r_ref).abs() / r_ref).max())


# ---- r4: the weight gradient on the bf16 instructions when one side is narrow (csrc/linear_bw_x6.hip; default from 32768 rows on, PTR_LIN_BW_X6=2 forces it)
@pytest.mark.parametrize("R,K,N", [(2049, 136, 408), (640, 136, 136), (1000, 136, 128), (777, 128, 256), (333, 100, 100), (65, 100, 4), (4097, 140, 1536),
                                   (31, 8, 400), (1, 136, 136), (515, 512, 136), (300, 700, 100), (900, 44, 100), (40000, 136, 408), (70003, 100, 100),
                                   (257, 256, 112), (130, 4, 8), (1000, 140, 144)])
@pytest.mark.parametrize("bias", [True, False])
def test_weight_gradient_bf16x6_matches_float64_and_the_fp32_kernel(R, K, N, bias, monkeypatch):
    """dW = dY^T X and db through `ptr_linear_backward_weight` with the bf16x6 kernel forced (narrow dY / narrow X + ones column, 7- and 9-tile
    narrow images, one and several passes over the wide side, ragged last slab) against float64 and against the fp32-MFMA kernel: its error is
    not above the fp32 kernel's, and every entry of dW / db is written."""
    from ptranking_amd.linear import _bwd_weight
    torch.manual_seed(R + K + N)
    x = torch.randn(R, K, device="cuda")
    dy = torch.randn(R, N, device="cuda")
    ref_w = dy.double().cpu().t() @ x.double().cpu()
    ref_b = dy.double().cpu().sum(0)
    out = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("PTR_LIN_BW_X6", mode)
        dw = torch.full((N, K), float("nan"), device="cuda")
        db = torch.full((N,), float("nan"), device="cuda") if bias else None
        _bwd_weight(x, K, dy, bias, dw_out=dw, db_out=db)
        torch.cuda.synchronize()
        assert torch.isfinite(dw).all() and (db is None or torch.isfinite(db).all()), mode
        out[mode] = (dw.double().cpu(), None if db is None else db.double().cpu())
    sw = max(1.0, float(ref_w.abs().max()))
    e32, e6 = float((out["0"][0] - ref_w).abs().max()), float((out["2"][0] - ref_w).abs().max())
    assert e6 <= max(2e-5 * sw, 1.5 * e32), (e6, e32, sw)
    if bias:
        sb = max(1.0, float(ref_b.abs().max()))
        b32, b6 = float((out["0"][1] - ref_b).abs().max()), float((out["2"][1] - ref_b).abs().max())
        assert b6 <= max(2e-5 * sb, 1.5 * b32), (b6, b32, sb)


@pytest.mark.parametrize("K,N", [(136, 408), (100, 100), (408, 136)])
def test_weight_gradient_bf16x6_reads_strided_rows(K, N, monkeypatch):
    """X as a column slice of a wider matrix (row stride > K): the narrow-side kernel takes leading dimensions like the fp32 kernel does."""
    from ptranking_amd.linear import _bwd_weight
    monkeypatch.setenv("PTR_LIN_BW_X6", "2")
    torch.manual_seed(K + N)
    R, pad = 3000, 12
    big = torch.randn(R, K + pad, device="cuda")
    x = big[:, 4:4 + K]                                      # 16-byte aligned column offset, row stride K + 12
    dy = torch.randn(R, N, device="cuda")
    dw = torch.full((N, K), float("nan"), device="cuda"); db = torch.full((N,), float("nan"), device="cuda")
    _bwd_weight(x, K + pad, dy, True, dw_out=dw, db_out=db)
    ref_w = dy.double().cpu().t() @ x.double().cpu()
    close(dw, ref_w, tol=2e-5, what="dw")
    close(db, dy.double().cpu().sum(0), tol=2e-5, what="db")


# ---- r6: forward / backward-input on the bf16 instructions (csrc/linear_x6.hip; default for K <= 256 in 16-byte rows from 1024 rows on; PTR_LIN_X6=0 selects
# the fp32-MFMA kernel, =2 the general form where the default takes the whole-tile form)
X6_LIN_RATIO = 1.1      # bf16x6 error against float64 may be this much above the fp32-MFMA kernel's ... (measured 0.9-1.05, printed below)
X6_LIN_FLOOR = 2e-6     # ... plus this much of the tensor's maximum


def _x6_lin_err(out, ref):
    return float((out.double().cpu() - ref).abs().max())


@pytest.mark.parametrize("R,K,N", [(4133, 136, 136), (2049, 136, 408), (1024, 100, 100), (70003, 100, 100), (3000, 128, 256), (1500, 136, 128), (5000, 112, 112),
                                   (2500, 144, 144), (1777, 64, 100), (1200, 200, 136), (3001, 256, 96), (66000, 36, 200), (1030, 16, 64), (2000, 132, 120),
                                   # the chunked form (K in chunks of 128): whole-K image too large, or too few output tiles per block with it
                                   (2000, 512, 136), (1500, 256, 512), (1100, 300, 200), (1024, 408, 136), (40000, 512, 136), (1300, 700, 100)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_linear_forward_bf16x6_matches_float64_and_the_fp32_kernel(R, K, N, act, monkeypatch):
    """Y = epi(X W^T + b) through `ptr_linear_forward`: the bf16x6 kernels (whole-tile form: K16 in {112, 128, 144} with 7-9 output tiles; general form: the
    rest, incl. 1-3 blocks of outputs, a 16-deep tail or none, a ragged last tile) against float64 and against the fp32-MFMA kernel — same dropout mask
    (counter-based: a function of (seed, site, row, column)), every entry written, error not above the fp32 kernel's."""
    from ptranking_amd.linear import _fwd
    torch.manual_seed(R + K + N + act)
    x = torch.randn(R, K, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    z = x.double().cpu() @ w.double().cpu().t() + b.double().cpu()
    p = 0.25 if act == 2 else 0.0
    out = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("PTR_LIN_X6", mode)
        y = _fwd(x, K, w, b, act=act, p=p, seed=1234, site=3)
        torch.cuda.synchronize()
        assert torch.isfinite(y).all(), mode
        out[mode] = y
    if act == 0:
        ref = z
    else:
        ref = z.clamp(min=0.0)
        if act == 2:   # the kernel's own mask: kept entries are scaled by 1 / (1 - p), and the three forms must drop the same ones
            kept = out["0"].cpu() != 0
            for mode in ("1", "2"):
                flips = (out[mode].cpu() != 0) != kept
                assert float(ref[flips].abs().max() if flips.any() else 0.0) < 1e-5, "the forms drop different entries"
            ref = torch.where(kept, ref / (1.0 - p), torch.zeros_like(ref))
    s = max(1.0, float(ref.abs().max()))
    e32 = _x6_lin_err(out["0"], ref)
    for mode in ("1", "2"):
        e6 = _x6_lin_err(out[mode], ref)
        print(f"MEASURED linear fwd x6 R={R} K={K} N={N} act={act} mode={mode}: err {e6:.3e} fp32 {e32:.3e} ratio {e6 / max(e32, 1e-12):.2f}")
        assert e6 <= X6_LIN_RATIO * e32 + X6_LIN_FLOOR * s, (mode, e6, e32, s)


@pytest.mark.parametrize("R,K,N", [(4133, 136, 136), (1500, 128, 136), (1024, 100, 100), (70003, 100, 100), (3000, 256, 128), (2049, 136, 408), (1777, 100, 64),
                                   (2500, 144, 144), (1200, 136, 200), (1500, 136, 512), (1200, 200, 300), (40000, 136, 408)])
@pytest.mark.parametrize("gated", [False, True])
def test_linear_backward_input_bf16x6_matches_float64_and_the_fp32_kernel(R, K, N, gated, monkeypatch):
    """dX = (dY W) * [gate > 0] / (1 - p) through `ptr_linear_backward_input` (the same kernels on W^T; the gate read through a buffer resource): layer
    K inputs -> N outputs, so the product contracts over N (whole-N weight image up to 256, the chunked form beyond: 408, 512, 300)."""
    from ptranking_amd.linear import _bwd_input
    torch.manual_seed(R + K + N)
    dy = torch.randn(R, N, device="cuda")
    w = torch.randn(N, K, device="cuda") / K ** 0.5
    gate = torch.randn(R, K, device="cuda") if gated else None
    p = 0.1 if gated else 0.0
    ref = dy.double().cpu() @ w.double().cpu()
    if gated:
        ref = ref * (gate.cpu() > 0).double() / (1.0 - p)
    out = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("PTR_LIN_X6", mode)
        dx = _bwd_input(dy, w, gate=gate, p=p)
        torch.cuda.synchronize()
        assert torch.isfinite(dx).all(), mode
        out[mode] = dx
    s = max(1.0, float(ref.abs().max()))
    e32 = _x6_lin_err(out["0"], ref)
    for mode in ("1", "2"):
        e6 = _x6_lin_err(out[mode], ref)
        print(f"MEASURED linear bwd-input x6 R={R} K={K} N={N} gated={gated} mode={mode}: err {e6:.3e} fp32 {e32:.3e} ratio {e6 / max(e32, 1e-12):.2f}")
        assert e6 <= X6_LIN_RATIO * e32 + X6_LIN_FLOOR * s, (mode, e6, e32, s)


def test_linear_bf16x6_reads_strided_rows_and_is_bit_stable():
    """X as a column slice of a wider matrix (ldx > K, 16-byte aligned), Y into a freshly allocated tensor; two launches give identical bits (the tile loop's
    hand-counted loads: a miscounted wait would read a fragment before it arrives, and not the same way twice)."""
    from ptranking_amd.linear import _fwd
    torch.manual_seed(3)
    big = torch.randn(9000, 200, device="cuda")
    x = big[:, 32:168]                                    # 136 columns at a 128-byte offset, ldx = 200
    w = torch.randn(136, 136, device="cuda") / 136 ** 0.5
    b = torch.randn(136, device="cuda")
    y1 = _fwd(x, 200, w, b)
    y2 = _fwd(x, 200, w, b)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    ref = x.double().cpu() @ w.double().cpu().t() + b.double().cpu()
    close(y1, ref, tol=5e-6, what="strided x6")


@pytest.mark.parametrize("R,K,N", [(4133, 136, 136), (2050, 512, 136), (3000, 128, 256), (1500, 200, 96)])
def test_linear_bf16x6_writes_strided_outputs_in_place(R, K, N, monkeypatch):
    """Y (forward) and dX (backward-input, with a strided gate) as column slices of wider buffers through the raw C ABI: only the slice is written (the buffer
    stores of linear_x6.hip drop a lane by an out-of-range offset — a wrong leading dimension there would scribble over the neighbours), whole-tile, general
    and chunked forms."""
    import ctypes as C
    from ptranking_amd import _lib
    torch.manual_seed(R + K)
    x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    ldy = N + 8
    for mode in ("1", "2", "0"):
        monkeypatch.setenv("PTR_LIN_X6", mode)
        ybuf = torch.full((R, ldy), 7.25, device="cuda")
        y = ybuf[:, 4:4 + N]
        _lib.call("ptr_linear_forward", _lib.ptr(x), K, _lib.ptr(w), _lib.ptr(b), R, K, N, 0, C.c_float(0.0), C.c_uint64(0), 0, _lib.ptr(y), ldy,
                  _lib.current_stream(x.device))
        torch.cuda.synchronize()
        assert bool((ybuf[:, :4] == 7.25).all()) and bool((ybuf[:, 4 + N:] == 7.25).all()), mode
        close(y, x.double().cpu() @ w.double().cpu().t() + b.double().cpu(), tol=5e-6, what=f"strided y mode {mode}")
        # backward-input: dX[R][K] = dY[R][N] W, gated by a strided gate, into a strided dX
        dy = torch.randn(R, N, device="cuda")
        gbuf = torch.randn(R, K + 4, device="cuda"); gate = gbuf[:, 4:]
        dbuf = torch.full((R, K + 12), -3.5, device="cuda"); dx = dbuf[:, 8:8 + K]
        _lib.call("ptr_linear_backward_input", _lib.ptr(dy), N, _lib.ptr(w), R, K, N, _lib.ptr(gate), K + 4, C.c_float(0.2), _lib.ptr(dx), K + 12,
                  _lib.current_stream(x.device))
        torch.cuda.synchronize()
        assert bool((dbuf[:, :8] == -3.5).all()) and bool((dbuf[:, 8 + K:] == -3.5).all()), mode
        ref = (dy.double().cpu() @ w.double().cpu()) * (gate.cpu() > 0).double() / 0.8
        close(dx, ref, tol=5e-6, what=f"strided dx mode {mode}")
