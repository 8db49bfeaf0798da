"""CPU: the C-ABI library loads and exports every symbol include/ptranking_amd.h declares; the product path refuses to
run without a GPU (no CPU fallback) and never touches oracle/."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ptranking_amd.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ptr_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    from ptranking_amd import build
    return build.build()


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ("ptr_lambdarank_fwd_bwd", "ptr_ranknet_fwd_bwd", "ptr_lambdaloss_fwd_bwd", "ptr_approxndcg_fwd_bwd",
                 "ptr_listnet_fwd_bwd", "ptr_listmle_fwd_bwd", "ptr_sort_desc", "ptr_metrics_at_ks", "ptr_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"{lib_path} lacks {missing}"
    lib.ptr_abi_version.restype = ctypes.c_int
    assert lib.ptr_abi_version() == int(re.search(r"#define PTR_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    lib.ptr_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.ptr_last_error(), bytes)


def test_python_binding_matches_header(lib_path):
    from ptranking_amd import _lib
    declared = set(declared_symbols())
    bound = set(_lib.SIGNATURES)
    assert declared <= bound, f"unbound: {sorted(declared - bound)}"
    assert bound - declared <= _lib.OPTIONAL | declared
    handle = _lib.load()
    # the C prototypes and the ctypes argtypes agree on arity
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name in declared:
        proto = re.search(name + r"\s*\(([^)]*)\)", src).group(1).strip()
        n = 0 if proto in ("void", "") else proto.count(",") + 1
        assert n == len(_lib.SIGNATURES[name]), name
    assert handle.ptr_abi_version() == _lib.ABI_VERSION
    assert _lib.MAX_LIST_LEN == int(re.search(r"#define PTR_MAX_LIST_LEN (\d+)", src).group(1))


def test_argument_errors_need_no_gpu(lib_path):
    """Argument validation happens before any launch, so it can be exercised on a GPU-less box."""
    from ptranking_amd import _lib
    lib = _lib.load()
    rc = lib.ptr_lambdarank_fwd_bwd(None, None, None, 4, 8, ctypes.c_float(1.0), None, None, None, None)
    assert rc == 1001 and b"NULL" in lib.ptr_last_error()
    one = ctypes.c_void_p(16)
    rc = lib.ptr_sort_desc(one, None, 2, 10 ** 6, one, one, None)
    assert rc == 1002 and b"PTR_MAX_LIST_LEN" in lib.ptr_last_error()
    rc = lib.ptr_lambdaloss_fwd_bwd(one, one, None, 1, 8, 5, ctypes.c_float(1.0), ctypes.c_float(5.0), 7, 1, None, one, one, None)
    assert rc == 1001 and b"loss_type" in lib.ptr_last_error()
    rc = lib.ptr_approxndcg_fwd_bwd(one, one, None, 1, 8, ctypes.c_float(-1.0), 1, 1, ctypes.c_float(0.0), one, one, one, one, one, None)
    assert rc == 1001 and b"alpha" in lib.ptr_last_error()
    ks = (ctypes.c_int32 * 40)(*range(1, 41))
    rc = lib.ptr_metrics_at_ks(one, one, None, 1, 8, ks, 40, 1, 0, ctypes.c_float(4.0), None, one, None, None, None, None)
    assert rc == 1002
    rc = lib.ptr_metrics_at_ks(one, one, None, 1, 8, ks, 3, 1, 1, ctypes.c_float(4.0), None, one, one, None, None, None)
    assert rc == 1002 and b"nERR" in lib.ptr_last_error()          # nERR is undefined for LABEL_TYPE.Permutation
    rc = lib.ptr_metrics_at_ks(one, one, None, 1, 8, ks, 3, 1, 7, ctypes.c_float(4.0), None, one, None, None, None, None)
    assert rc == 1001 and b"label_type" in lib.ptr_last_error()


def test_x6_entry_point_validates_without_a_gpu():
    """ABI v3: `ptr_mlp_x6_ws_bytes` / `ptr_mlp_forward_x6` — the configurations the bf16x6 forward does not serve report 0 bytes /
    PTR_ERR_UNSUPPORTED, bad arguments PTR_ERR_INVALID_ARG, an empty batch succeeds, all before any launch."""
    from ptranking_amd import _lib
    lib = _lib.load()
    lib.ptr_mlp_x6_ws_bytes.restype = ctypes.c_size_t
    assert lib.ptr_mlp_x6_ws_bytes(136, 3) == 13 * 21504 + 16384          # 5 + 4 + 4 slices of the weight image + the trace area
    assert lib.ptr_mlp_x6_ws_bytes(46, 3) == 0 and lib.ptr_mlp_x6_ws_bytes(136, 1) == 0 and lib.ptr_mlp_x6_ws_bytes(136, 9) == 0
    one = ctypes.c_void_p(4096)
    args = lambda R, F, NL, p, X=one, acts=one: (X, one, R, F, NL, 1, ctypes.c_float(p), ctypes.c_uint64(1), one, acts, one, None)
    assert lib.ptr_mlp_forward_x6(*args(8, 46, 3, 0.1)) == 1002 and b"bf16x6" in lib.ptr_last_error()
    assert lib.ptr_mlp_forward_x6(*args(8, 136, 3, 1.5)) == 1001 and b"dropout" in lib.ptr_last_error()
    assert lib.ptr_mlp_forward_x6(*args(8, 136, 3, 0.1, X=None)) == 1001 and b"NULL" in lib.ptr_last_error()
    assert lib.ptr_mlp_forward_x6(*args(8, 136, 3, 0.1, X=ctypes.c_void_p(4100))) == 1001 and b"aligned" in lib.ptr_last_error()
    assert lib.ptr_mlp_forward_x6(*args(8, 136, 3, 0.1, acts=None)) == 1001           # a training forward stores its activations
    assert lib.ptr_mlp_forward_x6(*args(2 ** 23, 136, 3, 0.1)) == 1002 and b"4 GB" in lib.ptr_last_error()
    assert lib.ptr_mlp_forward_x6(*args(0, 136, 3, 0.1)) == 0


def test_opt_step_loss_validates_without_a_gpu():
    """ABI v4: `ptr_opt_step_loss` (the data-parallel step's optimiser step + loss-slot sum in one launch) checks its arguments before any launch."""
    from ptranking_amd import _lib
    lib = _lib.load()
    one = ctypes.c_void_p(4096)
    f = ctypes.c_float
    args = lambda kind=1, n=8, step=1, p=one, s2=one, lq=one, nq=4: (p, one, ctypes.c_int64(n), kind, f(1e-3), f(0.9), f(0.999), f(1e-8), f(1e-3), step,
                                                                     one, s2, lq, nq, one, None)
    assert lib.ptr_opt_step_loss(*args(kind=7)) == 1001 and b"optimiser" in lib.ptr_last_error()
    assert lib.ptr_opt_step_loss(*args(step=0)) == 1001
    assert lib.ptr_opt_step_loss(*args(p=None)) == 1001
    assert lib.ptr_opt_step_loss(*args(s2=None)) == 1001            # Adam needs both moment buffers
    assert lib.ptr_opt_step_loss(*args(lq=None)) == 1001            # a loss sum needs its slots
    assert lib.ptr_opt_step_loss(one, one, ctypes.c_int64(0), 1, f(1e-3), f(0.9), f(0.999), f(1e-8), f(0.0), 1, one, one, None, 0, None, None) == 0


def test_train_step_descriptor_layout_and_validation_without_a_gpu():
    """ABI v5: `ptr_train_step` takes ONE descriptor; the ctypes mirror has the header's size, and a descriptor of another size, an unknown
    loss or missing scratch is refused before any launch."""
    from ptranking_amd import _lib
    lib = _lib.load()
    hdr = open(HEADER).read()
    body = re.search(r"typedef struct ptr_train_step_desc \{(.*?)\} ptr_train_step_desc;", hdr, flags=re.S).group(1)
    names = re.findall(r"[\s\*,]([A-Za-z_0-9]+)(?:\[4\])?\s*(?=[,;])", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert names == [f[0] for f in _lib.TrainStepDesc._fields_], names          # same fields, same order
    assert ctypes.sizeof(_lib.TrainStepDesc) == 256
    d = _lib.TrainStepDesc()
    assert lib.ptr_train_step(None, None) == 1001
    d.struct_bytes = 8
    assert lib.ptr_train_step(ctypes.addressof(d), None) == 1001 and b"descriptor of 8 bytes" in lib.ptr_last_error()
    d.struct_bytes = ctypes.sizeof(_lib.TrainStepDesc)
    d.B, d.L, d.loss_kind = 2, 8, 0
    assert lib.ptr_train_step(ctypes.addressof(d), None) == 1001 and b"unknown loss" in lib.ptr_last_error()
    d.loss_kind = 2
    assert lib.ptr_train_step(ctypes.addressof(d), None) == 1001 and b"NULL scratch" in lib.ptr_last_error()


def test_product_path_fails_loudly_on_cpu_tensors():
    import ptranking_amd as pa
    p, y = torch.zeros(2, 8), torch.zeros(2, 8)
    for fn in (pa.functional.lambdarank_loss, pa.functional.ranknet_loss, pa.functional.listnet_loss,
               pa.functional.approxndcg_loss, pa.functional.lambdaloss_loss):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            fn(p, y)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pa.functional.metrics_at_ks(p, y, [1, 5])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pa.functional.sort_desc(p)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ptranking_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "liboracle" not in text, f


def test_missing_library_is_a_loud_error(monkeypatch, tmp_path):
    from ptranking_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.NativeLibraryError, match="no CPU / eager fallback"):
        _lib.load()
