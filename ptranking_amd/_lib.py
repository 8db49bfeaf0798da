"""ctypes binding of libptranking_amd.so (the C ABI declared in include/ptranking_amd.h).

The product path has exactly one implementation: the HIP kernels behind this library.  There is no CPU or eager
fallback — if the library is missing, or a tensor is not on the GPU, the call fails loudly.
"""
import ctypes as C
import os

import torch  # noqa: F401  — must be imported first so that OUR .so binds to the HIP runtime torch already loaded

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTR_LIB") or os.path.join(_PKG, "libptranking_amd.so")   # PTR_LIB: an experiment build (build.py --variant)

ABI_VERSION = 5
MAX_LIST_LEN = 4096
MAX_CUTOFFS = 32

_vp, _i, _f, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_uint64

# name -> argtypes (restype is int unless listed in _RESTYPES).  Mirrors include/ptranking_amd.h one to one.
SIGNATURES = {
    "ptr_abi_version": [],
    "ptr_last_error": [],
    "ptr_ranknet_fwd_bwd": [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp],
    "ptr_lambdarank_fwd_bwd": [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp],
    "ptr_lambdaloss_fwd_bwd": [_vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp],
    "ptr_approxndcg_fwd_bwd": [_vp, _vp, _vp, _i, _i, _f, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp],
    "ptr_softrank_fwd_bwd": [_vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp, _vp],
    "ptr_listnet_fwd_bwd": [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "ptr_mdprank_fwd_bwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp],
    "ptr_listmle_fwd_bwd": [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "ptr_stlistnet_fwd_bwd": [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp],
    "ptr_rankmse_fwd_bwd": [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "ptr_rankcosine_fwd_bwd": [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "ptr_shuffle_ties_order": [_vp, _vp, _i, _i, _u64, _vp, _vp],
    "ptr_sort_desc": [_vp, _vp, _i, _i, _vp, _vp, _vp],
    "ptr_metrics_at_ks": [_vp, _vp, _vp, _i, _i, C.POINTER(C.c_int32), _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp],
    "ptr_sum_f32": [_vp, _i, _f, _vp, _vp],
    "ptr_mlp_num_params": [_i, _i],
    "ptr_mlp_backward_ws_floats": [_i, _i],
    "ptr_mlp_backward_dz_floats": [_i, _i, _i],
    "ptr_mlp_acts_floats": [_i, _i],
    "ptr_mlp_forward": [_vp, _vp, _i, _i, _i, _i, _f, _u64, _vp, _vp, _vp],
    "ptr_mlp_backward": [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _u64, _vp, _vp, _vp, _vp],
    "ptr_mlp_backward_step": [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _u64, _vp, _vp, _vp, _i, _f, _f, _f, _f, _f, _i, _vp, _vp, _vp, _i, _vp, _vp],
    "ptr_opt_step_loss": [_vp, _vp, C.c_int64, _i, _f, _f, _f, _f, _f, _i, _vp, _vp, _vp, _i, _vp, _vp],
    "ptr_train_step": [_vp, _vp],
    "ptr_mlp_x6_ws_bytes": [_i, _i],
    "ptr_mlp_forward_x6": [_vp, _vp, _i, _i, _i, _i, _f, _u64, _vp, _vp, _vp, _vp],
    "ptr_adam_step": [_vp, _vp, _vp, _vp, C.c_int64, _f, _f, _f, _f, _f, _i, _vp],
    "ptr_adagrad_step": [_vp, _vp, _vp, C.c_int64, _f, _f, _f, _f, _i, _vp],
    "ptr_rmsprop_step": [_vp, _vp, _vp, C.c_int64, _f, _f, _f, _f, _vp],
    "ptr_mlp_dropout_mask": [_i, _i, _i, _f, _u64, _vp, _vp],
    "ptr_linear_forward": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _u64, _i, _vp, _i, _vp],
    "ptr_linear_backward_input": [_vp, _i, _vp, _i, _i, _i, _vp, _i, _f, _vp, _i, _vp],
    "ptr_linear_backward_weight_ws_floats": [_i, _i, _i],
    "ptr_linear_backward_weight": [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "ptr_bn_ws_floats": [_i, _i, _i],
    "ptr_bn_stats": [_vp, _i, _i, _i, _i, _vp, _i, _f, _vp, _vp, _vp, _vp],
    "ptr_bnact_forward": [_vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _f, _u64, _i, _vp, _vp],
    "ptr_bnact_backward": [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _f, _u64, _i, _vp, _vp, _vp, _vp, _vp],
    "ptr_dropout_apply": [_vp, _i, _i, _i, _f, _u64, _i, _vp, _i, _vp],
    "ptr_relu_gate": [_vp, _vp, C.c_int64, _vp, _vp],
    "ptr_mhsa_forward": [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _u64, _i, _vp, _vp, _vp],
    "ptr_mhsa_backward": [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _u64, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "ptr_mhsa_dropout_mask": [_i, _i, _i, _f, _u64, _i, _vp, _vp],
    "ptr_layernorm_forward": [_vp, _vp, _vp, C.c_int64, _i, _f, _vp, _vp, _vp],
    "ptr_layernorm_backward_ws_floats": [_i],
    "ptr_layernorm_backward": [_vp, _vp, _vp, _vp, C.c_int64, _i, _vp, _vp, _vp, _vp, _vp],
    "ptr_letor_scan": [C.c_char_p, _i, _vp, _vp, _vp],
    "ptr_letor_load": [C.c_char_p, _i, _f, C.c_int64, C.c_int32, C.c_int64, _vp, _i, _vp, _vp, _vp],
}
_RESTYPES = {"ptr_last_error": C.c_char_p, "ptr_mlp_num_params": C.c_size_t, "ptr_mlp_backward_ws_floats": C.c_size_t,
             "ptr_mlp_backward_dz_floats": C.c_size_t, "ptr_mlp_acts_floats": C.c_size_t, "ptr_mlp_x6_ws_bytes": C.c_size_t, "ptr_linear_backward_weight_ws_floats": C.c_size_t, "ptr_bn_ws_floats": C.c_size_t,
             "ptr_layernorm_backward_ws_floats": C.c_size_t}
OPTIONAL = set()

_lib = None

LOSS_KINDS = {"ptr_ranknet_fwd_bwd": 1, "ptr_lambdarank_fwd_bwd": 2, "ptr_lambdaloss_fwd_bwd": 3, "ptr_listnet_fwd_bwd": 4}   # PTR_LOSS_*


class TrainStepDesc(C.Structure):
    """`ptr_train_step_desc` of include/ptranking_amd.h (ABI v5), field for field."""
    _fields_ = [("struct_bytes", C.c_int32), ("loss_kind", C.c_int32),
                ("B", C.c_int32), ("L", C.c_int32), ("F", C.c_int32), ("NL", C.c_int32),
                ("opt_kind", C.c_int32), ("step", C.c_int32),
                ("loss_i", C.c_int32 * 4), ("loss_f", C.c_float * 4),
                ("p_drop", C.c_float), ("lr", C.c_float), ("hyper1", C.c_float), ("hyper2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float),
                ("seed", C.c_uint64),
                ("X", C.c_void_p), ("labels", C.c_void_p), ("lens", C.c_void_p),
                ("params", C.c_void_p), ("grad", C.c_void_p), ("state1", C.c_void_p), ("state2", C.c_void_p),
                ("preds", C.c_void_p), ("acts", C.c_void_p), ("loss_q", C.c_void_p), ("dpreds", C.c_void_p), ("dz", C.c_void_p), ("ws", C.c_void_p),
                ("wimg", C.c_void_p), ("loss_out", C.c_void_p), ("events", C.c_void_p * 4), ("wimg_current", C.c_int32), ("reserved", C.c_int32)]


class NativeLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises NativeLibraryError when the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -m ptranking_amd.build, or "
            f"__graft_entry__.build()).  ptranking_amd has no CPU / eager fallback by design.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if name in OPTIONAL:
                continue
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}; rebuild it")
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    v = lib.ptr_abi_version()
    if v != ABI_VERSION:
        raise NativeLibraryError(f"ABI version mismatch: library {v}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def has(name):
    return hasattr(load(), name)


# Optional profiling hook (bench.py): TIMING = {} makes call() bracket every entry point with HIP events recorded on
# torch's current stream (the stream the kernels are enqueued on); TIMING[name] collects (start, end) event pairs.
TIMING = None


def query(name, *args):
    """Invoke a size-returning helper (no error code)."""
    return getattr(load(), name)(*args)


def call(name, *args):
    """Invoke an int-returning entry point; raise RuntimeError(ptr_last_error()) on a non-zero return."""
    lib = load()
    if TIMING is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        rc = getattr(lib, name)(*args)
        ev1.record()
        TIMING.setdefault(name, []).append((ev0, ev1))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.ptr_last_error()
        raise RuntimeError(f"{name} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
