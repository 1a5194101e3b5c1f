"""ctypes binding of libivae_amd.so (the C ABI declared in include/ivae_amd.h).

The library is the product: if it is missing or fails to load this module raises --
there is no CPU or PyTorch fallback for the hot path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# IVAE_AMD_LIB selects another build of the same ABI (the `make prof` diagnostic library)
LIB_PATH = os.environ.get("IVAE_AMD_LIB") or os.path.join(_HERE, "libivae_amd.so")

ACT_IDS = {"none": 0, "lrelu": 1, "xtanh": 2, "sigmoid": 3}

ERRORS = {
    -1: "bad dimensions", -2: "unknown activation", -3: "null pointer", -4: "shape too large for the kernels",
    -5: "HIP error", -6: "workspace too small",
}


class IvaeDims(C.Structure):
    _fields_ = [("batch", C.c_int32), ("data_dim", C.c_int32), ("latent_dim", C.c_int32), ("aux_dim", C.c_int32),
                ("hidden_dim", C.c_int32), ("n_layers", C.c_int32), ("activation", C.c_int32), ("slope", C.c_float),
                ("n_models", C.c_int32), ("tile_rows", C.c_int32), ("waves", C.c_int32), ("wgrad_rows", C.c_int32)]


class IvaeCoef(C.Structure):
    _fields_ = [("a", C.c_float), ("b", C.c_float), ("c", C.c_float), ("d", C.c_float),
                ("dataset_size", C.c_float), ("decoder_var", C.c_float), ("pairwise", C.c_int32),
                ("global_batch", C.c_int32)]


class IvaeWsLayout(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("per_model", "g", "v", "l", "z", "f", "lpx", "lqc", "lp", "lq", "lqi",
                                         "rz", "cg", "cv", "lse_s", "lse_k")]


EXPORTS = ["ivae_param_count", "ivae_workspace_bytes", "ivae_workspace_layout", "ivae_elbo_fwd_bwd", "ivae_forward",
           "ivae_pairwise", "ivae_pairwise_workspace_bytes", "ivae_backward", "ivae_adam_step", "ivae_elbo_adam_steps", "ivae_train_steps", "ivae_train_steps_supported", "ivae_corr_workspace_bytes", "ivae_corr", "ivae_corr_static", "ivae_corr_batched",
           "ivae_last_hip_error", "ivae_version", "ivae_debug_fault_inject", "ivae_debug_launch_events"]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `make` (or __graft_entry__.build()); "
                           "ivae_amd has no fallback path")
    L = C.CDLL(LIB_PATH)
    vp, f64 = C.c_void_p, C.c_double
    pd, pc = C.POINTER(IvaeDims), C.POINTER(IvaeCoef)
    L.ivae_param_count.argtypes = [pd, C.POINTER(C.c_int64)]
    L.ivae_workspace_bytes.argtypes = [pd, C.c_int32, C.POINTER(C.c_size_t)]
    L.ivae_workspace_layout.argtypes = [pd, C.c_int32, C.POINTER(IvaeWsLayout)]
    L.ivae_elbo_fwd_bwd.argtypes = [pd, pc, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    L.ivae_forward.argtypes = [pd, pc, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    L.ivae_pairwise_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]
    L.ivae_pairwise.argtypes = [C.c_int32, C.c_int32, C.c_int32, pc, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    L.ivae_backward.argtypes = [pd, pc, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    L.ivae_adam_step.argtypes = [vp, vp, vp, vp, C.c_int64, f64, f64, f64, f64, C.c_int64, vp]
    L.ivae_elbo_adam_steps.argtypes = [pd, pc, pc, vp, vp, vp, vp, vp, vp, C.c_int32, C.c_int64, f64, C.POINTER(C.c_double), f64, f64,
                                       f64, vp, vp, vp, vp, C.c_size_t, vp]
    L.ivae_train_steps_supported.argtypes = [pd]
    L.ivae_train_steps.argtypes = [pd, pc, vp, vp, vp, vp, vp, C.c_int64, vp, C.c_int64, vp, C.c_uint64, C.c_int32, C.c_int64,
                                   f64, C.POINTER(C.c_double), f64, f64, f64, vp, vp, vp, vp]
    L.ivae_debug_fault_inject.argtypes = [C.c_int32]
    L.ivae_debug_launch_events.argtypes = [vp, vp]
    L.ivae_corr_workspace_bytes.argtypes = [C.c_int64, C.c_int32, C.POINTER(C.c_size_t)]
    L.ivae_corr.argtypes = [vp, vp, C.c_int64, C.c_int32, vp, vp, C.c_size_t, vp]
    L.ivae_corr_static.argtypes = [vp, vp, C.c_int64, C.c_int32, vp, vp]
    L.ivae_corr_batched.argtypes = [vp, vp, C.c_int64, C.c_int64, C.c_int32, vp, vp]
    L.ivae_last_hip_error.restype = C.c_char_p
    L.ivae_version.restype = C.c_char_p
    for n in EXPORTS:
        getattr(L, n)   # AttributeError if the .so does not export a declared symbol
    if b"DIAG" in L.ivae_version() and not os.environ.get("IVAE_AMD_LIB"):
        raise RuntimeError(f"{LIB_PATH} is a diagnostic build (make diag); it is loaded only when IVAE_AMD_LIB names it")
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = ERRORS.get(rc, f"error {rc}")
        if rc == -5:
            msg += ": " + lib().ivae_last_hip_error().decode()
        raise RuntimeError(f"{what} failed: {msg}")
