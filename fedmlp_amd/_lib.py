"""ctypes binding of include/fedmlp_hip.h (the C-ABI HIP library).

There is no CPU fallback: if ``libfedmlp_hip.so`` has not been built
(``make`` / ``__graft_entry__.build()``) loading fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FEDMLP_HIP_LIB selects another build of the same library (timing probes under tools/)
LIB_PATH = os.environ.get("FEDMLP_HIP_LIB") or os.path.join(_HERE, "libfedmlp_hip.so")

FM_MAX_CLASSES = 32
FM_COMM_ID_BYTES = 128


class FmConfig(C.Structure):
    _fields_ = [("model", C.c_int32), ("n_classes", C.c_int32), ("in_h", C.c_int32),
                ("in_w", C.c_int32), ("max_images", C.c_int32), ("reserved", C.c_int32 * 3),
                ("stream", C.c_void_p)]


class FmAdam(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("weight_decay", C.c_float)]


_P = C.c_void_p
_F = C.POINTER(C.c_float)
_I32 = C.c_int32
_I64 = C.c_int64

# name -> (restype, argtypes); every symbol include/fedmlp_hip.h declares
SYMBOLS = {
    "fm_last_error": (C.c_char_p, []),
    "fm_version": (C.c_char_p, []),
    "fm_create": (C.c_int, [C.POINTER(FmConfig), C.POINTER(_P)]),
    "fm_destroy": (C.c_int, [_P]),
    "fm_sync": (C.c_int, [_P]),
    "fm_state_sizes": (C.c_int, [_P, C.POINTER(_I64), C.POINTER(_I64)]),
    "fm_set_state": (C.c_int, [_P, _P, _P]),
    "fm_get_state": (C.c_int, [_P, _P, _P]),
    "fm_state_device": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_I64)]),
    "fm_counters": (C.c_int, [_P, _P, _I32]),
    "fm_state_scale": (C.c_int, [_P, C.c_float]),
    "fm_fedavg_fold": (C.c_int, [_P, C.POINTER(_P), _F, _I32, _P]),
    "fm_stream_mode": (C.c_int, [_P]),
    "fm_mfma_products": (C.c_int, []),
    "fm_products": (C.c_int, [_P]),
    "fm_planes_mode": (C.c_int, [_P]),
    "fm_teacher_snapshot": (C.c_int, [_P]),
    "fm_adam_reset": (C.c_int, [_P, C.POINTER(FmAdam)]),
    "fm_forward_eval": (C.c_int, [_P, _P, _I32, _I32, _P, _P]),
    "fm_step_bce": (C.c_int, [_P, _P, _P, _I32, _F, _I32, _P]),
    "fm_step_stage1": (C.c_int, [_P, _P, _P, _P, _I32, _F, _I32, _I32, _P]),
    "fm_step_stage2": (C.c_int, [_P, _P, _P, _P, _I32, _P]),
    "fm_step_fixmatch": (C.c_int, [_P, _P, _P, _P, _I32, _F, _F, _F, _I32, _I32, _P]),
    "fm_proto_reset": (C.c_int, [_P]),
    "fm_proto_accumulate": (C.c_int, [_P, _P, _P, _P, _I32, _F, _F, C.c_float, C.c_float]),
    "fm_proto_finalize": (C.c_int, [_P, _I32, _I64, _F, _P, _P]),
    "fm_cos_tag": (C.c_int, [_P, _P, _I64, _P, C.POINTER(_I32), _I32, _P]),
    "fm_select_topk": (C.c_int, [_P, _P, _I64, C.c_double, C.c_double, _I32, C.POINTER(_I32),
                                 C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "fm_select_topk_rows": (C.c_int, [_P, _P, _I64, _I32, C.POINTER(_I32), C.POINTER(_I32), _I32, C.c_double, C.c_double, _I32,
                                      C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "fm_augment": (C.c_int, [_P, _P, _P, _P, _I32, _F, _F, _P]),
    "fm_forward_train": (C.c_int, [_P, _P, _P, _I32, _P, _P]),
    "fm_backward_step": (C.c_int, [_P, _P]),
    "fm_teacher_axpby": (C.c_int, [_P, C.c_float, C.c_float]),
    "fm_teacher_swap": (C.c_int, [_P]),
    "fm_set_stochastic": (C.c_int, [_P, _P, _P]),
    "fm_feature_dim": (C.c_int, [_P]),
    "fm_profile_enable": (C.c_int, [_P, _I32]),
    "fm_profile_read": (C.c_int, [_P, _I32, C.POINTER(_I64), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double)]),
    "fm_comm_preflight": (C.c_int, []),
    "fm_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "fm_comm_init": (C.c_int, [_P, C.POINTER(C.c_uint8), _I32, _I32]),
    "fm_comm_destroy": (C.c_int, [_P]),
    "fm_comm_size": (C.c_int, [_P]),
    "fm_fedavg_allreduce": (C.c_int, [_P, C.c_float]),
    "fm_fedavg_tao": (C.c_int, [_P, C.POINTER(C.c_double), C.c_double, _F, C.POINTER(C.c_double)]),
    "fm_fedavg_proto": (C.c_int, [_P, _P, C.c_double, _F, _P]),
    "fm_profile_ops": (C.c_int, [_P, _I32, C.c_char_p, _I32]),
    "fm_debug_num_convs": (C.c_int, [_P]),
    "fm_debug_conv_info": (C.c_int, [_P, _I32, C.POINTER(_I32)]),
    "fm_debug_conv": (C.c_int, [_P, _I32, _I32, _P, _P, _P, _I32, _I32, _P]),
    "fm_debug_pw": (C.c_int, [_P, _I32, _I32, _P, _P, _P, _I32, _I32, _P, _P, _P, _P]),
    "fm_debug_proj_bwd": (C.c_int, [_P, _I32, _I32, _P, _P, _P, _P, _P, _I32, _I32, _P, _P]),
    "fm_debug_exp_bwd": (C.c_int, [_P, _I32, _P, _P, _P, _P, _P, _I32, _I32, _P, _P]),
    "fm_debug_get_grads": (C.c_int, [_P, _P]),
    "fm_debug_activation": (C.c_int, [_P, _I32, _I32, _I32, _P, C.POINTER(_I32)]),
    "fm_debug_stem_masks": (C.c_int, [_P, _I32, _I32, _P, _P]),
    "fm_debug_lose_part": (C.c_int, [_I32]),
}

_lib = None


def load():
    """Load the shared library (once) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built and fedmlp_amd has no CPU "
            "fallback. Build it with `make` (or __graft_entry__.build()).")
    # torch bundles its own ROCm runtime; it must be the first HIP runtime in the process
    # (loading ours first leaves two runtimes and hipMalloc reports "no ROCm-capable device")
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)    # AttributeError if the .so lacks a declared symbol
        except AttributeError:
            # an OLDER build selected for a same-box A/B (tools/) may lack a newer test hook; the shipped library may not
            if os.environ.get("FEDMLP_HIP_LIB") and name.startswith("fm_debug_"):
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class FmError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = load().fm_last_error()
        raise FmError(f"fedmlp_hip error {rc}: {msg.decode() if msg else ''}")


def fvec(values, n=None):
    """host float[n] argument"""
    vals = [float(v) for v in values]
    if n is not None:
        assert len(vals) == n, (len(vals), n)
    return (C.c_float * len(vals))(*vals)
