"""ctypes binding of libisbfsar_hip.so (include/isbfsar.h).

There is no CPU fallback: if the shared library is missing or a call fails this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ISB_LIB_PATH: load another build of the SAME library (A/B timing of two builds inside one GPU session)
LIB_PATH = os.environ.get("ISB_LIB_PATH") or os.path.join(_HERE, "csrc", "libisbfsar_hip.so")

ISB_AR_PREC_DEFAULT = 0     # = ISB_AR_PREC_F16 (ABI version 2)
ISB_AR_PREC_BF16X3 = 1
ISB_AR_PREC_F16 = 2
ISB_AR_PREC_BF16 = 3


class IsbError(RuntimeError):
    pass


class isb_ar_cfg(C.Structure):
    _fields_ = [("seq_len", C.c_int32), ("n_joints", C.c_int32), ("way_max", C.c_int32),
                ("device", C.c_int32), ("precision", C.c_int32), ("max_batch", C.c_int32)]


class isb_hpe_cfg(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("ppx", C.c_float), ("ppy", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32), ("device", C.c_int32),
                ("max_batch", C.c_int32), ("n_out_joints", C.c_int32), ("precision", C.c_int32)]


class isb_rgb_cfg(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_batch", C.c_int32)]


class isb_det_cfg(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("device", C.c_int32), ("max_batch", C.c_int32)]


_lib = None

# name -> (restype, argtypes); every symbol include/isbfsar.h declares
_P = C.c_void_p
SIGNATURES = {
    "isb_last_error": (C.c_char_p, []),
    "isb_version": (C.c_int, []),
    "isb_wsreg_verified": (C.c_int, []),
    "isb_device_count": (C.c_int, []),
    "isb_hw_queues": (C.c_int, [C.POINTER(C.c_int32)]),
    "isb_ar_create": (C.c_int, [C.POINTER(isb_ar_cfg), C.POINTER(_P)]),
    "isb_ar_destroy": (None, [_P]),
    "isb_ar_precision": (C.c_int, [_P]),
    "isb_ar_load_weights": (C.c_int, [_P, _P, C.c_size_t]),
    "isb_ar_set_support": (C.c_int, [_P, _P, _P, C.c_int32]),
    "isb_ar_get_support_features": (C.c_int, [_P, _P]),
    "isb_ar_infer": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, _P]),
    "isb_ar_infer_host": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P]),
    "isb_ar_last_chosen": (C.c_int, [_P, _P, C.c_int32]),
    "isb_ar_set_input_type": (C.c_int, [_P, C.c_int32]),
    "isb_ar_set_support_hybrid": (C.c_int, [_P, _P, _P, C.c_int32]),
    "isb_ar_infer_hybrid": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, _P, _P]),
    "isb_ar_profile": (C.c_int, [_P, C.c_int32]),
    "isb_ar_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "isb_hpe_create": (C.c_int, [C.POINTER(isb_hpe_cfg), C.POINTER(_P)]),
    "isb_hpe_destroy": (None, [_P]),
    "isb_hpe_load_weights": (C.c_int, [_P, _P, C.c_size_t]),
    "isb_hpe_set_joint_map": (C.c_int, [_P, _P, _P, C.c_int32]),
    "isb_hpe_forward": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, _P]),
    "isb_hpe_forward_host": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P]),
    "isb_hpe_submit_host": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P]),
    "isb_hpe_wait_host": (C.c_int, [_P]),
    "isb_hpe_set_augmentations": (C.c_int, [_P, C.c_int32, _P, _P]),
    "isb_hpe_set_lanes": (C.c_int, [_P, C.c_int32]),
    "isb_hpe_create_shared": (C.c_int, [_P, C.POINTER(_P)]),
    "isb_hpe_memory": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    "isb_hpe_crop_params_host": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P]),
    "isb_hpe_warp_host": (C.c_int, [_P, _P, _P, C.c_int32, _P]),
    "isb_hpe_backbone_host": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "isb_hpe_post_host": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, _P]),
    "isb_hpe_profile": (C.c_int, [_P, C.c_int32]),
    "isb_hpe_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "isb_hpe_profile_read_dw": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "isb_pose_windows": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "isb_pose_distance": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    "isb_hpe_select_person": (C.c_int, [_P, _P, _P, C.c_int32, C.c_float, _P, _P, _P]),
    "isb_hpe_select_person_host": (C.c_int, [_P, _P, _P, C.c_int32, C.c_float, _P, _P]),
    "isb_rgb_create": (C.c_int, [C.POINTER(isb_rgb_cfg), C.POINTER(_P)]),
    "isb_rgb_destroy": (None, [_P]),
    "isb_rgb_load_weights": (C.c_int, [_P, _P, C.c_size_t]),
    "isb_rgb_forward": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    "isb_rgb_forward_host": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P]),
    "isb_dist_unique_id": (C.c_int, [_P]),
    "isb_dist_create": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "isb_dist_destroy": (None, [_P]),
    "isb_dist_info": (C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "isb_dist_comm_count": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "isb_dist_all_gather": (C.c_int, [_P, _P, _P, C.c_size_t, _P]),
    "isb_det_create": (C.c_int, [C.POINTER(isb_det_cfg), C.POINTER(_P)]),
    "isb_det_destroy": (None, [_P]),
    "isb_det_n_convs": (C.c_int, []),
    "isb_det_describe": (C.c_int, [C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]),
    "isb_det_load_weights": (C.c_int, [_P, _P, C.c_size_t]),
    "isb_det_forward": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P]),
    "isb_det_forward_host": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "isb_det_debug_host": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, _P]),
    "isb_debug_fused_mb": (C.c_int, [C.c_int32] + [_P] * 8 + [C.c_int32] * 7 + [_P, C.POINTER(C.c_float)]),
    "isb_debug_hpe_mb8_stamps": (C.c_int, [_P, C.c_int32, _P]),
    "isb_debug_ar_stamps": (C.c_int, [_P, C.c_int32, _P]),
    "isb_debug_gemm_f32": (C.c_int, [C.c_int32] + [_P] * 5 + [C.c_int32] * 10 + [_P, C.POINTER(C.c_float)]),
    "isb_debug_mbfront": (C.c_int, [C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, _P, _P, C.POINTER(C.c_float)]),
    "isb_debug_se_fcs": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.POINTER(C.c_float)]),
    "isb_debug_dwconv": (C.c_int, [C.c_int32] + [_P] * 4 + [C.c_int32] * 5 + [_P, _P, C.POINTER(C.c_float)]),
    "isb_debug_dwconv_fc1": (C.c_int, [C.c_int32] + [_P] * 4 + [C.c_int32] * 5 + [_P, _P, C.POINTER(C.c_float), _P, C.c_int32, _P,
                                       C.POINTER(C.c_int32)]),
    "isb_debug_conv": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, _P] + [C.c_int32] * 10 + [_P, C.POINTER(C.c_float)]),
}
# NOTE: every entry point must be listed here BEFORE the first lib() call: a function without
# argtypes silently truncates Python-int pointers (torch data_ptr()) to 32 bits.


def lib() -> C.CDLL:
    """Load the HIP library (once). Raises IsbError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IsbError(f"{LIB_PATH} is missing: run `python -m isbfsar_amd.build` (hipcc, gfx950). "
                           "There is no CPU fallback.")
        # torch bundles its own libamdhip64.so (same SONAME as /opt/rocm's). Both the device
        # pointers torch hands us and our kernels must live in ONE HIP runtime instance, so let
        # torch's copy load first; the dynamic linker then binds our NEEDED entry to it.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().isb_last_error()
        raise IsbError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
