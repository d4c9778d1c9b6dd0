"""ctypes binding of libocflow_hip.so (C ABI declared in include/ocflow_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If it is missing it is built once with
hipcc (ocflow_amd/build.py); if that fails, importing the ops raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

# ONE HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so; importing torch first makes the dynamic linker
# bind libocflow_hip.so's NEEDED libamdhip64.so.7 to that already-loaded copy.  Loaded the other way round (our library
# first, against /opt/rocm) the process ends up with two runtimes and torch's stream handles are invalid in ours
# (every launch fails with OCF_ELAUNCH) -- observed on the MI355X box, see DESIGN.md "Boundary".
import torch  # noqa: F401  (load order, not an API dependency: the C ABI has no torch types)

from . import build as _build

ABI_VERSION = 3  # include/ocflow_hip.h OCF_ABI_VERSION
OCF_F32, OCF_BF16 = 0, 1
# per-call kernel policies (enum ocf_corr_policy, ocf_corr_bwd_policy, ocf_warp_bwd_policy, ocf_range_policy of the header)
CORR_AUTO, CORR_BAND_ONLY, CORR_NO_WAVE, CORR_WAVE = 0, 1, 2, 3
CORR_VARIANT_WAVE = 4  # enum ocf_corr_variant
CORR_VARIANTS_WITH_GATHER = (1, 2, 4)  # row kernels (C <= 64) and wave kernel: the warp gathered inside the correlation launch
CORR_BWD_AUTO, CORR_BWD_VECTOR, CORR_BWD_MFMA = 0, 1, 2
WARP_BWD_AUTO, WARP_BWD_SCATTER, WARP_BWD_GATHER, WARP_BWD_BINNED, WARP_BWD_SMALL = 0, 1, 2, 3, 4
WARP_BWD_GIMG_NATIVE = 0x100  # flag: grad_img in the feature dtype (when ocf_warp_bwd_plan answers bit 1)
RANGE_AUTO, RANGE_SPLAT, RANGE_GATHER = 0, 1, 2
OCF_OK = 0
OCF_EUNSUPPORTED = -2


class Tensor4(C.Structure):
    _fields_ = [("data", C.c_void_p), ("sb", C.c_int64), ("sc", C.c_int64), ("sh", C.c_int64), ("sw", C.c_int64)]


_T4P = C.POINTER(Tensor4)
_i, _f, _vp = C.c_int, C.c_float, C.c_void_p

# name -> argtypes; mirrors include/ocflow_hip.h one to one (tests/test_abi.py checks the header against this)
SIGNATURES = {
    "ocf_abi_version": [],
    "ocf_device_info": [C.c_char_p, _i, C.POINTER(_i)],
    "ocf_profile_next_launch": [_vp, _vp],
    "ocf_stream_capture_id": [_vp, C.POINTER(C.c_ulonglong)],
    "ocf_corr_fwd": [_T4P, _T4P, _T4P, _T4P, _T4P, _i, _i, _i, _i, _i, _f, _i, _f, _i, _i, _vp],
    "ocf_corr_fwd_variant": [_T4P, _T4P, _T4P, _T4P, _T4P, _i, _i, _i, _i, _i, _i, _i],
    "ocf_corr_bwd": [_T4P, _T4P, _T4P, _T4P, _T4P, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ocf_corr_bwd_variant": [_T4P, _T4P, _T4P, _T4P, _T4P, _i, _i, _i, _i, _i, _i, _i],
    "ocf_corr_occ_bwd": [_T4P, _T4P, _T4P, _T4P, _T4P, _T4P, _T4P, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ocf_warp_fwd": [_T4P, _T4P, _T4P, _i, _i, _i, _i, _f, _i, _i, _vp],
    "ocf_warp_bwd": [_T4P, _T4P, _T4P, _T4P, _T4P, _i, _i, _i, _i, _f, _i, _i, _i, _vp, _vp],
    "ocf_warp_bwd_workspace_bytes": [_i, _i, _i, _i, C.POINTER(C.c_int64)],
    "ocf_warp_bwd_plan": [_T4P, _T4P, _T4P, _T4P, _i, _i, _i, _i, _i, _i],
    "ocf_warp_bwd_det_workspace_bytes": [_i, _i, _i, _i, C.POINTER(C.c_int64)],
    "ocf_warp_bwd_det": [_T4P, _T4P, _T4P, _T4P, _T4P, _vp, _i, _i, _i, _i, _f, _i, _i, _vp],
    "ocf_range_map_det_workspace_bytes": [_i, _i, _i, C.POINTER(C.c_int64)],
    "ocf_range_map_det": [_T4P, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ocf_photo_loss_fwd": [_T4P, _T4P, _T4P, _T4P, _vp, _i, _i, _i, _i, _vp],
    "ocf_photo_loss_bwd": [_T4P, _T4P, _T4P, _T4P, _vp, _T4P, _T4P, _i, _i, _i, _i, _vp],
    "ocf_masked_l1_fwd": [_T4P, _T4P, _T4P, _vp, _i, _i, _i, _i, _vp],
    "ocf_masked_l1_bwd": [_T4P, _T4P, _T4P, _vp, _T4P, _i, _i, _i, _i, _vp],
    "ocf_smooth_loss_fwd": [_T4P, _T4P, _vp, _i, _i, _i, _i, _f, _vp],
    "ocf_smooth_loss_bwd": [_T4P, _T4P, _vp, _T4P, _i, _i, _i, _i, _f, _vp],
    "ocf_range_map": [_T4P, _vp, _vp, _i, _i, _i, _i, _vp],
    "ocf_moments": [_T4P, _vp, _i, _i, _i, _i, _i, _vp],
    "ocf_norm_fwd": [_T4P, _T4P, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ocf_norm_bwd_sums": [_T4P, _T4P, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ocf_norm_bwd": [_T4P, _T4P, _T4P, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ocf_ssim_fwd": [_T4P, _T4P, _vp, _i, _i, _i, _i, _i, _vp],
    "ocf_ssim_bwd_workspace_bytes": [_i, _i, _i, _i, _i, C.POINTER(C.c_int64)],
    "ocf_ssim_bwd": [_T4P, _T4P, _vp, _T4P, _vp, _i, _i, _i, _i, _i, _vp],
    "ocf_sqdiff": [_T4P, _T4P, _vp, _i, _i, _i, _i, _vp],
    "ocf_conv3x3_wgrad": [_T4P, _T4P, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ocf_conv3x3_plan": [_i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)],
    "ocf_conv3x3_igemm": [_T4P, _vp, _i, _i, _T4P, _vp, _T4P, _i, _i, _i, _i, _i, _i, _f, _f, _i, _i, _vp],
    "ocf_conv3x3_igemm_ex": [_T4P, _vp, _i, _i, _T4P, _vp, _T4P, _T4P, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, _i, _vp],
    "ocf_conv3x3_pack_bytes": [_i, C.POINTER(C.c_int64)],
    "ocf_conv3x3_pack_describe": [_vp, _i, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _i, _i, _i, _vp, _i, _i, _i, C.c_int64,
                                  C.POINTER(C.c_int64)],
    "ocf_conv3x3_pack_weights": [_vp, _i, C.c_int64, _vp],
    "ocf_conv3x3_wgrad_bias": [_T4P, _T4P, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ocf_conv3x3_wgrad_plan": [_i, _i, _i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)],
    "ocf_concat_channels": [C.POINTER(_T4P), C.POINTER(_i), _i, _T4P, _i, _i, _i, _i, _i, _vp],
    "ocf_deconv3x3s2_fwd": [_T4P, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _i, _vp, _T4P, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ocf_deconv3x3s2_bwd": [_T4P, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _i, _T4P, _T4P, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _vp, _T4P,
                            _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ocf_deconv4x4s2_fwd_workspace_bytes": [_i, _i, _i, _i, _i, C.POINTER(C.c_int64)],
    "ocf_deconv4x4s2_fwd": [_T4P, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _i, _vp, _T4P, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "ocf_resize_bilinear": [_T4P, _T4P, _i, _i, _i, _i, _i, _i, _i, _vp],
    "ocf_pair_input": [_T4P, _vp, _i, _i, _i, _vp],
    "ocf_flow_occ_head_fwd": [_T4P, _T4P, _T4P, _T4P, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "ocf_flow_occ_head_bwd": [_vp, _vp, _vp, _T4P, _T4P, _i, _i, _i, _i, _i, _vp],
    "ocf_conv3x3_c16": [_T4P, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _i, _T4P, _i, _i, _i, _i, _i, _vp],
    "ocf_conv3x3_c16_wgrad_workspace_bytes": [_i, _i, _i, C.POINTER(C.c_int64)],
    "ocf_conv3x3_c16_wgrad": [_T4P, _T4P, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _vp, _i, _i, _i, _i, _vp],
    "ocf_conv3x3s2_c3": [_T4P, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _i, _T4P, _i, _i, _i, _i, _vp],
    "ocf_conv3x3s2_c3_wgrad_workspace_bytes": [_i, _i, _i, C.POINTER(C.c_int64)],
    "ocf_conv3x3s2_c3_wgrad": [_T4P, _T4P, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _vp, _i, _i, _i, _i, _vp],
    "ocf_adam_flat": [_vp, _vp, _vp, _vp, _vp, C.c_int64, _f, C.c_double, C.c_double, _f, _f, _vp, _vp],
    "ocf_bias_act_fwd": [_T4P, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "ocf_bias_act_bwd": [_T4P, _T4P, _T4P, _vp, _i, _i, _i, _i, _f, _i, _i, _vp],
    "ocf_bn_act_workspace_bytes": [_i, _i, C.POINTER(C.c_int64)],
    "ocf_bn_act_fwd": [_T4P, _T4P, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _f, _i, _i, _i, _vp],
    "ocf_bn_act_bwd": [_T4P, _T4P, _T4P, _T4P, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp],
    "ocf_bn_act_bwd_x": [_T4P, _T4P, _T4P, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp],
}

_lock = threading.Lock()
_lib = None


def lib_path() -> str:
    return _build.LIB_PATH


def load():
    """Return the loaded library, building it first if the .so is absent.  Raises on failure."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not os.path.exists(path):
            _build.build()
        try:
            lib = C.CDLL(path)
        except OSError as e:  # pragma: no cover
            raise RuntimeError(f"ocflow_amd: cannot load {path}: {e}. The HIP extension is required; "
                               "there is no CPU fallback (python -m ocflow_amd.build)") from e
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        lib.ocf_status_string.argtypes = [C.c_int]
        lib.ocf_status_string.restype = C.c_char_p
        if lib.ocf_abi_version() != ABI_VERSION:
            raise RuntimeError(f"ocflow_amd: {path} implements ABI version {lib.ocf_abi_version()}, this package binds version {ABI_VERSION} "
                               "(stale build: python -m ocflow_amd.build --force)")
        _lib = lib
    return _lib


def check(status: int, what: str):
    if status != OCF_OK:
        msg = load().ocf_status_string(status).decode()
        raise RuntimeError(f"ocflow_amd: {what} failed: {msg} (status {status})")
