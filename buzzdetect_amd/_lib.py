"""ctypes binding of libbuzzdetect_hip.so — the declarations mirror include/buzzdetect_hip.h."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

from . import build as _build

ABI_VERSION = 5
EMBEDDER_BLOB_FLOATS = 3_217_344
EMBEDDING_SIZE = 1024
MEL_BANDS = 64
PATCH_FRAMES = 96
NUM_STAGES = 27
PROFILE_SLOTS = 29
MAX_CLASSES = 64

ERROR_NAMES = {-1: "BD_EINVAL", -2: "BD_ENODEVICE", -3: "BD_EHIP", -4: "BD_EWORKSPACE",
               -5: "BD_ERANGE", -6: "BD_EWEIGHTS"}


class BuzzdetectHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {message}")
        self.code = code


class bd_weights(C.Structure):
    _fields_ = [
        ("embedder_blob", C.POINTER(C.c_float)),
        ("embedder_floats", C.c_int64),
        ("mel", C.POINTER(C.c_float)),
        ("head_kernel", C.POINTER(C.c_float)),
        ("head_bias", C.POINTER(C.c_float)),
        ("n_classes", C.c_int32),
    ]


# name -> (restype, argtypes); one entry per prototype in include/buzzdetect_hip.h
PROTOTYPES = {
    "bd_abi_version": (C.c_int, []),
    "bd_last_error": (C.c_char_p, []),
    "bd_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(bd_weights)]),
    "bd_destroy": (C.c_int, [C.c_void_p]),
    "bd_set_group_windows": (C.c_int, [C.c_void_p, C.c_int32]),
    "bd_padded_length": (C.c_int64, [C.c_int64, C.c_int32]),
    "bd_num_frames": (C.c_int64, [C.c_int64, C.c_int32]),
    "bd_num_windows": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "bd_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32]),
    "bd_resample_length": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "bd_resample_taps": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "bd_set_resample_quality": (C.c_int, [C.c_void_p, C.c_int32]),
    "bd_format_rows": (C.c_int64, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_int64]),
    "bd_resample_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "bd_debug_fir_plan": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    "bd_resample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                              C.c_void_p]),
    "bd_resample_s16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                  C.c_void_p]),
    "bd_frontend": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "bd_patches": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "bd_embed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                           C.c_void_p, C.c_void_p]),
    "bd_predict": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                             C.c_void_p, C.c_void_p, C.c_void_p]),
    "bd_batch_num_windows": (C.c_int64, [C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    "bd_batch_workspace_bytes": (C.c_int64, [C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32]),
    "bd_predict_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bd_predict_chunks": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "bd_calibrate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "bd_get_scales": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "bd_set_activation_exponents": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "bd_stage_shape": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "bd_stage_tap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                               C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "bd_debug_pointwise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_void_p]),
    "bd_set_pointwise_variant": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "bd_set_pointwise_mode": (C.c_int, [C.c_void_p, C.c_int32]),
    "bd_range_flag": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_void_p]),
    "bd_range_flag_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "bd_set_fusion": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "bd_debug_pointwise_f16x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                           C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "bd_stager_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.c_int32]),
    "bd_stager_destroy": (C.c_int, [C.c_void_p]),
    "bd_stager_read": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "bd_stager_acquire": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]),
    "bd_stager_submit": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]),
    "bd_profile_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "bd_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32]),
}

_lib: Optional[C.CDLL] = None


def library_path() -> str:
    return os.environ.get("BUZZDETECT_HIP_LIB", _build.LIB_PATH)


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load the shared object (building it with hipcc first if it is not there).

    There is deliberately no fallback: if the library cannot be built or loaded this raises.
    """
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    external = "BUZZDETECT_HIP_LIB" in os.environ
    if not os.path.exists(path):
        if not build_if_missing or external:
            raise FileNotFoundError(f"{path} not found; run `python -m buzzdetect_amd.build`")
        _build.build()
    elif not external and _build.needs_build():
        # the in-tree library is older than its sources (it is git-ignored, so a checkout does not refresh it):
        # rebuild where hipcc exists, refuse to run stale kernels where it does not
        if not build_if_missing:
            raise RuntimeError(f"{path} is older than its sources; run `python -m buzzdetect_amd.build`")
        try:
            _build.build()
        except (RuntimeError, OSError, subprocess.CalledProcessError) as exc:
            raise RuntimeError(f"{path} is older than its sources and could not be rebuilt: {exc}") from exc
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.bd_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {lib.bd_abi_version()} != {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int) -> int:
    if rc < 0:
        raise BuzzdetectHipError(int(rc), load().bd_last_error().decode(errors="replace"))
    return int(rc)
