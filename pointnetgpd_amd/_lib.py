"""ctypes binding of libpngpd.so (include/pngpd.h).  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpngpd.so")

OK = 0
FP32, BF16, BF16X3 = 0, 1, 2
LAYOUT_C3P, LAYOUT_P3 = 0, 1
PART_STN, PART_TRUNK, PART_HEAD = 1, 2, 4
NOALLOC_MAX_B = 256
EDGES_F64, EDGES_F32 = 0, 1
UNITS = ("stn_conv1", "stn_conv2", "stn_conv3", "stn_fc1", "stn_fc2", "stn_fc3",
         "feat_conv1", "feat_conv2", "feat_conv3", "fc1", "fc2", "fc3")
FIELDS = ("weight", "bias", "bn_weight", "bn_bias", "bn_mean", "bn_var")
SLOTS = len(UNITS) * len(FIELDS)

PRECISIONS = {"fp32": FP32, "float32": FP32, "bf16": BF16, "bfloat16": BF16, "bf16x3": BF16X3}


class PngpdError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"pngpd error {code}: {message}")
        self.code = code


_lib = None

# name -> (restype, argtypes): every symbol include/pngpd.h declares
SIGNATURES = {
    "pngpd_last_error": (c_char_p, []),
    "pngpd_abi_version": (c_int, []),
    "pngpd_model_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int]),
    "pngpd_model_destroy": (c_int, [c_void_p]),
    "pngpd_model_load": (c_int, [c_void_p, POINTER(c_void_p), c_int, c_void_p]),
    "pngpd_model_load_device": (c_int, [c_void_p, POINTER(c_void_p), c_int, c_void_p]),
    "pngpd_forward_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "pngpd_forward_infer": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    "pngpd_forward_infer_noalloc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                            c_void_p, c_void_p]),
    "pngpd_forward_infer_timed": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_size_t, c_void_p, c_void_p]),
    "pngpd_crop_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pngpd_crop_count": (c_int, [c_void_p, c_int, c_void_p, c_int, POINTER(c_double), c_void_p, c_size_t, c_void_p,
                                 c_void_p, c_void_p]),
    "pngpd_crop_emit": (c_int, [c_void_p, c_int, c_int, POINTER(c_double), c_void_p, c_size_t, c_void_p, c_void_p,
                                c_void_p, c_int64, c_void_p, c_void_p]),
    "pngpd_crop_gather": (c_int, [c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_int,
                                  c_int, c_int, c_void_p, c_uint64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pngpd_crop_subsample": (c_int, [c_void_p, c_int, c_void_p, c_int, POINTER(c_double), c_void_p, c_size_t, c_int,
                                     c_int, c_int, c_uint64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pngpd_crop_count_frames": (c_int, [c_void_p, c_int, c_void_p, c_int, POINTER(c_double), c_void_p, c_size_t,
                                        c_void_p, c_void_p, c_void_p]),
    "pngpd_dataset_frames": (c_int, [c_void_p, c_int, c_int, POINTER(c_double), c_double, c_void_p, c_void_p]),
    "pngpd_crop_gather_dataset": (c_int, [c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                                          c_int, c_int, c_void_p, c_uint64, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p]),
    "pngpd_vote": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pngpd_voxel_workspace_bytes": (c_size_t, [c_int]),
    "pngpd_voxel_unique": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p, c_int64, c_void_p, c_void_p]),
    "pngpd_train_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "pngpd_train_forward": (c_int, [c_int, c_int, c_int, c_void_p, POINTER(c_void_p), c_int, c_float, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pngpd_train_forward_timed": (c_int, [c_int, c_int, c_int, c_void_p, POINTER(c_void_p), c_int, c_float, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "pngpd_train_backward": (c_int, [c_int, c_int, c_int, c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_int,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pngpd_train_backward_part": (c_int, [c_int, c_int, c_int, c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "pngpd_train_forward_parts": (c_int, [c_int, c_int, c_int, c_int, c_void_p, POINTER(c_void_p), c_int, c_float,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pngpd_train_saved": (c_int, [c_int, c_int, c_int, c_void_p, c_size_t, c_int, POINTER(c_void_p), POINTER(c_size_t)]),
    "pngpd_train_argmax": (c_int, [c_int, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "pngpd_nll_loss": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pngpd_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                                c_int, c_float, c_void_p]),
    "pngpd_score": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
}


def load() -> ctypes.CDLL:
    """Load the library once.  torch must be imported first so both share one HIP runtime."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP library has not been built "
            "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != OK:
        raise PngpdError(rc, load().pngpd_last_error().decode("utf-8", "replace"))
