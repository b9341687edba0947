"""ctypes binding of libfgvc_hip.so (include/fgvc_hip.h).

There is NO fallback: if the shared library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import contextlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FGVC_HIP_LIB: another build of the same library (A/B timing of two kernel versions on one box: tools/experiments)
LIB_PATH = os.environ.get("FGVC_HIP_LIB") or os.path.join(_HERE, "lib", "libfgvc_hip.so")

FGVC_OK = 0
ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_LAUNCH = 1, 2, 3
NO_LIMIT = 0x3FFFFFFF
PAIR_MASKED = 1
WEIGHT_SOFTMAX, WEIGHT_COSINE, WEIGHT_RAW = 0, 1, 2

_p = C.c_void_p
_i = C.c_int
_f = C.c_float

# name -> (restype, argtypes); mirrors include/fgvc_hip.h one to one
SIGNATURES = {
    "fgvc_version": (C.c_char_p, []),
    "fgvc_last_error": (C.c_char_p, []),
    "fgvc_set_option": (_i, [C.c_char_p, _i]),
    "fgvc_r2max_for_radius": (_i, [_f]),
    "fgvc_normalize_chw_to_hwc_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "fgvc_pair_topk_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "fgvc_pair_topk_f16x3": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "fgvc_pair_topk_f16x3_runs": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _p]),
    "fgvc_split_f16x2": (_i, [_p, _p, C.c_int64, _i, _p]),
    "fgvc_split_f16f6p": (_i, [_p, _p, C.c_int64, _i, _p]),
    "fgvc_pair_topk_f16f6": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "fgvc_pair_topk_f16f6_runs": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _p]),
    "fgvc_split_f16f6x": (_i, [_p, _p, C.c_int64, _i, _p]),
    "fgvc_pair_topk_f16f6x": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "fgvc_pair_topk_f16f6x_runs": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _p]),
    "fgvc_merge_refine_workspace_bytes": (C.c_size_t, [_i, _i]),
    "fgvc_merge_refine_topk_f32": (_i, [_p, _p, _p, _p, _p, C.c_int64, _i, _p, C.c_int64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _f, _i, _i, _i,
                                        _p, _p, _p, _p, _p]),
    "fgvc_pair_topk_f16x3_timed_out": (_i, []),
    "fgvc_pair_topk_f16x3_probe": (_i, [_p]),
    "fgvc_conv64_probe": (_i, [_p]),
    "fgvc_nchw_to_split_nhwc_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "fgvc_conv_split_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fgvc_conv_split_fmt_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "fgvc_conv_split_proj_fmt_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "fgvc_conv_split_bank_f16f6p_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fgvc_conv_split_bank_f16f6x_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fgvc_conv64_split_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "fgvc_conv64_split_res_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "fgvc_conv64_split_fmt_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "fgvc_conv_s2_split_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fgvc_conv_s2_split_fmt_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "fgvc_stem7_split_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "fgvc_stem7_split_fmt_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "fgvc_nhwc_to_split_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fgvc_normalize_nhwc_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "fgvc_normalize_split_nhwc_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "fgvc_normalize_split_f16x2_nhwc_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "fgvc_normalize_split_f16f6p_nhwc_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "fgvc_normalize_split_f16f6x_nhwc_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "fgvc_merge_topk_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p, _p, _p, _p]),
    "fgvc_propagate_topk_f32": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "fgvc_corr_volume_f32": (_i, [_p, _p, _i, _i, _i, _f, _p, _p]),
    "fgvc_split_bf16": (_i, [_p, _p, C.c_int64, _i, _p]),
    "fgvc_corr_volume_bf16x3": (_i, [_p, _p, _i, _i, _i, _f, _p, _p]),
    "fgvc_corr_volume_bf16": (_i, [_p, _p, _i, _i, _i, _f, _p, _p]),
    "fgvc_split_f16f8": (_i, [_p, _p, C.c_int64, _i, _p]),
    "fgvc_corr_volume_f16f8": (_i, [_p, _p, _i, _i, _i, _f, _p, _p]),
    "fgvc_split_f16f6": (_i, [_p, _p, C.c_int64, _i, _p]),
    "fgvc_corr_volume_f16f6": (_i, [_p, _p, _i, _i, _i, _f, _p, _p]),
    "fgvc_debug_store_sweep_f32": (_i, [_p, C.c_int64, _i, _p]),
    "fgvc_dense_attend_splits": (_i, [_i, _i]),
    "fgvc_dense_attend_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p]),
    "fgvc_dense_attend_finish_f32": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "fgvc_dense_kth_f32": (_i, [_p, _i, _i, _i, _p, _i, _p, _p]),
    "fgvc_dense_propagate_f32": (_i, [_p, _p, _i, _i, _i, _p, _p, _i, _p, _p]),
    "fgvc_local_corr_topk_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p, _p]),
    "fgvc_local_corr_topk_f16x3": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p, _p]),
    "fgvc_topk_coord_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "fgvc_c2f_refine_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _p]),
    "fgvc_c2f_refine_mode_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _p, _p, _p, _p]),
    "fgvc_bn_act_f32": (_i, [_p, _p, _p, _p, _p, _p, _f, _i, _p, _i, _i, _i, _p]),
    "fgvc_gaussian_labels_f32": (_i, [_p, _i, _i, _i, _i, _f, _p, _p]),
    "fgvc_softargmax_workspace_bytes": (C.c_size_t, [_i, _i]),
    "fgvc_softargmax_top5_f32": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _f, _p, _p, _p]),
}

_lib = None


class FgvcHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64; it must be in the process BEFORE this library is dlopen'ed so
        # that both resolve to ONE HIP runtime (otherwise the system runtime is loaded first and kernels
        # launched from here see "no ROCm-capable device").
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise FgvcHipError(
                f"{LIB_PATH} is missing: build it with `python -m fgvc_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


ABLATIONS_PATH = os.path.join(_HERE, "lib", "libfgvc_hip_ablations.so")
_ablations = None


@contextlib.contextmanager
def ablations():
    """Inside the block every call goes to libfgvc_hip_ablations.so -- the same kernels, whose fgvc_set_option also accepts the profiling
    ablations that give WRONG results (the production library refuses them).  For measurements only (bench.py's store-stream replay,
    tools/experiments); each library has its own option words.  Raises FgvcHipError when that library was not built."""
    global _lib, _ablations
    prod = load()
    if _ablations is None:
        if not os.path.exists(ABLATIONS_PATH):
            raise FgvcHipError(f"{ABLATIONS_PATH} is missing: `python -m fgvc_amd.build` makes it beside the production library")
        lib = C.CDLL(ABLATIONS_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _ablations = lib
    _lib = _ablations
    try:
        yield _ablations
    finally:
        _lib = prod


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != FGVC_OK:
        raise FgvcHipError(f"{name} failed (code {rc}): {lib.fgvc_last_error().decode()}")
