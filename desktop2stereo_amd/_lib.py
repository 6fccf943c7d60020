"""ctypes binding of libd2s_hip.so (include/d2s.h).

The library is the product path: if it is missing or a symbol is absent this module raises --
there is no CPU or PyTorch fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("D2S_LIB") or os.path.join(HERE, "libd2s_hip.so")     # D2S_LIB: another build of the same library (A/B runs)

OK = 0
MODE = {"Half-SBS": 0, "Full-SBS": 1, "Half-TAB": 2, "Full-TAB": 3}
FMT_U8_HWC, FMT_F32_CHW, FMT_F32_HWC, FMT_U8_CHW = 0, 1, 2, 3
PREC_FP32, PREC_BF16, PREC_FP8, PREC_BF16X3, PREC_FP8_MLP = 0, 1, 2, 3, 4


class ModelDesc(C.Structure):
    _fields_ = [("hidden", C.c_int32), ("heads", C.c_int32), ("layers", C.c_int32),
                ("out_indices", C.c_int32 * 4), ("neck", C.c_int32 * 4), ("fusion", C.c_int32),
                ("head_hidden", C.c_int32), ("mlp", C.c_int32), ("patch", C.c_int32),
                ("pos_grid", C.c_int32), ("ln_eps", C.c_float), ("precision", C.c_int32), ("temporal", C.c_int32),
                ("max_depth", C.c_float)]


class PostParams(C.Structure):
    _fields_ = [("percentile", C.c_float), ("subsample_cap", C.c_int32), ("gamma", C.c_float),
                ("foreground_scale", C.c_float), ("aa_strength", C.c_float), ("ema_alpha", C.c_float),
                ("metric", C.c_int32)]


RESAMPLE = {"bilinear": 0, "bicubic_aa": 1}      # D2S_RESAMPLE_*: _resize_patch_aligned_t's CPU branch / IS_CUDA branch


class PreParams(C.Structure):
    _fields_ = [("mean", C.c_float * 3), ("std", C.c_float * 3), ("resample", C.c_int32), ("square", C.c_int32)]


class SbsParams(C.Structure):
    _fields_ = [("ipd_uv", C.c_double), ("depth_ratio", C.c_float), ("convergence", C.c_float),
                ("display_mode", C.c_int32), ("fill_16_9", C.c_int32)]


class DibrParams(C.Structure):
    _fields_ = [("ipd_uv", C.c_double), ("depth_strength", C.c_float), ("convergence", C.c_float), ("roll", C.c_float),
                ("search_radius", C.c_float), ("depth_tolerance", C.c_float), ("blur_radius", C.c_float),
                ("res_w", C.c_float), ("res_h", C.c_float), ("display_mode", C.c_int32),
                ("feather_enabled", C.c_int32), ("feather_width", C.c_float), ("corner_radius", C.c_float),
                ("viewport", C.c_float * 4), ("alpha_mode", C.c_int32), ("struct_size", C.c_uint32)]


DIBR_ALPHA = {"window": 0, "premultiplied": 1, "rgba": 2}      # D2S_DIBR_ALPHA_*


# every symbol include/d2s.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "d2s_last_error": (C.c_char_p, []),
    "d2s_version": (C.c_int, []),
    "d2s_debug_reload_env": (C.c_int, []),
    "d2s_debug_lds_poison": (C.c_int, []),
    "d2s_debug_pp_tail_timeouts": (C.c_int, [C.c_int, C.POINTER(C.c_uint)]),
    "d2s_engine_create": (C.c_int, [C.POINTER(ModelDesc), C.c_int, C.POINTER(_P)]),
    "d2s_engine_set_weight": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int]),
    "d2s_engine_finalize": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "d2s_engine_destroy": (C.c_int, [_P]),
    "d2s_engine_memory": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "d2s_preprocess": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(PreParams), _P]),
    "d2s_process_shape": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "d2s_process": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "d2s_process_rgb": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "d2s_process_area_shape": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "d2s_process_area": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "d2s_overlay_text": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_char_p, _P]),
    "d2s_model_forward": (C.c_int, [_P, _P, _P, C.c_int, _P]),
    "d2s_engine_calibrate": (C.c_int, [_P, _P, C.c_int, _P]),
    "d2s_post_process": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(PostParams), _P, C.c_uint64, _P]),
    "d2s_post_process_workspace": (C.c_uint64, [C.c_int, C.c_int, C.c_int]),
    "d2s_post_process_to": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(PostParams), _P, C.c_uint64, _P]),
    "d2s_ema_update": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P]),
    "d2s_upsample_depth": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P]),
    "d2s_make_sbs": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.POINTER(SbsParams), _P, C.c_int, _P]),
    "d2s_sbs_shape": (C.c_int, [C.c_int, C.c_int, C.POINTER(SbsParams), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "d2s_dibr_shape": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "d2s_dibr_warp": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(DibrParams), _P, C.c_int, _P]),
    "d2s_jpeg_bound": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "d2s_jpeg_encode": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int64, _P, _P, C.c_int64, _P]),
    "d2s_present_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(_P)]),
    "d2s_present_bind": (C.c_int, [_P, C.c_int, _P, C.c_uint64]),
    "d2s_present_bind_gl_buffer": (C.c_int, [_P, C.c_int, C.c_uint]),
    "d2s_present_acquire": (C.c_int, [_P, _P, C.POINTER(C.c_int), C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "d2s_present_publish": (C.c_int, [_P, C.c_int, _P]),
    "d2s_present_cancel": (C.c_int, [_P, C.c_int, _P]),
    "d2s_present_consume": (C.c_int, [_P, _P, C.POINTER(C.c_int), C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "d2s_present_release": (C.c_int, [_P, C.c_int, _P]),
    "d2s_present_destroy": (C.c_int, [_P]),
    "d2s_pipeline": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(PreParams), C.POINTER(PostParams),
                               C.POINTER(SbsParams), C.c_int, _P, C.c_int, _P, _P]),
    "d2s_engine_reset_stream": (C.c_int, [_P]),
    "d2s_engine_tap": (C.c_int, [_P, C.c_char_p, _P, C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_int), _P]),
    "d2s_engine_profile": (C.c_int, [_P, C.c_int]),
    "d2s_engine_profile_read": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "d2s_profile_class_name": (C.c_char_p, [C.c_int]),
    "d2s_gemm_probe": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "d2s_attention_probe": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
}

_lib = None


class D2SError(RuntimeError):
    pass


def load(path: str = LIB_PATH):
    """Load the library and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise D2SError(f"{path} not found: build it with `python -m desktop2stereo_amd.build` "
                       "(there is no fallback path)")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str = ""):
    if status != OK:
        msg = load().d2s_last_error()
        raise D2SError(f"{what or 'libd2s_hip'} failed (status {status}): {msg.decode() if msg else ''}")
