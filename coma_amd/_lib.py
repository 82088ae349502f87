"""ctypes binding of libcoma_hip.so (the C ABI declared in include/coma_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call is made with tensors
that are not on a HIP device, this module raises.  The CPU oracle under oracle/ is test
infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# COMA_HIP_LIB=<path>: tuning aid -- load another BUILD of the library (A/B of two builds inside one GPU call: box-to-box spread is larger
# than most of the effects being measured); the product and the tests use the in-tree library
LIB_PATH = os.environ.get("COMA_HIP_LIB") or os.path.join(_HERE, "libcoma_hip.so")
ABI_VERSION = 8                  # = COMA_ABI_VERSION of include/coma_hip.h: bumped with every change of the SIGNATURES table below

_lib = None

_vp, _i, _i64, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
_fp3 = C.POINTER(C.c_float)

# name -> (restype, argtypes); must list every function declared in include/coma_hip.h
SIGNATURES = {
    "coma_abi_version": (_i, []),
    "coma_last_error": (C.c_char_p, []),
    "coma_contact_accumulate_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _i, _i, _i, _i, _fp3, _fp3, _f, _f, _f, _f,
                                         _vp, _vp, _vp, _vp, _vp, _vp]),
    "coma_contact_map_f32": (_i, [_vp, _vp, _fp3, _vp, _vp, _i64, _i, _f, _vp, _vp]),
    "coma_significant_pairs_u8": (_i, [_vp, _f, _i, _i, _vp, _vp, _vp, _vp]),
    "coma_masked_max_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "coma_entropy_f32": (_i, [_vp, _i64, _i, _f, _f, _vp, _vp]),
    "coma_row_argmax_i64": (_i, [_vp, _i64, _i, _i64, _i64, _vp, _vp, _vp]),
    "coma_contact_select_u8": (_i, [_vp, _vp, _i, _i, _f, _vp, _vp]),
    "coma_occupancy_splat": (_i, [_vp, _i, _i, _i, _vp, _d, _d, _vp, _vp]),
    "coma_occupancy_reduce": (_i, [_vp, _vp, _i, _i64, _vp, _vp, _vp]),
    "coma_occupancy_fused_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i]),
    "coma_occupancy_fused": (_i, [_vp, _i, _i, _i, _vp, _d, _d, _d, _i, _vp, _i, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "coma_nearest_vertex_i64": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "coma_dlt_score_f64": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "coma_ransac_mse_f64": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _d, _vp, _vp, _vp]),
    "coma_vertex_normals_f64": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _d, _vp, _vp]),
    # include/sd_hip.h
    "sd_conv_gemm_f16": (_i, [_vp, _vp]),
    "sd_conv_gemm_workspace_bytes": (C.c_size_t, []),
    "sd_debug_timestamps": (_i, [_vp, _i]),
    "sd_groupnorm_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp, _vp, _vp]),
    "sd_groupnorm_colstats_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "sd_layernorm_f16": (_i, [_vp, _i64, _i, _f, _vp, _vp, _vp, _vp]),
    "sd_attention_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]),
    "sd_attention_wide_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "sd_xattn_chain_f16": (_i, [_vp] * 15 + [_i64, _i, _i, _i, _f, _vp, _i, _vp]),
    "sd_groupnorm_table_f16": (_i, [_vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _i, _vp]),
    "sd_xfront_f16": (_i, [_vp] * 11 + [_i64, _i, _i, _f, _vp]),
    "sd_xtail_f16": (_i, [_vp] * 11 + [_i64, _vp]),
    "sd_winograd_input_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _f, _vp, _vp]),
    "sd_winograd_weight_f16": (_i, [_vp, _i, _i, _f, _vp, _vp]),
    "sd_winograd_output_f16": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _i, _f, _vp, _vp]),
    "sd_groupnorm_table_cat_f16": (_i, [_i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sd_conv3x3_small_n_f16": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "sd_conv3x3_halo_f16": (_i, [_vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "sd_gn_winograd_input_f16": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _i, _f, _vp, _vp]),
    "sd_im2col3x3_c3_f16": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "sd_conv3x3_c3_f16": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "sd_softmax_f16": (_i, [_vp, _i64, _i, _i, _f, _vp]),
    "sd_cfg_ddim_step": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _i, _vp]),
    "sd_timestep_embedding_f16": (_i, [_vp, _i, _i, _vp, _vp]),
    "sd_nchw_to_nhwc_f16": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "sd_nhwc_to_nchw_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "sd_image_to_u8": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "sd_vae_sample": (_i, [_vp, _i, _vp, _f, _i64, _vp, _vp, _vp]),
    "sd_add_noise": (_i, [_vp, _vp, _f, _i64, _vp, _vp]),
    "sd_mask_adapt": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "sd_model_create": (_i, [C.POINTER(_vp)]),
    "sd_model_destroy": (_i, [_vp]),
    "sd_model_register_buffer": (_i, [_vp, _vp, C.c_size_t, _i]),
    "sd_model_bind": (_i, [_vp, C.c_char_p, _vp, C.c_size_t]),
    "sd_model_binding": (_i, [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "sd_model_record_begin": (_i, [_vp, C.c_char_p]),
    "sd_model_record_end": (_i, [_vp]),
    "sd_model_num_launches": (_i, [_vp, C.c_char_p]),
    "sd_model_run": (_i, [_vp, C.c_char_p, _vp]),
    "sd_model_replay": (_i, [_vp, C.c_char_p, _vp]),
    "sd_model_prepare": (_i, [_vp, C.c_char_p]),
    "sd_model_save": (_i, [_vp, C.c_char_p]),
    "sd_model_load": (_i, [C.c_char_p, C.POINTER(_vp)]),
    "sd_copy_d2d": (_i, [_vp, _vp, C.c_size_t, _vp]),
    "sd_unet_set_context": (_i, [_vp, _vp, _vp]),
    "sd_unet_forward": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "sd_vae_decode": (_i, [_vp, _vp, _vp, _vp]),
    "sd_vae_encode": (_i, [_vp, _vp, _vp, _vp]),
    "sd_mask_adapt_batched": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _d, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    # include/seg_hip.h
    "seg_conv_gemm_f32": (_i, [_vp, _vp]),
    "seg_resize_normalize_u8": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "seg_maxpool3x3s2_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "seg_subsample2_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "seg_memset": (_i, [_vp, _i, C.c_size_t, _vp]),
    "seg_rpn_select": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp]),
    "seg_rpn_select_levels": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "seg_sort_candidates": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "seg_nms": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "seg_roi_align_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "seg_box_predict": (_i, [_vp, _i, _vp, _vp, _i, _i, _f, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "seg_finalize_detections": (_i, [_vp, _vp, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp]),
    "seg_point_sample_f32": (_i, [_vp, _i, _i, _i, _i, _f, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _i, C.c_longlong, _vp]),
    "seg_upsample2x_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "seg_topk_points": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "seg_point_logit_scatter": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "seg_paste_masks": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
}


class ComaHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ComaHipError(
                f"{LIB_PATH} is missing: build it with `python -m coma_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        if os.environ.get("COMA_HIP_LIB"):
            import sys
            print(f"coma_amd: COMA_HIP_LIB override active -- loading {LIB_PATH} instead of the in-tree library (tuning aid)", file=sys.stderr, flush=True)
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)      # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        if h.coma_abi_version() != ABI_VERSION:
            raise ComaHipError(f"ABI version mismatch: {LIB_PATH} reports {h.coma_abi_version()}, this package binds version {ABI_VERSION} "
                               "(rebuild with `python -m coma_amd.build`)")
        _lib = h
    return _lib


_switched = threading.local()       # device that was current before stream_ptr() switched it for one launch


def check(rc: int, what: str):
    """Raise on a non-zero return code.  Every wrapper calls this right after its launch, so it is also where the device that
    stream_ptr() made current for that launch is handed back to the caller (the process-wide current device is not ours to change)."""
    _restore()
    if rc != 0:
        raise ComaHipError(f"{what} failed ({rc}): {lib().coma_last_error().decode()}")


def ptr(t: torch.Tensor | None, dtype=None, name="tensor"):
    """Device pointer of a contiguous HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ComaHipError(f"{name} must live on a HIP device (got {t.device}); there is no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise ComaHipError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ComaHipError(f"{name} must be contiguous")
    return C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """Current stream of `device`, and that device made current for the ONE launch that follows (the C ABI launches under HIP's
    current device, so a tensor on cuda:1 with cuda:0 current would otherwise meet a stream of another device); check() puts
    the caller's device back."""
    _restore()              # a launch that raised between its stream_ptr() and its check() left the switch behind: undo it first
    if device is not None:
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is not None and dev.index != torch.cuda.current_device():
            _switched.dev = torch.cuda.current_device()          # restored by check() after the launch
            torch.cuda.set_device(dev)
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _restore():
    prev = getattr(_switched, "dev", None)
    if prev is not None:
        _switched.dev = None
        torch.cuda.set_device(prev)


import contextlib  # noqa: E402


@contextlib.contextmanager
def on_device(device):
    """`with _lib.on_device(t.device) as stream:` -- the stream pointer for ANY NUMBER of launches on `device`, that device current
    inside the block and the caller's device back afterwards whatever happens (exceptions included).  The one-shot stream_ptr() /
    check() pair is for wrappers with exactly one launch; a wrapper that launches twice uses this."""
    _restore()
    dev = torch.device(device) if device is not None else None
    prev = None
    if dev is not None and dev.type == "cuda" and dev.index is not None and dev.index != torch.cuda.current_device():
        prev = torch.cuda.current_device()
        torch.cuda.set_device(dev)
    try:
        yield C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    finally:
        if prev is not None:
            torch.cuda.set_device(prev)


def vec3(v):
    return (C.c_float * 3)(float(v[0]), float(v[1]), float(v[2]))
