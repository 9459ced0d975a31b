"""ctypes binding of ``csrc/libmars_hip.so`` -- the C-ABI declared in ``include/mars_hip.h``.

There is no CPU fallback: if the library is missing, ``lib()`` raises and every op of
the package with it.
"""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MARS_HIP_LIB") or os.path.join(_HERE, "csrc", "libmars_hip.so")   # override: A/B runs of two builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mars_hip.h")

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_QUICKGELU = 0, 1, 2


class MarsAxisTable(Structure):
    _fields_ = [("lo", c_void_p), ("size", c_void_p), ("w", c_void_p), ("kmax", c_int32)]


class MarsGemm(Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p),
        ("lda", c_int64), ("ldb", c_int64), ("ldc", c_int64),
        ("sA0", c_int64), ("sA1", c_int64), ("sB0", c_int64), ("sB1", c_int64), ("sC0", c_int64), ("sC1", c_int64),
        ("M", c_int32), ("N", c_int32), ("K", c_int32),
        ("nb0", c_int32), ("nb1", c_int32),
        ("bias", c_void_p), ("gamma", c_void_p), ("residual", c_void_p),
        ("ldr", c_int64), ("sR0", c_int64), ("sR1", c_int64),
        ("alpha", c_float), ("act", c_int32), ("in_dtype", c_int32), ("out_dtype", c_int32),
        ("ln_copy", c_void_p), ("ldcopy", c_int64), ("ln_part", c_void_p), ("ln_stat", c_void_p), ("ln_colsum", c_void_p),
        ("ln_lo", c_void_p),
    ]


_P, _I, _L, _F, _D = c_void_p, c_int, c_int64, c_float, c_double

# name -> argtypes (restype is int unless listed in _RESTYPES)
_SIGNATURES = {
    "marship_gemm": [POINTER(MarsGemm), _P],
    "marship_set_gemm_variant": [_I],
    "marship_gemm_variant_for": [_P],
    "marship_set_gemm_split": [_I],
    "marship_set_gemm_slots": [_I],
    "marship_gemm_split_rows": [_P],
    "marship_layernorm": [_P, _L, _P, _P, _P, _L, _I, _I, _F, _I, _P],
    "marship_ln_finish": [_P, _I, _I, _I, _F, _P, _P, _F, _P],
    "marship_row_stats": [_P, _L, _I, _I, _F, _P, _P, _L, _P, _P, _F, _P],
    "marship_im2col": [_P, _I, _I, _I, _I, _I, _I, _P, _P, _F, _P, _L, _I, _I, _P],
    "marship_attn_small": [_P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _F, _I, _I, _P],
    "marship_embed_tokens": [_P, _I, _P, _P, _P, _L, _I, _I, _P],
    "marship_write_prefix_tokens": [_P, _I, _I, _I, _P, _P, _P, _I, _P],
    "marship_transpose": [_P, _L, _L, _L, _P, _L, _L, _L, _I, _I, _I, _I, _I, _P],
    "marship_softmax_rows": [_P, _L, _P, _L, _I, _I, _I, _I, _P],
    "marship_attn_mean_accum": [_P, _L, _L, _I, _I, _I, _F, _P, _L, _I, _P],
    "marship_swiglu": [_P, _P, _L, _I, _I, _I, _P],
    "marship_cast": [_P, _P, _L, _I, _I, _P],
    "marship_flash_attn": [_P, _P, _P, _I, _I, _I, _I, _F, _P],
    "marship_flash_attn_mean": [_P, _P, _I, _I, _I, _I, _F, _P, _L, _F, _P],
    "marship_flash_attn_fp8": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _F, _P],
    "marship_l2norm_rows": [_P, _L, _L, _I, _I, _I, _I, _P, _L, _I, _P],
    "marship_vva_prior": [_P, _L, _P, _I, _I, _P, _I, _P, _P],
    "marship_vva_prior_static": [_P, _L, _P, _I, _I, _P, _P],
    "marship_minmax": [_P, _P, _I, _P],
    "marship_box_mask": [_P, _I, _D, _P, _P],
    "marship_sinkhorn1": [_P, _L, _P, _L, _I, _P, _P],
    "marship_max_and_transpose": [_P, _P, _L, _I, _P, _P, _P],
    "marship_masked_matvec": [_P, _L, _P, _P, _I, _P, _P],
    "marship_resample_axis": [_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _F, _F, _P, _I, _P],
    "marship_resample_axis_u8": [_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P],
    "marship_f32_to_u8": [_P, _P, _L, _P],
    "marship_pool_masks": [_P, _I, _I, _I, _P, _P],
    "marship_proposal_stats": [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    "marship_cosine_scores": [_P, _L, _I, _I, _P, _P, _P],
    "marship_rank": [_P, _P, _P, _P, _P, _P, _I, _D, _D, _D, _P, _P, _P, _P],
    "marship_merge_masks": [_P, _P, _I, _L, _P, _P],
    "marship_iou_counts": [_P, _P, _P, _L, _P, _P],
    "marship_points_in_masks": [_P, _I, _I, _I, _P, _I, _P, _P],
    "marship_gather_rows": [_P, _L, _I, _P, _P, _L, _I, _L, _I, _I, _P],
    "marship_relpos_bias": [_P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P],
    "marship_relpos_bias_rows": [_P, _L, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P],
    "marship_sam_attn": [_P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P],
    "marship_sam_attn_rows": [_P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "marship_softmax_bias_rows": [_P, _L, _P, _L, _L, _I, _P, _P, _I, _I, _I, _P],
    "marship_fourier_pe": [_P, _P, _I, _P, _P, _P, _L, _L, _P],
    "marship_add_cast": [_P, _P, _L, _P, _L, _I, _I, _P],
    "marship_act_cast": [_P, _P, _L, _I, _I, _P],
    "marship_stream_copies": [_P, _P, _P, _F, _P, _P, _L, _P, _P, _L, _I, _I, _P],
    "marship_layernorm_act_rows": [_P, _P, _P, _F, _P, _L, _L, _I, _I, _I, _P],
    "marship_sam_upscale_ln_gelu": [_P, _L, _P, _P, _F, _P, _L, _L, _I, _I, _P],
    "marship_sam_upscale_masks": [_P, _L, _P, _P, _P, _L, _P, _L, _I, _P],
    "marship_sam_i2t_kv_operands": [_P, _L, _P, _L, _I, _I, _P, _P],
    "marship_sam_i2t_block": [_P, _L, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _F, _P, _P, _I, _P],
    "marship_sam_t2i_q_operands": [_P, _L, _I, _I, _P, _P],
    "marship_sam_t2i_block": [_P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _P, _L, _I, _I, _P],
    "marship_hyper_product": [_P, _L, _P, _L, _P, _I, _L, _I, _P],
    "marship_mask_embed": [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P],
    "marship_resample_mask_stats": [_P, _L, _L, _I, _I, _I, _I, _I, POINTER(MarsAxisTable), POINTER(MarsAxisTable), _I, _I, _F, _F, _P, _P, _P],
    "marship_resample_threshold": [_P, _L, _L, _I, _P, _I, _I, _I, _I, POINTER(MarsAxisTable), POINTER(MarsAxisTable), _I, _I, _F, _P, _P],
    "marship_mask_stats": [_P, _I, _I, _I, _F, _F, _P, _P],
    "marship_threshold_masks": [_P, _F, _P, _L, _P],
    "marshost_box_nms": [_P, _P, _I, _F, _P, _P],
    "marshost_remove_small_regions": [_P, _I, _I, _F, _I, _P],
    "marshost_lap": [_P, _L, _I, _I, _I, _P, _P],
    "marship_pack_masks": [_P, _I, _L, _P, _P],
    "marship_unpack_masks": [_P, _I, _L, _P, _P],
    "marship_cosine_table": [_P, _L, _I, _I, _P, _I, _F, _P, _P],
    "marship_layernorm_bwd": [_P, _P, _P, _L, _F, _I, _P, _P, _I, _I, _F, _P],
    "marship_quickgelu_bwd": [_P, _P, _P, _L, _P],
    "marship_softmax_bwd": [_P, _P, _P, _L, _I, _I, _F, _P],
    "marship_col_mean": [_P, _L, _I, _I, _I, _P, _P],
    "marship_cam_head": [_P, _P, _P, _I, _I, _F, _P, _P, _P],
    "marship_cam_rows": [_P, _L, _P, _I, _I, _I, _P, _P],
    "marshost_emd_scores": [_P, _L, _I, _I, _P, _P, _I, _I, _P, _I],
    "marshost_emd2_uniform": [_P, _I, _I],
    "marship_last_error": [],
    "marship_version": [],
}
_RESTYPES = {"marshost_emd2_uniform": c_double, "marship_last_error": c_char_p}

_lib = None


class MarsHipError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load the shared library once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MarsHipError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                f"(make -C .../csrc). There is no CPU fallback for the ranking path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, c_int)
        _lib = handle
    return _lib


def declared_symbols():
    """Entry points named in include/mars_hip.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mars(?:hip|host)_\w+)\s*\(", text)))


def missing_symbols():
    handle = ctypes.CDLL(LIB_PATH)
    out = []
    for name in declared_symbols():
        try:
            getattr(handle, name)
        except AttributeError:
            out.append(name)
    unbound = [n for n in declared_symbols() if n not in _SIGNATURES]
    return out + [f"{n} (no ctypes signature)" for n in unbound]


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().marship_last_error()
        raise MarsHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

