"""Host-side coefficient tables for the separable resampling kernels.

Two families, both input-independent (built once per (in_size, out_size) and cached):

* ``torch_aa_table``  -- the weights ``F.interpolate(..., antialias=True)`` uses
  (ATen ``_compute_indices_min_size_weights_aa``): that is what torchvision's tensor
  ``Resize`` lowers to at /root/reference/mars/components/VisualTextAlignmentModule.py:88-92
  (bicubic, a = -0.5) and /root/reference/utils/backbone_loader.py:182-186 (bilinear).
  The float32 / float64 mix of the C++ source is mirrored step by step.
* ``pil_table_u8``    -- Pillow's 8-bit path (``precompute_coeffs`` +
  ``normalize_coeffs_8bpc`` of libImaging/Resample.c, 22-bit fixed point) used by the
  PIL ``Resize(336, BICUBIC)`` of /root/reference/utils/backbone_loader.py:172-176.
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import Tuple

import numpy as np

f32 = np.float32
f64 = np.float64


def _aa_bilinear(x: np.float32) -> np.float32:
    x = f32(abs(x))
    return f32(1.0 - f64(x)) if x < 1.0 else f32(0.0)


def _aa_bicubic(x: np.float32) -> np.float32:
    a = f32(-0.5)
    x = f32(abs(x))
    if x < 1.0:
        return f32(f32(f32(f32(f32(a + f32(2)) * x) - f32(a + f32(3))) * x) * x + f32(1))
    if x < 2.0:
        return f32(f32(f32(f32(f32(f32(a * x) - f32(f32(5) * a)) * x) + f32(f32(8) * a)) * x) - f32(f32(4) * a))
    return f32(0.0)


@lru_cache(maxsize=None)
def torch_aa_table(in_size: int, out_size: int, mode: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(xmin int32 [out], xsize int32 [out], weights float32 [out, kmax])."""
    interp_size = {"bilinear": 2, "bicubic": 4}[mode]
    filt = {"bilinear": _aa_bilinear, "bicubic": _aa_bicubic}[mode]
    scale = f32(f32(in_size) / f32(out_size))
    support = f32(f64(interp_size * 0.5) * f64(scale)) if scale >= 1.0 else f32(interp_size * 0.5)
    kmax = int(math.ceil(float(support))) * 2 + 1
    invscale = f32(1.0 / f64(scale)) if scale >= 1.0 else f32(1.0)
    xmin = np.zeros(out_size, np.int32)
    xsize = np.zeros(out_size, np.int32)
    wts = np.zeros((out_size, kmax), np.float32)
    for i in range(out_size):
        center = f32(f64(scale) * (i + 0.5))
        lo = max(int(f64(center) - f64(support) + 0.5), 0)
        size = min(int(f64(center) + f64(support) + 0.5), in_size) - lo
        total = f32(0.0)
        w = np.zeros(kmax, np.float32)
        for j in range(size):
            arg = f32((f64(f32(f32(j + lo) - center)) + 0.5) * f64(invscale))
            w[j] = filt(arg)
            total = f32(total + w[j])
        if total != 0.0:
            for j in range(size):
                w[j] = f32(w[j] / total)
        xmin[i], xsize[i] = lo, size
        wts[i] = w
    return xmin, xsize, wts


def _pil_bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _pil_bilinear(x: float) -> float:
    if x < 0.0:
        x = -x
    return 1.0 - x if x < 1.0 else 0.0


@lru_cache(maxsize=None)
def pil_table_u8(in_size: int, out_size: int, filt: str = "bicubic") -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Pillow 8-bit coefficients, BICUBIC (support 2) or BILINEAR (support 1):
    (xmin int32, xsize int32, kk int32 [out, ksize])."""
    _pil_filter, base_support = {"bicubic": (_pil_bicubic, 2.0), "bilinear": (_pil_bilinear, 1.0)}[filt]
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = base_support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    xsize = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        n = hi - lo
        k = [_pil_filter((x + lo - center + 0.5) * ss) for x in range(n)]
        ww = 0.0
        for v in k:
            ww += v
        if ww != 0.0:
            k = [v / ww for v in k]
        for x, v in enumerate(k):
            kk[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
        xmin[xx], xsize[xx] = lo, n
    return xmin, xsize, kk


@lru_cache(maxsize=None)
def torch_bilinear_matrix(in_size: int, out_size: int) -> np.ndarray:
    """Dense [out, in] float64 matrix of F.interpolate(mode='bilinear', align_corners=False) along one
    axis (ATen area_pixel_compute_source_index: float32 source coordinate, clamped at 0)."""
    scale = f32(f32(in_size) / f32(out_size))
    M = np.zeros((out_size, in_size), np.float64)
    for o in range(out_size):
        src = f32(f32(scale * f32(o + 0.5)) - f32(0.5))
        if src < 0:
            src = f32(0.0)
        i0 = int(src)
        i1 = min(i0 + 1, in_size - 1)
        l1 = f32(src - f32(i0))
        l0 = f32(f32(1.0) - l1)
        M[o, i0] += float(l0)
        M[o, i1] += float(l1)
    return M


def matrix_to_table(M: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Banded [out, in] matrix -> (xmin, xsize, weights [out, kmax]) of the resample kernels."""
    out_size = M.shape[0]
    nz = [np.nonzero(M[o])[0] for o in range(out_size)]
    xmin = np.array([int(v[0]) if len(v) else 0 for v in nz], np.int32)
    xsize = np.array([int(v[-1]) - int(v[0]) + 1 if len(v) else 0 for v in nz], np.int32)
    kmax = max(1, int(xsize.max()))
    wts = np.zeros((out_size, kmax), np.float32)
    for o in range(out_size):
        wts[o, : xsize[o]] = M[o, xmin[o]: xmin[o] + xsize[o]]
    return xmin, xsize, wts


@lru_cache(maxsize=None)
def torch_bilinear_table(in_size: int, out_size: int):
    return matrix_to_table(torch_bilinear_matrix(in_size, out_size))


@lru_cache(maxsize=None)
def sam_postprocess_table(low_size: int, img_size: int, input_len: int, orig_len: int):
    """One axis of Sam.postprocess_masks (/root/reference/segment_anything/modeling/sam.py:134-162):
    bilinear low_size -> img_size, keep the first input_len samples, bilinear -> orig_len, composed
    into a single banded table."""
    up = torch_bilinear_matrix(low_size, img_size)[:input_len]
    return matrix_to_table(torch_bilinear_matrix(input_len, orig_len) @ up)


def band_plan(table_y, in_rows: int, width: int, lds_bytes: int = 64 * 1024, bands=(16, 8, 4, 2, 1)):
    """For the banded evaluation of a two-axis resampling of square planes of side ``in_rows`` (source rows into LDS, x into a
    second LDS array, then y out of that): the largest band of output rows whose source rows (``in_rows`` + ``width`` f32
    samples each) fit in ``lds_bytes``.  table_y = (lo, size, weights) of the y axis on
    the host.  Returns (band, max_rows) or None."""
    lo, size = np.asarray(table_y[0], np.int64), np.asarray(table_y[1], np.int64)
    if lo.size == 0 or int((lo + size).max()) > in_rows or int(lo.min()) < 0:
        return None
    for band in bands:
        rows = 1
        for a in range(0, lo.size, band):
            rows = max(rows, int((lo[a:a + band] + size[a:a + band]).max() - lo[a:a + band].min()))
        if rows * (width + in_rows) * 4 <= lds_bytes:
            return band, rows
    return None
