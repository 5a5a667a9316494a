"""CPU restatement (numpy, integer arithmetic) of Pillow's bicubic resize of an 8-bit RGB image - what
``T.Resize((4h, 4w), BICUBIC)`` does to the PIL input in the reference (inference.py:66-73) - and of the
ToTensor / ToPILImage conversions around the sampler (inference.py:18-19, :73, :93).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The algorithm lives in a third-party dependency that is not under
/root/reference: Pillow (12.2.0 in this image), ``src/libImaging/Resample.c``: ``precompute_coeffs`` (double
coefficients, support 2.0 for bicubic with a = -0.5, window clipped to the image and renormalised),
``normalize_coeffs_8bpc`` (fixed point, PRECISION_BITS = 32 - 8 - 2 = 22, round half away from zero),
``ImagingResampleHorizontal_8bpc`` then ``ImagingResampleVertical_8bpc`` (accumulator starts at 1 << 21, result
``clip8(acc >> 22)``; the horizontal pass is rounded to 8 bits before the vertical pass reads it).
Pinned here against Pillow itself (tests/test_oracle_golden.py::test_pil_bicubic_restatement_is_bit_exact)."""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float, a: float = -0.5) -> float:
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """-> (bounds [out,2] int32 = (xmin, count), coeffs [out, ksize] int32 fixed point)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C cast: truncation toward zero
        xmin = max(xmin, 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)                                  # accumulated left to right, like the C loop
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray, axis: int) -> np.ndarray:
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for xx in range(bounds.shape[0]):
        xmin, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img_hwc: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 [H,W,C] -> uint8 [out_h,out_w,C], bit-exact with ``Image.resize((out_w, out_h), Image.BICUBIC)``."""
    h, w, _ = img_hwc.shape
    bw, kw = precompute_coeffs(w, out_w)
    bh, kh = precompute_coeffs(h, out_h)
    tmp = _pass(img_hwc, bw, kw, axis=1) if out_w != w else img_hwc     # horizontal first (ImagingResample)
    return _pass(tmp, bh, kh, axis=0) if out_h != h else tmp


def to_unit_chw(img_hwc_u8: np.ndarray) -> np.ndarray:
    """torchvision ToTensor on an RGB PIL image: uint8 HWC -> float32 CHW / 255 (inference.py:73)."""
    return (img_hwc_u8.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1)


def to_u8_hwc(img01_chw: np.ndarray) -> np.ndarray:
    """torchvision ToPILImage on a float tensor: ``pic.mul(255).byte()`` - truncation, no rounding, no clamp beyond the
    sampler's own clamp to [0,1] (inference.py:93)."""
    return (img01_chw.astype(np.float32) * np.float32(255.0)).astype(np.uint8).transpose(1, 2, 0)
