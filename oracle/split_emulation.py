"""CPU emulation of the split-operand convolution arithmetic (TEST INFRASTRUCTURE, like everything under oracle/).

The product's f16x3 precision (srgd_amd/csrc/conv3x3_split.hip, conv_igemm.hip) evaluates a convolution on fp32 tensors as

    x = x_hi + x_lo,  w * s = w_hi + w_lo          (hi = 16-bit round-to-nearest of the value, lo = round(value - hi))
    conv(x, w) ~= [conv(x_hi, w_hi) + conv(x_lo, w_hi) + conv(x_hi, w_lo)] / s             (fp32 accumulation)

with f16 halves and s = the power of two that puts max|w| into [2^10, 2^11) (bf16 halves: s = 1).  A product of two 16-bit values
is exact in fp32, so three fp32 convolutions on the rounded operands reproduce the three MFMAs up to summation order.  There is
no reference code for this (the reference computes plain fp32, model.py:246): the emulation pins the KERNELS to the documented
arithmetic, and tools/split_numerics_study.py uses it to price the precision against the reference's fixtures.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def halves(x: torch.Tensor, kind: str):
    dt = torch.float16 if kind == "f16" else torch.bfloat16
    if kind == "f16":
        x = x.clamp(-65504.0, 65504.0)          # the kernels saturate instead of producing inf - inf
    hi = x.to(dt).float()
    lo = (x - hi).to(dt).float()
    return hi, lo


def weight_scale(w: torch.Tensor, kind: str) -> float:
    if kind != "f16":
        return 1.0
    m = float(w.abs().max())
    if not (m > 0.0) or not math.isfinite(m):
        return 1.0
    return 2.0 ** (10 - math.floor(math.log2(m)))


def split_conv2d(x, w, b=None, stride=1, padding=0, kind="f16", terms=3, scale=True):
    s = weight_scale(w, kind) if scale else 1.0
    wh, wl = halves(w * s, kind)
    xh, xl = halves(x, kind)
    y = F.conv2d(xl, wh, None, stride=stride, padding=padding) + F.conv2d(xh, wl, None, stride=stride, padding=padding)
    if terms == 4:
        y = y + F.conv2d(xl, wl, None, stride=stride, padding=padding)
    y = (y + F.conv2d(xh, wh, None, stride=stride, padding=padding)) * (1.0 / s)
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y


def mixed_split_conv2d(x, w, b=None, stride=1, padding=0, mode="f16x2_w1"):
    """Cheaper relatives of f16x3, for pricing only (no kernel computes these):
    f16x2_w1  x = x_hi + x_lo, w rounded to ONE f16 value:   conv(x_hi, w_hi) + conv(x_lo, w_hi)                       (2 MFMAs)
    f16x2_x1  x rounded to ONE f16 value, w = w_hi + w_lo:   conv(x_hi, w_hi) + conv(x_hi, w_lo)                       (2 MFMAs)
    f16mx2    f16x3 with the two cross terms on MX-fp8 operands (e4m3 + E8M0 per 32 channels; the scaled MFMA runs at twice the
              f16 rate, so the three products cost two): conv(x_hi, w_hi) + conv(Q(x_lo), Q(w_hi)) + conv(Q(x_hi), Q(w_lo))"""
    from oracle import mxfp8
    s = weight_scale(w, "f16")
    wh, wl = halves(w * s, "f16")
    xh, xl = halves(x, "f16")
    if mode == "f16mx2":
        xl = x.clamp(-65504.0, 65504.0) - xh     # the prototype kernel (conv3x3_mx2.hip) quantises x - x_hi itself, not its f16 rounding
    y = F.conv2d(xh, wh, None, stride=stride, padding=padding)
    if mode == "f16x2_w1":
        y = y + F.conv2d(xl, wh, None, stride=stride, padding=padding)
    elif mode == "f16x2_x1":
        y = y + F.conv2d(xh, wl, None, stride=stride, padding=padding)
    elif mode == "f16mx2":
        qa = lambda t: mxfp8.quantize(t.permute(0, 2, 3, 1).contiguous())[2].permute(0, 3, 1, 2).contiguous()
        y = y + F.conv2d(qa(xl), mxfp8.quantize_conv_weight(wh), None, stride=stride, padding=padding) \
              + F.conv2d(qa(xh), mxfp8.quantize_conv_weight(wl), None, stride=stride, padding=padding)
    else:
        raise ValueError(mode)
    y = y * (1.0 / s)
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y
