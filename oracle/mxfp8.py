"""CPU emulation of OCP Microscaling fp8 (MX-fp8, e4m3 elements + one E8M0 scale per 32 elements) - TEST INFRASTRUCTURE ONLY.

The reference has no fp8 path: BASELINE configs[4] asks for an fp8 compute mode of THIS engine with a parity-vs-bf16
check.  This file restates the published format (OCP Microscaling Formats v1.0: e4m3 elements = round-to-nearest-even of
x / 2^exp saturated to +-448, one E8M0 exponent per 32 elements) with the engine's scale rule (block_exponent: the recipe's
floor(log2(max|x|)) - 8, one step up when the block maximum would saturate) so the GPU quantisation kernel (quant_mxfp8.hip), the host weight packing (conv3x3_mxfp8.hip) and the convolution on
v_mfma_scale_f32_16x16x128_f8f6f4 can be checked bit for bit / to fp32 summation order.  Imported by tests only.
"""
from __future__ import annotations

import torch


def block_exponent(amax: torch.Tensor, rule: str = "engine") -> torch.Tensor:
    """E8M0 byte of a block with maximum magnitude `amax` (float32 tensor): floor(log2 amax) - 8 + 127, plus one when the
    maximum's mantissa exceeds 1.75 (it would land above 448 = the e4m3 maximum after scaling): the smallest power-of-two scale
    that does not saturate the block maximum.  (The OCP conversion recipe stops at floor(log2 amax) - 8 and clamps; the engine's
    rule keeps the format and trades one bit of resolution in those blocks for an unclipped maximum: +2.5 dB on configs[4].)
    Zero/denormal blocks get byte 0.  Read from the float's bit fields exactly as the kernels do.
    rule="ocp": the published conversion recipe itself (no step up; maxima above 1.75 * 2^k saturate at 448) - kept for the
    known-answer test and the scale-rule study (tools/mx_scale_rules.py), not used by the engine.
    Pinned by tests/test_mxfp8_cpu.py: hand-computed E8M0 bytes for both rules, and the element rounding (torch.float8_e4m3fn)
    against a from-the-definition nearest-even encoder over the 256 code points."""
    if rule not in ("engine", "ocp"):
        raise ValueError(rule)
    bits = amax.contiguous().view(torch.int32)
    bexp = (bits >> 23) & 0xFF
    over = ((bits & 0x7FFFFF) > 0x600000).to(torch.int32) if rule == "engine" else torch.zeros_like(bexp)
    return (bexp - 8 + over).clamp(min=0, max=254).to(torch.uint8)


def quantize(x: torch.Tensor, block: int = 32, rule: str = "engine"):
    """x: float32 [..., C] with C % 32 == 0 -> (q uint8 [..., C] e4m3 bit patterns, s uint8 [..., C/32], dequantised float32)."""
    x = x.float()
    shape = x.shape
    xb = x.reshape(-1, shape[-1] // block, block)
    sb = block_exponent(xb.abs().amax(dim=-1), rule)
    inv = torch.ldexp(torch.ones_like(sb, dtype=torch.float32), (127 - sb.int()))             # 2^(127 - byte)
    scaled = (xb * inv[..., None]).clamp(-448.0, 448.0)
    q8 = scaled.to(torch.float8_e4m3fn)                                                         # RNE (inputs already saturated)
    deq = q8.float() * torch.ldexp(torch.ones_like(inv), (sb.int() - 127))[..., None]
    return (q8.view(torch.uint8).reshape(shape), sb.reshape(*shape[:-1], shape[-1] // block), deq.reshape(shape))


def quantize_conv_weight(w: torch.Tensor):
    """OIHW float32 -> dequantised float32 with one scale per (output channel, tap, 32 input channels)."""
    o, i, kh, kw = w.shape
    wt = w.permute(0, 2, 3, 1).contiguous()            # [O, kh, kw, I]: blocks of 32 run along the input channels
    _, _, deq = quantize(wt)
    return deq.permute(0, 3, 1, 2).contiguous()
