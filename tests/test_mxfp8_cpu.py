"""oracle/mxfp8.py pinned against things that are not the engine: hand-computed known answers for the E8M0
block scale under both rules (the engine's non-saturating rule and the OCP conversion recipe), hand-computed e4m3 code points,
and torch.float8_e4m3fn's rounding checked against a nearest-even encoder written from the format definition (OCP 8-bit
floating point spec: 1-4-3, bias 7, no infinity, one NaN, max 448, subnormals = multiples of 2^-9).  CPU only."""
import math

import numpy as np
import pytest
import torch

from oracle import mxfp8


def _decode(byte):
    s, e, m = byte >> 7, (byte >> 3) & 15, byte & 7
    if e == 15 and m == 7:
        return float("nan")
    v = (m / 8.0) * 2.0 ** -6 if e == 0 else (1.0 + m / 8.0) * 2.0 ** (e - 7)
    return -v if s else v


CODE = np.array([_decode(b) for b in range(128)])               # non-negative code points, index = byte (0x7f = NaN)


def _encode_rne(x):
    """nearest e4m3 code point of a finite |x| <= 448, ties to the even mantissa - straight from the definition"""
    a = abs(x)
    d = np.abs(CODE[:127] - a)
    lo = int(np.argmin(d))
    ties = [b for b in range(127) if d[b] == d[lo]]
    b = min(ties, key=lambda t: (t & 1, t)) if len(ties) > 1 else lo
    return b | (0x80 if math.copysign(1.0, x) < 0 else 0)


# amax -> (OCP recipe byte, engine byte); byte = exponent + 127
E8M0_KNOWN = [
    (1.0, 119, 119),            # floor(log2 1) - 8 = -8
    (1.75, 119, 119),           # 1.75 * 2^8 = 448 exactly: fits, no step up
    (1.875, 119, 120),          # 1.875 * 2^8 = 480 > 448: the recipe clamps, the engine steps the scale up
    (448.0, 127, 127),          # 448 = 1.75 * 2^8 -> exponent 0
    (449.0, 127, 128),
    (0.001, 109, 109),          # 0.001 = 1.024 * 2^-10
    (3.0, 120, 120),
    (255.9, 126, 127),          # 1.999 * 2^7
    (0.0, 0, 0),                # zero block: smallest scale
    (2.0 ** -130, 0, 0),        # denormal amax: biased exponent field 0 -> clamped
    (2.0 ** 127, 246, 246),
    (1.9 * 2.0 ** 127, 246, 247),
]

# value -> e4m3 byte at scale 1 (hand-computed from the 1-4-3 layout)
E4M3_KNOWN = [
    (1.0, 0x38), (-1.0, 0xB8), (0.5, 0x30), (1.125, 0x39), (448.0, 0x7E), (-448.0, 0xFE), (240.0, 0x77),
    (1.0625, 0x38),             # halfway 1.0 / 1.125: even mantissa (0)
    (1.1875, 0x3A),             # halfway 1.125 / 1.25: even mantissa (2)
    (2.0 ** -6, 0x08),          # smallest normal
    (2.0 ** -9, 0x01),          # smallest subnormal
    (2.0 ** -10, 0x00),         # halfway 0 / 2^-9: even (0)
    (3 * 2.0 ** -10, 0x02),     # halfway 2^-9 / 2^-8: even (2)
    (0.0, 0x00),
]


@pytest.mark.parametrize("amax,ocp,engine", E8M0_KNOWN)
def test_e8m0_block_scale_known_answers(amax, ocp, engine):
    a = torch.tensor([amax], dtype=torch.float32)
    assert int(mxfp8.block_exponent(a, "ocp")) == ocp
    assert int(mxfp8.block_exponent(a, "engine")) == engine
    assert int(mxfp8.block_exponent(a)) == engine                   # the engine's rule is the default


def test_e4m3_code_points_known_answers():
    for v, byte in E4M3_KNOWN:
        got = int(torch.tensor([v], dtype=torch.float32).to(torch.float8_e4m3fn).view(torch.uint8))
        assert got == byte, (v, hex(got), hex(byte))
        assert _encode_rne(v) == byte, (v, hex(_encode_rne(v)), hex(byte))
        if v != 0.0 and abs(v) >= 2.0 ** -9 and byte == _encode_rne(_decode(byte & 0x7F) * (1 if byte < 128 else -1)):
            assert _decode(byte) == pytest.approx(torch.tensor([byte], dtype=torch.uint8).view(torch.float8_e4m3fn).float().item())


def test_torch_e4m3_cast_is_round_to_nearest_even_over_the_code_table():
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(4000, generator=g) * s for s in (0.01, 1.0, 30.0, 150.0)]).clamp(-448, 448)
    mids = torch.tensor([(CODE[b] + CODE[b + 1]) / 2 for b in range(126)], dtype=torch.float32)      # every tie point
    x = torch.cat([x, mids, -mids, torch.tensor(CODE[:127], dtype=torch.float32)])
    got = x.to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    want = np.array([_encode_rne(float(v)) for v in x.numpy()], dtype=np.uint8)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, [(float(x[i]), hex(got[i]), hex(want[i])) for i in bad[:5]]


def test_quantize_round_trip_and_rules():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(64, 128, generator=g) * 3.0
    for rule in ("engine", "ocp"):
        q, s, deq = mxfp8.quantize(x, rule=rule)
        assert q.shape == x.shape and s.shape == (64, 4) and q.dtype == torch.uint8 and s.dtype == torch.uint8
        # dequantised = code point * 2^(s - 127), element by element
        code = torch.tensor([_decode(int(b)) for b in q.flatten()], dtype=torch.float64).reshape(x.shape)
        scale = torch.ldexp(torch.ones(64, 4, dtype=torch.float64), s.to(torch.int32) - 127).repeat_interleave(32, dim=1)
        assert torch.equal((code * scale).float(), deq)
        # quantising the dequantised tensor again changes nothing (idempotence)
        q2, s2, deq2 = mxfp8.quantize(deq, rule=rule)
        assert torch.equal(deq2, deq)
    # the engine's rule never saturates the block maximum; the recipe clamps maxima whose mantissa exceeds 1.75
    _, _, deq_e = mxfp8.quantize(x, rule="engine")
    _, _, deq_o = mxfp8.quantize(x, rule="ocp")
    amax = x.reshape(64, 4, 32).abs().amax(-1)
    rel_e = (deq_e.reshape(64, 4, 32).abs().amax(-1) - amax).abs() / amax
    assert float(rel_e.max()) <= 2.0 ** -4 + 1e-6                     # half an e4m3 ulp of a mantissa in [1, 2)
    clipped = (torch.frexp(amax)[0] * 2 > 1.75)
    assert clipped.any() and not torch.equal(deq_e, deq_o)
    assert torch.all(deq_o.reshape(64, 4, 32).abs().amax(-1)[clipped] < amax[clipped])


def test_weight_quantisation_blocks_run_along_input_channels():
    g = torch.Generator().manual_seed(2)
    w = torch.randn(8, 64, 1, 1, generator=g)
    deq = mxfp8.quantize_conv_weight(w)
    _, _, want = mxfp8.quantize(w.reshape(8, 64))
    assert torch.equal(deq.reshape(8, 64), want)
