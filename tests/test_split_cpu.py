"""CPU tests of the split-operand arithmetic's emulation (oracle/split_emulation.py - the checker of the f16x3 kernels) and of the
full-length configs[1] fixture's integrity.  The kernels themselves are tested on the GPU (tests/test_split_gpu.py)."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle.split_emulation import halves, split_conv2d, weight_scale
from tests.golden import cases as C

G = os.path.join(os.path.dirname(__file__), "golden")


def test_halves_carry_22_bits_and_are_exact_on_small_integers():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096, generator=g) * torch.logspace(-2, 2, 4096)
    hi, lo = halves(x, "f16")
    err = ((hi.double() + lo.double()) - x.double()).abs()
    # 11 + 11 significand bits while the lo half is a normal f16 number (|x| >= 2^-3); below that the lo half is subnormal and the
    # error is absolute: half of f16's subnormal spacing 2^-24 (why the WEIGHTS are pre-scaled; activations are O(1))
    assert (err <= torch.maximum(x.double().abs() * 2.0 ** -21.9, torch.tensor(2.0 ** -24, dtype=torch.float64))).all()
    big_enough = x.abs() >= 0.125
    assert (err[big_enough] / x.double().abs()[big_enough]).max() <= 2.0 ** -21.9
    hb, lb = halves(x, "bf16")
    relb = ((hb.double() + lb.double()) - x.double()).abs() / x.double().abs()
    assert relb.max() <= 2.0 ** -15 and relb.max() > (err[big_enough] / x.double().abs()[big_enough]).max()
    i = torch.arange(-2048, 2049).float()
    hi, lo = halves(i, "f16")
    assert torch.equal(hi, i) and not lo.any()
    big = torch.tensor([1e6, -1e6, 65504.0])                        # beyond f16's range: saturates, stays finite
    hi, lo = halves(big, "f16")
    assert torch.isfinite(hi).all() and torch.isfinite(lo).all() and hi[0] == 65504.0


def test_weight_scale_is_a_power_of_two_that_puts_the_maximum_at_2_10():
    for m in (1e-3, 0.03, 0.9, 1.0, 7.5, 1500.0):
        w = torch.tensor([m, -m / 3, m / 1000])
        s = weight_scale(w, "f16")
        assert np.log2(s) == round(np.log2(s)) and 1024.0 <= m * s < 2048.0
    assert weight_scale(torch.zeros(4), "f16") == 1.0 and weight_scale(torch.ones(4), "bf16") == 1.0


def test_split_conv_is_as_close_to_float64_as_an_fp32_convolution():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 64, 16, 16, generator=g)
    w = torch.randn(32, 64, 3, 3, generator=g) / 24
    b = torch.randn(32, generator=g)
    want = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    e32 = float((F.conv2d(x, w, b, padding=1).double() - want).abs().max())
    e16 = float((split_conv2d(x, w, b, padding=1, kind="f16").double() - want).abs().max())
    ebf = float((split_conv2d(x, w, b, padding=1, kind="bf16").double() - want).abs().max())
    ens = float((split_conv2d(x, w, b, padding=1, kind="f16", scale=False).double() - want).abs().max())
    assert e16 <= 3 * max(e32, 1e-6) and e16 < ens < ebf            # f16 halves + weight scale ~ fp32; without the scale worse; bf16 halves worst
    xi = torch.randint(-3, 4, (1, 32, 8, 8), generator=g).float()
    wi = torch.randint(-2, 3, (16, 32, 3, 3), generator=g).float()
    assert torch.equal(split_conv2d(xi, wi, None, padding=1, kind="f16"), F.conv2d(xi, wi, None, padding=1))


def test_full_length_config2_fixture_is_intact():
    case = C.FULL_CASES[0]
    z = np.load(os.path.join(G, f"sample_{case['name']}.npz"))
    img = z["image_u16"].astype(np.float64) / 65535.0
    assert img.shape == (1, 3, 1024, 1024)
    assert abs(img.sum() - float(z["checksum"])) < img.size * 7.7e-6                 # uint16 rounding only
    n = case["steps"]
    assert z["xt_abs"].shape == (n,) and z["x0_abs"].shape == (n,) and list(z["trace_steps"]) == list(C.FULL_TRACE_STEPS)
    for i in C.FULL_TRACE_STEPS:
        assert z[f"xt_{i}"].shape == (3, 160, 160) and z[f"x0_{i}"].shape == (3, 160, 160)
        assert np.abs(z[f"x0_{i}"]).max() <= 1.0 + 1e-6                              # clip_sample_denoised
    # the trajectory does what a DDPM run does: the noise level of x_t falls, x_start settles
    assert z["xt_abs"][0] > z["xt_abs"][-1] and abs(z["x0_abs"][-1] - z["x0_abs"][-2]) / z["x0_abs"][-1] < 1e-2
    cond = C.sampler_condition(case)
    assert abs(cond.double().sum().item() - float(z["cond_sum"])) < 1e-6             # the seeded input is reproducible here


def test_cheaper_relatives_of_f16x3_rank_as_the_design_note_says():
    # dropping a cross term costs 2^-12 of a product, cross terms on MX-fp8 operands 2^-15, all three on f16 2^-22
    from oracle.split_emulation import mixed_split_conv2d
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 64, 12, 12, generator=g)
    w = torch.randn(32, 64, 3, 3, generator=g) / 24.0
    want = F.conv2d(x.double(), w.double(), padding=1)
    scale = float(want.abs().max())
    err = lambda y: float((y.double() - want).abs().max()) / scale
    e3 = err(split_conv2d(x, w, padding=1, kind="f16"))
    emx = err(mixed_split_conv2d(x, w, padding=1, mode="f16mx2"))
    ew1 = err(mixed_split_conv2d(x, w, padding=1, mode="f16x2_w1"))
    ex1 = err(mixed_split_conv2d(x, w, padding=1, mode="f16x2_x1"))
    assert e3 < 2e-6 and e3 < emx / 8 and emx < min(ew1, ex1) / 4 and max(ew1, ex1) < 2e-3, (e3, emx, ew1, ex1)
