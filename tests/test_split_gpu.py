"""GPU tests of the split-operand precision (SRGD_PRECISION_F16X3): fp32 tensors, every convolution product as three f16 MFMAs on
(hi, lo) operand pairs (srgd_amd/csrc/conv3x3_split.hip, conv_igemm.hip: conv_igemm_split_kernel), through the kernel-level C ABI.

Three kinds of check:
  * against torch's float64 convolution (the quantity the reference's fp32 Block.proj approximates, model.py:246), with the
    tolerance of an fp32 convolution - the mode's claim;
  * against the CPU emulation of the documented arithmetic (oracle/split_emulation.py) to fp32 summation-order noise - the kernels
    compute what the header says, nothing looser;
  * exactness on small integers (every fragment map, swizzle and tap shift).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.split_emulation import split_conv2d
from tests.test_kernels_gpu import DEV, L, from_dev_nhwc, ptr, run_conv, stream, to_dev_nhwc, _report_k

pytestmark = pytest.mark.gpu

IMPL3 = {"f16": 6, "bf16": 8, "f16_256": 12}       # conv3x3_split (6: the engine's form, 512 threads; 12: the 256-thread form)
IMPLG = {"f16": 7, "bf16": 9}       # conv_igemm_split


def conv64(x, w, b, **kw):
    return F.conv2d(x.double(), w.double(), None if b is None else b.double(), **kw)


@pytest.mark.parametrize("kind", ["f16", "bf16", "f16_256"])
def test_conv3x3_split_integer_exact(kind):
    g = torch.Generator().manual_seed(12)
    x = torch.randint(-3, 4, (2, 64, 16, 64), generator=g).float()
    w = torch.randint(-2, 3, (128, 64, 3, 3), generator=g).float()
    b = torch.randint(-4, 5, (128,), generator=g).float()
    got, _ = run_conv(x, None, w, b, ks=3, stride=1, pad=1, kind=0, bf16=False, impl=IMPL3[kind])
    assert torch.equal(got, F.conv2d(x, w, b, padding=1))


@pytest.mark.parametrize("cfg", [(2, 32, 0, 128, 8, 32), (1, 64, 32, 256, 16, 64), (3, 128, 0, 128, 32, 32),
                                 (1, 256, 128, 1024, 8, 32), (1, 32, 0, 2048, 8, 32), (2, 32, 0, 512, 16, 32), (1, 32, 0, 2048, 16, 32, 8)],
                         ids=lambda s: "B%d_C%d+%d_Cout%d_%dx%d" % s[:6] + ("_g%d" % s[6] if len(s) > 6 else ""))
def test_conv3x3_split_borders_sources_stats(cfg):
    # the shapes of test_conv3x3_bf16_fast_path_borders_sources_stats: tiles touching every image border, two sources, every
    # channels-per-group class of the register-direct epilogue
    B, c0, c1, cout, H, W = cfg[:6]
    lib = L().lib()
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(B, c0, H, W, generator=g)
    x1 = torch.randn(B, c1, H, W, generator=g) if c1 else None
    w = torch.randn(cout, c0 + c1, 3, 3, generator=g) / (3 * (c0 + c1) ** 0.5)
    b = torch.randn(cout, generator=g)
    groups = cfg[6] if len(cfg) > 6 else (8 if cout <= 1024 else 0)
    xin = x0 if x1 is None else torch.cat((x0, x1), 1)
    want64 = conv64(xin, w, b, padding=1)
    scale = max(1.0, float(want64.abs().max()))
    err = {}
    for kind in ("f16", "bf16", "f16_256"):
        got, part, nslots = run_conv(x0, x1, w, b, ks=3, stride=1, pad=1, kind=0, bf16=False, groups=groups, impl=IMPL3[kind],
                                     want_slots=True)
        err[kind] = float((got.double() - want64).abs().max())
        emu = split_conv2d(xin, w, b, padding=1, kind=kind[:4].rstrip("_"))
        assert (got - emu).abs().max() <= 4e-6 * scale, (kind, float((got - emu).abs().max()))      # summation order only
        if kind.startswith("f16") and groups:
            assert torch.isfinite(part).all()
            s = part.sum(2).cpu().double()
            want_s1 = want64.reshape(B, groups, -1).sum(-1)
            want_s2 = (want64 ** 2).reshape(B, groups, -1).sum(-1)
            assert (s[..., 0] - want_s1).abs().max() <= 1e-4 * max(1.0, float(want_s1.abs().max()))
            assert (s[..., 1] - want_s2).abs().max() <= 1e-4 * float(want_s2.abs().max())
            gamma, beta = 1 + 0.2 * torch.randn(cout, generator=g), 0.3 * torch.randn(cout, generator=g)
            d = to_dev_nhwc(got, False)
            dg, db_ = gamma.to(DEV), beta.to(DEV)
            part = part.contiguous()
            L().check(lib.srgd_k_groupnorm_silu(ptr(d), ptr(d), ptr(None), ptr(part), B, H * W, cout, groups, ptr(dg),
                                                ptr(db_), ptr(None), nslots, 0, stream()), "groupnorm")
            y = F.silu(F.group_norm(want64.float(), groups, gamma, beta, eps=1e-5))
            assert (from_dev_nhwc(d) - y).abs().max() <= 2e-5 * max(1.0, float(y.abs().max()))
    e32 = float((F.conv2d(xin, w, b, padding=1).double() - want64).abs().max())
    _report_k(test="conv3x3_split", cfg=list(cfg), f16x3_max_abs=err["f16"], bf16x3_max_abs=err["bf16"], torch_fp32_max_abs=e32,
              ref_max=scale)
    assert err["f16"] <= 4e-6 * scale and err["f16_256"] <= 4e-6 * scale, err          # an fp32 convolution's own error on these shapes is ~1e-6 of the range
    assert err["bf16"] <= 3e-4 * scale, err
    assert err["f16"] < err["bf16"]


@pytest.mark.parametrize("cfg", [(2, 64, 128, 16, 32), (1, 128, 256, 16, 64), (3, 32, 128, 8, 32)], ids=lambda s: "B%d_C%d_Cout%d_%dx%d" % s)
def test_conv3x3_split_groupnorm_in_staging(cfg):
    # impl 11: conv(silu(a[b][c] * x + b[b][c])) with the activation applied to the fp32 halo pieces in registers; the zero padding
    # applies to the ACTIVATED tensor (silu(b) != 0 must not leak into the halo)
    B, cin, cout, H, W = cfg
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, cin, H, W, generator=g) * 2
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(cout, generator=g)
    ca = 1 + 0.3 * torch.randn(B, cin, generator=g)
    cb = 0.5 * torch.randn(B, cin, generator=g)
    coef = torch.stack([ca, cb]).contiguous().to(DEV)          # one allocation, 16-byte aligned rows
    act = F.silu(ca.double()[:, :, None, None] * x.double() + cb.double()[:, :, None, None])
    want64 = F.conv2d(act, w.double(), b.double(), padding=1)
    scale = max(1.0, float(want64.abs().max()))
    for impl in (11, 13):                            # the engine's form (512 threads) and the 256-thread form
        got, part, nslots = run_conv(x, None, w, b, ks=3, stride=1, pad=1, kind=0, bf16=False, groups=8, impl=impl, want_slots=True,
                                     gn_tail=(None, coef[0], coef[1]))
        err = float((got.double() - want64).abs().max())
        _report_k(test="conv3x3_split_gnin", cfg=list(cfg), impl=impl, max_abs=err, ref_max=scale)
        assert err <= 6e-6 * scale, (impl, err)
        s = part.sum(2).cpu().double()
        assert (s[..., 0] - want64.reshape(B, 8, -1).sum(-1)).abs().max() <= 1e-4 * max(1.0, float(want64.abs().sum(1).max()))


def test_conv3x3_split_small_and_large_magnitudes():
    # what the power-of-two weight scale is for: weights of 1e-3 (w_lo deep in f16's subnormal range without it), activations
    # spanning 1e-3 .. 1e2 in one tensor; and saturation instead of NaN beyond f16's range
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 64, 8, 32, generator=g) * torch.logspace(-3, 2, 64).view(1, 64, 1, 1)
    w = torch.randn(128, 64, 3, 3, generator=g) * 1e-3
    want64 = conv64(x, w, None, padding=1)
    got, _ = run_conv(x, None, w, None, ks=3, stride=1, pad=1, kind=0, bf16=False, impl=6)
    assert (got.double() - want64).abs().max() <= 4e-6 * float(want64.abs().max())
    x[0, 0, 0, 0] = 1e6
    got, _ = run_conv(x, None, w, None, ks=3, stride=1, pad=1, kind=0, bf16=False, impl=6)
    assert torch.isfinite(got).all()


@pytest.mark.parametrize("variant", ["plain", "residual", "two_sources", "pixel_shuffle", "unshuffle", "stats", "k_heavy", "ragged_m"])
def test_conv_igemm_split_variants(variant):
    g = torch.Generator().manual_seed(21)
    B, H, W = 2, 16, 24
    kw = dict(ks=1, stride=1, pad=0, kind=0)
    c0, c1, cout = 64, 0, 128
    residual, groups = None, 0
    if variant == "two_sources":
        c0, c1, cout = 96, 32, 192
    elif variant == "k_heavy":
        c0, cout, H, W = 1024, 256, 8, 8
    elif variant == "ragged_m":
        B, H, W, cout = 3, 10, 10, 96                   # 100 pixels per sample: a masked 128-row tile; Cout < CoutPad
    elif variant == "stats":
        groups, cout = 8, 256
    elif variant == "unshuffle":
        kw = dict(ks=2, stride=2, pad=0, kind=1)
    elif variant == "pixel_shuffle":
        kw = dict(ks=1, stride=1, pad=0, kind=2)
        cout = 256
    x0 = torch.randn(B, c0, H, W, generator=g)
    x1 = torch.randn(B, c1, H, W, generator=g) if c1 else None
    cin = c0 + c1
    if variant == "unshuffle":
        w = torch.randn(cout, 4 * cin, 1, 1, generator=g) / (4 * cin) ** 0.5
    else:
        w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g)
    xin = x0 if x1 is None else torch.cat((x0, x1), 1)
    if variant == "unshuffle":
        want64 = conv64(F.pixel_unshuffle(xin, 2), w, b)
    else:
        want64 = conv64(xin, w, b)
    if variant == "pixel_shuffle":
        want64 = F.pixel_shuffle(F.silu(want64), 2)
    if variant == "residual":
        residual = torch.randn(B, cout, H, W, generator=g)
        want64 = want64 + residual.double()
    scale = max(1.0, float(want64.abs().max()))
    for impl, tolr in ((7, 4e-6), (9, 3e-4)):
        got, part = run_conv(x0, x1, w, b, bf16=False, residual=residual, groups=groups, impl=impl, **kw)
        err = float((got.double() - want64).abs().max())
        _report_k(test="conv_igemm_split", variant=variant, impl=impl, max_abs=err, ref_max=scale)
        assert err <= tolr * scale, (variant, impl, err)
        if groups and impl == 7:
            s = part.sum(2).cpu().double()
            pre = conv64(xin, w, b)
            assert (s[..., 0] - pre.reshape(B, groups, -1).sum(-1)).abs().max() <= 1e-4 * float(pre.abs().sum(1).max())
            assert (s[..., 1] - (pre ** 2).reshape(B, groups, -1).sum(-1)).abs().max() <= 1e-4 * float((pre ** 2).reshape(B, groups, -1).sum(-1).max())


@pytest.mark.parametrize("variant", ["plain", "residual", "two_sources", "gn_tail", "pixel_shuffle", "unshuffle", "k_heavy", "one_step", "integers"])
def test_conv1x1_split_streaming_kernel(variant):
    # conv1x1_split.hip (fp32 pixel rows and split weights through a 3-deep LDS-DMA ring, split in registers, register-direct epilogue)
    # against float64 and against the generic split kernel (same arithmetic, different tiling)
    g = torch.Generator().manual_seed(31)
    B, H, W = 2, 16, 32                                     # 512 pixels per image: two 256-pixel tiles
    kw = dict(ks=1, stride=1, pad=0, kind=0)
    c0, c1, cout = 64, 0, 128
    residual = None
    if variant == "two_sources":
        c0, c1, cout = 96, 32, 256
    elif variant == "k_heavy":
        c0, cout, H, W = 1024, 256, 16, 16
    elif variant == "one_step":
        c0 = 32                                             # a single K-step: prologue and last step only
    elif variant == "unshuffle":
        kw = dict(ks=2, stride=2, pad=0, kind=1)
        H, W = 32, 32                                       # output 16 x 16 = 256 pixels per image
    elif variant == "pixel_shuffle":
        kw = dict(ks=1, stride=1, pad=0, kind=2)
        cout = 1024                                         # Cout / 4 = 256: two n-tiles per sub-pixel
    cin = c0 + c1
    if variant == "integers":
        x0 = torch.randint(-3, 4, (B, c0, H, W), generator=g).float()
        w = torch.randint(-2, 3, (cout, cin, 1, 1), generator=g).float()
        b = torch.randint(-4, 5, (cout,), generator=g).float()
    else:
        x0 = torch.randn(B, c0, H, W, generator=g)
        w = torch.randn(cout, (4 if variant == "unshuffle" else 1) * cin, 1, 1, generator=g) / ((4 if variant == "unshuffle" else 1) * cin) ** 0.5
        b = torch.randn(cout, generator=g)
    x1 = torch.randn(B, c1, H, W, generator=g) if c1 else None
    xin = x0 if x1 is None else torch.cat((x0, x1), 1)
    want64 = conv64(F.pixel_unshuffle(xin, 2), w, b) if variant == "unshuffle" else conv64(xin, w, b)
    if variant == "pixel_shuffle":
        want64 = F.pixel_shuffle(F.silu(want64), 2)
    if variant == "residual":
        residual = torch.randn(B, cout, H, W, generator=g)
        want64 = want64 + residual.double()
    gn_tail = None
    if variant == "gn_tail":            # out = conv(x) + silu(a * h + b): the ResnetBlock tail (model.py:250-259, :285) in the res_conv's epilogue
        hsrc = torch.randn(B, cout, H, W, generator=g) * 2
        ta, tb = 1 + 0.3 * torch.randn(B, cout, generator=g), 0.5 * torch.randn(B, cout, generator=g)
        gn_tail = (hsrc, ta, tb)
        want64 = want64 + F.silu(ta.double()[:, :, None, None] * hsrc.double() + tb.double()[:, :, None, None])
    got, _ = run_conv(x0, x1, w, b, bf16=False, residual=residual, impl=10, gn_tail=gn_tail, **kw)
    if variant == "integers":
        assert torch.equal(got.double(), want64)
        return
    scale = max(1.0, float(want64.abs().max()))
    err = float((got.double() - want64).abs().max())
    if variant == "gn_tail":            # (the generic split kernel has no tail epilogue: float64 is the only check)
        assert err <= 4e-6 * scale, (variant, err)
        return
    gen, _ = run_conv(x0, x1, w, b, bf16=False, residual=residual, impl=7, **kw)
    _report_k(test="conv1x1_split", variant=variant, max_abs=err, ref_max=scale, vs_generic_split=float((got - gen).abs().max()))
    assert err <= 4e-6 * scale, (variant, err)
    assert (got - gen).abs().max() <= 3e-6 * scale


@pytest.mark.parametrize("cfg", [("pre", 128, 384, 1, 512), ("pre", 256, 384, 2, 256), ("pre", 1024, 384, 1, 1024), ("post", 128, 128, 2, 512)],
                         ids=lambda c: "%s_C%d_Cout%d_B%d_N%d" % c)
def test_conv1x1_split_with_rmsnorm_folded_in(cfg):
    # the RMSNorms around the attention projections inside the projection's kernel (reference model.py:201-207, :300-303, :311-312)
    kind, cin, cout, B, N = cfg
    import ctypes as C
    lib = L().lib()
    g = torch.Generator().manual_seed(41)
    x = torch.randn(B, N, cin, generator=g) * torch.logspace(-1, 1, N).view(1, N, 1)      # pixel norms over two decades
    w = torch.randn(cout, cin, generator=g) / cin ** 0.5
    gain = 1 + 0.3 * torch.randn(cin if kind == "pre" else cout, generator=g)
    bias = None if kind == "pre" else torch.randn(cout, generator=g)
    res = None if kind == "pre" else torch.randn(B, N, cout, generator=g)

    def rms64(v, gg):
        v = v.double()
        return v / v.norm(dim=-1, keepdim=True).clamp_min(1e-12) * gg.double() * v.shape[-1] ** 0.5

    if kind == "pre":
        want = rms64(x, gain) @ w.double().t()
    else:
        want = rms64(x.double() @ w.double().t() + bias.double(), gain) + res.double()
    dx = x.contiguous().to(DEV)
    dres = None if res is None else res.contiguous().to(DEV)
    out = torch.empty(B, N, cout, device=DEV)
    wh, gh = w.contiguous(), gain.contiguous()
    bh = None if bias is None else bias.contiguous()
    L().check(lib.srgd_k_conv1x1_split_rms(ptr(dx), cin, B, N, ptr(wh), ptr(bh), cout, ptr(gh if kind == "pre" else None),
                                          ptr(gh if kind == "post" else None), ptr(dres), ptr(out), stream()), "conv1x1_split_rms")
    got = out.cpu().double()
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max())
    _report_k(test="conv1x1_split_rms", cfg=list(cfg), max_abs=err, ref_max=scale)
    assert err <= 6e-6 * scale, err
    # a zero pixel stays finite (F.normalize's eps)
    dx[0, 0].zero_()
    L().check(lib.srgd_k_conv1x1_split_rms(ptr(dx), cin, B, N, ptr(wh), ptr(bh), cout, ptr(gh if kind == "pre" else None),
                                          ptr(gh if kind == "post" else None), ptr(dres), ptr(out), stream()), "conv1x1_split_rms")
    assert torch.isfinite(out).all()


def test_conv_igemm_split_3x3_equals_the_halo_kernel_to_summation_order():
    # the same 3x3 layer through both split kernels (different tiling and K order, same arithmetic)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 16, 32, generator=g)
    w = torch.randn(128, 64, 3, 3, generator=g) / 24
    b = torch.randn(128, generator=g)
    a, _ = run_conv(x, None, w, b, ks=3, stride=1, pad=1, kind=0, bf16=False, impl=6)
    c, _ = run_conv(x, None, w, b, ks=3, stride=1, pad=1, kind=0, bf16=False, impl=7)
    assert (a - c).abs().max() <= 3e-6 * float(a.abs().max())


def test_split_kernels_reject_what_they_do_not_cover():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 48, 8, 32, generator=g)          # Cin % 32 != 0
    w = torch.randn(128, 48, 3, 3, generator=g)
    for impl in (6, 7):
        with pytest.raises(Exception):
            run_conv(x, None, w, None, ks=3, stride=1, pad=1, kind=0, bf16=False, impl=impl)
    with pytest.raises(Exception):                      # bf16 tensors
        run_conv(torch.randn(1, 64, 8, 32), None, torch.randn(128, 64, 3, 3), None, ks=3, stride=1, pad=1, kind=0, bf16=True, impl=6)


@pytest.mark.parametrize("cfg", [(2, 32, 0, 128, 8, 32), (1, 64, 32, 256, 16, 64), (3, 128, 0, 128, 32, 32), (1, 256, 128, 1024, 8, 32),
                                 (1, 96, 0, 128, 16, 32)],
                         ids=lambda s: "B%d_C%d+%d_Cout%d_%dx%d" % s[:6])
def test_conv3x3_mx2_prototype_matches_its_emulation(cfg):
    # impl 14 (conv3x3_mx2.hip, a prototype behind the kernel ABI): x_hi.w_hi on the f16 MFMA, both cross terms on MX-fp8 operands
    # (two taps per scaled MFMA) - against the emulation of exactly that arithmetic, and an order of magnitude from fp32
    from oracle.split_emulation import mixed_split_conv2d
    B, c0, c1, cout, H, W = cfg[:6]
    g = torch.Generator().manual_seed(13)
    x0 = torch.randn(B, c0, H, W, generator=g)
    x1 = torch.randn(B, c1, H, W, generator=g) if c1 else None
    w = torch.randn(cout, c0 + c1, 3, 3, generator=g) / (3 * (c0 + c1) ** 0.5)
    b = torch.randn(cout, generator=g)
    xin = x0 if x1 is None else torch.cat((x0, x1), 1)
    want64 = conv64(xin, w, b, padding=1)
    scale = max(1.0, float(want64.abs().max()))
    groups = 8 if cout <= 1024 else 0
    got, part, nslots = run_conv(x0, x1, w, b, ks=3, stride=1, pad=1, kind=0, bf16=False, groups=groups, impl=14, want_slots=True)
    emu = mixed_split_conv2d(xin, w, b, padding=1, mode="f16mx2")
    err_emu, err64 = float((got - emu).abs().max()), float((got.double() - want64).abs().max())
    e3 = float((split_conv2d(xin, w, b, padding=1, kind="f16").double() - want64).abs().max())
    _report_k(test="conv3x3_mx2", cfg=list(cfg), vs_emulation=err_emu, vs_fp64=err64, f16x3_vs_fp64=e3, ref_max=scale)
    assert err_emu <= 4e-6 * scale, (err_emu, err64)
    assert err64 <= 2e-4 * scale and err64 > e3
    if groups:
        s = part.sum(2).cpu().double()
        want_s1 = got.double().reshape(B, groups, -1).sum(-1)
        assert (s[..., 0] - want_s1).abs().max() <= 1e-4 * max(1.0, float(want_s1.abs().max()))


def test_conv3x3_mx2_prototype_integer_exact():
    # small integers are exact in every operand format involved (f16, e4m3 with power-of-two block scales): every fragment map, plane,
    # scale byte and tap pairing must be right for this to hold
    g = torch.Generator().manual_seed(12)
    x = torch.randint(-3, 4, (2, 64, 16, 64), generator=g).float()
    w = torch.randint(-2, 3, (128, 64, 3, 3), generator=g).float()
    b = torch.randint(-4, 5, (128,), generator=g).float()
    got, _ = run_conv(x, None, w, b, ks=3, stride=1, pad=1, kind=0, bf16=False, impl=14)
    assert torch.equal(got, F.conv2d(x, w, b, padding=1))


@pytest.mark.parametrize("cfg", [(2, 64, 128, 16, 32), (1, 128, 256, 16, 64), (3, 32, 128, 8, 32)], ids=lambda s: "B%d_C%d_Cout%d_%dx%d" % s)
def test_conv3x3_mx2_prototype_groupnorm_in_staging(cfg):
    # impl 15: conv3x3_mx2 with the producer's GroupNorm + SiLU applied to the fp32 halo pieces ahead of the split (coefficients through
    # an LDS slot); against the emulation of the same arithmetic on the activated tensor (v_exp / v_rcp: ~3e-7 relative before the split)
    from oracle.split_emulation import mixed_split_conv2d
    B, cin, cout, H, W = cfg
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, cin, H, W, generator=g) * 2
    w = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(cout, generator=g)
    ca = 1 + 0.3 * torch.randn(B, cin, generator=g)
    cb = 0.5 * torch.randn(B, cin, generator=g)
    coef = torch.stack([ca, cb]).contiguous().to(DEV)
    act = F.silu(ca[:, :, None, None] * x + cb[:, :, None, None])
    want64 = F.conv2d(act.double(), w.double(), b.double(), padding=1)
    scale = max(1.0, float(want64.abs().max()))
    got, part, nslots = run_conv(x, None, w, b, ks=3, stride=1, pad=1, kind=0, bf16=False, groups=8, impl=15, want_slots=True,
                                 gn_tail=(None, coef[0], coef[1]))
    emu = mixed_split_conv2d(act, w, b, padding=1, mode="f16mx2")
    err_emu, err64 = float((got - emu).abs().max()), float((got.double() - want64).abs().max())
    _report_k(test="conv3x3_mx2_gnin", cfg=list(cfg), vs_emulation=err_emu, vs_fp64=err64, ref_max=scale)
    # an activation that differs by an ulp can land on the other side of an e4m3 rounding boundary of a cross-term operand: 2^-15 of a product
    assert err_emu <= 2e-5 * scale and err64 <= 2e-4 * scale, (err_emu, err64)
    s = part.sum(2).cpu().double()
    assert (s[..., 0] - got.double().reshape(B, 8, -1).sum(-1)).abs().max() <= 1e-4 * max(1.0, float(want64.abs().sum(1).max()))
