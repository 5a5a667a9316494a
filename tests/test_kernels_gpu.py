"""GPU parity tests, one kernel family at a time, through the kernel-level C ABI
(include/srgd_hip_kernels.h) against the CPU oracle / plain torch fp32 ops on the same seeded inputs.

fp32 mode: exact-fp32 MFMA, tolerance = summation-order noise.  bf16 mode: inputs are rounded to
bf16 first and the oracle runs on the rounded values, so the tolerance only covers bf16 output
rounding + accumulation order (2^-8 relative).
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import srgd_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda"


def L():
    from srgd_amd import _lib
    return _lib


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(0) if t is None else C.c_void_p(t.data_ptr())


def to_dev_nhwc(x_nchw, bf16):
    """NCHW fp32 (CPU) -> NHWC contiguous on the GPU in the activation type."""
    t = x_nchw.permute(0, 2, 3, 1).contiguous().to(DEV)
    return t.to(torch.bfloat16) if bf16 else t


def from_dev_nhwc(t):
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def rnd(x, bf16):
    return x.to(torch.bfloat16).float() if bf16 else x


def _report_k(**kw):
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.jsonl")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "a") as f:
        f.write(json.dumps(kw) + "\n")


def tol(bf16, ref, k=1.0):
    scale = max(1.0, float(ref.abs().max()))
    return (1.2e-2 if bf16 else 2e-5) * scale * k


def run_conv(x0, x1, w, b, *, ks, stride, pad, kind, bf16, residual=None, groups=0, impl=0, want_slots=False,
             gn_tail=None):
    lib = L().lib()
    B, C0, H, W = x0.shape
    C1 = 0 if x1 is None else x1.shape[1]
    cout = w.shape[0]
    d0 = to_dev_nhwc(x0, bf16)
    d1 = None if x1 is None else to_dev_nhwc(x1, bf16)
    ho = (H + 2 * pad - ks) // stride + 1
    wo = (W + 2 * pad - ks) // stride + 1
    if kind == 2:
        out = torch.empty(B, 2 * ho, 2 * wo, cout // 4, device=DEV, dtype=d0.dtype)
    else:
        out = torch.empty(B, ho, wo, cout, device=DEV, dtype=d0.dtype)
    dres = None if residual is None else to_dev_nhwc(residual, bf16)
    part = None
    if groups:
        # capacity: the 3x3 fast paths write one slot per contributing wave (4 per 256-pixel patch, 8 per 128-channel tile of a
        # group that spans whole tiles); the call reports the count it used
        part = torch.full((B, groups, (ho * wo // 32) * max(1, cout // groups // 64), 2), float("nan"), device=DEV)
    wh = w.contiguous().float()
    bh = None if b is None else b.contiguous().float()
    slots = C.c_int()
    tail_src = tail_a = tail_b = None
    if gn_tail is not None:                 # (h [B,C,H,W], a [B,C], b [B,C]): out = silu(a*h + b) + conv(x); h None: impl 5 / 11 (GNIN)
        tail_src = None if gn_tail[0] is None else to_dev_nhwc(gn_tail[0], bf16)
        tail_a = gn_tail[1] if gn_tail[1].is_cuda else gn_tail[1].float().contiguous().to(DEV)
        tail_b = gn_tail[2] if gn_tail[2].is_cuda else gn_tail[2].float().contiguous().to(DEV)
    L().check(lib.srgd_k_conv2d_timed(ptr(d0), ptr(d1), C0, C1, B, H, W, ks, stride, pad, kind, ptr(wh), ptr(bh), cout,
                                      ptr(out), ptr(dres), ptr(part), groups, int(bf16), impl, 0, None,
                                      C.byref(slots), ptr(tail_src), ptr(tail_a), ptr(tail_b), stream()),
              "srgd_k_conv2d_timed")
    torch.cuda.synchronize()
    if groups:
        part = part.reshape(-1)[:B * groups * slots.value * 2].reshape(B, groups, slots.value, 2)
    if want_slots:
        return from_dev_nhwc(out), part, slots.value
    return from_dev_nhwc(out), part


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
def test_mfma_layout_exact_small_integers(bf16):
    # asymmetric integer data: every product and sum is exact in bf16/fp32, so the result must be
    # bit-identical - catches any row/col/k-pairing mistake in the MFMA fragment maps.
    g = torch.Generator().manual_seed(0)
    x = torch.randint(-3, 4, (1, 32, 16, 16), generator=g).float()
    w = torch.randint(-2, 3, (48, 32, 3, 3), generator=g).float()
    b = torch.randint(-4, 5, (48,), generator=g).float()
    got, _ = run_conv(x, None, w, b, ks=3, stride=1, pad=1, kind=0, bf16=bf16)
    want = F.conv2d(x, w, b, padding=1)
    if bf16:
        want = want.to(torch.bfloat16).float()       # outputs up to ~|2600|: bf16 output rounding only
    assert torch.equal(got, want)


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 16, 16, 32, 16), (1, 128, 128, 16, 32), (3, 64, 32, 48, 24), (1, 256, 128, 32, 32), (1, 32, 40, 10, 10)],
                         ids=lambda s: "B%d_Cin%d_Cout%d_%dx%d" % s)
def test_conv3x3_bias(bf16, shape):
    B, cin, cout, H, W = shape
    g = torch.Generator().manual_seed(1)
    x = rnd(torch.randn(B, cin, H, W, generator=g), bf16)
    w = rnd(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5), bf16)
    b = torch.randn(cout, generator=g)
    got, _ = run_conv(x, None, w, b, ks=3, stride=1, pad=1, kind=0, bf16=bf16)
    want = F.conv2d(x, w, b, padding=1)
    assert (got - want).abs().max() <= tol(bf16, want)


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
def test_conv_two_sources_equals_concat_and_residual(bf16):
    g = torch.Generator().manual_seed(2)
    x0 = rnd(torch.randn(2, 64, 16, 32, generator=g), bf16)
    x1 = rnd(torch.randn(2, 32, 16, 32, generator=g), bf16)
    w = rnd(torch.randn(48, 96, 3, 3, generator=g) / 30, bf16)
    b = torch.randn(48, generator=g)
    res = rnd(torch.randn(2, 48, 16, 32, generator=g), bf16)
    got, _ = run_conv(x0, x1, w, b, ks=3, stride=1, pad=1, kind=0, bf16=bf16, residual=res)
    want = F.conv2d(torch.cat((x0, x1), 1), w, b, padding=1) + res
    assert (got - want).abs().max() <= tol(bf16, want)
    # 1x1 over a concat (res_conv of the up path), no bias
    w1 = rnd(torch.randn(160, 96, 1, 1, generator=g) / 10, bf16)
    got, _ = run_conv(x0, x1, w1, None, ks=1, stride=1, pad=0, kind=0, bf16=bf16)
    want = F.conv2d(torch.cat((x0, x1), 1), w1)
    assert (got - want).abs().max() <= tol(bf16, want)


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
def test_downsample_is_unshuffle_plus_1x1(bf16):
    g = torch.Generator().manual_seed(3)
    x = rnd(torch.randn(2, 32, 32, 32, generator=g), bf16)
    w = rnd(torch.randn(64, 128, 1, 1, generator=g) / 11, bf16)
    b = torch.randn(64, generator=g)
    got, _ = run_conv(x, None, w, b, ks=2, stride=2, pad=0, kind=1, bf16=bf16)
    want = O.space_to_depth_conv({"d.1.weight": w, "d.1.bias": b}, "d", x)
    assert got.shape == want.shape
    assert (got - want).abs().max() <= tol(bf16, want)


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
def test_pixel_shuffle_upsample(bf16):
    g = torch.Generator().manual_seed(4)
    x = rnd(torch.randn(2, 64, 16, 16, generator=g), bf16)
    w = rnd(torch.randn(128, 64, 1, 1, generator=g) / 8, bf16)
    b = torch.randn(128, generator=g)
    got, _ = run_conv(x, None, w, b, ks=1, stride=1, pad=0, kind=2, bf16=bf16)
    want = O.pixel_shuffle_up({"u.net.0.weight": w, "u.net.0.bias": b}, "u", x)
    assert got.shape == want.shape
    assert (got - want).abs().max() <= tol(bf16, want)


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
@pytest.mark.parametrize("cfg", [(2, 32, 64, 16, 16, 8), (1, 16, 128, 32, 32, 8), (2, 64, 16, 16, 8, 8), (3, 32, 96, 16, 8, 6)],
                         ids=lambda s: "B%d_Cin%d_Cout%d_%dx%d_g%d" % s)
def test_conv_groupnorm_scale_shift_silu_residual(bf16, cfg):
    # (Cout = 96, 6 groups: 12 / 24 vectors per pixel - the GroupNorm-apply kernel's non-power-of-two channel-offset path)
    # Block.forward (model.py:250-259) + the ResnetBlock residual (:285) on top of the conv's fused statistics
    B, cin, cout, H, W, ngroups = cfg
    lib = L().lib()
    g = torch.Generator().manual_seed(5)
    x = rnd(torch.randn(B, cin, H, W, generator=g) * 2 + 0.5, bf16)
    w = rnd(torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5), bf16)
    b = torch.randn(cout, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(cout, generator=g), 0.3 * torch.randn(cout, generator=g)
    ss = 0.5 * torch.randn(B, 2 * cout, generator=g)
    res = rnd(torch.randn(B, cout, H, W, generator=g), bf16)
    conv_out, part, nslots = run_conv(x, None, w, b, ks=3, stride=1, pad=1, kind=0, bf16=bf16, groups=ngroups, want_slots=True)
    assert torch.isfinite(part).all(), "conv epilogue did not fill every GroupNorm partial slot"
    ref_conv = F.conv2d(x, w, b, padding=1)
    # statistics must come from the fp32 accumulators
    s1 = part[..., 0].sum(-1).cpu()
    want_s1 = ref_conv.reshape(B, ngroups, -1).sum(-1)
    assert (s1 - want_s1).abs().max() <= 1e-3 * max(1.0, float(want_s1.abs().max()))
    d = to_dev_nhwc(conv_out, bf16)
    dres = to_dev_nhwc(res, bf16)
    dg, db_, dss = gamma.to(DEV), beta.to(DEV), ss.to(DEV)      # keep the device copies alive across the call
    part = part.contiguous()
    L().check(lib.srgd_k_groupnorm_silu(ptr(d), ptr(d), ptr(dres), ptr(part), B, H * W, cout, ngroups, ptr(dg), ptr(db_),
                                        ptr(dss), nslots, int(bf16), stream()), "groupnorm")
    got = from_dev_nhwc(d)
    y = F.group_norm(conv_out if bf16 else ref_conv, ngroups, gamma, beta, eps=1e-5)
    y = y * (ss[:, :cout, None, None] + 1) + ss[:, cout:, None, None]
    want = F.silu(y) + res
    assert (got - want).abs().max() <= tol(bf16, want, k=2.0)


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
@pytest.mark.parametrize("C", [16, 128, 1024])
def test_rmsnorm(bf16, C):
    lib = L().lib()
    g = torch.Generator().manual_seed(6)
    x = rnd(torch.randn(2, C, 8, 16, generator=g) * 3, bf16)
    x[0, :, 0, 0] = 0.0                       # eps path: zero vector stays zero
    gain = 1 + 0.1 * torch.randn(1, C, 1, 1, generator=g)
    res = rnd(torch.randn(2, C, 8, 16, generator=g), bf16)
    d, dres = to_dev_nhwc(x, bf16), to_dev_nhwc(res, bf16)
    out = torch.empty_like(d)
    dgain = gain.reshape(-1).to(DEV)
    L().check(lib.srgd_k_rmsnorm(ptr(d), ptr(out), ptr(dres), ptr(dgain), 2 * 8 * 16, C, int(bf16),
                                 stream()), "rmsnorm")
    got = from_dev_nhwc(out)
    want = O.rms_norm(x, gain) + res
    assert (got - want).abs().max() <= tol(bf16, want)
    # in place, no residual
    L().check(lib.srgd_k_rmsnorm(ptr(d), ptr(d), ptr(None), ptr(dgain), 2 * 8 * 16, C, int(bf16),
                                 stream()), "rmsnorm")
    torch.cuda.synchronize()
    assert (from_dev_nhwc(d) - O.rms_norm(x, gain)).abs().max() <= tol(bf16, want)


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
@pytest.mark.parametrize("hw", [(16, 16), (64, 64), (24, 40)], ids=lambda s: "%dx%d" % s)
def test_linear_attention_core(bf16, hw):
    lib = L().lib()
    H, W = hw
    g = torch.Generator().manual_seed(7)
    qkv = torch.randn(2, 384, H, W, generator=g) * 2
    qkv[0, 128 + 5, 3, 3] = 9.0               # a spike in k: exercises the cross-chunk max merge
    qkv = rnd(qkv, bf16)
    d = to_dev_nhwc(qkv, bf16)
    out = torch.empty(2, H, W, 128, device=DEV, dtype=d.dtype)
    L().check(lib.srgd_k_linear_attention(ptr(d), ptr(out), 2, H * W, 4, int(bf16), stream()), "linattn")
    got = from_dev_nhwc(out)
    want = O.linear_attention_core(qkv, 4, 32)
    assert (got - want).abs().max() <= tol(bf16, want)


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
@pytest.mark.parametrize("hw", [(16, 16), (32, 32), (8, 24)], ids=lambda s: "%dx%d" % s)
def test_full_attention_core(bf16, hw):
    lib = L().lib()
    H, W = hw
    g = torch.Generator().manual_seed(8)
    qkv = torch.randn(2, 384, H, W, generator=g) * 1.5
    qkv[1, :32, 2, 2] *= 6.0                  # one sharp query row: online-softmax rescale path
    qkv = rnd(qkv, bf16)
    d = to_dev_nhwc(qkv, bf16)
    out = torch.empty(2, H, W, 128, device=DEV, dtype=d.dtype)
    L().check(lib.srgd_k_full_attention(ptr(d), ptr(out), 2, H * W, 4, int(bf16), stream()), "fullattn")
    torch.cuda.synchronize()
    got = from_dev_nhwc(out)
    want = O.full_attention_core(qkv, 4, 32)
    assert (got - want).abs().max() <= tol(bf16, want)


@pytest.mark.parametrize("cfg", [(2, 32, 0, 128, 8, 32), (1, 64, 32, 256, 16, 64), (3, 128, 0, 128, 32, 32),
                                 (1, 256, 128, 1024, 8, 32), (1, 32, 0, 2048, 8, 32), (2, 32, 0, 512, 16, 32), (1, 32, 0, 2048, 16, 32, 8)],
                         ids=lambda s: "B%d_C%d+%d_Cout%d_%dx%d" % s[:6] + ("_g%d" % s[6] if len(s) > 6 else ""))
def test_conv3x3_bf16_fast_path_borders_sources_stats(cfg):
    # conv3x3_bf16.hip (halo patch in LDS, LDS-DMA staging, zero fill by the buffer range check) against the
    # oracle's conv2d on bf16-rounded operands, incl. tiles touching every image border and two sources.  GroupNorm partials
    # for every channels-per-group class of the register-direct epilogue: 16 (two groups per 32-channel half), 32, 64 (one
    # group per wave), 128 and 256 (a group spans one / two whole 128-channel tiles).
    B, c0, c1, cout, H, W = cfg[:6]
    lib = L().lib()
    g = torch.Generator().manual_seed(11)
    x0 = rnd(torch.randn(B, c0, H, W, generator=g), True)
    x1 = rnd(torch.randn(B, c1, H, W, generator=g), True) if c1 else None
    w = rnd(torch.randn(cout, c0 + c1, 3, 3, generator=g) / (3 * (c0 + c1) ** 0.5), True)
    b = torch.randn(cout, generator=g)
    groups = cfg[6] if len(cfg) > 6 else (8 if cout <= 1024 else 0)
    got, part, nslots = run_conv(x0, x1, w, b, ks=3, stride=1, pad=1, kind=0, bf16=True, groups=groups, impl=2,
                                 want_slots=True)
    xin = x0 if x1 is None else torch.cat((x0, x1), 1)
    want = F.conv2d(xin, w, b, padding=1)
    assert (got - want).abs().max() <= tol(True, want)
    if groups:
        assert torch.isfinite(part).all()
        cpg = cout // groups
        s = part.sum(2).cpu()
        want_s1 = want.reshape(B, groups, -1).sum(-1)
        want_s2 = (want ** 2).reshape(B, groups, -1).sum(-1)
        assert (s[..., 0] - want_s1).abs().max() <= 2e-3 * max(1.0, float(want_s1.abs().max()))
        assert (s[..., 1] - want_s2).abs().max() <= 2e-3 * float(want_s2.abs().max())
        # and the whole Block: GroupNorm + SiLU on top of those statistics
        gamma, beta = 1 + 0.2 * torch.randn(cout, generator=g), 0.3 * torch.randn(cout, generator=g)
        d = to_dev_nhwc(got, True)
        dg, db_ = gamma.to(DEV), beta.to(DEV)
        part = part.contiguous()
        L().check(lib.srgd_k_groupnorm_silu(ptr(d), ptr(d), ptr(None), ptr(part), B, H * W, cout, groups, ptr(dg),
                                            ptr(db_), ptr(None), nslots, 1, stream()), "groupnorm")
        y = F.silu(F.group_norm(got, groups, gamma, beta, eps=1e-5))
        assert (from_dev_nhwc(d) - y).abs().max() <= tol(True, y, k=2.0)


def test_conv3x3_bf16_integer_exact():
    g = torch.Generator().manual_seed(12)
    x = torch.randint(-3, 4, (2, 64, 16, 64), generator=g).float()
    w = torch.randint(-2, 3, (128, 64, 3, 3), generator=g).float()
    b = torch.randint(-4, 5, (128,), generator=g).float()
    got, _ = run_conv(x, None, w, b, ks=3, stride=1, pad=1, kind=0, bf16=True, impl=2)
    assert torch.equal(got, F.conv2d(x, w, b, padding=1).to(torch.bfloat16).float())


@pytest.mark.parametrize("C", [128, 256])
@pytest.mark.parametrize("cfg", [(1, 16, 16), (2, 32, 64), (1, 128, 128), (3, 8, 8)], ids=lambda s: "B%d_%dx%d" % s)
def test_linear_attention_block_fused(cfg, C):
    # linattn_fused.hip (C = 128) / linattn_fused256.hip (C = 256): RMSNorm -> qkv -> linear attention -> to_out ->
    # RMSNorm -> + x in two kernels, against the oracle's LinearAttention block on bf16-rounded inputs/weights.
    B, H, W = cfg
    lib = L().lib()
    g = torch.Generator().manual_seed(21)
    x = rnd(torch.randn(B, C, H, W, generator=g) * 1.5, True)
    sd = {"a.norm.g": 1 + 0.1 * torch.randn(1, C, 1, 1, generator=g),
          "a.to_qkv.weight": torch.randn(384, C, 1, 1, generator=g) / C ** 0.5,
          "a.to_out.0.weight": torch.randn(C, 128, 1, 1, generator=g) / 128 ** 0.5,
          "a.to_out.0.bias": 0.1 * torch.randn(C, generator=g),
          "a.to_out.1.g": 1 + 0.1 * torch.randn(1, C, 1, 1, generator=g)}
    want = O.linear_attention(sd, "a", x, 4, 32) + x
    d = to_dev_nhwc(x, True)
    y = torch.empty_like(d)
    hw = [sd["a.to_qkv.weight"].reshape(384, C).contiguous(), sd["a.norm.g"].reshape(C).contiguous(),
          sd["a.to_out.0.weight"].reshape(C, 128).contiguous(), sd["a.to_out.0.bias"].contiguous(),
          sd["a.to_out.1.g"].reshape(C).contiguous()]
    L().check(lib.srgd_k_linattn_block_fused(ptr(d), ptr(y), B, H * W, C, *[ptr(t) for t in hw], stream()), "fused")
    torch.cuda.synchronize()
    got = from_dev_nhwc(y)
    err = (got - want).abs().max().item()
    assert err <= 3e-2 * max(1.0, float(want.abs().max())), err


def test_conv_rejects_bad_shapes():
    lib = L().lib()
    x = torch.zeros(1, 8, 8, 24, device=DEV)           # 24 channels: not a multiple of 16
    w = torch.zeros(16, 24, 3, 3)
    out = torch.zeros(1, 8, 8, 16, device=DEV)
    rc = lib.srgd_k_conv2d(ptr(x), ptr(None), 24, 0, 1, 8, 8, 3, 1, 1, 0, ptr(w), ptr(None), 16, ptr(out), ptr(None),
                           ptr(None), 0, 0, stream())
    assert rc != 0 and b"multiple of 16" in lib.srgd_last_error()


# ------------------------------------------------------------------ conv1x1_bf16 (pointwise streaming GEMM)
def test_conv1x1_fast_path_exact_small_integers():
    # integer data: exact in bf16 -> bit-identical to torch; catches A/B swizzle, K-walk (two sources) and
    # transposed-epilogue mistakes.  M = 2*16*32 = 1024 rows (4 tiles), Cout 256 (2 n-tiles), K = 64 + 32.
    g = torch.Generator().manual_seed(11)
    x0 = torch.randint(-3, 4, (2, 64, 16, 32), generator=g).float()
    x1 = torch.randint(-3, 4, (2, 32, 16, 32), generator=g).float()
    w = torch.randint(-2, 3, (256, 96, 1, 1), generator=g).float()
    b = torch.randint(-4, 5, (256,), generator=g).float()
    got, _ = run_conv(x0, x1, w, b, ks=1, stride=1, pad=0, kind=0, bf16=True, impl=3)
    want = F.conv2d(torch.cat((x0, x1), 1), w, b).to(torch.bfloat16).float()
    assert torch.equal(got, want)


@pytest.mark.parametrize("variant", ["plain", "residual", "gn_tail", "pixel_shuffle", "unshuffle"])
def test_conv1x1_fast_path_matches_torch_and_generic(variant):
    g = torch.Generator().manual_seed(12)
    B, H, W = 2, 16, 32
    kw = dict(ks=1, stride=1, pad=0, kind=0, bf16=True)
    if variant == "pixel_shuffle":
        x0, x1 = rnd(torch.randn(B, 64, H, W, generator=g), True), None
        w = rnd(torch.randn(512, 64, 1, 1, generator=g) / 8, True)
        b = torch.randn(512, generator=g)
        kw["kind"] = 2
        want = O.pixel_shuffle_up({"u.net.0.weight": w, "u.net.0.bias": b}, "u", x0)
    elif variant == "unshuffle":
        x0, x1 = rnd(torch.randn(B, 32, 2 * H, 2 * W, generator=g), True), None
        w = rnd(torch.randn(128, 128, 1, 1, generator=g) / 11, True)
        b = torch.randn(128, generator=g)
        kw.update(ks=2, stride=2, kind=1)
        want = O.space_to_depth_conv({"d.1.weight": w, "d.1.bias": b}, "d", x0)
    else:
        x0 = rnd(torch.randn(B, 64, H, W, generator=g), True)
        x1 = rnd(torch.randn(B, 32, H, W, generator=g), True)
        w = rnd(torch.randn(128, 96, 1, 1, generator=g) / 10, True)
        b = torch.randn(128, generator=g)
        want = F.conv2d(torch.cat((x0, x1), 1), w, b)
        if variant == "residual":
            kw["residual"] = rnd(torch.randn(B, 128, H, W, generator=g), True)
            want = want + kw["residual"]
        if variant == "gn_tail":
            h = rnd(torch.randn(B, 128, H, W, generator=g), True)
            ca, cb = 1 + 0.3 * torch.randn(B, 128, generator=g), 0.5 * torch.randn(B, 128, generator=g)
            kw["gn_tail"] = (h, ca, cb)
            want = F.silu(ca[:, :, None, None] * h + cb[:, :, None, None]) + want
    fast, _ = run_conv(x0, x1, w, b, impl=3, **kw)
    generic, _ = run_conv(x0, x1, w, b, impl=1, **kw)
    assert fast.shape == want.shape
    assert (fast - want).abs().max() <= tol(True, want, k=2.0)
    assert (fast - generic).abs().max() <= tol(True, want, k=2.0)


@pytest.mark.parametrize("variant", ["plain", "residual", "gn_tail", "pixel_shuffle", "unshuffle"])
def test_conv1x1_streaming_kernel_equals_generic_kernel_bitwise_on_a_multi_tile_grid(variant):
    # conv1x1_bf16 walks K in the same order and rounds at the same points as the generic implicit-GEMM kernel: every epilogue,
    # the two-source K walk and the 2x2 / stride-2 gather must agree to the last bit (impl 3 vs impl 1), over a grid with several
    # m- and n-tiles per XCD.  (Round 4 ran this comparison for a 256 x 256 tile instance, which lost its A/B and was removed.)
    g = torch.Generator().manual_seed(31)
    B, H, W = 3, 16, 32
    kw = dict(ks=1, stride=1, pad=0, kind=0, bf16=True)
    if variant == "pixel_shuffle":
        x0, x1 = rnd(torch.randn(B, 128, H, W, generator=g), True), None
        w = rnd(torch.randn(1024, 128, 1, 1, generator=g) / 11, True)         # Cout / 4 = 256
        b = torch.randn(1024, generator=g)
        kw["kind"] = 2
    elif variant == "unshuffle":
        x0, x1 = rnd(torch.randn(B, 64, 2 * H, 2 * W, generator=g), True), None
        w = rnd(torch.randn(512, 256, 1, 1, generator=g) / 16, True)
        b = torch.randn(512, generator=g)
        kw.update(ks=2, stride=2, kind=1)
    else:
        x0 = rnd(torch.randn(B, 160, H, W, generator=g), True)
        x1 = rnd(torch.randn(B, 96, H, W, generator=g), True)
        w = rnd(torch.randn(512, 256, 1, 1, generator=g) / 16, True)
        b = torch.randn(512, generator=g)
        if variant == "residual":
            kw["residual"] = rnd(torch.randn(B, 512, H, W, generator=g), True)
        if variant == "gn_tail":
            h = rnd(torch.randn(B, 512, H, W, generator=g), True)
            kw["gn_tail"] = (h, 1 + 0.3 * torch.randn(B, 512, generator=g), 0.5 * torch.randn(B, 512, generator=g))
    fast, _ = run_conv(x0, x1, w, b, impl=3, **kw)
    generic, _ = run_conv(x0, x1, w, b, impl=1, **kw)
    assert torch.isfinite(fast).all() and fast.abs().max() > 0.5
    assert torch.equal(fast, generic)


def test_conv1x1_production_shape_agrees_with_generic():
    # res_conv of the last up stage: 128+128 -> 128 at 256^2 with the GroupNorm tail, in place (out aliases the tail source
    # in the engine; here separate buffers), batch 3
    g = torch.Generator().manual_seed(13)
    x0 = rnd(torch.randn(3, 128, 256, 256, generator=g), True)
    x1 = rnd(torch.randn(3, 128, 256, 256, generator=g), True)
    w = rnd(torch.randn(128, 256, 1, 1, generator=g) / 16, True)
    b = torch.randn(128, generator=g)
    h = rnd(torch.randn(3, 128, 256, 256, generator=g), True)
    ca, cb = 1 + 0.3 * torch.randn(3, 128, generator=g), 0.5 * torch.randn(3, 128, generator=g)
    kw = dict(ks=1, stride=1, pad=0, kind=0, bf16=True, gn_tail=(h, ca, cb))
    fast, _ = run_conv(x0, x1, w, b, impl=3, **kw)
    generic, _ = run_conv(x0, x1, w, b, impl=1, **kw)
    assert (fast - generic).abs().max() <= tol(True, generic, k=2.0)


# ------------------------------------------------------------------ kernels vs the reference's own sub-modules (G3 fixtures)
def test_kernel_chains_match_reference_submodules_fp32():
    # tests/golden/modules_dim16.npz holds outputs of the REFERENCE nn.Modules (dim-16 U-Net, seeded weights); here the HIP
    # kernels, driven one by one through the kernel-level C ABI in fp32 mode, must reproduce them.
    import json
    import os
    from srgd_amd.synth import synth_state_dict
    from tests.golden import cases as GC
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(gdir, f"modules_dim{GC.MODULE_DIM}.npz"))
    with open(os.path.join(gdir, f"schema_dim{GC.MODULE_DIM}.json")) as f:
        schema = {k: tuple(v) for k, v in json.load(f).items()}
    sd = O.strip_model_prefix(synth_state_dict(schema, seed=0))
    lib = L().lib()

    def check(name, got, k=1.0):
        want = torch.from_numpy(z[name])
        assert got.shape == want.shape, name
        assert (got - want).abs().max() <= tol(False, want, k=k), (name, float((got - want).abs().max()))

    def rmsnorm(x, gain, residual=None):
        d = to_dev_nhwc(x, False)
        out = torch.empty_like(d)
        dres = None if residual is None else to_dev_nhwc(residual, False)
        dg = gain.reshape(-1).float().contiguous().to(DEV)
        B, Cc, H, W = x.shape
        L().check(lib.srgd_k_rmsnorm(ptr(d), ptr(out), ptr(dres), ptr(dg), B * H * W, Cc, 0, stream()), "rmsnorm")
        return from_dev_nhwc(out)

    def attention(x, p, full):
        B, Cc, H, W = x.shape
        xn = rmsnorm(x, sd[p + ".norm.g"])
        qkv, _ = run_conv(xn, None, sd[p + ".to_qkv.weight"], None, ks=1, stride=1, pad=0, kind=0, bf16=False)
        d = to_dev_nhwc(qkv, False)
        out = torch.empty(B, H, W, 128, device=DEV, dtype=torch.float32)
        fn = lib.srgd_k_full_attention if full else lib.srgd_k_linear_attention
        L().check(fn(ptr(d), ptr(out), B, H * W, 4, 0, stream()), "attention core")
        core = from_dev_nhwc(out)
        if full:
            y, _ = run_conv(core, None, sd[p + ".to_out.weight"], sd[p + ".to_out.bias"], ks=1, stride=1, pad=0, kind=0, bf16=False)
            return y
        y, _ = run_conv(core, None, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"], ks=1, stride=1, pad=0, kind=0, bf16=False)
        return rmsnorm(y, sd[p + ".to_out.1.g"])

    check("rms_norm", rmsnorm(GC.module_input("rms_norm"), sd["downs.0.2.norm.g"]))
    got, _ = run_conv(GC.module_input("downsample"), None, sd["downs.0.3.1.weight"], sd["downs.0.3.1.bias"], ks=2, stride=2,
                      pad=0, kind=1, bf16=False)
    check("downsample", got)
    got, _ = run_conv(GC.module_input("pixel_shuffle_up"), None, sd["ups.0.3.net.0.weight"], sd["ups.0.3.net.0.bias"], ks=1,
                      stride=1, pad=0, kind=2, bf16=False)
    check("pixel_shuffle_up", got)
    got, _ = run_conv(GC.module_input("last_down_conv3x3"), None, sd["downs.3.3.weight"], sd["downs.3.3.bias"], ks=3, stride=1,
                      pad=1, kind=0, bf16=False)
    check("last_down_conv3x3", got)
    check("linear_attention", attention(GC.module_input("linear_attention"), "downs.0.2", False), k=2.0)
    check("full_attention", attention(GC.module_input("full_attention"), "downs.3.2", True), k=2.0)
    check("mid_attention", attention(GC.module_input("mid_attention"), "mid_attn", True), k=2.0)


# ------------------------------------------------------------------ MX-fp8 (BASELINE configs[4] compute path)
def test_quant_mxfp8_is_bit_exact_with_the_format_emulation():
    # quant_mxfp8.hip vs oracle/mxfp8.py (OCP MX format; E8M0 = floor(log2 amax) - 8, + 1 when the block maximum would saturate; e4m3 RNE): every byte equal.
    from oracle import mxfp8 as MX
    lib = L().lib()
    g = torch.Generator().manual_seed(31)
    npix, C = 4096, 256
    x = torch.randn(npix, C, generator=g) * torch.exp(3 * torch.randn(npix, 1, generator=g))     # wide dynamic range across pixels
    x[5] = 0                                                         # an all-zero pixel (scale byte 0)
    x[6, :32] = 1e-30                                                # below bf16's normal range after rounding? (stays tiny)
    x[7, :32] = torch.tensor([2.0 ** (k % 9 - 4) for k in range(32)])       # exact powers of two: block maximum on a boundary
    x[8, 32:64] = 1.9990234375 * 2.0 ** 3                           # mantissa > 1.75: scaled maximum exceeds 448 -> saturates
    x = x.to(torch.bfloat16)
    d = x.to(DEV)
    q = torch.empty(npix, C, dtype=torch.uint8, device=DEV)
    s = torch.empty(npix, C // 32, dtype=torch.uint8, device=DEV)
    L().check(lib.srgd_k_quant_mxfp8(ptr(d), ptr(q), ptr(s), npix, C, stream()), "quant")
    torch.cuda.synchronize()
    wq, ws, _ = MX.quantize(x.float())
    assert torch.equal(s.cpu(), ws)
    assert torch.equal(q.cpu(), wq)


def test_quant_mxfp8_keeps_a_nan_visible():
    # a NaN activation must not become a finite e4m3 value (the convolution consumes the fp8 twin, so the blow-up would
    # vanish): it is stored as the e4m3 NaN byte (0x7f / 0xff), its block's scale comes from the finite elements, and an
    # infinity saturates to +-448 like any out-of-range value.
    from oracle import mxfp8 as MX
    lib = L().lib()
    g = torch.Generator().manual_seed(32)
    npix, C = 64, 64
    x = torch.randn(npix, C, generator=g).to(torch.bfloat16)
    x[3, 5] = float("nan")
    x[9, 40] = float("inf")
    d = x.to(DEV)
    q = torch.empty(npix, C, dtype=torch.uint8, device=DEV)
    s = torch.empty(npix, C // 32, dtype=torch.uint8, device=DEV)
    L().check(lib.srgd_k_quant_mxfp8(ptr(d), ptr(q), ptr(s), npix, C, stream()), "quant")
    torch.cuda.synchronize()
    q, s = q.cpu(), s.cpu()
    assert (int(q[3, 5]) & 0x7f) == 0x7f
    clean = x.float().clone()
    clean[3, 5] = 0.0
    wq, ws, _ = MX.quantize(clean)
    keep = torch.ones(npix, C, dtype=torch.bool)
    keep[3, 5] = False
    keep[9, 32:] = False                                   # the infinity's block: its scale byte saturates, checked below
    assert torch.equal(s[:, 0], ws[:, 0]) and torch.equal(s[torch.arange(npix) != 9, 1], ws[torch.arange(npix) != 9, 1])
    assert torch.equal(q[keep], wq[keep])
    assert q[9, 40] == 0x7e                                # +448, the largest finite e4m3 value
    assert torch.isfinite(q.view(torch.float8_e4m3fn).float()[keep]).all()


def _mx_conv_reference(x0, x1, w, b):
    """The fp8 convolution's arithmetic on the CPU: MX-quantised activations (per pixel, per 32 channels) and weights (per
    output channel, tap, 32 input channels), exact products, fp32 sums; each source quantised on its own as the engine does."""
    from oracle import mxfp8 as MX
    qs = []
    for t in (x0, x1):
        if t is not None:
            _, _, deq = MX.quantize(t.permute(0, 2, 3, 1).contiguous())
            qs.append(deq.permute(0, 3, 1, 2))
    return F.conv2d(torch.cat(qs, 1).double(), MX.quantize_conv_weight(w).double(), b.double(), padding=1).float()


def test_conv3x3_mxfp8_exact_small_integers():
    # small integers are exact in e4m3 with power-of-two block scales -> the result must equal the fp32 convolution bit for bit
    # (after the bf16 store); catches operand / scale lane maps, swizzles, the two-source K walk and the transposed epilogue.
    lib = L().lib()
    g = torch.Generator().manual_seed(12)
    x0 = torch.randint(-3, 4, (2, 128, 16, 32), generator=g).float()
    x1 = torch.randint(-3, 4, (2, 256, 16, 32), generator=g).float()
    w = torch.randint(-2, 3, (256, 384, 3, 3), generator=g).float()
    b = torch.randint(-4, 5, (256,), generator=g).float()
    d0, d1 = to_dev_nhwc(x0, True), to_dev_nhwc(x1, True)
    out = torch.empty(2, 16, 32, 256, dtype=torch.bfloat16, device=DEV)
    L().check(lib.srgd_k_conv3x3_mxfp8(ptr(d0), ptr(d1), 128, 256, 2, 16, 32, ptr(w), ptr(b), 256, ptr(out), ptr(None), 8, 0,
                                       None, None, stream()), "conv3x3_mxfp8")
    want = F.conv2d(torch.cat([x0, x1], 1), w, b, padding=1).to(torch.bfloat16).float()
    assert torch.equal(from_dev_nhwc(out), want)


@pytest.mark.parametrize("shape", [(1, 128, 128, 32, 64), (3, 256, 128, 8, 32), (1, 512, 1024, 32, 32)],
                         ids=lambda s: "B%d_%dto%d_%dx%d" % s)
def test_conv3x3_mxfp8_matches_the_quantised_reference(shape):
    # random data: the kernel must reproduce conv(dequant(MX(x)), dequant(MX(w))) up to fp32 summation order and the bf16 store;
    # the distance to the UNQUANTISED convolution is reported (that is the price of e4m3, not a kernel property)
    B, Cin, Cout, H, W = shape
    lib = L().lib()
    g = torch.Generator().manual_seed(13)
    x = rnd(torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, 1, H, W, generator=g)), True)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    b = 0.1 * torch.randn(Cout, generator=g)
    d = to_dev_nhwc(x, True)
    out = torch.empty(B, H, W, Cout, dtype=torch.bfloat16, device=DEV)
    nslots = C.c_int(0)
    part = torch.zeros(B * 8 * (H * W // 32) * 2 + 16, device=DEV)
    L().check(lib.srgd_k_conv3x3_mxfp8(ptr(d), ptr(None), Cin, 0, B, H, W, ptr(w), ptr(b), Cout, ptr(out), ptr(part), 8, 0, None,
                                       C.byref(nslots), stream()), "conv3x3_mxfp8")
    got = from_dev_nhwc(out)
    want = _mx_conv_reference(x, None, w, b)
    scale = float(want.abs().max())
    err = (got - want).abs().max().item()
    assert err <= 2.0 ** -8 * scale + 1e-5, (err, scale)                  # bf16 store + summation order
    plain = F.conv2d(x, w, b, padding=1)
    rel = float(((got - plain) ** 2).mean().sqrt() / (plain ** 2).mean().sqrt())
    _report_k(test="conv3x3_mxfp8_vs_unquantised", shape=list(shape), rel_rms=rel)
    assert rel < 0.08                                                      # e4m3: ~2^-4 / sqrt(3) per operand
    # GroupNorm partial sums written by the epilogue: per (sample, group) over all slots = sums of the fp32 outputs
    p = part[: B * 8 * nslots.value * 2].cpu().reshape(B, 8, nslots.value, 2).sum(2)
    wsum = want.reshape(B, 8, -1).sum(-1)
    wsq = (want.double() ** 2).reshape(B, 8, -1).sum(-1).float()
    assert torch.allclose(p[..., 0], wsum, rtol=2e-3, atol=2e-2 * float(wsq.max()) ** 0.5)
    assert torch.allclose(p[..., 1], wsq, rtol=2e-3)


# ------------------------------------------------------------------ MX-fp8 pointwise layers (conv1x1_mxfp8.hip, impl 4)
def _mx_pointwise_reference(x0, x1, w_oi, b, taps=1):
    """conv1x1_mxfp8's arithmetic on the CPU: MX-quantised activations (per pixel, per 32 channels of each source) and weights
    (per output channel, tap, 32 input channels), exact products, fp32 sums.  taps = 4: 2x2 / stride-2 gather (Downsample)."""
    from oracle import mxfp8 as MX
    qs = []
    for t in (x0, x1):
        if t is not None:
            _, _, deq = MX.quantize(t.permute(0, 2, 3, 1).contiguous())
            qs.append(deq.permute(0, 3, 1, 2))
    xq = torch.cat(qs, 1)
    if taps == 1:
        _, _, wq = MX.quantize(w_oi.reshape(w_oi.shape[0], -1))
        return F.conv2d(xq.double(), wq.reshape(*w_oi.shape[:2], 1, 1).double(), b.double()).float()
    # Downsample: rearrange 'b c (h p1) (w p2) -> b (c p1 p2) h w' then 1x1 over 4c channels (reference model.py:106-110); the
    # engine walks it tap-major (tap = p1 * 2 + p2, c inside), one weight scale per (output channel, tap, 32 channels)
    cout, c4 = w_oi.shape[:2]
    c = c4 // 4
    w4 = w_oi.reshape(cout, c, 2, 2).permute(0, 2, 3, 1).contiguous()          # [o, p1, p2, c]
    _, _, w4q = MX.quantize(w4)
    return F.conv2d(xq.double(), w4q.permute(0, 3, 1, 2).double(), b.double(), stride=2).float()


def test_conv1x1_mxfp8_exact_small_integers():
    # small integers are exact in e4m3 with power-of-two block scales: bit-identical to torch after the bf16 store; catches the
    # A / B swizzles, both scale layouts (bytes per pixel, one dword per lane for the weights), the two-source K walk (128 + 256
    # channels = 3 K-steps through the 3-deep ring) and the shared transposed epilogue.  M = 2*16*32 = 1024 rows, 2 n-tiles.
    g = torch.Generator().manual_seed(21)
    x0 = torch.randint(-3, 4, (2, 128, 16, 32), generator=g).float()
    x1 = torch.randint(-3, 4, (2, 256, 16, 32), generator=g).float()
    w = torch.randint(-2, 3, (256, 384, 1, 1), generator=g).float()
    b = torch.randint(-4, 5, (256,), generator=g).float()
    got, _ = run_conv(x0, x1, w, b, ks=1, stride=1, pad=0, kind=0, bf16=True, impl=4)
    want = F.conv2d(torch.cat((x0, x1), 1), w, b).to(torch.bfloat16).float()
    assert torch.equal(got, want)


@pytest.mark.parametrize("variant", ["plain", "residual", "gn_tail", "pixel_shuffle", "unshuffle", "k_heavy"])
def test_conv1x1_mxfp8_matches_the_quantised_reference(variant):
    # random data: conv(dequant(MX(x)), dequant(MX(w))) up to fp32 summation order and the bf16 store, through every epilogue the
    # engine uses; the distance to the bf16 kernel's result is reported (the price of e4m3, not a kernel property)
    g = torch.Generator().manual_seed(22)
    B, H, W = 2, 16, 32
    kw = dict(ks=1, stride=1, pad=0, kind=0, bf16=True)
    x1 = None
    if variant == "pixel_shuffle":
        x0 = rnd(torch.randn(B, 128, H, W, generator=g), True)
        w = torch.randn(512, 128, 1, 1, generator=g) / 11
        b = torch.randn(512, generator=g)
        kw["kind"] = 2
        pre = _mx_pointwise_reference(x0, None, w[:, :, 0, 0], b)
        want = F.pixel_shuffle(F.silu(pre), 2)
    elif variant == "unshuffle":
        x0 = rnd(torch.randn(B, 128, 2 * H, 2 * W, generator=g), True)
        w = torch.randn(128, 512, 1, 1, generator=g) / 22
        b = torch.randn(128, generator=g)
        kw.update(ks=2, stride=2, kind=1)
        want = _mx_pointwise_reference(x0, None, w[:, :, 0, 0], b, taps=4)
    elif variant == "k_heavy":
        x0 = rnd(torch.randn(1, 1024, H, W, generator=g) * torch.exp(torch.randn(1, 1, H, W, generator=g)), True)
        x1 = rnd(torch.randn(1, 512, H, W, generator=g), True)
        w = torch.randn(1024, 1536, 1, 1, generator=g) / 39
        b = torch.randn(1024, generator=g)
        want = _mx_pointwise_reference(x0, x1, w[:, :, 0, 0], b)
    else:
        x0 = rnd(torch.randn(B, 128, H, W, generator=g), True)
        x1 = rnd(torch.randn(B, 128, H, W, generator=g), True)
        w = torch.randn(128, 256, 1, 1, generator=g) / 16
        b = torch.randn(128, generator=g)
        want = _mx_pointwise_reference(x0, x1, w[:, :, 0, 0], b)
        if variant == "residual":
            kw["residual"] = rnd(torch.randn(B, 128, H, W, generator=g), True)
            want = want.to(torch.bfloat16).float() + kw["residual"]
        if variant == "gn_tail":
            h = rnd(torch.randn(B, 128, H, W, generator=g), True)
            ca, cb = 1 + 0.3 * torch.randn(B, 128, generator=g), 0.5 * torch.randn(B, 128, generator=g)
            kw["gn_tail"] = (h, ca, cb)
            want = F.silu(ca[:, :, None, None] * h + cb[:, :, None, None]) + want.to(torch.bfloat16).float()
    got, _ = run_conv(x0, x1, w, b, impl=4, **kw)
    assert got.shape == want.shape
    scale = float(want.abs().max())
    err = (got - want).abs().max().item()
    assert err <= 2.0 ** -7 * scale + 1e-5, (variant, err, scale)          # bf16 staging + store, fp32 summation order
    bf, _ = run_conv(x0, x1, rnd(w, True), b, impl=3, **kw)
    rel = float(((got - bf) ** 2).mean().sqrt() / (bf ** 2).mean().sqrt())
    _report_k(test="conv1x1_mxfp8_vs_bf16_kernel", variant=variant, rel_rms=rel)
    assert rel < 0.08
