"""CPU tests: the oracle (oracle/srgd_oracle.py) against the committed outputs of the reference
(tests/golden/*.npz|json, produced by tests/golden/make_golden.py running /root/reference)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import srgd_oracle as O
from srgd_amd.synth import synth_state_dict
from tests.golden import cases as C

G = os.path.join(os.path.dirname(__file__), "golden")


def _schema(dim):
    with open(os.path.join(G, f"schema_dim{dim}.json")) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def test_geometry_tables():
    with open(os.path.join(G, "geometry.json")) as f:
        geo = json.load(f)
    for (h, w) in C.GEOMETRY_SIZES:
        want = geo[f"{h}x{w}"]
        box, pad = O.canvas_box_and_pad(h, w)
        assert list(box) == want["box"] and list(pad) == want["pad"]
        hp, wp = h + pad[2] + pad[3], w + pad[0] + pad[1]
        assert [hp, wp] == want["canvas"]
        even, odd = O.sampling_grids(hp, wp)
        assert len(even) == want["n_even"] and len(odd) == want["n_odd"]
        if want["truncated"]:
            assert [list(c) for c in even[:3] + even[-3:]] == want["even"]
            assert [list(c) for c in odd[:3] + odd[-3:]] == want["odd"]
        else:
            assert [list(c) for c in even] == want["even"]
            assert [list(c) for c in odd] == want["odd"]
        inner, ipad = O.grid_bbox(odd, hp, wp)
        assert list(inner) == want["inner"] and list(ipad) == want["inner_pad"]


def test_geometry_known_answers_survey_appendix_d():
    # SURVEY.md Appendix D
    assert O.canvas_box_and_pad(1024, 1024) == ((128, 128, 1152, 1152), (128, 128, 128, 128))
    assert O.canvas_box_and_pad(300, 500) == ((134, 234, 634, 534), (134, 134, 234, 234))
    e, o = O.sampling_grids(1280, 1280)
    assert (len(e), len(o)) == (25, 16)
    e, o = O.sampling_grids(8448, 8448)
    assert (len(e), len(o)) == (1089, 1024)


def test_schedule_bit_exact():
    z = np.load(os.path.join(G, "schedule.npz"))
    for n in C.SCHEDULE_STEPS:
        steps = torch.linspace(1.0, 0.0, n + 1)
        got = torch.stack([O.log_snr_linear(steps[i]) for i in range(n + 1)]).numpy()
        assert np.array_equal(got.view(np.uint32), z[f"log_snr_{n}"].view(np.uint32))
    # range -10.00005 (t=1) ... 9.21029 (t=0)
    assert abs(float(O.log_snr_linear(torch.tensor(1.0))) + 10.00005) < 1e-4
    assert abs(float(O.log_snr_linear(torch.tensor(0.0))) - 9.21029) < 1e-4


@pytest.mark.parametrize("case", C.UNET_CASES, ids=lambda c: c["name"])
def test_unet_eps_matches_reference(case):
    z = np.load(os.path.join(G, "unet_eps.npz"))
    sd = synth_state_dict(_schema(case["dim"]), seed=case["weight_seed"])
    w_sum = sum(v.double().abs().sum().item() for v in sd.values())
    assert abs(w_sum - float(z[f"{case['name']}.w_sum"])) < 1e-9 * w_sum, "synthetic weights drifted"
    x, cnd, ls = C.unet_inputs(case)
    assert abs(x.double().sum().item() - float(z[f"{case['name']}.x_sum"])) < 1e-6
    usd = O.strip_model_prefix(sd)
    cfg = O.UnetCfg(dim=case["dim"])
    with torch.inference_mode():
        for mode in case["modes"]:
            label, c = C.unet_mode_args(mode, case, cnd)
            got = O.unet_forward(usd, cfg, x, ls, label, c).numpy()
            want = z[f"{case['name']}.{mode}"]
            assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("case", C.SAMPLER_CASES, ids=lambda c: c["name"])
def test_tiled_sample_matches_reference(case):
    z = np.load(os.path.join(G, f"sample_{case['name']}.npz"))
    sd = synth_state_dict(_schema(case["dim"]), seed=case["weight_seed"])
    cond = C.sampler_condition(case)
    assert abs(cond.double().sum().item() - float(z["cond_sum"])) < 1e-6
    torch.manual_seed(case["seed"])
    assert np.array_equal(torch.randn(16).numpy(), z["first_draw"]), "torch CPU generator stream changed"
    label = torch.tensor([case["label"]]) if case["label"] is not None else None
    torch.manual_seed(case["seed"])
    with torch.inference_mode():
        got = O.tiled_sample(O.strip_model_prefix(sd), O.UnetCfg(dim=case["dim"]), cond, label,
                             batch_size=case["batch_size"], num_sample_steps=case["steps"],
                             cond_scale=case["cond_scale"], class_cond_scale=case["class_cond_scale"],
                             **C.extra_kwargs(case))
    want = z["image"]
    assert got.shape == want.shape
    assert np.abs(got.numpy() - want).max() <= 1e-4      # thread-count noise floor is ~1e-5


def test_both_guidance_scales_raise():
    sd = synth_state_dict(_schema(16), seed=0)
    with pytest.raises(NotImplementedError):
        O.tiled_sample(O.strip_model_prefix(sd), O.UnetCfg(dim=16), torch.rand(1, 3, 256, 256),
                       torch.tensor([0]), num_sample_steps=2, cond_scale=2.0, class_cond_scale=2.0)


def test_noise_stream_is_batch_size_independent():
    # draws are multiples of 16 elements, so the stream seen by each tile
    # does not depend on how tiles are grouped into minibatches.
    torch.manual_seed(5)
    a = torch.randn(8, 3, 256, 256)
    torch.manual_seed(5)
    b = torch.cat([torch.randn(3, 3, 256, 256), torch.randn(5, 3, 256, 256)])
    assert torch.equal(a, b)


def test_pil_bicubic_restatement_is_bit_exact():
    # oracle/pil_resample.py restates Pillow's Resample.c (un-vendored dependency of the reference's T.Resize on a PIL
    # image); pinned here against Pillow itself on ragged, tiny, constant and x4 production-like inputs.
    import numpy as np
    from PIL import Image
    from oracle.pil_resample import resize_bicubic_u8, to_u8_hwc, to_unit_chw
    rng = np.random.default_rng(0)
    for (h, w, sh, sw) in [(64, 64, 4, 4), (37, 53, 4, 4), (5, 7, 4, 4), (1, 9, 4, 4), (33, 20, 3, 3), (40, 40, 1, 1),
                           (17, 31, 2, 5), (96, 128, 4, 4)]:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = np.asarray(Image.fromarray(a, "RGB").resize((w * sw, h * sh), Image.BICUBIC))
        assert np.array_equal(resize_bicubic_u8(a, h * sh, w * sw), want), (h, w)
    flat = np.full((9, 9, 3), 255, np.uint8)
    assert np.array_equal(resize_bicubic_u8(flat, 36, 36), np.asarray(Image.fromarray(flat, "RGB").resize((36, 36), Image.BICUBIC)))
    # ToTensor / ToPILImage conversions (torch semantics: /255 in fp32, mul(255).byte() truncation)
    import torch
    u = rng.integers(0, 256, (6, 5, 3), dtype=np.uint8)
    assert np.array_equal(to_unit_chw(u), torch.from_numpy(u).permute(2, 0, 1).float().div(255).numpy())
    f = torch.rand(3, 6, 5)
    assert np.array_equal(to_u8_hwc(f.numpy()), f.mul(255).byte().permute(1, 2, 0).numpy())


def _edm_state_dict(dim, seed):
    """The EDM wrapper names its U-Net ``net`` (model.py:2099): same tensors, prefix ``net.`` instead of ``model.``."""
    schema = {"net." + k[len("model."):]: v for k, v in _schema(dim).items()}
    return synth_state_dict(schema, seed=seed)


@pytest.mark.parametrize("case", C.EDM_CASES, ids=lambda c: c["name"])
def test_edm_tiled_sample_matches_reference(case):
    z = np.load(os.path.join(G, f"sample_edm_{case['name']}.npz"))
    sd = _edm_state_dict(case["dim"], case["weight_seed"])
    assert abs(sum(v.double().abs().sum().item() for v in sd.values()) - float(z["w_sum"])) < 1e-6
    usd = {k[len("net."):]: v for k, v in sd.items()}
    cond = C.sampler_condition(case)
    assert abs(cond.double().sum().item() - float(z["cond_sum"])) < 1e-6
    label = torch.tensor([case["label"]]) if case["label"] is not None else None
    torch.manual_seed(case["seed"])
    with torch.inference_mode():
        got = O.edm_tiled_sample(usd, O.UnetCfg(dim=case["dim"]), O.EdmCfg(num_sample_steps=case.get("ctor_steps", case["steps"])), cond, label, batch_size=case["batch_size"],
                                 num_sample_steps=case["steps"], cond_scale=case["cond_scale"],
                                 class_cond_scale=case["class_cond_scale"], **C.edm_extra_kwargs(case))
    want = z["image"]
    assert got.shape == want.shape
    assert np.abs(got.numpy() - want).max() <= 1e-4


def test_oracle_building_blocks_match_the_reference_submodules():
    # G3: each oracle function against the output of the reference's own nn.Module (dim-16 U-Net, seeded weights)
    z = np.load(os.path.join(G, f"modules_dim{C.MODULE_DIM}.npz"))
    sd = O.strip_model_prefix(synth_state_dict(_schema(C.MODULE_DIM), seed=0))
    t = C.module_time_embedding()
    heads, dh, groups = 4, 32, 8

    def check(name, got, tol=2e-5):
        want = torch.from_numpy(z[name])
        assert got.shape == want.shape, name
        assert (got - want).abs().max().item() <= tol * max(1.0, want.abs().max().item()), name

    with torch.inference_mode():
        x = C.module_input("resnet_same")
        check("resnet_same", O.resnet_block(sd, "downs.0.0", x, t, groups))
        x = C.module_input("resnet_concat")
        check("resnet_concat", O.resnet_block(sd, "ups.0.0", x, t, groups))
        check("linear_attention", O.linear_attention(sd, "downs.0.2", C.module_input("linear_attention"), heads, dh))
        check("full_attention", O.full_attention(sd, "downs.3.2", C.module_input("full_attention"), heads, dh))
        check("mid_attention", O.full_attention(sd, "mid_attn", C.module_input("mid_attention"), heads, dh))
        check("downsample", O.space_to_depth_conv(sd, "downs.0.3", C.module_input("downsample")))
        x = C.module_input("last_down_conv3x3")
        check("last_down_conv3x3", torch.nn.functional.conv2d(x, sd["downs.3.3.weight"], sd["downs.3.3.bias"], padding=1))
        check("pixel_shuffle_up", O.pixel_shuffle_up(sd, "ups.0.3", C.module_input("pixel_shuffle_up")))
        check("rms_norm", O.rms_norm(C.module_input("rms_norm"), sd["downs.0.2.norm.g"]))
        check("time_mlp", O.time_embedding(sd, torch.tensor([-3.0, 2.5])))
        check("class_mlp", O.class_embedding(sd, torch.tensor([1])))


@pytest.mark.parametrize("case", C.SAMPLE_CASES, ids=lambda c: c["name"])
def test_untiled_sample_matches_reference(case):
    z = np.load(os.path.join(G, f"sample_untiled_{case['name']}.npz"))
    sd = synth_state_dict(_schema(case["dim"]), seed=case["weight_seed"])
    cond = C.sample_condition(case)
    assert abs(cond.double().sum().item() - float(z["cond_sum"])) < 1e-6
    torch.manual_seed(case["seed"])
    with torch.inference_mode():
        got = O.sample(O.strip_model_prefix(sd), O.UnetCfg(dim=case["dim"]), cond, torch.tensor([case["label"]]),
                       num_sample_steps=case["steps"], cond_scale=case["cond_scale"],
                       class_cond_scale=case["class_cond_scale"], **C.sample_extra_kwargs(case))
    assert got.shape == z["image"].shape
    assert np.abs(got.numpy() - z["image"]).max() <= 1e-4


@pytest.mark.parametrize("case", C.EDM_UNTILED_CASES, ids=lambda c: c["name"])
def test_edm_untiled_sample_matches_reference(case):
    # sample -> sample_org (Heun, model.py:2212-2306) / sample_using_dpmpp (DPM-Solver++(2M), :2479-2557)
    z = np.load(os.path.join(G, f"sample_edm_untiled_{case['name']}.npz"))
    sd = _edm_state_dict(case["dim"], case["weight_seed"])
    usd = {k[len("net."):]: v for k, v in sd.items()}
    cond = C.sample_condition(case)
    assert abs(cond.double().sum().item() - float(z["cond_sum"])) < 1e-6
    fn = O.edm_sample_dpmpp if case["dpmpp"] else O.edm_sample
    torch.manual_seed(case["seed"])
    with torch.inference_mode():
        got = fn(usd, O.UnetCfg(dim=case["dim"]), O.EdmCfg(num_sample_steps=case.get("ctor_steps", case["steps"])), cond,
                 torch.tensor([case["label"]]), num_sample_steps=case["steps"], cond_scale=case["cond_scale"],
                 class_cond_scale=case["class_cond_scale"], **C.edm_extra_kwargs(case))
    assert got.shape == z["image"].shape
    assert np.abs(got.numpy() - z["image"]).max() <= 1e-4


def test_mxfp8_emulation_scale_rule_and_round_trip():
    # oracle/mxfp8.py (the format emulation the GPU quantiser is checked against bit for bit): known answers of the scale rule -
    # floor(log2 amax) - 8, one step up exactly when the block maximum would land above 448 - and round-trip properties
    from oracle import mxfp8 as M
    be = lambda v: int(M.block_exponent(torch.tensor([v], dtype=torch.float32))[0])
    assert be(448.0) == 127 and be(449.0) == 128            # 448 = 1.75 * 2^8 still fits, 449 does not
    assert be(1.0) == 119 and be(1.75) == 119 and be(1.76) == 120
    assert be(0.0) == 0 and be(1e-40) == 0                  # zero / denormal blocks: smallest scale
    g = torch.Generator().manual_seed(3)
    x = torch.randn(64, 128, generator=g) * torch.logspace(-3, 3, 64).unsqueeze(1)      # rows over six decades
    q, s, deq = M.quantize(x)
    assert q.shape == x.shape and s.shape == (64, 4) and q.dtype == torch.uint8 and s.dtype == torch.uint8
    blocks = x.reshape(64, 4, 32)
    scale = torch.ldexp(torch.ones(64, 4), s.int() - 127)
    # never saturates: the block maximum is representable (<= 448 after scaling), and the scale is the smallest such power of two
    assert float((blocks.abs().amax(-1) / scale).max()) <= 448.0
    assert float((blocks.abs().amax(-1) / (scale / 2)).min()) > 448.0
    # e4m3 carries 3 mantissa bits: relative error of a normal element <= 2^-4; dequantise -> requantise is idempotent
    rel = ((deq - x).abs() / x.abs().clamp(min=1e-30)).reshape(64, 4, 32)
    big = blocks.abs() >= blocks.abs().amax(-1, keepdim=True) / 64       # elements within 6 binades of the block maximum
    assert float(rel[big].max()) <= 2.0 ** -4 + 1e-6
    # (as values: a maximum that rounded down onto mantissa 1.75 is re-encoded one scale step lower with the same value)
    assert torch.equal(M.quantize(deq)[2], deq)
