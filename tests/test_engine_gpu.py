"""GPU parity tests of the product path (srgd_amd -> C ABI -> HIP kernels) against the committed
outputs of the reference (tests/golden) and against the CPU oracle on identical seeded inputs/noise.

Tolerances: fp32 mode is gated at the north-star bar (1e-3 max abs on final pixels in [0,1]; the
measured gap is ~1e-5, the reference's own thread-count noise).  bf16 mode is a
throughput mode: its error vs the fp32 reference is *reported* and only loosely bounded here.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import srgd_oracle as O
from srgd_amd.synth import synth_state_dict
from tests.golden import cases as C

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
REPORT = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out", "parity_report.jsonl")


def _report(**kw):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(json.dumps(kw) + "\n")


def _schema(dim):
    with open(os.path.join(G, f"schema_dim{dim}.json")) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


_MODELS = {}


# MX-fp8 modes vs the REFERENCE's fixtures.  FROZEN: a change that lowers PSNR must fail here, not move the gate.
# fp8 gates are round 3's (33.5 / 25.0).  Round 4 measured, with MX-e4m3 attention weights (9 sites in fp8, the 7 below the top
# resolution in fp8_mixed) and fp8_mixed's pointwise layers back on conv1x1_bf16: fp8 35.06 / 26.83 dB (round 3, bf16 attention
# weights: 36.48 / 27.98), fp8_mixed 52.62 / 39.96 dB (round 3 with the MX pointwise kernel: 52.14 / 39.63) - fp8_mixed's
# gates moved UP to 3 dB under those.
FP8_CONFIG5_GATE_DB, FP8_MIXED_CONFIG5_GATE_DB = 33.8, 49.6          # configs[4], one tile, 100 steps, CFG 2.0 (round 5: fp8 36.9 dB with e4m3 attention weights at eight sites)
FP8_CONFIG5_FULL_GATE_DB, FP8_MIXED_CONFIG5_FULL_GATE_DB = 33.2, 49.2   # configs[4] at its full geometry, against the bf16 ENGINE on the same noise
FP8_CONFIG2_GATE_DB, FP8_MIXED_CONFIG2_GATE_DB = 25.0, 36.9          # configs[1] geometry, 2 steps from noise (bf16: 43.7)


def build_sampler(dim, steps=50, weight_seed=0, fresh=False):
    """The product objects, built exactly as inference.py does (get_model -> .module.eval().to(cuda)).  ``fresh``: a sampler
    (and engine) of the caller's own instead of the cached one - ranks-as-threads tests need one per thread."""
    key = (dim, weight_seed) if not fresh else (dim, weight_seed, object())
    if key not in _MODELS:
        import logging
        from srgd_amd.config import load_config
        from srgd_amd.model import get_model
        conf = load_config(os.path.join(os.path.dirname(G), "..", "conf", "conditional_continuous_linear_df8kost_dim128.yaml"))
        conf.unet_dim = dim
        conf.num_sample_steps = steps
        ema = get_model(conf, logging.getLogger("test"))
        ema.module.load_state_dict(synth_state_dict(_schema(dim), seed=weight_seed), strict=True)
        _MODELS[key] = ema.module.eval().to(torch.device("cuda"))
    return _MODELS.pop(key) if fresh else _MODELS[key]


def test_library_is_loaded_in_process():
    from srgd_amd import _lib
    _lib.lib()
    with open("/proc/self/maps") as f:
        assert "libsrgd_hip.so" in f.read(), "the HIP engine must be the thing that runs"


@pytest.mark.parametrize("case", C.UNET_CASES, ids=lambda c: c["name"])
@pytest.mark.parametrize("precision", ["fp32", "f16x3", "f16mx2", "bf16"])
def test_unet_forward_matches_reference(case, precision):
    z = np.load(os.path.join(G, "unet_eps.npz"))
    sampler = build_sampler(case["dim"], weight_seed=case["weight_seed"])
    unet = sampler.model
    x, cnd, ls = C.unet_inputs(case)
    unet.precision = precision
    try:
        for mode in case["modes"]:
            label, c = C.unet_mode_args(mode, case, cnd)
            got = unet(x.cuda(), ls.cuda(), None if label is None else label.cuda(), None if c is None else c.cuda())
            torch.cuda.synchronize()
            want = torch.from_numpy(z[f"{case['name']}.{mode}"])
            err = (got.cpu() - want).abs().max().item()
            scale = max(1.0, want.abs().max().item())
            _report(test="unet_forward", case=case["name"], mode=mode, precision=precision, max_abs=err, ref_max=scale)
            assert torch.isfinite(got).all()
            # bf16 measured on MI355X: 2.4e-2 (dim 16) / 1.3e-2 (dim 128) of the eps range; gate at ~1.7x that
            # f16mx2 (prototype): the three-MFMA blocks at the input's own resolution, two-MFMA below - eps within 5e-4 of its range
            assert err <= (4e-2 if precision == "bf16" else 5e-4 if precision == "f16mx2" else 1e-4) * scale, (mode, err)
    finally:
        unet.precision = "fp32"


# The two precisions that claim the north-star bar (<= 1e-3 max-abs against the reference's output): exact-fp32 MFMA, and the
# split-operand mode (three f16 MFMAs per product on fp32 tensors; dim-16 cases mix it with fp32 layers where Cin % 32 != 0) - and the
# f16mx2 prototype in its default placement (dim 16: identical to f16x3, no layer is eligible; dim 128: two-MFMA arithmetic below 256^2).
@pytest.mark.parametrize("precision", ["fp32", "f16x3", "f16mx2"])
@pytest.mark.parametrize("case", C.SAMPLER_CASES, ids=lambda c: c["name"])
def test_tiled_sample_fp32_matches_reference(case, precision):
    z = np.load(os.path.join(G, f"sample_{case['name']}.npz"))
    sampler = build_sampler(case["dim"], weight_seed=case["weight_seed"])
    cond = C.sampler_condition(case)
    label = torch.tensor([case["label"]]).cuda() if case["label"] is not None else None
    torch.manual_seed(case["seed"])
    assert np.array_equal(torch.randn(16).numpy(), z["first_draw"]), "torch CPU generator stream differs from the fixture's"
    torch.manual_seed(case["seed"])
    sampler.noise_source = "host"
    got = sampler.tiled_sample(batch_size=case["batch_size"], condition_x=cond.cuda(), class_label=label,
                               cond_scale=case["cond_scale"], class_cond_scale=case["class_cond_scale"],
                               num_sample_steps=case["steps"], precision=precision, **C.extra_kwargs(case))
    torch.cuda.synchronize()
    want = torch.from_numpy(z["image"])
    assert got.shape == want.shape and got.dtype == torch.float32
    err = (got.cpu() - want).abs().max().item()
    _report(test="tiled_sample", case=case["name"], precision=precision, max_abs=err)
    assert err <= 1e-3, err           # north-star bar
    assert err <= 2e-4, err           # and in practice an order of magnitude inside it


@pytest.mark.parametrize("case", [c for c in C.SAMPLER_CASES if c["name"] in ("dim16_256_cfg1", "dim128_config1")],
                         ids=lambda c: c["name"])
def test_tiled_sample_bf16_vs_reference_reported(case):
    z = np.load(os.path.join(G, f"sample_{case['name']}.npz"))
    sampler = build_sampler(case["dim"], weight_seed=case["weight_seed"])
    cond = C.sampler_condition(case)
    label = torch.tensor([case["label"]]).cuda()
    torch.manual_seed(case["seed"])
    sampler.noise_source = "host"
    got = sampler.tiled_sample(batch_size=case["batch_size"], condition_x=cond.cuda(), class_label=label,
                               num_sample_steps=case["steps"], precision="bf16").cpu()
    want = torch.from_numpy(z["image"])
    err = (got - want).abs()
    mse = float((err ** 2).mean())
    psnr = 10 * np.log10(1.0 / max(mse, 1e-20))
    _report(test="tiled_sample", case=case["name"], precision="bf16", max_abs=float(err.max()),
            mean_abs=float(err.mean()), psnr_db=psnr)
    assert torch.isfinite(got).all() and got.min() >= 0 and got.max() <= 1
    # measured on MI355X: 54.3 dB (dim 16) / 55.9 dB (dim 128) against the REFERENCE's image; gate 3 dB below
    assert psnr > 51.0, psnr
    assert float(err.max()) < 0.1, float(err.max())      # measured 4.4e-2 ... 5.1e-2


def test_result_is_independent_of_batch_size_and_bitwise_repeatable():
    # tiles are independent and the noise stream does not depend on the
    # minibatch grouping -> the fp32 engine must give bit-identical images for any sub-batch size.
    case = C.SAMPLER_CASES[2]
    sampler = build_sampler(case["dim"])
    cond = C.sampler_condition(case).cuda()
    label = torch.tensor([case["label"]]).cuda()
    outs = []
    for bs in (1, 4, 9, 4):
        torch.manual_seed(3)
        outs.append(sampler.tiled_sample(batch_size=bs, condition_x=cond, class_label=label,
                                         num_sample_steps=3, precision="fp32").cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]) and torch.equal(outs[1], outs[3])


def test_two_step_lanes_are_bitwise_identical_to_one():
    # srgd_amd.lanes: a small step runs as two concurrent halves on two HIP streams through two engines.  Tiles of a step are
    # independent, so the images must not change - in both noise modes, with and without guidance (two passes per tile), in
    # the eager and the hipGraph path (device noise, >= 3 steps: direct launch, capture, replay), DDPM and EDM.
    sampler = build_sampler(16)
    cond = C.synthetic_lr_condition(0, 256, 256).cuda()          # configs[1] geometry: 25 / 16 tiles per step
    label = torch.tensor([0]).cuda()
    keep = sampler.step_lanes
    try:
        for noise, scale, prec in (("device", 1.0, "bf16"), ("host", 2.0, "fp32"), ("device", 2.0, "fp8")):
            sampler.noise_source = noise
            outs = []
            for lanes in (1, 2, None):                            # None: automatic (two lanes here: one launch of <= 100 samples)
                sampler.step_lanes = lanes
                sampler.device_noise_seed = 5
                torch.manual_seed(5)
                outs.append(sampler.tiled_sample(batch_size=25, condition_x=cond, class_label=label, class_cond_scale=scale,
                                                 num_sample_steps=5, precision=prec).cpu())
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (noise, scale, prec)
    finally:
        sampler.step_lanes = keep
        sampler.noise_source = "host"
    edm = build_edm_sampler(16)
    keep = edm.step_lanes
    try:
        for noise in ("device", "host"):
            edm.noise_source = noise
            outs = []
            for lanes in (1, 2):
                edm.step_lanes = lanes
                edm.device_noise_seed = 9
                torch.manual_seed(9)
                outs.append(edm.tiled_sample(batch_size=25, condition_x=cond, class_label=label, num_sample_steps=5, precision="bf16").cpu())
            assert torch.equal(outs[0], outs[1]), noise
    finally:
        edm.step_lanes = keep
        edm.noise_source = "host"


@pytest.mark.parametrize("precision", ["f16x3", "f16mx2"])
def test_f16x3_lanes_graphs_and_lockstep_are_bitwise_neutral_at_dim128(precision):
    # the split-operand kernels (persistent conv1x1_split, conv3x3_split with GroupNorm-in-staging, the RMSNorm / GroupNorm-tail
    # epilogues) under everything the sampler does around them, at the production width: two concurrent step lanes (automatic for
    # f16x3), hipGraph replay vs eager launches, two images in lock-step vs their solo runs - all bit-identical; the same for the
    # f16mx2 prototype (conv3x3_mx2 below the tile's resolution)
    import os
    sampler = build_sampler(128)
    conds = torch.cat([C.synthetic_lr_condition(i, 64, 64) for i in range(2)]).cuda()        # 256^2 images: one tile per step each
    big = C.synthetic_lr_condition(0, 256, 256).cuda()                                       # configs[1] geometry: 25 / 16 tiles
    label = torch.tensor([0]).cuda()
    keep = sampler.step_lanes
    try:
        sampler.noise_source = "device"
        sampler.device_noise_seed = 11
        outs = []
        for lanes in (1, 2, None):
            sampler.step_lanes = lanes
            outs.append(sampler.tiled_sample(batch_size=25, condition_x=big, class_label=label, num_sample_steps=5, precision=precision).cpu())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        sampler.step_lanes = None
        for mode in ("0",):
            os.environ["SRGD_GRAPHS"] = mode
            sampler.model._invalidate_engines()
            eager = sampler.tiled_sample(batch_size=25, condition_x=big, class_label=label, num_sample_steps=5, precision=precision).cpu()
        assert torch.equal(eager, outs[0])
        os.environ.pop("SRGD_GRAPHS", None)
        sampler.model._invalidate_engines()
        both = sampler.tiled_sample(batch_size=8, condition_x=conds, class_label=label, num_sample_steps=4, class_cond_scale=1.5,
                                    precision=precision).cpu()
        for i in range(2):
            solo = sampler.tiled_sample(batch_size=8, condition_x=conds[i:i + 1], class_label=label, num_sample_steps=4,
                                        class_cond_scale=1.5, precision=precision).cpu()
            assert torch.equal(both[i:i + 1], solo), i
        assert torch.isfinite(both).all()
    finally:
        os.environ.pop("SRGD_GRAPHS", None)
        sampler.model._invalidate_engines()
        sampler.step_lanes = keep
        sampler.noise_source = "host"


def test_device_noise_mode_full_size_properties():
    # BASELINE config-2 geometry (256^2 LR -> 1024^2, canvas 1280^2, 25/16 tiles), few steps:
    # size-independent properties - output range, determinism per seed, seed sensitivity.
    sampler = build_sampler(16)
    cond = C.synthetic_lr_condition(0, 256, 256).cuda()
    label = torch.tensor([0]).cuda()
    sampler.noise_source = "device"
    try:
        outs = []
        for seed in (71, 71, 72):
            sampler.device_noise_seed = seed
            outs.append(sampler.tiled_sample(batch_size=25, condition_x=cond, class_label=label,
                                             num_sample_steps=4, precision="bf16").cpu())
        assert outs[0].shape == (1, 3, 1024, 1024)
        assert torch.isfinite(outs[0]).all() and outs[0].min() >= 0 and outs[0].max() <= 1
        assert torch.equal(outs[0], outs[1])
        assert not torch.equal(outs[0], outs[2])
    finally:
        sampler.noise_source = "host"


def test_device_rng_moments():
    sampler = build_sampler(16)
    eng = sampler.model.engine("fp32")
    z = eng.randn_(torch.empty(1 << 22, device="cuda"), seed=5, stream_id=9).cpu()
    assert abs(z.mean().item()) < 3e-3 and abs(z.std().item() - 1) < 3e-3
    assert abs((z ** 4).mean().item() - 3.0) < 0.05
    z2 = eng.randn_(torch.empty(1 << 22, device="cuda"), seed=5, stream_id=10).cpu()
    assert abs(float((z * z2).mean())) < 3e-3


def test_error_behaviour_matches_reference():
    sampler = build_sampler(16)
    cond = torch.rand(1, 3, 256, 256).cuda()
    with pytest.raises(NotImplementedError):
        sampler.tiled_sample(condition_x=cond, class_label=torch.tensor([0]).cuda(), num_sample_steps=2,
                             cond_scale=2.0, class_cond_scale=2.0)
    with pytest.raises(RuntimeError):                      # reflect pad >= dim (H <= 256 < W), model.py:3303
        sampler.tiled_sample(condition_x=torch.rand(1, 3, 100, 300).cuda(), num_sample_steps=2)
    with pytest.raises(AssertionError):                    # model.py:679
        sampler.model(torch.zeros(1, 3, 100, 100).cuda(), torch.zeros(1).cuda())
    from srgd_amd._lib import SrgdHipError
    with pytest.raises(SrgdHipError):                      # strict load: engine refuses a missing tensor
        from srgd_amd.engine import HipEngine
        e = HipEngine(dim=16, dim_mults=(1, 2, 4, 8), full_attn=(False, False, False, True), precision="fp32")
        e.load_state_dict({}, strict=True)


def test_with_images_trajectories():
    sampler = build_sampler(16)
    cond = torch.rand(1, 3, 256, 256).cuda()
    torch.manual_seed(1)
    out, imgs, x0s = sampler.tiled_sample(condition_x=cond, class_label=torch.tensor([1]).cuda(), num_sample_steps=3,
                                          with_images=True, with_x0_images=True)
    assert len(imgs) == 4 and len(x0s) == 4 and imgs[-1].shape == (1, 3, 256, 256)
    assert x0s[-1].abs().max() <= 1.0 + 1e-6
    torch.manual_seed(1)
    again = sampler.tiled_sample(condition_x=cond, class_label=torch.tensor([1]).cuda(), num_sample_steps=3)
    assert torch.equal(out, again)


def test_hipgraph_replay_is_bitwise_identical_to_eager_launches():
    # device-noise mode: steps 0/1 run eagerly, 2/3 are captured, 4+ replay the graphs (step-dependent values
    # are read through a device-side step counter).  Must equal the all-eager run bit for bit.
    import os
    sampler = build_sampler(16)
    cond = C.synthetic_lr_condition(3, 96, 96).cuda()            # 384x384 -> canvas 768^2 (9 / 4 tiles)
    label = torch.tensor([2]).cuda()
    outs = {}
    try:
        sampler.noise_source = "device"
        sampler.device_noise_seed = 5
        for mode in ("1", "0"):
            os.environ["SRGD_GRAPHS"] = mode
            sampler.model._invalidate_engines()                   # the switch is read at engine creation
            outs[mode] = sampler.tiled_sample(batch_size=9, condition_x=cond, class_label=label, num_sample_steps=9,
                                              class_cond_scale=1.5, class_guidance_start_steps=3, precision="bf16").cpu()
    finally:
        os.environ.pop("SRGD_GRAPHS", None)
        sampler.model._invalidate_engines()
        sampler.noise_source = "host"
    assert torch.isfinite(outs["1"]).all()
    assert torch.equal(outs["1"], outs["0"])


@pytest.mark.parametrize("prec", ["bf16", "fp8_mixed"])
def test_output_conv_fused_into_last_resnet_block_matches_separate_pass(prec):
    # dim 128: the last ResnetBlock's res_conv epilogue (conv1x1_bf16 EPI_GNTAIL_FINAL) applies the 1x1 output convolution to
    # the bf16 values it would have stored and hands final_step 16 B per pixel (ConvArgs::eps4).  Same inputs, fp32 sums in a
    # different order: after ONE step the predicted x0 agrees with the unfused run (SRGD_FINAL_FUSION=0) to fp32 rounding
    # (x 1/alpha ~ 150 at t = 1); over several steps the bf16 U-Net amplifies that like any other rounding, so the images are
    # compared by PSNR.  One pass, two passes (class CFG) and the EDM wrapper's Heun pair.
    import os
    sampler = build_sampler(128)
    edm = build_edm_sampler(128)
    cond = C.synthetic_lr_condition(4, 64, 64).cuda()            # 256x256: one tile
    label = torch.tensor([1]).cuda()
    outs = {}
    try:
        for smp in (sampler, edm):
            smp.noise_source = "device"
            smp.device_noise_seed = 9
        for mode in ("1", "0"):
            os.environ["SRGD_FINAL_FUSION"] = mode
            sampler.model._invalidate_engines()
            edm.net._invalidate_engines()
            one = sampler.tiled_sample(batch_size=4, condition_x=cond, class_label=label, num_sample_steps=1, precision=prec,
                                       with_images=True, with_x0_images=True)[2][-1]
            outs[mode] = [one.cpu(),
                          sampler.tiled_sample(batch_size=4, condition_x=cond, class_label=label, num_sample_steps=5, precision=prec).cpu(),
                          sampler.tiled_sample(batch_size=4, condition_x=cond, class_label=label, num_sample_steps=5,
                                               class_cond_scale=2.0, precision=prec).cpu(),
                          edm.tiled_sample(batch_size=4, condition_x=cond, class_label=label, num_sample_steps=4, cond_scale=1.5,
                                           precision=prec).cpu()]
    finally:
        os.environ.pop("SRGD_FINAL_FUSION", None)
        sampler.model._invalidate_engines()
        edm.net._invalidate_engines()
        for smp in (sampler, edm):
            smp.noise_source = "host"
    d1 = (outs["1"][0] - outs["0"][0]).abs().max().item()
    psnrs = [float(10 * np.log10(1.0 / max(float(((a - b) ** 2).mean()), 1e-20))) for a, b in zip(outs["1"][1:], outs["0"][1:])]
    _report(test="output_conv_fused_vs_separate", precision=prec, x0_after_one_step_max_abs=d1, psnr_db=psnrs)
    assert all(torch.isfinite(a).all() for a in outs["1"])
    assert d1 <= 5e-4, d1                                        # measured 2.7e-5
    # (sanity only: two fp32-rounding-different but equivalent runs diverge chaotically through the bf16 U-Net - 44 ... 57 dB
    # measured depending on unrelated summation orders elsewhere; the sharp check is the one-step comparison above)
    assert min(psnrs) > 38.0, psnrs


@pytest.mark.parametrize("prec", ["bf16", "fp8_mixed"])
def test_bench_geometry_is_bitwise_repeatable(prec):
    # bench.py's operating point: dim 128, five 256^2 LR images in lock-step -> 125 / 80 tiles per U-Net launch on 1280^2 canvases,
    # device noise, hipGraph replay from step 4 on.  Two runs with the same seed must agree bit for bit, a third with another seed
    # must not; outputs finite and in range (size-independent properties at BASELINE's full geometry).
    sampler = build_sampler(128)
    conds = torch.cat([C.synthetic_lr_condition(i, 256, 256) for i in range(5)]).cuda()
    label = torch.tensor([0]).cuda()
    sampler.noise_source = "device"
    try:
        outs = []
        for seed in (71, 71, 72):
            sampler.device_noise_seed = seed
            outs.append(sampler.tiled_sample(batch_size=125, condition_x=conds, class_label=label, num_sample_steps=8,
                                             precision=prec).cpu())
    finally:
        sampler.noise_source = "host"
    assert outs[0].shape == (5, 3, 1024, 1024)
    assert torch.isfinite(outs[0]).all() and outs[0].min() >= 0 and outs[0].max() <= 1
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("noise,prec", [("host", "fp32"), ("device", "bf16")])
def test_lockstep_images_equal_their_solo_runs(noise, prec):
    # [B,3,H,W] condition: B same-sized images advance together, every U-Net launch spanning tiles of all of them.
    # Each must come out bit-identical to sampling it alone with the same seed (what the reference's per-image
    # seed_everything gives, inference.py:73), for a sub-batch that straddles image boundaries too.
    sampler = build_sampler(16)
    conds = torch.cat([C.synthetic_lr_condition(s, 96, 96) for s in (0, 1, 2)]).cuda()     # 3 x (384^2 -> canvas 768^2)
    label = torch.tensor([1]).cuda()
    sampler.noise_source = noise
    sampler.device_noise_seed = 11
    try:
        solo = []
        for i in range(3):
            torch.manual_seed(9)
            solo.append(sampler.tiled_sample(batch_size=9, condition_x=conds[i:i + 1], class_label=label,
                                             num_sample_steps=5, class_cond_scale=1.3, precision=prec).cpu())
        for bs in (27, 7):
            torch.manual_seed(9)
            both = sampler.tiled_sample(batch_size=bs, condition_x=conds, class_label=label, num_sample_steps=5,
                                        class_cond_scale=1.3, precision=prec).cpu()
            assert both.shape == (3, 3, 384, 384)
            for i in range(3):
                assert torch.equal(both[i:i + 1], solo[i]), (bs, i)
    finally:
        sampler.noise_source = "host"


def test_lockstep_full_size_batch_of_five():
    # the bench shape: 5 x (256^2 LR -> 1024^2) in one run = 125 / 80 tiles per U-Net launch at dim 128, bf16
    # (activation buffers of ~2 GB: exercises 64-bit indexing); image 3 must equal its solo run.
    sampler = build_sampler(128)
    conds = torch.cat([C.synthetic_lr_condition(s, 256, 256) for s in range(5)]).cuda()
    label = torch.tensor([0]).cuda()
    sampler.noise_source = "device"
    sampler.device_noise_seed = 71
    try:
        five = sampler.tiled_sample(batch_size=125, condition_x=conds, class_label=label, num_sample_steps=3, precision="bf16").cpu()
        solo = sampler.tiled_sample(batch_size=25, condition_x=conds[3:4], class_label=label, num_sample_steps=3, precision="bf16").cpu()
    finally:
        sampler.noise_source = "host"
    assert torch.isfinite(five).all()
    assert torch.equal(five[3:4], solo)
    assert not torch.equal(five[0], five[1])


def test_streaming_pointwise_path_equals_generic_path_bitwise():
    # bf16 engine, dim 128: every pointwise layer (res_conv + GroupNorm tail, to_qkv / to_out, both resamplers, the 7x1
    # route of the input conv) runs on conv1x1_bf16.hip; SRGD_CONV1X1=0 sends them through the generic implicit GEMM.
    # Same MFMA shape, same K order, same rounding points -> the eps prediction must not change by a single bit.
    import os
    case = next(c for c in C.UNET_CASES if c["dim"] == 128)
    sampler = build_sampler(128, weight_seed=case["weight_seed"])
    unet = sampler.model
    x, cnd, ls = C.unet_inputs(case)
    label, c = C.unet_mode_args(case["modes"][0], case, cnd)
    outs = {}
    unet.precision = "bf16"
    try:
        for mode in ("1", "0"):
            os.environ["SRGD_CONV1X1"] = mode
            unet._invalidate_engines()
            outs[mode] = unet(x.cuda(), ls.cuda(), None if label is None else label.cuda(),
                              None if c is None else c.cuda()).cpu()
    finally:
        os.environ.pop("SRGD_CONV1X1", None)
        unet._invalidate_engines()
        unet.precision = "fp32"
    assert torch.isfinite(outs["1"]).all()
    assert torch.equal(outs["1"], outs["0"])


@pytest.mark.parametrize("size", [(64, 64), (37, 53), (5, 7), (256, 256), (1, 9)], ids=lambda s: "%dx%d" % s)
def test_device_bicubic_x4_is_bit_exact_with_pillow(size):
    # integer path: the GPU result must equal Pillow's (= T.Resize on a PIL image + ToTensor) bit for bit, and the
    # CPU restatement in oracle/pil_resample.py as well
    from PIL import Image
    from oracle.pil_resample import resize_bicubic_u8, to_unit_chw
    from srgd_amd.inference import pil_to_unit_tensor, upsample_bicubic_on_device
    h, w = size
    rng = np.random.default_rng(h * 1000 + w)
    arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    img = Image.fromarray(arr, "RGB")
    got = upsample_bicubic_on_device(img, 4, torch.device("cuda")).cpu()
    want = pil_to_unit_tensor(img.resize((w * 4, h * 4), Image.BICUBIC)).unsqueeze(0)
    assert got.shape == want.shape
    assert torch.equal(got, want)
    assert np.array_equal(got[0].numpy(), to_unit_chw(resize_bicubic_u8(arr, 4 * h, 4 * w)))


def test_device_unit_to_u8_matches_topilimage():
    from oracle.pil_resample import to_u8_hwc
    from srgd_amd.inference import unit_tensor_to_pil, unit_tensor_to_pil_on_device
    g = torch.Generator().manual_seed(4)
    t = torch.rand(3, 123, 77, generator=g)
    t[0, 0, :4] = torch.tensor([0.0, 1.0, 254.999 / 255, 0.5])
    got = np.asarray(unit_tensor_to_pil_on_device(t.cuda()))
    assert np.array_equal(got, np.asarray(unit_tensor_to_pil(t)))
    assert np.array_equal(got, to_u8_hwc(t.numpy()))


def test_groupnorm_fused_into_conv_staging_matches_separate_pass():
    # SRGD_GN_FUSION=1: GroupNorm1-apply + SiLU runs inside conv2's LDS staging (conv3x3_bf16 GNIN variant) instead of
    # the separate gn_apply pass.  Same formula on the same bf16 inputs, rounded to bf16 at the same point.
    import os
    case = next(c for c in C.UNET_CASES if c["dim"] == 128)
    sampler = build_sampler(128, weight_seed=case["weight_seed"])
    unet = sampler.model
    x, cnd, ls = C.unet_inputs(case)
    label, c = C.unet_mode_args(case["modes"][0], case, cnd)
    outs = {}
    unet.precision = "bf16"
    try:
        for mode in ("1", "0"):
            os.environ["SRGD_GN_FUSION"] = mode
            unet._invalidate_engines()
            outs[mode] = unet(x.cuda(), ls.cuda(), None if label is None else label.cuda(),
                              None if c is None else c.cuda()).cpu()
    finally:
        os.environ.pop("SRGD_GN_FUSION", None)
        unet._invalidate_engines()
        unet.precision = "fp32"
    assert torch.isfinite(outs["1"]).all()
    d = (outs["1"] - outs["0"]).abs().max().item()
    _report(test="gn_fusion_vs_separate", max_abs=d, ref_max=outs["0"].abs().max().item())
    assert d <= 2e-2 * max(1.0, outs["0"].abs().max().item()), d


# ------------------------------------------------------------------ EDM sampler (model.py:2309-2475)
_EDM_MODELS = {}


def build_edm_sampler(dim, steps=32, weight_seed=0, fresh=False):
    key = (dim, weight_seed) if not fresh else (dim, weight_seed, object())
    if key not in _EDM_MODELS:
        import logging
        from srgd_amd.config import load_config
        from srgd_amd.model import get_model
        conf = load_config(os.path.join(os.path.dirname(G), "..", "conf", "conditional_continuous_linear_df8kost_dim128.yaml"))
        conf.unet_dim = dim
        conf.num_sample_steps = steps
        conf.model = "conditional_elucidated"
        ema = get_model(conf, logging.getLogger("test"))
        schema = {"net." + k[len("model."):]: v for k, v in _schema(dim).items()}
        assert list(ema.module.state_dict().keys()) == list(schema.keys())
        ema.module.load_state_dict(synth_state_dict(schema, seed=weight_seed), strict=True)
        _EDM_MODELS[key] = ema.module.eval().to(torch.device("cuda"))
    return _EDM_MODELS.pop(key) if fresh else _EDM_MODELS[key]


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
@pytest.mark.parametrize("case", C.EDM_CASES, ids=lambda c: c["name"])
def test_edm_tiled_sample_fp32_matches_reference(case, precision):
    z = np.load(os.path.join(G, f"sample_edm_{case['name']}.npz"))
    sampler = build_edm_sampler(case["dim"], weight_seed=case["weight_seed"])
    cond = C.sampler_condition(case).cuda()
    label = torch.tensor([case["label"]]).cuda() if case["label"] is not None else None
    torch.manual_seed(case["seed"])
    ctor_steps = sampler.num_sample_steps
    try:
        # the constructor's step count matters when it differs from the per-call one (noised start / ring sigmas)
        sampler.num_sample_steps = case.get("ctor_steps", case["steps"])
        got = sampler.tiled_sample(batch_size=case["batch_size"], condition_x=cond, class_label=label,
                                   cond_scale=case["cond_scale"], class_cond_scale=case["class_cond_scale"],
                                   num_sample_steps=case["steps"], precision=precision, **C.edm_extra_kwargs(case)).cpu()
    finally:
        sampler.num_sample_steps = ctor_steps
    want = torch.from_numpy(z["image"])
    err = (got - want).abs().max().item()
    _report(test="edm_tiled_sample", case=case["name"], precision=precision, max_abs=err)
    assert got.shape == want.shape
    assert err <= 1e-3, err                 # north-star bar
    assert err <= 3e-4, err                 # regression guard


@pytest.mark.parametrize("case", C.EDM_UNTILED_CASES, ids=lambda c: c["name"])
def test_edm_untiled_sample_fp32_matches_reference(case):
    # ConditionalElucidatedDiffusionSR.sample (model.py:2196-2209): the Heun loop sample_org (:2212-2306) or, with
    # use_dpmpp_solver, sample_using_dpmpp (:2479-2557, srgd_edm_dpmpp_step); [B,3,256,256] batches, per-image noise
    z = np.load(os.path.join(G, f"sample_edm_untiled_{case['name']}.npz"))
    sampler = build_edm_sampler(case["dim"], weight_seed=case["weight_seed"])
    cond = C.sample_condition(case).cuda()
    assert abs(cond.double().sum().item() - float(z["cond_sum"])) < 1e-6
    label = torch.tensor([case["label"]]).cuda()
    ctor_steps, ctor_solver = sampler.num_sample_steps, sampler.use_dpmpp_solver
    torch.manual_seed(case["seed"])
    try:
        sampler.num_sample_steps = case.get("ctor_steps", case["steps"])
        sampler.use_dpmpp_solver = case["dpmpp"]
        got, imgs, x0s = sampler.sample(batch_size=case["batch"], condition_x=cond, class_label=label,
                                        cond_scale=case["cond_scale"], class_cond_scale=case["class_cond_scale"],
                                        num_sample_steps=case["steps"], precision="fp32", with_images=True,
                                        with_x0_images=True, **C.edm_extra_kwargs(case))
        if case.get("generation_start_steps", 0) == 0:
            # get_noised_images broadcasts a [B] sigma vector against [B,3,h,w] (model.py:2191-2193): B = 3 fails upstream
            with pytest.raises(RuntimeError):
                sampler.sample(batch_size=3, condition_x=torch.rand(3, 3, 256, 256).cuda(), class_label=label,
                               num_sample_steps=case["steps"], generation_start_steps=1)
    finally:
        sampler.num_sample_steps, sampler.use_dpmpp_solver = ctor_steps, ctor_solver
    got = got.cpu()
    want = torch.from_numpy(z["image"])
    err = (got - want).abs().max().item()
    _report(test="edm_untiled_sample", case=case["name"], precision="fp32", max_abs=err)
    assert got.shape == want.shape
    assert err <= 1e-3 and err <= 3e-4, err
    executed = case["steps"] - case.get("generation_start_steps", 0)
    assert len(imgs) == len(x0s) == executed + 1 and imgs[-1].shape == want.shape
    # the last trajectory entry is the un-clamped final image (model.py:2291 / :2545 before :2298 / :2549)
    assert ((imgs[-1].clamp(-1, 1) + 1) * 0.5 - got).abs().max().item() <= 1e-6


def test_noised_starts_in_device_noise_mode_are_deterministic():
    # generation_start_steps > 0 (q_sample of the condition, model.py:3305-3310) and start_white_noise=False (:3311-3315) with the
    # in-engine Philox generator: finite, in range, bit-repeatable per seed, seed-sensitive, different from the white-noise start;
    # same for the EDM wrapper's noised start (get_noised_images, :2186-2194)
    ddpm, edm = build_sampler(16), build_edm_sampler(16)
    cond = C.synthetic_lr_condition(2, 75, 75).cuda()           # 300 x 300 -> 768^2 canvas
    label = torch.tensor([0]).cuda()
    try:
        for smp in (ddpm, edm):
            smp.noise_source = "device"

        def run(smp, seed, **kw):
            smp.device_noise_seed = seed
            return smp.tiled_sample(batch_size=9, condition_x=cond, class_label=label, num_sample_steps=6, precision="bf16", **kw).cpu()

        base = run(ddpm, 5)
        for kw in (dict(generation_start_steps=2), dict(start_white_noise=False)):
            a, b, c = run(ddpm, 5, **kw), run(ddpm, 5, **kw), run(ddpm, 6, **kw)
            assert torch.isfinite(a).all() and a.min() >= 0 and a.max() <= 1
            assert torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, base), kw
        ebase = run(edm, 5)
        a, b, c = run(edm, 5, generation_start_steps=2), run(edm, 5, generation_start_steps=2), run(edm, 6, generation_start_steps=2)
        assert torch.isfinite(a).all() and torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a, ebase)
    finally:
        for smp in (ddpm, edm):
            smp.noise_source = "host"


def test_untiled_entry_points_bf16_dim128_close_to_fp32_engine():
    # the un-tiled loops (DDPM sample, EDM sample_org / sample_using_dpmpp) on the production kernels: dim 128, batch of two
    # 256^2 images as stacked tiles, bf16 against the fp32 engine on the same host noise
    ddpm, edm = build_sampler(128), build_edm_sampler(128)
    cond = torch.cat([C.synthetic_lr_condition(i, 64, 64) for i in (0, 1)]).cuda()
    label = torch.tensor([1]).cuda()
    ctor_solver = edm.use_dpmpp_solver
    res = {}
    try:
        for name, fn in (("ddpm_sample", lambda p: ddpm.sample(batch_size=2, condition_x=cond, class_label=label, num_sample_steps=6,
                                                                class_cond_scale=1.5, precision=p)),
                         ("edm_heun", lambda p: edm.sample(batch_size=2, condition_x=cond, class_label=label, num_sample_steps=5,
                                                           precision=p)),
                         ("edm_dpmpp", lambda p: edm.sample(batch_size=2, condition_x=cond, class_label=label, num_sample_steps=6,
                                                            cond_scale=1.5, precision=p))):
            edm.use_dpmpp_solver = name == "edm_dpmpp"
            outs = {}
            for prec in ("fp32", "bf16"):
                torch.manual_seed(11)
                outs[prec] = fn(prec).cpu()
            mse = float(((outs["bf16"] - outs["fp32"]) ** 2).mean())
            res[name] = float(10 * np.log10(1.0 / max(mse, 1e-20)))
            assert outs["bf16"].shape == (2, 3, 256, 256) and torch.isfinite(outs["bf16"]).all()
            assert outs["bf16"].min() >= 0 and outs["bf16"].max() <= 1
    finally:
        edm.use_dpmpp_solver = ctor_solver
    _report(test="untiled_bf16_vs_fp32_engine_dim128", psnr_db=res)
    assert min(res.values()) > 48.0, res          # measured 51.5 / 55.3 / 55.8 dB


def test_edm_bf16_and_device_noise_modes_run():
    case = C.EDM_CASES[0]
    z = np.load(os.path.join(G, f"sample_edm_{case['name']}.npz"))
    sampler = build_edm_sampler(case["dim"], weight_seed=case["weight_seed"])
    cond = C.sampler_condition(case).cuda()
    label = torch.tensor([case["label"]]).cuda()
    torch.manual_seed(case["seed"])
    bf = sampler.tiled_sample(batch_size=4, condition_x=cond, class_label=label, num_sample_steps=case["steps"], precision="bf16").cpu()
    mse = float(((bf - torch.from_numpy(z["image"])) ** 2).mean())
    _report(test="edm_tiled_sample", case=case["name"], precision="bf16", psnr_db=10 * np.log10(1.0 / max(mse, 1e-20)))
    assert torch.isfinite(bf).all() and bf.min() >= 0 and bf.max() <= 1
    sampler.noise_source = "device"
    try:
        outs = []
        for seed in (3, 3, 4):
            sampler.device_noise_seed = seed
            outs.append(sampler.tiled_sample(batch_size=4, condition_x=cond, class_label=label, num_sample_steps=4, precision="bf16").cpu())
        assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])
    finally:
        sampler.noise_source = "host"


def test_config5_cfg2_100_steps_fp32_parity_and_bf16_fp8_weight_reports():
    # BASELINE configs[4] at one-tile geometry: 100 DDPM steps, class_cond_scale 2.0 (both passes in one launch),
    # dim-128 U-Net.  fp32 engine gated at the north-star bar against the reference's own output; bf16 and the
    # fp8-e4m3-weight mode (bf16 kernels, weights rounded through e4m3 per output channel) are reported against it
    # and against each other ("parity-vs-bf16 check").
    case = C.LONG_CASES[0]
    z = np.load(os.path.join(G, f"sample_{case['name']}.npz"))
    want = torch.from_numpy(z["image"])
    sampler = build_sampler(case["dim"], weight_seed=case["weight_seed"])
    cond = C.sampler_condition(case).cuda()
    assert abs(cond.double().sum().item() - float(z["cond_sum"])) < 1e-6
    label = torch.tensor([case["label"]]).cuda()
    outs = {}
    for mode in ("fp32", "f16x3", "bf16", "bf16_w8", "fp8", "fp8_mixed"):
        torch.manual_seed(case["seed"])
        outs[mode] = sampler.tiled_sample(batch_size=case["batch_size"], condition_x=cond, class_label=label,
                                          class_cond_scale=case["class_cond_scale"], num_sample_steps=case["steps"],
                                          precision=mode).cpu()
    psnr = lambda a, b: float(10 * np.log10(1.0 / max(float(((a - b) ** 2).mean()), 1e-20)))
    err32 = (outs["fp32"] - want).abs().max().item()
    errx3 = (outs["f16x3"] - want).abs().max().item()
    assert errx3 <= 1e-3, errx3
    _report(test="config5_256", fp32_max_abs=err32, f16x3_max_abs=errx3, bf16_psnr_vs_ref=psnr(outs["bf16"], want),
            w8_psnr_vs_ref=psnr(outs["bf16_w8"], want), w8_psnr_vs_bf16=psnr(outs["bf16_w8"], outs["bf16"]),
            w8_max_abs_vs_bf16=(outs["bf16_w8"] - outs["bf16"]).abs().max().item(),
            fp8_psnr_vs_ref=psnr(outs["fp8"], want), fp8_mixed_psnr_vs_ref=psnr(outs["fp8_mixed"], want),
            fp8_psnr_vs_bf16=psnr(outs["fp8"], outs["bf16"]), fp8_mixed_psnr_vs_bf16=psnr(outs["fp8_mixed"], outs["bf16"]),
            fp8_max_abs_vs_ref=(outs["fp8"] - want).abs().max().item(),
            fp8_mixed_max_abs_vs_ref=(outs["fp8_mixed"] - want).abs().max().item())
    assert err32 <= 1e-3, err32
    for k in ("bf16", "bf16_w8", "fp8", "fp8_mixed"):
        assert torch.isfinite(outs[k]).all() and outs[k].min() >= 0 and outs[k].max() <= 1
    # the MX-fp8 modes against the REFERENCE's own output for configs[4], gated within 3 dB of the
    # measurement on MI355X (profiles/r3_parity_report.jsonl)
    assert psnr(outs["fp8"], want) > FP8_CONFIG5_GATE_DB, psnr(outs["fp8"], want)
    assert psnr(outs["fp8_mixed"], want) > FP8_MIXED_CONFIG5_GATE_DB, psnr(outs["fp8_mixed"], want)
    assert psnr(outs["fp8_mixed"], want) > psnr(outs["fp8"], want)
    assert not torch.equal(outs["bf16"], outs["bf16_w8"])
    assert psnr(outs["bf16"], want) > 54.0                # measured 57.9 dB vs the reference
    assert psnr(outs["bf16_w8"], outs["bf16"]) > 32.0     # measured 35.2 dB (random-init weights, per-channel e4m3 scales)


# Gates of the throughput modes against the REFERENCE's configs[1] output at full length (50 steps): set 3 dB under the first
# measurement on MI355X (profiles/r6_parity_report.jsonl); moved up only.
CONFIG2_FULL_GATES_DB = {"bf16": 55.5, "fp8_mixed": 49.5, "fp8": 33.7}      # measured 58.50 / 52.56 / 36.72 dB (round 6)


def _full_length_check(case, gates, tag):
    """A full-length run against the REFERENCE's own output and trajectory (tests/golden/make_golden.py --config2-full /
    --config5-full): the parity modes (fp32, f16x3) gated at the north-star bar on the final image and followed through the
    trajectory (with_images / with_x0_images, model.py:3398-3401); the throughput modes reported as PSNR against the reference."""
    path = os.path.join(G, f"sample_{case['name']}.npz")
    if not os.path.exists(path):
        pytest.skip(f"fixture {os.path.basename(path)} not generated in this tree")
    z = np.load(path)
    want = torch.from_numpy(z["image_u16"].astype(np.float32) / 65535.0)
    assert abs(want.double().sum().item() - float(z["checksum"])) < 3.2e6 * 7.7e-6
    sampler = build_sampler(case["dim"], weight_seed=case["weight_seed"])
    cond = C.sampler_condition(case).cuda()
    assert abs(cond.double().sum().item() - float(z["cond_sum"])) < 1e-6
    label = torch.tensor([case["label"]]).cuda()
    sampler.noise_source = "host"
    psnr_of = lambda a: float(10 * np.log10(1.0 / max(float(((a - want) ** 2).mean()), 1e-20)))
    steps = [int(i) for i in z["trace_steps"]]
    kw = dict(batch_size=case["batch_size"], condition_x=cond, class_label=label, num_sample_steps=case["steps"],
              cond_scale=case["cond_scale"], class_cond_scale=case["class_cond_scale"])
    for prec in ("fp32", "f16x3"):
        torch.manual_seed(case["seed"])
        out, xts, x0s = sampler.tiled_sample(precision=prec, with_images=True, with_x0_images=True, **kw)
        out = out.cpu()
        assert len(xts) == case["steps"] + 1 and len(x0s) == case["steps"] + 1
        err = float((out - want).abs().max())
        # trajectory: the reference's canvases after the traced steps (8x-subsampled planes) and fp64 checksums of every step
        xt_err = {i: float((C.trace_planes(xts[i + 1]) - torch.from_numpy(z[f"xt_{i}"])).abs().max()) for i in steps}
        x0_err = {i: float((C.trace_planes(x0s[i + 1]) - torch.from_numpy(z[f"x0_{i}"])).abs().max()) for i in steps}
        xt_abs = np.array([t.double().abs().sum().item() for t in xts[1:]])
        x0_abs = np.array([t.double().abs().sum().item() for t in x0s[1:]])
        rel_xt = float(np.max(np.abs(xt_abs - z["xt_abs"]) / z["xt_abs"]))
        rel_x0 = float(np.max(np.abs(x0_abs - z["x0_abs"]) / z["x0_abs"]))
        del xts, x0s
        _report(test=tag, precision=prec, max_abs=err, mean_abs=float((out - want).abs().mean()),
                xt_max_abs_at=xt_err, x0_max_abs_at=x0_err, xt_abs_checksum_rel=rel_xt, x0_abs_checksum_rel=rel_x0)
        assert out.shape == (1, 3, case["h"], case["w"])
        assert err <= 1e-3, (prec, err)                              # north-star bar, against the reference itself, at full length
        # x_t carries N(0,1)-scale noise: summation-order differences stay ~1e-5; x_start = (x - sigma eps) / alpha amplifies eps
        # differences by 1 / alpha (148x at step 0) before the clamp
        assert max(xt_err.values()) <= 2e-3, (prec, xt_err)
        assert max(x0_err.values()) <= 2e-2 and x0_err[steps[-1]] <= 2e-3, (prec, x0_err)
        assert rel_xt <= 1e-5 and rel_x0 <= 1e-4, (prec, rel_xt, rel_x0)
    rep = {}
    for prec in gates:
        torch.manual_seed(case["seed"])
        out = sampler.tiled_sample(precision=prec, **kw).cpu()
        assert torch.isfinite(out).all() and out.min() >= 0 and out.max() <= 1
        rep[prec] = dict(psnr_db=psnr_of(out), max_abs=float((out - want).abs().max()), mean_abs=float((out - want).abs().mean()))
    _report(test=tag, throughput_modes=rep)
    for prec, gate in gates.items():
        assert rep[prec]["psnr_db"] > gate, (prec, rep[prec])


def test_config2_full_length_against_the_reference():
    # BASELINE configs[1] exactly as written - 256^2 LR -> 1024^2, canvas 1280^2, 25 / 16 tiles, 50 DDPM steps, class_cond_scale 1.0,
    # dim 128, batch_size 8, seed 71: 1,025 tile-forwards through /root/reference/model.py:3288-3413 (about two hours of CPU in the
    # build container).  Until round 5 this run was checked engine-against-engine only.
    _full_length_check(C.FULL_CASES[0], CONFIG2_FULL_GATES_DB, "config2_full_vs_reference")


# configs[4] at its full geometry against the reference (measured on MI355X: fp32 1.10e-5, f16x3 1.03e-5 max-abs; bf16 57.51 dB,
# fp8_mixed 52.23 dB, fp8 36.32 dB): gates 3 dB under the measurement
CONFIG5_FULL_GATES_DB = {"bf16": 54.5, "fp8_mixed": 49.2, "fp8": 33.3}


def test_config5_full_geometry_against_the_reference():
    # BASELINE configs[4] at full geometry: the same 1024^2 image, 100 DDPM steps, class_cond_scale 2.0 (both passes in one launch) =
    # 4,100 tile-forwards through the reference (about four hours of CPU: make_golden.py --config5-full)
    _full_length_check(C.FULL5_CASES[0], CONFIG5_FULL_GATES_DB, "config5_full_vs_reference")


def test_edm_lockstep_images_equal_their_solo_runs():
    sampler = build_edm_sampler(16)
    conds = torch.cat([C.synthetic_lr_condition(s, 96, 96) for s in (0, 1)]).cuda()       # 2 x (384^2 -> canvas 768^2)
    label = torch.tensor([1]).cuda()
    solo = []
    for i in range(2):
        torch.manual_seed(9)
        solo.append(sampler.tiled_sample(batch_size=9, condition_x=conds[i:i + 1], class_label=label, num_sample_steps=4,
                                         class_cond_scale=1.3, precision="fp32").cpu())
    for bs in (18, 5):
        torch.manual_seed(9)
        both = sampler.tiled_sample(batch_size=bs, condition_x=conds, class_label=label, num_sample_steps=4,
                                    class_cond_scale=1.3, precision="fp32").cpu()
        assert both.shape == (2, 3, 384, 384)
        for i in range(2):
            assert torch.equal(both[i:i + 1], solo[i]), (bs, i)


@pytest.mark.parametrize("case", C.SAMPLE_CASES, ids=lambda c: c["name"])
def test_untiled_sample_fp32_matches_reference(case):
    # sample() / p_sample_loop (model.py:3191-3247, :3417-3432): a batch of independent 256^2 images, per-image noise
    z = np.load(os.path.join(G, f"sample_untiled_{case['name']}.npz"))
    sampler = build_sampler(case["dim"], steps=case["steps"], weight_seed=case["weight_seed"])
    cond = C.sample_condition(case).cuda()
    torch.manual_seed(case["seed"])
    got = sampler.sample(batch_size=case["batch"], condition_x=cond, class_label=torch.tensor([case["label"]]).cuda(),
                         cond_scale=case["cond_scale"], class_cond_scale=case["class_cond_scale"],
                         num_sample_steps=case["steps"], **C.sample_extra_kwargs(case)).cpu()
    want = torch.from_numpy(z["image"])
    err = (got - want).abs().max().item()
    _report(test="untiled_sample", case=case["name"], precision="fp32", max_abs=err)
    assert got.shape == want.shape
    assert err <= 1e-3 and err <= 2e-4, err
    with pytest.raises(ValueError):
        sampler.sample(batch_size=case["batch"] + 1, condition_x=cond, num_sample_steps=2)


def test_edm_hipgraph_replay_is_bitwise_identical_to_eager_launches():
    import os
    sampler = build_edm_sampler(16)
    cond = C.synthetic_lr_condition(3, 96, 96).cuda()            # 384x384 -> canvas 768^2 (9 / 4 tiles)
    label = torch.tensor([2]).cuda()
    outs = {}
    try:
        sampler.noise_source = "device"
        sampler.device_noise_seed = 5
        for mode in ("1", "0"):
            os.environ["SRGD_GRAPHS"] = mode
            sampler.net._invalidate_engines()
            outs[mode] = sampler.tiled_sample(batch_size=9, condition_x=cond, class_label=label, num_sample_steps=9,
                                              class_cond_scale=1.5, class_guidance_start_steps=3, precision="bf16").cpu()
    finally:
        os.environ.pop("SRGD_GRAPHS", None)
        sampler.net._invalidate_engines()
        sampler.noise_source = "host"
    assert torch.isfinite(outs["1"]).all()
    assert torch.equal(outs["1"], outs["0"])


def test_graph_cache_survives_scratch_growth_between_guidance_modes():
    # with device noise, a passes=2 step grows the per-launch scratch (GroupNorm partials, coefficient rows, ...);
    # graphs captured for passes=1 before that hold the OLD pointers.  Guidance toggled 1 -> 2 -> 1 inside one run (legal for
    # a C-ABI caller) must still equal the all-eager run bit for bit.
    import os
    from srgd_amd.model import _schedule, get_coord_and_pad, get_coords, get_area
    from srgd_amd._lib import SamplerGeometry
    sampler = build_sampler(16)
    dev = sampler.device
    cond = C.synthetic_lr_condition(4, 96, 96).to(dev)             # 384^2 -> canvas 768^2 (9 / 4 tiles)
    _, _, h, w = cond.shape
    (left, top, right, bottom), pad = get_coord_and_pad(h, w)
    hp, wp = h + pad[2] + pad[3], w + pad[0] + pad[1]
    c0 = get_coords(hp, wp, 256, 256)
    c1 = get_coords(hp - 256, wp - 256, 256, 256, diff=128)
    (sl, st_, sr, sb), _ = get_area(c1, hp, wp)
    geo = SamplerGeometry(H=h, W=w, Hp=hp, Wp=wp, left=left, top=top, inner_l=sl, inner_t=st_, inner_r=sr, inner_b=sb,
                          tile=256, n_even=len(c0), n_odd=len(c1), n_images=1)
    n = 14
    scalars, log_snrs = _schedule(n)
    passes_of = [1] * 5 + [2] * 4 + [1] * 5          # 1 (eager, captured, replayed) -> 2 (scratch grows) -> 1 (replay again)
    outs = {}
    try:
        for mode in ("1", "0"):
            os.environ["SRGD_GRAPHS"] = mode
            sampler.model._invalidate_engines()
            eng = sampler.model.engine("bf16")
            cc = torch.empty(1, 3, hp, wp, device=dev)
            eng.sampler_begin(geo, cond, cc, [(a, c_) for (a, _, c_, _) in c0], [(a, c_) for (a, _, c_, _) in c1], scalars,
                              log_snrs, 1)
            img = eng.randn_(torch.empty(1, 3, hp, wp, device=dev), 9, 0)
            for i in range(n):
                p = passes_of[i]
                eng.sampler_step(i, img, cc, None, None, None, p, 1 if p == 2 else 0, 1.5 if p == 2 else 1.0, 9, seed=9)
            out = torch.empty(1, 3, h, w, device=dev)
            eng.sampler_end(img, out)
            outs[mode] = out.cpu()
    finally:
        os.environ.pop("SRGD_GRAPHS", None)
        sampler.model._invalidate_engines()
    assert torch.isfinite(outs["1"]).all()
    assert torch.equal(outs["1"], outs["0"])


# ------------------------------------------------------------------ BASELINE configs[1] at its real geometry and width
def test_config2_geometry_dim128_matches_reference_fp32_and_bf16():
    # 256^2 LR -> 1024^2, canvas 1280^2, 25 (even) / 16 (odd) tiles, dim 128, batch_size 8 (ragged
    # minibatches), 2 DDPM steps = 41 tile-forwards through the reference's minibatch loop, scatter and odd-step ring re-noise
    # (model.py:3364-3396).  The fixture is the REFERENCE's output (uint16 steps of 1/65535: +-7.7e-6).
    case = C.WIDE_CASES[0]
    z = np.load(os.path.join(G, f"sample_{case['name']}.npz"))
    want = torch.from_numpy(z["image_u16"].astype(np.float32) / 65535.0)
    assert abs(want.double().sum().item() - float(z["checksum"])) < 3.2e6 * 7.7e-6     # the dequantised fixture is intact
    sampler = build_sampler(case["dim"], weight_seed=case["weight_seed"])
    cond = C.sampler_condition(case).cuda()
    assert abs(cond.double().sum().item() - float(z["cond_sum"])) < 1e-6
    label = torch.tensor([case["label"]]).cuda()
    sampler.noise_source = "host"
    outs = {}
    for prec in ("fp32", "f16x3", "bf16", "fp8", "fp8_mixed"):
        torch.manual_seed(case["seed"])
        outs[prec] = sampler.tiled_sample(batch_size=case["batch_size"], condition_x=cond, class_label=label,
                                          num_sample_steps=case["steps"], precision=prec).cpu()
    e32 = (outs["fp32"] - want).abs()
    ex3 = (outs["f16x3"] - want).abs()
    assert float(ex3.max()) <= 1e-3 and float(ex3.max()) <= 3e-4, float(ex3.max())     # the split-operand mode: same bar, same guard
    ebf = (outs["bf16"] - want).abs()
    psnr_of = lambda a: float(10 * np.log10(1.0 / max(float(((a - want) ** 2).mean()), 1e-20)))
    psnr = psnr_of(outs["bf16"])
    _report(test="config2_geometry_2steps", case=case["name"], fp32_max_abs=float(e32.max()), f16x3_max_abs=float(ex3.max()), bf16_max_abs=float(ebf.max()),
            bf16_mean_abs=float(ebf.mean()), bf16_psnr_vs_reference=psnr, fp8_psnr_vs_reference=psnr_of(outs["fp8"]),
            fp8_mixed_psnr_vs_reference=psnr_of(outs["fp8_mixed"]))
    # MX-fp8 modes against the reference at configs[1]'s real geometry; gates within 3 dB of the measurement
    for k in ("fp8", "fp8_mixed"):
        assert torch.isfinite(outs[k]).all() and outs[k].min() >= 0 and outs[k].max() <= 1
    assert psnr_of(outs["fp8"]) > FP8_CONFIG2_GATE_DB, psnr_of(outs["fp8"])
    assert psnr_of(outs["fp8_mixed"]) > FP8_MIXED_CONFIG2_GATE_DB, psnr_of(outs["fp8_mixed"])
    assert outs["fp32"].shape == (1, 3, 1024, 1024)
    assert float(e32.max()) <= 1e-3                        # north-star bar, against the reference itself
    assert float(e32.max()) <= 3e-4                        # regression guard (fp32 residual is summation order only)
    assert torch.isfinite(outs["bf16"]).all() and outs["bf16"].min() >= 0 and outs["bf16"].max() <= 1
    assert psnr > 40.5, psnr                               # measured 43.5 dB on MI355X (2 steps from pure noise: the image is still mostly noise, a clamp flip costs up to 0.67)
    # sub-batching is invisible: 25 tiles in one launch == the reference's minibatches of 8
    torch.manual_seed(case["seed"])
    again = sampler.tiled_sample(batch_size=25, condition_x=cond, class_label=label, num_sample_steps=case["steps"],
                                 precision="fp32").cpu()
    assert torch.equal(again, outs["fp32"])


# ------------------------------------------------------------------ BASELINE configs[3] geometry on one GPU (property test)
def test_config4_geometry_8448_canvas_properties():
    # 2048^2 LR -> 8192^2 HR: canvas 8448^2, 1089 (even) / 1024 (odd) tiles per step.  The reference needs ~15 h of CPU for
    # one step here, so the properties the domain offers are checked instead: finite, in range, deterministic, and invariant to
    # how the tile list is cut into launches (125 vs 64 tiles per launch, bit-identical) - plus the activation tensors of a
    # 125-tile launch (4.2 GB at 256^2 x 128 ch bf16) exercise every >2^31-byte offset path of the kernels.
    sampler = build_sampler(128)
    lr = 2048
    cond = C.synthetic_lr_condition(0, lr, lr).cuda()
    assert cond.shape == (1, 3, 8192, 8192)
    label = torch.tensor([0]).cuda()
    sampler.noise_source = "device"
    outs = []
    try:
        for sub in (125, 64, 125):
            sampler.device_noise_seed = 71
            outs.append(sampler.tiled_sample(batch_size=sub, condition_x=cond, class_label=label, num_sample_steps=3,
                                             precision="bf16"))
            torch.cuda.synchronize()
    finally:
        sampler.noise_source = "host"
    a, b, c = outs
    assert a.shape == (1, 3, 8192, 8192)
    assert bool(torch.isfinite(a).all()) and float(a.min()) >= 0.0 and float(a.max()) <= 1.0
    assert torch.equal(a, c), "same seed, same launch shape: must repeat bit for bit"
    assert torch.equal(a, b), "tiles are independent within a step: 125 vs 64 tiles per launch must not change a bit"
    assert float(a.std()) > 0.05                           # not a constant image
    del outs, a, b, c
    torch.cuda.empty_cache()


@pytest.mark.parametrize("hw", [(64, 64), (192, 320), (8, 8)], ids=lambda s: f"{s[0]}x{s[1]}")
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_unet_forward_accepts_any_size_divisible_by_8(hw, precision):
    # model.py:679 only asks for H, W divisible by the down-sampling factor (8); 64x64 is BASELINE configs[0]'s tile.
    # Sizes whose pixel counts are not multiples of the GEMM tiles run with masked tail tiles.
    dim = 16
    sampler = build_sampler(dim)
    unet = sampler.model
    sd = O.strip_model_prefix(synth_state_dict(_schema(dim), seed=0))
    g = torch.Generator().manual_seed(77)
    h, w = hw
    x = torch.randn(2, 3, h, w, generator=g)
    cnd = torch.rand(2, 3, h, w, generator=g) * 2 - 1
    ls = torch.tensor([-1.5, 3.0])
    lab = torch.tensor([2])
    with torch.inference_mode():
        want = O.unet_forward(sd, O.UnetCfg(dim=dim), x, ls, lab, cnd)
    unet.precision = precision
    try:
        got = unet(x.cuda(), ls.cuda(), lab.cuda(), cnd.cuda()).cpu()
    finally:
        unet.precision = "fp32"
    err = (got - want).abs().max().item()
    scale = max(1.0, want.abs().max().item())
    _report(test="unet_forward_any_size", hw=list(hw), precision=precision, max_abs=err, ref_max=scale)
    assert err <= (1e-4 if precision == "fp32" else 4e-2) * scale, err


def test_mixed_class_labels_are_refused_not_collapsed():
    sampler = build_sampler(16)
    cond = torch.rand(2, 3, 256, 256).cuda()
    with pytest.raises(NotImplementedError):
        sampler.sample(batch_size=2, condition_x=cond, class_label=torch.tensor([0, 2]).cuda(), num_sample_steps=2)
    out = sampler.sample(batch_size=2, condition_x=cond, class_label=torch.tensor([1, 1]).cuda(), num_sample_steps=2)
    assert out.shape == (2, 3, 256, 256)


# ------------------------------------------------------------------ BASELINE configs[4]: fp8 (MX) compute path
def test_fp8_unet_forward_vs_reference_and_bf16():
    # one U-Net evaluation in fp8 mode (every 3x3 convolution on v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 weights and activations
    # with a scale per 32 channels) against the REFERENCE's eps (fixture) and against the bf16 engine
    case = next(c for c in C.UNET_CASES if c["name"] == "dim128_128")
    z = np.load(os.path.join(G, "unet_eps.npz"))
    sampler = build_sampler(case["dim"], weight_seed=case["weight_seed"])
    unet = sampler.model
    x, cnd, ls = C.unet_inputs(case)
    label, c = C.unet_mode_args("label_cond", case, cnd)
    outs = {}
    try:
        for prec in ("bf16", "fp8"):
            unet.precision = prec
            outs[prec] = unet(x.cuda(), ls.cuda(), label.cuda(), c.cuda()).cpu()
    finally:
        unet.precision = "fp32"
    want = torch.from_numpy(z[f"{case['name']}.label_cond"])
    rel = lambda a, b: float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt())
    _report(test="fp8_unet_forward", rel_rms_vs_reference=rel(outs["fp8"], want), rel_rms_vs_bf16=rel(outs["fp8"], outs["bf16"]),
            bf16_rel_rms_vs_reference=rel(outs["bf16"], want), max_abs_vs_reference=float((outs["fp8"] - want).abs().max()))
    assert torch.isfinite(outs["fp8"]).all()
    assert not torch.equal(outs["fp8"], outs["bf16"])          # the fp8 kernels really ran
    assert rel(outs["fp8"], want) < 0.085                       # frozen at round 3's gate; measured 0.0630 with MX-e4m3 attention weights (round 3: 0.0566; 0.052 with bf16 pointwise layers): e4m3 carries 3 mantissa bits, 40 + 13 layers deep


def test_fp8_mode_uses_the_mxfp8_kernels():
    sampler = build_sampler(128)
    eng = sampler.model.engine("fp8")
    cond = C.synthetic_lr_condition(0, 64, 64).cuda()
    sampler.noise_source = "device"
    try:
        eng.profile_begin()
        out = sampler.tiled_sample(batch_size=4, condition_x=cond, class_label=torch.tensor([0]).cuda(), num_sample_steps=2,
                                   precision="fp8")
        prof = eng.profile_end()
    finally:
        sampler.noise_source = "host"
    assert torch.isfinite(out).all() and out.min() >= 0 and out.max() <= 1
    assert prof["launches"]["conv3x3_mxfp8"] == 2 * 40 and prof["launches"]["conv3x3_bf16"] == 0     # all 40 3x3 convs per forward
    # every tensor a 3x3 convolution reads gets its MX-fp8 twin from its producer's epilogue (conv1x1 variants, GroupNorm2 +
    # residual, both fused LinearAttention kernels, the 3x3 resamplers): no stand-alone quantisation pass is left
    assert prof["launches"]["quantize_mxfp8"] == 0
    # the pointwise layers whose inputs have MX-fp8 twins run on conv1x1_mxfp8 (13 of the 32 pointwise launches per forward)
    assert prof["launches"]["conv1x1_mxfp8"] == 2 * 13, prof["launches"]
    # fp8_mixed: the 11 convolutions at the tile's own resolution (first down stage 4, last up stage 4 + its 3x3 resampler, final
    # block 2) stay on the bf16 kernel, the other 29 run MX-fp8; twins are written only for tensors an MX convolution reads, and
    # still no stand-alone quantisation pass is needed
    eng = sampler.model.engine("fp8_mixed")
    sampler.noise_source = "device"
    try:
        eng.profile_begin()
        out = sampler.tiled_sample(batch_size=4, condition_x=cond, class_label=torch.tensor([0]).cuda(), num_sample_steps=2,
                                   precision="fp8_mixed")
        prof = eng.profile_end()
    finally:
        sampler.noise_source = "host"
    assert torch.isfinite(out).all() and out.min() >= 0 and out.max() <= 1
    assert prof["launches"]["conv3x3_mxfp8"] == 2 * 29 and prof["launches"]["conv3x3_bf16"] == 2 * 11
    assert prof["launches"]["quantize_mxfp8"] == 0
    # round 4: fp8_mixed is the quality-oriented fp8 mode - its pointwise layers stay on conv1x1_bf16 unless SRGD_MX1X1=1 asks
    # for the MX kernel (the next test): +3 % throughput was not worth 1.2 dB there
    assert prof["launches"]["conv1x1_mxfp8"] == 0, prof["launches"]


def test_fp8_pointwise_layers_on_the_mx_kernel_and_the_switch_back():
    # ("fp8 ... attention weights"; SRGD_MX1X1=0 switches back): pointwise layers whose input tensors have MX-fp8 twins run
    # on conv1x1_mxfp8 - the 8 res_convs of the up stages + the final block's (with the fused output convolution), the 3
    # Downsample 1x1s, the PixelShuffle 1x1 behind the first up stage's softmax attention = 13 of the 32 pointwise launches per
    # forward; to_qkv / to_out of the unfused attention sites (inputs come out of RMSNorm / the attention core) and the input
    # convolution stay on conv1x1_bf16.  In fp8_mixed the 256^2 zones' pointwise layers stay bf16 as well: 9.  The image must
    # stay close to the bf16-pointwise path's (same 3x3 kernels, 13 more e4m3 layers).
    import os
    sampler = build_sampler(128)
    cond = C.synthetic_lr_condition(0, 64, 64).cuda()
    label = torch.tensor([0]).cuda()
    sampler.noise_source = "device"
    outs = {}
    try:
        for knob in ("0", "1"):
            os.environ["SRGD_MX1X1"] = knob
            sampler.model._invalidate_engines()                     # the switch is read at engine creation
            for prec, want in (("fp8", 13), ("fp8_mixed", 9)):
                eng = sampler.model.engine(prec)
                eng.profile_begin()
                out = sampler.tiled_sample(batch_size=4, condition_x=cond, class_label=label, num_sample_steps=2, precision=prec)
                prof = eng.profile_end()
                assert torch.isfinite(out).all() and out.min() >= 0 and out.max() <= 1
                assert prof["launches"]["conv1x1_mxfp8"] == (2 * want if knob == "1" else 0), (knob, prec, prof["launches"])
                assert prof["launches"]["quantize_mxfp8"] == 0
                outs[knob, prec] = out.cpu()
    finally:
        os.environ.pop("SRGD_MX1X1", None)
        sampler.model._invalidate_engines()
        sampler.noise_source = "host"
    for prec in ("fp8", "fp8_mixed"):
        mse = float(((outs["1", prec] - outs["0", prec]) ** 2).mean())
        psnr = 10 * np.log10(1.0 / max(mse, 1e-12))
        _report(test="fp8_mx1x1_vs_default", precision=prec, psnr_db=psnr)
        assert psnr > 25.0, (prec, psnr)            # 2 steps from pure noise: the image is mostly noise, small eps errors show


def test_fp8_modes_carry_mx_e4m3_attention_weights():
    # BASELINE configs[4] "fp8 conv + attention weights" (VERDICT r3 item 3): in the fp8 modes to_qkv / to_out of the attention
    # sites (model.py:300-303, 341-342) - incl. the operands the fused LinearAttention kernels keep in registers - are MX-fp8
    # values (e4m3 elements, E8M0 scale per 32 input channels, the engine's scale rule), dequantised at pack time: eight of the
    # nine sites in "fp8" (round 5, tools/fp8_attn_site_study.py: all nine cost 1.4 dB against the reference, all but the first
    # down stage's 256x256 LinearAttention site cost nothing measurable), the seven below the tile's own resolution in "fp8_mixed"
    # (its 256x256 zones keep bf16 weights like their 3x3 convolutions do).  Pinned against the format emulation: a checkpoint whose attention weights were rounded by
    # oracle/mxfp8.py, run with the engine's own rounding switched off (SRGD_FP8_ATTN_W=0), must give the same eps bit for bit;
    # and the rounding must actually move the result against the bf16-weight run.
    import os
    from oracle import mxfp8
    case = next(c for c in C.UNET_CASES if c["name"] == "dim128_128")
    sd = synth_state_dict(_schema(128), seed=case["weight_seed"])
    attn_keys = [k for k in sd if k.endswith(("to_qkv.weight", "to_out.weight", "to_out.0.weight"))]
    assert len(attn_keys) == 18                                   # 9 sites x (to_qkv, to_out)
    kept_bf16 = {"fp8": ("model.downs.0.",), "fp8_mixed": ("model.downs.0.", "model.ups.3.")}     # the LinearAttention sites at 256x256 (zones 0 and 2n)
    x, cnd, ls = C.unet_inputs(case)
    label, c = C.unet_mode_args("label_cond", case, cnd)
    rel = lambda a, b: float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt())
    try:
        for prec, n_sites in (("fp8", 8), ("fp8_mixed", 7)):
            rounded = [k for k in attn_keys if not k.startswith(kept_bf16[prec])]
            assert len(rounded) == 2 * n_sites
            sd_q = {k: (mxfp8.quantize_conv_weight(v) if k in rounded else v.clone()) for k, v in sd.items()}
            assert all(not torch.equal(sd_q[k], sd[k]) for k in rounded)
            outs = {}
            for name, weights, knob in (("engine_rounds", sd, "1"), ("pre_rounded", sd_q, "0"), ("bf16_weights", sd, "0")):
                os.environ["SRGD_FP8_ATTN_W"] = knob
                sampler = build_sampler(128, weight_seed=case["weight_seed"], fresh=True)
                sampler.load_state_dict(weights, strict=True)
                sampler.model._invalidate_engines()
                sampler.model.precision = prec
                outs[name] = sampler.model(x.cuda(), ls.cuda(), label.cuda(), c.cuda()).cpu()
                del sampler
            assert torch.isfinite(outs["engine_rounds"]).all()
            assert torch.equal(outs["engine_rounds"], outs["pre_rounded"]), prec
            assert not torch.equal(outs["engine_rounds"], outs["bf16_weights"]), prec
            _report(test="fp8_attention_weights", precision=prec, sites=n_sites,
                    rel_rms_vs_bf16_attention_weights=rel(outs["engine_rounds"], outs["bf16_weights"]))
    finally:
        os.environ.pop("SRGD_FP8_ATTN_W", None)


def test_config5_full_geometry_fp8_vs_bf16_parity_report():
    # BASELINE configs[4] as named: 256^2 LR -> 1024^2 (canvas 1280^2, 25/16 tiles), 100 DDPM steps, class_cond_scale 2.0 (both
    # passes in one launch), fp8 weights + activations vs the bf16 engine on the identical (device) noise stream.
    sampler = build_sampler(128)
    cond = C.synthetic_lr_condition(0, 256, 256).cuda()
    label = torch.tensor([0]).cuda()
    sampler.noise_source = "device"
    outs = {}
    try:
        for prec in ("bf16", "fp8", "fp8_mixed"):
            sampler.device_noise_seed = 71
            outs[prec] = sampler.tiled_sample(batch_size=25, condition_x=cond, class_label=label, class_cond_scale=2.0,
                                              num_sample_steps=100, precision=prec).cpu()
    finally:
        sampler.noise_source = "host"
    err = (outs["fp8"] - outs["bf16"]).abs()
    psnr = float(10 * np.log10(1.0 / max(float((err ** 2).mean()), 1e-20)))
    errm = (outs["fp8_mixed"] - outs["bf16"]).abs()
    psnr_mixed = float(10 * np.log10(1.0 / max(float((errm ** 2).mean()), 1e-20)))
    _report(test="config5_full_1024_fp8_vs_bf16", psnr_db=psnr, max_abs=float(err.max()), mean_abs=float(err.mean()),
            mixed_psnr_db=psnr_mixed, mixed_max_abs=float(errm.max()))
    # fp8 below the top resolution only (the 256x256-resolution zones keep bf16 3x3 convolutions): measured 53.2 dB
    assert torch.isfinite(outs["fp8_mixed"]).all() and psnr_mixed > FP8_MIXED_CONFIG5_FULL_GATE_DB, psnr_mixed      # round 4: 52.2 dB (7 attention sites in e4m3, pointwise layers bf16); round 3: 52.1 dB
    assert outs["fp8"].shape == (1, 3, 1024, 1024)
    assert torch.isfinite(outs["fp8"]).all() and outs["fp8"].min() >= 0 and outs["fp8"].max() <= 1
    assert psnr > FP8_CONFIG5_FULL_GATE_DB, psnr           # round 4: 34.5 dB with MX-e4m3 attention weights at all nine sites (round 3: 36.0 dB; 36.6 with the pointwise layers in bf16) (random-init weights; 3 mantissa bits on weights AND activations;
                                       # 34.1 dB with the OCP recipe's clamping scale rule)


def test_fp8_fused_twins_equal_separate_quantisation_passes_bitwise():
    # fp8 mode: producers write the MX-fp8 twin of a tensor in their own epilogue (conv1x1 variants, GroupNorm2 + residual,
    # fused LinearAttention, the 3x3 resamplers).  The twin is defined as quant(stored bf16 values), so switching the fusion
    # off (every conv input quantised by quant_mxfp8 in a pass of its own) must not change a single bit of the result.
    import os
    sampler = build_sampler(128)
    cond = C.synthetic_lr_condition(2, 96, 96).cuda()               # 384^2 -> canvas 768^2, 9 / 4 tiles
    label = torch.tensor([1]).cuda()
    outs = {}
    try:
        sampler.noise_source = "device"
        sampler.device_noise_seed = 3
        for mode in ("1", "0"):
            os.environ["SRGD_Q_FUSED"] = mode
            sampler.model._invalidate_engines()                     # the switch is read at engine creation
            outs[mode] = sampler.tiled_sample(batch_size=9, condition_x=cond, class_label=label, num_sample_steps=4,
                                              class_cond_scale=1.5, precision="fp8").cpu()
    finally:
        os.environ.pop("SRGD_Q_FUSED", None)
        sampler.model._invalidate_engines()
        sampler.noise_source = "host"
    assert torch.isfinite(outs["1"]).all()
    assert torch.equal(outs["1"], outs["0"])


def test_f16mx2_prototype_mode_meets_the_bar_on_the_reference_fixtures():
    # SRGD_PRECISION_F16MX2 (prototype, conv3x3_mx2.hip): f16x3 with the 3x3 convolutions' cross terms on MX-fp8 operands.  Against the
    # REFERENCE's own outputs: configs[0] (10 steps), configs[1]'s geometry (2 steps) and, when the fixture is in the tree, configs[1]
    # and configs[4] at full length (50 steps / 100 steps with class guidance) - the north-star bar 1e-3 on the finished images; CPU emulation predicted 1.3e-4 on configs[0]
    rows = {}
    for case in (next(c for c in C.SAMPLER_CASES if c["name"] == "dim128_config1"), C.WIDE_CASES[0], C.FULL_CASES[0], C.FULL5_CASES[0]):
        path = os.path.join(G, f"sample_{case['name']}.npz")
        if not os.path.exists(path):
            continue
        z = np.load(path)
        want = torch.from_numpy(z["image"] if "image" in z else z["image_u16"].astype(np.float32) / 65535.0)
        sampler = build_sampler(case["dim"], weight_seed=case["weight_seed"])
        cond = C.sampler_condition(case).cuda()
        label = torch.tensor([case["label"]]).cuda()
        sampler.noise_source = "host"
        errs = {}
        for prec in ("f16mx2", "f16x3"):
            torch.manual_seed(case["seed"])
            out = sampler.tiled_sample(batch_size=case["batch_size"], condition_x=cond, class_label=label, num_sample_steps=case["steps"],
                                       cond_scale=case["cond_scale"], class_cond_scale=case["class_cond_scale"], precision=prec,
                                       **C.extra_kwargs(case)).cpu()
            errs[prec] = float((out - want).abs().max())
        rows[case["name"]] = errs
        # with the engine's default placement (two-MFMA arithmetic below the tile's resolution, three-MFMA at 256^2: final block, last
        # up stage, first down stage) the mode meets the bar AND fp32's regression guard on the 2-step stress fixture too (measured
        # 9.4e-6 / 2.3e-4 / 1.3e-5; with the two-MFMA arithmetic everywhere 1.4e-4 / 2.3e-3 / 6.5e-5: tools/mx2_tail_study.py)
        assert errs["f16mx2"] <= (3e-4 if case["steps"] == 2 else 1e-4), (case["name"], errs)
        assert errs["f16mx2"] != errs["f16x3"]                   # it IS a different arithmetic
    _report(test="f16mx2_prototype_vs_reference", max_abs=rows)
    assert len(rows) >= 2
