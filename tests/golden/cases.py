"""Case definitions shared by ``make_golden.py`` (which runs the reference) and the tests
(which rebuild the same seeded inputs and compare the oracle / the HIP path with the stored
reference outputs).  Pure data + seeded input builders; no reference code."""
from __future__ import annotations

import numpy as np
import torch

GEOMETRY_SIZES = [(256, 256), (64, 64), (1024, 1024), (8192, 8192), (300, 500), (1000, 1500),
                  (257, 256), (512, 512), (513, 700)]
SCHEDULE_STEPS = [10, 50, 100]

UNET_CASES = [
    dict(name="dim16_128", dim=16, hw=128, batch=2, weight_seed=0, input_seed=7,
         log_snr=[-3.0, 2.5], label=1, modes=["label_cond", "null_class", "null_cond"]),
    dict(name="dim128_128", dim=128, hw=128, batch=1, weight_seed=0, input_seed=8,
         log_snr=[0.75], label=0, modes=["label_cond", "null_class"]),
]

SAMPLER_CASES = [
    # G5: single tile canvas, CFG off / class CFG 2.0
    dict(name="dim16_256_cfg1", dim=16, h=256, w=256, steps=10, batch_size=4, label=0,
         cond_scale=1.0, class_cond_scale=1.0, weight_seed=0, cond="rand", cond_seed=1234, seed=71),
    dict(name="dim16_256_cfg2", dim=16, h=256, w=256, steps=10, batch_size=4, label=0,
         cond_scale=1.0, class_cond_scale=2.0, weight_seed=0, cond="rand", cond_seed=1234, seed=71),
    # G6: ragged canvas -> 768x768, 9/4 tiles, ragged last minibatch, ring re-noise
    dict(name="dim16_300x500", dim=16, h=300, w=500, steps=4, batch_size=4, label=2,
         cond_scale=1.0, class_cond_scale=1.0, weight_seed=0, cond="rand", cond_seed=1235, seed=71),
    # q_sample start (generation_start_steps > 0 / start_white_noise=False) and delayed class guidance
    dict(name="dim16_256_genstart", dim=16, h=256, w=256, steps=8, batch_size=4, label=1,
         cond_scale=1.0, class_cond_scale=1.5, weight_seed=0, cond="rand", cond_seed=1236, seed=71,
         generation_start_steps=3, class_guidance_start_steps=5),
    dict(name="dim16_300x300_nowhite", dim=16, h=300, w=300, steps=3, batch_size=8, label=0,
         cond_scale=1.0, class_cond_scale=1.0, weight_seed=0, cond="rand", cond_seed=1237, seed=71,
         start_white_noise=False),
    # G7: BASELINE config 1 - 64x64 LR -> x4 bicubic -> 256x256, dim-128 U-Net, 10 steps, CFG off
    dict(name="dim128_config1", dim=128, h=256, w=256, steps=10, batch_size=8, label=0,
         cond_scale=1.0, class_cond_scale=1.0, weight_seed=0, cond="lr_bicubic", cond_seed=1234, seed=71),
]


# G3: outputs of the reference's own sub-modules (dim 16 U-Net, seeded weights) on seeded inputs - pins each oracle
# building block separately (tests/golden/modules_dim16.npz).  name -> (module path inside the U-Net, input shape)
MODULE_DIM = 16
MODULE_CASES = {
    "resnet_same": ("downs.0.0", (2, 16, 32, 32)),          # ResnetBlock without res_conv
    "resnet_concat": ("ups.0.0", (2, 128 + 64, 16, 16)),     # ResnetBlock with res_conv (skip concat widths)
    "linear_attention": ("downs.0.2", (2, 16, 32, 32)),
    "full_attention": ("downs.3.2", (2, 64, 16, 16)),
    "mid_attention": ("mid_attn", (1, 128, 8, 8)),
    "downsample": ("downs.0.3", (2, 16, 32, 32)),
    "last_down_conv3x3": ("downs.3.3", (1, 64, 16, 16)),
    "pixel_shuffle_up": ("ups.0.3", (2, 128, 8, 8)),
    "rms_norm": ("downs.0.2.norm", (2, 16, 8, 8)),
}


def module_input(name):
    g = torch.Generator().manual_seed(4000 + sorted(MODULE_CASES).index(name))
    return torch.randn(*MODULE_CASES[name][1], generator=g)


def module_time_embedding():
    g = torch.Generator().manual_seed(3999)
    return torch.randn(2, MODULE_DIM * 4, generator=g)


# BASELINE configs[4] at config-1 geometry: 100 DDPM steps, class_cond_scale = 2.0 (two passes), dim-128 U-Net, one 256^2
# tile.  200 CPU U-Net forwards: generated once by make_golden.py --config5-only; used by the GPU tests only (fp32 parity
# gate + bf16 and fp8-weight reports), the CPU suite skips it for time.
LONG_CASES = [
    dict(name="dim128_config5_256", dim=128, h=256, w=256, steps=100, batch_size=8, label=0, cond_scale=1.0,
         class_cond_scale=2.0, weight_seed=0, cond="lr_bicubic", cond_seed=1234, seed=71),
]

# un-tiled sample() (model.py:3417-3432): a batch of independent image_size^2 images, per-image noise
SAMPLE_CASES = [
    dict(name="dim16_b3_cfg", dim=16, batch=3, steps=6, label=2, cond_scale=1.0, class_cond_scale=1.5, weight_seed=0,
         cond_seed=1240, seed=71, class_guidance_start_steps=2),
    dict(name="dim16_b2_genstart", dim=16, batch=2, steps=6, label=0, cond_scale=1.0, class_cond_scale=1.0, weight_seed=0,
         cond_seed=1241, seed=71, generation_start_steps=2),
]


def sample_condition(case):
    g = torch.Generator().manual_seed(case["cond_seed"])
    return torch.rand(case["batch"], 3, 256, 256, generator=g)


def sample_extra_kwargs(case):
    keys = ("generation_start_steps", "class_guidance_start_steps", "guidance_start_steps")
    return {k: case[k] for k in keys if k in case}


# EDM sampler (ConditionalElucidatedDiffusionSR.tiled_sample, model.py:2309-2475) over the same U-Net; fixture files
# sample_edm_<name>.npz hold the reference's outputs (its un-vendored base class restated in oracle/refshim.py)
EDM_CASES = [
    dict(name="dim16_256", dim=16, h=256, w=256, steps=6, batch_size=4, label=0, cond_scale=1.0, class_cond_scale=1.0,
         weight_seed=0, cond="rand", cond_seed=1234, seed=71),
    dict(name="dim16_256_cfg2", dim=16, h=256, w=256, steps=6, batch_size=4, label=0, cond_scale=1.0, class_cond_scale=2.0,
         weight_seed=0, cond="rand", cond_seed=1234, seed=71),
    dict(name="dim16_300x500", dim=16, h=300, w=500, steps=4, batch_size=4, label=2, cond_scale=1.0, class_cond_scale=1.0,
         weight_seed=0, cond="rand", cond_seed=1235, seed=71),
    dict(name="dim16_256_genstart_lrcfg", dim=16, h=256, w=256, steps=8, batch_size=4, label=1, cond_scale=1.5,
         class_cond_scale=1.0, weight_seed=0, cond="rand", cond_seed=1236, seed=71, generation_start_steps=2),
    dict(name="dim16_300x300_zeroinit_noclamp", dim=16, h=300, w=300, steps=3, batch_size=8, label=0, cond_scale=1.0,
         class_cond_scale=1.0, weight_seed=0, cond="rand", cond_seed=1237, seed=71, zero_init=True, clamp=False),
]


def edm_extra_kwargs(case):
    keys = ("generation_start_steps", "class_guidance_start_steps", "guidance_start_steps", "zero_init", "clamp")
    return {k: case[k] for k in keys if k in case}


def unet_inputs(case):
    g = torch.Generator().manual_seed(case["input_seed"])
    b, hw = case["batch"], case["hw"]
    x = torch.randn(b, 3, hw, hw, generator=g)
    cnd = torch.rand(b, 3, hw, hw, generator=g) * 2 - 1
    ls = torch.tensor(case["log_snr"], dtype=torch.float32)
    return x, cnd, ls


def unet_mode_args(mode, case, cnd):
    label = torch.tensor([case["label"]])
    if mode == "label_cond":
        return label, cnd
    if mode == "null_class":
        return None, cnd
    if mode == "null_cond":
        return label, None
    raise KeyError(mode)


from srgd_amd.synth import synthetic_lr_condition  # noqa: E402,F401  (shared with bench.py)


def sampler_condition(case):
    if case["cond"] == "rand":
        g = torch.Generator().manual_seed(case["cond_seed"])
        return torch.rand(1, 3, case["h"], case["w"], generator=g)
    if case["cond"] == "lr_bicubic":
        return synthetic_lr_condition(0, case["h"] // 4, case["w"] // 4, seed_base=case["cond_seed"])
    raise KeyError(case["cond"])


def extra_kwargs(case):
    """Optional tiled_sample knobs a case may carry (same names in the reference, the oracle and the product)."""
    keys = ("generation_start_steps", "class_guidance_start_steps", "guidance_start_steps", "start_white_noise")
    return {k: case[k] for k in keys if k in case}
