"""fp8 mode: what does carrying each attention site's projection weights (to_qkv, to_out) as MX-e4m3 cost?  (GPU box)
Zones: 0-3 down stages (0-2 LinearAttention @256^2/128^2/64^2, 3 softmax attention @32^2), 4 middle (softmax), 5-8 up stages
(5 softmax @32^2, 6-8 LinearAttention @64^2/128^2/256^2).  Runs the configs[4] one-tile fixture (LONG_CASES[0]: 256^2 canvas,
100 DDPM steps, class CFG 2.0, host noise) in precision "fp8" with SRGD_FP8_ATTN_BF16_ZONES masks and reports PSNR against the
REFERENCE's image (tests/golden) and against the bf16 engine."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import logging  # noqa: E402

from srgd_amd.config import load_config  # noqa: E402
from srgd_amd.model import get_model  # noqa: E402
from srgd_amd.synth import synth_state_dict  # noqa: E402
from tests.golden import cases as C  # noqa: E402  (case tables and the golden image: data, not the oracle)


def build_sampler(dim, weight_seed):
    conf = load_config(os.path.join(ROOT, "conf", "conditional_continuous_linear_df8kost_dim128.yaml"))
    conf.unet_dim = dim
    ema = get_model(conf, logging.getLogger("study"))
    schema = {k: tuple(v.shape) for k, v in ema.module.state_dict().items()}
    ema.module.load_state_dict(synth_state_dict(schema, seed=weight_seed), strict=True)
    return ema.module.eval().to(torch.device("cuda"))

case = C.LONG_CASES[0]
z = np.load(os.path.join(ROOT, "tests", "golden", f"sample_{case['name']}.npz"))
want = torch.from_numpy(z["image"])
cond = C.sampler_condition(case).cuda()
label = torch.tensor([case["label"]]).cuda()
psnr = lambda a, b: float(10 * np.log10(1.0 / max(float(((a - b) ** 2).mean()), 1e-20)))


def run(prec, **env):
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        sampler = build_sampler(case["dim"], case["weight_seed"])
        torch.manual_seed(case["seed"])
        return sampler.tiled_sample(batch_size=case["batch_size"], condition_x=cond, class_label=label,
                                    class_cond_scale=case["class_cond_scale"], num_sample_steps=case["steps"], precision=prec).cpu()
    finally:
        for k in env:
            os.environ.pop(k, None)


bf16 = run("bf16")
rows = []


def row(name, out):
    rows.append(dict(config=name, psnr_vs_reference=round(psnr(out, want), 2), psnr_vs_bf16=round(psnr(out, bf16), 2)))
    print(rows[-1], flush=True)


row("bf16 engine", bf16)
T = dict(SRGD_FP8_ATTN_BF16_ZONES=0)                 # (the shipped default keeps zone 0's site in bf16: mask 0 = e4m3 everywhere)
row("fp8, attention weights bf16 at all nine sites", run("fp8", SRGD_FP8_ATTN_W=0))
row("fp8, e4m3 at all nine sites (E8M0 per 32 input channels)", run("fp8", **T))
# (one scale per output channel instead of per 32 input channels measured 35.06 dB in a study build: profiles/r5/fp8_attn_site_study.json)
ALL = (1 << 9) - 1
for zn in range(9):
    row(f"fp8, e4m3 ONLY at zone {zn}", run("fp8", SRGD_FP8_ATTN_BF16_ZONES=ALL & ~(1 << zn)))
row("fp8, e4m3 everywhere but zone 0", run("fp8", SRGD_FP8_ATTN_BF16_ZONES=1 << 0))
row("fp8, e4m3 everywhere but zone 8", run("fp8", SRGD_FP8_ATTN_BF16_ZONES=1 << 8))
row("fp8, e4m3 at the seven sites below the tile's resolution (zones 1-7)", run("fp8", SRGD_FP8_ATTN_BF16_ZONES=(1 << 0) | (1 << 8)))
row("fp8 as shipped (e4m3 everywhere but zone 0)", run("fp8"))
# (profiles/r5/fp8_attn_site_study.json also holds two rows of a study build that split the 256^2 sites by projection:
#  seven sites + to_qkv of both 36.42 / 27.57 dB, seven sites + to_out of both 36.65 / 27.54 dB)
row("fp8, e4m3 everywhere but the 256^2 and 128^2 sites (zones 0, 1, 7, 8)", run("fp8", SRGD_FP8_ATTN_BF16_ZONES=0b110000011))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "fp8_attn_site_study.json"), "w"), indent=1)


# ---- the candidate placements at configs[1]'s real geometry too (WIDE_CASES[0]: 256^2 LR -> 1024^2, 2 DDPM steps from noise)
case2 = C.WIDE_CASES[0]
z2 = np.load(os.path.join(ROOT, "tests", "golden", f"sample_{case2['name']}.npz"))
want2 = torch.from_numpy(z2["image_u16"].astype(np.float32) / 65535.0)
cond2 = C.sampler_condition(case2).cuda()
label2 = torch.tensor([case2["label"]]).cuda()


def run2(prec, **env):
    for k, v in env.items():
        os.environ[k] = str(v)
    try:
        sampler = build_sampler(case2["dim"], case2["weight_seed"])
        sampler.noise_source = "host"
        torch.manual_seed(case2["seed"])
        return sampler.tiled_sample(batch_size=case2["batch_size"], condition_x=cond2, class_label=label2,
                                    num_sample_steps=case2["steps"], precision=prec).cpu()
    finally:
        for k in env:
            os.environ.pop(k, None)


rows2 = []
for name, env in [("attention weights bf16 at all nine sites", dict(SRGD_FP8_ATTN_W=0)), ("e4m3 at all nine sites", T),
                  ("e4m3 everywhere but zone 0 (shipped)", {}), ("e4m3 everywhere but zone 8", dict(SRGD_FP8_ATTN_BF16_ZONES=256)),
                  ("seven sites (zones 1-7)", dict(SRGD_FP8_ATTN_BF16_ZONES=257))]:
    rows2.append(dict(config="configs[1] geometry, fp8, " + name, psnr_vs_reference=round(psnr(run2("fp8", **env), want2), 2)))
    print(rows2[-1], flush=True)
json.dump(dict(config5_256=rows, config2_geometry=rows2), open(os.path.join(ROOT, "gpurun_out", "fp8_attn_site_study.json"), "w"), indent=1)
