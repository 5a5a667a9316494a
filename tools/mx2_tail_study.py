"""f16mx2 prototype: which blocks have to stay on the three-MFMA arithmetic for the 2-step stress fixture to meet the bar?  (GPU box)

    SRGD_MX2_EXACT_TAIL=N SRGD_MX2_EXACT_HEAD=M python tools/mx2_tail_study.py [--full]

Prints one JSON row: max-abs against the REFERENCE's fixtures (configs[0], configs[1] geometry after 2 steps, optionally configs[1] at
full length) for precision f16mx2 under the environment's setting.  One process per setting: the knobs are read when an engine is created."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.golden import cases as C                      # noqa: E402
from tests.test_engine_gpu import G, build_sampler      # noqa: E402

cases = [next(c for c in C.SAMPLER_CASES if c["name"] == "dim128_config1"), C.WIDE_CASES[0]]
if "--full" in sys.argv:
    cases.append(C.FULL_CASES[0])
row = dict(tail=int(os.environ.get("SRGD_MX2_EXACT_TAIL", "0")), head=int(os.environ.get("SRGD_MX2_EXACT_HEAD", "0")))
for case in cases:
    z = np.load(os.path.join(G, f"sample_{case['name']}.npz"))
    want = torch.from_numpy(z["image"] if "image" in z else z["image_u16"].astype(np.float32) / 65535.0)
    sampler = build_sampler(case["dim"], weight_seed=case["weight_seed"])
    sampler.noise_source = "host"
    torch.manual_seed(case["seed"])
    out = sampler.tiled_sample(batch_size=case["batch_size"], condition_x=C.sampler_condition(case).cuda(),
                               class_label=torch.tensor([case["label"]]).cuda(), num_sample_steps=case["steps"], cond_scale=case["cond_scale"],
                               class_cond_scale=case["class_cond_scale"], precision="f16mx2", **C.extra_kwargs(case)).cpu()
    row[case["name"]] = float((out - want).abs().max())
print(json.dumps(row), flush=True)
