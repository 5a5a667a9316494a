"""The CPU oracle at BASELINE configs[1]'s full length against the reference's fixture (about an hour of CPU; build container).

    python tools/oracle_vs_full_fixture.py [threads]

Test tooling: runs oracle/srgd_oracle.tiled_sample on the seeds of tests/golden/cases.py:FULL_CASES and compares the final image
and the per-step fp64 checksums with tests/golden/sample_dim128_config2_1024_50steps.npz (the reference's own output)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import srgd_oracle as O                 # noqa: E402
from srgd_amd.synth import synth_state_dict         # noqa: E402
from tests.golden import cases as C                 # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def main():
    torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
    case = C.FULL_CASES[0]
    z = np.load(os.path.join(G, f"sample_{case['name']}.npz"))
    with open(os.path.join(G, f"schema_dim{case['dim']}.json")) as f:
        schema = {k: tuple(v) for k, v in json.load(f).items()}
    sd = O.strip_model_prefix(synth_state_dict(schema, seed=case["weight_seed"]))
    cond = C.sampler_condition(case)
    t0 = time.time()
    torch.manual_seed(case["seed"])
    trace = {}
    with torch.inference_mode():
        img = O.tiled_sample(sd, O.UnetCfg(dim=case["dim"]), cond, torch.tensor([case["label"]]), batch_size=case["batch_size"],
                             num_sample_steps=case["steps"], cond_scale=case["cond_scale"], class_cond_scale=case["class_cond_scale"],
                             trace=trace)
    want = z["image_u16"].astype(np.float32) / 65535.0
    got = img.numpy()
    # the oracle's own trajectory (canvas after every step) against the reference's fp64 checksums and subsampled planes
    xt_abs = np.array([t.double().abs().sum().item() for t in trace["img"]])
    x0_abs = np.array([t.double().abs().sum().item() for t in trace["x_start"]])
    planes = {int(i): float((C.trace_planes(trace["img"][int(i)]) - torch.from_numpy(z[f"xt_{int(i)}"])).abs().max()) for i in z["trace_steps"]}
    out = dict(case=case["name"], seconds=round(time.time() - t0, 1), max_abs_vs_u16_fixture=float(np.abs(got - want).max()),
               checksum_diff=abs(float(img.double().sum()) - float(z["checksum"])),
               xt_abs_checksum_max_diff=float(np.abs(xt_abs - z["xt_abs"]).max()), x0_abs_checksum_max_diff=float(np.abs(x0_abs - z["x0_abs"]).max()),
               xt_plane_max_abs=planes)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
