"""End-to-end rate of the command line (PNG in -> PNG out, checkpoint load included and reported separately) on the GPU box:
10 synthetic 256x256 LR images -> 1024x1024, dim-128 U-Net, 50 steps, seeded synthetic checkpoint.
    python tools/cli_throughput.py [--precision bf16] [--lockstep 5]"""
import argparse, json, os, subprocess, sys, tempfile, time
import numpy as np
import torch
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from srgd_amd.synth import synth_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16")
ap.add_argument("--lockstep", type=int, default=5)
ap.add_argument("--images", type=int, default=10)
args = ap.parse_args()
with tempfile.TemporaryDirectory() as td:
    schema = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", "schema_dim128.json"))).items()}
    ckpt = os.path.join(td, "ckpt.pth")
    torch.save({"ema_model": synth_state_dict(schema, seed=0), "epoch": 300}, ckpt)
    ind, outd = os.path.join(td, "in"), os.path.join(td, "out")
    os.makedirs(ind)
    rng = np.random.default_rng(0)
    for i in range(args.images):
        Image.fromarray(rng.integers(0, 256, (256, 256, 3), dtype=np.uint8), "RGB").save(os.path.join(ind, f"im{i:02d}.png"))
    base = [sys.executable, os.path.join(ROOT, "inference.py"), "-c", os.path.join(ROOT, "conf", "conditional_continuous_linear_df8kost_dim128.yaml"),
            "-m", ckpt, "--input_dir", ind, "--num_sample_steps", "50", "--test_label", "0", "--batch_size", "25",
            "--precision", args.precision, "--device_noise"]
    rows = {}
    for tag, extra, n in (("load_only", ["--end_index", "0"], 0), ("lockstep1", [], args.images), (f"lockstep{args.lockstep}", ["--lockstep", str(args.lockstep)], args.images)):
        out = outd + "_" + tag
        t0 = time.time()
        r = subprocess.run(base + ["--output_dir", out] + extra, cwd=ROOT, capture_output=True, text=True)
        dt = time.time() - t0
        assert r.returncode == 0, r.stderr[-2000:]
        rows[tag] = dt
        print(tag, f"{dt:.2f} s", flush=True)
    load = rows["load_only"]
    for tag in rows:
        if tag != "load_only":
            print(f"{tag}: {(rows[tag] - load) / args.images:.3f} s per 1024^2 image after the {load:.1f} s start-up (import, checkpoint load, weight packing)")
