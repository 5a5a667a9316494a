"""The reference's command line end to end on the GPU (inference.py:21-168 semantics): YAML config + .pth checkpoint
(`ema_model` state_dict, weights_only load, strict) + a directory of PNGs -> `*_out.png`, skip-if-exists, unreadable files
reported and skipped, and pixel-exact agreement with the CPU oracle pipeline (Pillow bicubic -> oracle tiled_sample ->
ToPILImage) up to the fp32 tolerance."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import pil_resample as PR
from oracle import srgd_oracle as O
from srgd_amd.synth import synth_state_dict
from tests.test_engine_gpu import _schema

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_inference_cli_writes_out_pngs_and_matches_the_oracle_pipeline(tmp_path):
    dim, steps, seed, label = 16, 4, 71, 1
    conf_src = open(os.path.join(ROOT, "conf", "conditional_continuous_linear_df8kost_dim128.yaml")).read()
    assert "unet_dim: 128" in conf_src
    conf = tmp_path / "dim16.yaml"
    conf.write_text(conf_src.replace("unet_dim: 128", f"unet_dim: {dim}"))
    sd = synth_state_dict(_schema(dim), seed=3)
    ckpt = tmp_path / "ckpt.pth"
    torch.save({"ema_model": sd, "epoch": 300}, ckpt)
    indir, outdir = tmp_path / "in", tmp_path / "out"
    indir.mkdir()
    rng = np.random.default_rng(5)
    lr = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    Image.fromarray(lr, "RGB").save(indir / "a.png")
    (indir / "broken.png").write_bytes(b"not a png")
    cmd = [sys.executable, os.path.join(ROOT, "inference.py"), "-c", str(conf), "-m", str(ckpt), "--input_dir", str(indir),
           "--output_dir", str(outdir), "--num_sample_steps", str(steps), "--test_label", str(label),
           "--seed", str(seed), "--batch_size", "3"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Invalid image or unable to open image" in r.stdout
    assert "engine precision: f16x3" in r.stdout     # the DEFAULT run computes at the reference's accuracy (split-operand mode; no --no_amp needed)
    out_png = outdir / "a_out.png"
    assert out_png.exists() and not (outdir / "broken_out.png").exists()
    got = np.asarray(Image.open(out_png).convert("RGB"))
    assert got.shape == (160, 224, 3)
    # the same pipeline on the CPU: Pillow bicubic x4 (restated) -> /255 -> oracle tiled_sample (same torch seed) -> *255 trunc
    cond = torch.from_numpy(PR.to_unit_chw(PR.resize_bicubic_u8(lr, 160, 224)))[None]
    torch.manual_seed(seed)
    with torch.inference_mode():
        ref = O.tiled_sample(O.strip_model_prefix(sd), O.UnetCfg(dim=dim), cond, torch.tensor([label]), batch_size=3,
                             num_sample_steps=steps)
    want = PR.to_u8_hwc(ref[0].numpy())
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3, (diff.max(), (diff > 0).mean())    # truncation may flip at x.9999
    # second run: everything already there -> "skip"
    r2 = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0 and "skip" in r2.stdout
    # --precision fp32 (exact-fp32 MFMA, the CLI default until round 5): the same image up to truncation flips
    out32 = tmp_path / "out_fp32"
    cmd32 = [c if c != str(outdir) else str(out32) for c in cmd] + ["--precision", "fp32"]
    r32 = subprocess.run(cmd32, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r32.returncode == 0 and "engine precision: fp32" in r32.stdout, r32.stderr[-2000:]
    d32 = np.abs(np.asarray(Image.open(out32 / "a_out.png").convert("RGB")).astype(int) - want.astype(int))
    assert d32.max() <= 1 and (d32 > 0).mean() < 2e-3
    # explicit opt-in to the throughput mode: same image within bf16's distance of the fp32 result
    out_bf = tmp_path / "out_bf16"
    cmd_bf = [c if c != str(outdir) else str(out_bf) for c in cmd] + ["--precision", "bf16", "--no_amp"]
    r3 = subprocess.run(cmd_bf, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r3.returncode == 0 and "engine precision: bf16" in r3.stdout, r3.stderr[-2000:]
    bf = np.asarray(Image.open(out_bf / "a_out.png").convert("RGB")).astype(np.float64)
    mse = ((bf - got.astype(np.float64)) ** 2).mean() / 255.0 ** 2
    assert 10 * np.log10(1.0 / max(mse, 1e-20)) > 40.0


def test_inference_cli_lockstep_groups_equal_one_image_at_a_time(tmp_path):
    # --lockstep N (engine extension): consecutive same-sized images sampled together, each bit-identical to its solo run (the
    # reference reseeds per image, inference.py:73); a differently sized image closes the group
    dim, steps = 16, 3
    conf_src = open(os.path.join(ROOT, "conf", "conditional_continuous_linear_df8kost_dim128.yaml")).read()
    conf = tmp_path / "dim16.yaml"
    conf.write_text(conf_src.replace("unet_dim: 128", f"unet_dim: {dim}"))
    ckpt = tmp_path / "ckpt.pth"
    torch.save({"ema_model": synth_state_dict(_schema(dim), seed=3), "epoch": 300}, ckpt)
    indir = tmp_path / "in"
    indir.mkdir()
    rng = np.random.default_rng(9)
    for name, (h, w) in (("a", (40, 56)), ("b", (40, 56)), ("c", (40, 56)), ("d", (72, 64)), ("e", (40, 56))):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB").save(indir / f"{name}.png")
    outs = {}
    for tag, extra in (("solo", []), ("lock", ["--lockstep", "2"])):
        outdir = tmp_path / f"out_{tag}"
        cmd = [sys.executable, os.path.join(ROOT, "inference.py"), "-c", str(conf), "-m", str(ckpt), "--input_dir", str(indir),
               "--output_dir", str(outdir), "--num_sample_steps", str(steps), "--test_label", "0", "--batch_size", "4"] + extra
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[tag] = {n: np.asarray(Image.open(outdir / f"{n}_out.png").convert("RGB")) for n in "abcde"}
    for n in "abcde":
        assert outs["solo"][n].shape == outs["lock"][n].shape
        assert np.array_equal(outs["solo"][n], outs["lock"][n]), n
    # under a launcher (RANK / WORLD_SIZE set): every rank takes a contiguous slice of the files - here two ranks sharing cuda:0 -
    # and together they produce the same five images (the automatic form of the reference's --start_index/--end_index per process)
    outdir = tmp_path / "out_2ranks"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29653", os.path.join(ROOT, "inference.py"), "-c", str(conf), "-m", str(ckpt), "--input_dir", str(indir),
           "--output_dir", str(outdir), "--num_sample_steps", str(steps), "--test_label", "0", "--batch_size", "4"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "rank 0/2: files [0:3] of 5" in r.stdout and "rank 1/2: files [3:5] of 5" in r.stdout
    for n in "abcde":
        assert np.array_equal(np.asarray(Image.open(outdir / f"{n}_out.png").convert("RGB")), outs["solo"][n]), n
