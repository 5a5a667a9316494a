"""Numerics study for the split-operand ("3-MFMA") precision, on the CPU, BEFORE any kernel is written.

    python tools/split_numerics_study.py [--case dim128_config1] [--threads 4] [--variants f16x3,bf16x3,...]

Tooling, not product: it runs the CPU oracle (oracle/srgd_oracle.py) with every convolution that the engine's MFMA kernels
takes in that mode (Cin % 32 == 0; the 7x7 input convolution, Cin = 6, keeps fp32) replaced by the emulated split product

    x = x_hi + x_lo,  w = w_hi + w_lo        (hi = round-to-nearest 16-bit value, lo = round(x - hi))
    conv(x, w) ~= conv(x_hi, w_hi) + conv(x_lo, w_hi) + conv(x_hi, w_lo)            (fp32 accumulation)

A product of two 16-bit values is exact in fp32 (bf16: 16 significand bits, f16: 22), so three fp32 convolutions on the rounded
operands reproduce what three MFMAs accumulate up to summation order.  Variants:
    bf16x3   bf16 halves (8 + 8 bits; fp32's exponent range, nothing to scale)
    bf16x4   ... plus the x_lo * w_lo term
    f16x3    f16 halves (11 + 11 bits); weights pre-scaled by a power of two per tensor so that w_lo stays out of f16's
             subnormal range (undone exactly after the accumulation)
    f16x3ns  f16 halves, no weight scaling (what the subnormal range costs)
    bf16     plain bf16 operands (one MFMA): the throughput mode's convolution arithmetic, for scale
    f16x2_w1 / f16x2_x1 / f16mx2   cheaper relatives of f16x3 (two MFMAs' worth of matrix work; oracle/split_emulation.py), for pricing
The report is the final-pixel max-abs / PSNR against the REFERENCE's fixture (tests/golden), next to the plain fp32 oracle's.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import srgd_oracle as O                 # noqa: E402
from srgd_amd.synth import synth_state_dict         # noqa: E402
from oracle.split_emulation import mixed_split_conv2d, split_conv2d    # noqa: E402
from tests.golden import cases as C                 # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def make_conv(variant):
    """conv2d replacement for one variant; layers the engine keeps on the exact-fp32 kernel (Cin % 32 != 0) stay fp32."""
    def conv2d(x, w, b=None, stride=1, padding=0):
        if x.shape[1] % 32 != 0 or variant == "fp32":
            return TF.conv2d(x, w, b, stride=stride, padding=padding)
        if variant == "bf16":
            y = TF.conv2d(x.to(torch.bfloat16).float(), w.to(torch.bfloat16).float(), None, stride=stride, padding=padding)
            return y if b is None else y + b.view(1, -1, 1, 1)
        if variant in ("f16x2_w1", "f16x2_x1", "f16mx2"):        # cheaper relatives of f16x3 (pricing only, no kernel)
            return mixed_split_conv2d(x, w, b, stride=stride, padding=padding, mode=variant)
        kind = "bf16" if variant.startswith("bf16") else "f16"
        return split_conv2d(x, w, b, stride=stride, padding=padding, kind=kind, terms=4 if variant.endswith("x4") else 3,
                            scale=variant != "f16x3ns")

    return conv2d


class _FShim(types.SimpleNamespace):
    """`F` of the oracle module with conv2d replaced"""
    def __init__(self, conv):
        super().__init__()
        self._conv = conv

    def __getattr__(self, name):
        if name == "conv2d":
            return self._conv
        return getattr(TF, name)


def schema(dim):
    with open(os.path.join(G, f"schema_dim{dim}.json")) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def find_case(name):
    for group in (C.SAMPLER_CASES, C.LONG_CASES, C.WIDE_CASES, getattr(C, "FULL_CASES", [])):
        for c in group:
            if c["name"] == name:
                return c
    raise KeyError(name)


def reference_image(case):
    z = np.load(os.path.join(G, f"sample_{case['name']}.npz"))
    if "image" in z:
        return z["image"]
    return z["image_u16"].astype(np.float32) / 65535.0


def run(case, variant):
    sd = O.strip_model_prefix(synth_state_dict(schema(case["dim"]), seed=case["weight_seed"]))
    cond = C.sampler_condition(case)
    label = torch.tensor([case["label"]]) if case["label"] is not None else None
    saved = O.F
    O.F = _FShim(make_conv(variant))
    try:
        torch.manual_seed(case["seed"])
        with torch.inference_mode():
            img = O.tiled_sample(sd, O.UnetCfg(dim=case["dim"]), cond, label, batch_size=case["batch_size"],
                                 num_sample_steps=case["steps"], cond_scale=case["cond_scale"],
                                 class_cond_scale=case["class_cond_scale"], **C.extra_kwargs(case))
    finally:
        O.F = saved
    return img.numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="dim128_config1")
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--variants", default="fp32,bf16x3,f16x3,f16x3ns,bf16x4,bf16")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    case = find_case(a.case)
    want = reference_image(case)
    rows = []
    for v in a.variants.split(","):
        t0 = time.time()
        got = run(case, v)
        d = np.abs(got.astype(np.float64) - want.astype(np.float64))
        mse = float((d ** 2).mean())
        row = dict(case=a.case, variant=v, max_abs=float(d.max()), mean_abs=float(d.mean()),
                   psnr_db=float(10 * np.log10(1.0 / mse)) if mse > 0 else float("inf"), seconds=round(time.time() - t0, 1))
        rows.append(row)
        print(json.dumps(row), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
