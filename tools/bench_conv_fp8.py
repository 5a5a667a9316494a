"""Per-shape timing of the MX-fp8 3x3 convolution next to the bf16 fast path (GPU box only), random data.

    python tools/bench_conv_fp8.py [--batch 25] [--iters 10] [--only SUBSTR]

Prints algorithmic TFLOP/s from HIP-event timing of the convolution kernel alone (srgd_k_conv3x3_mxfp8 / srgd_k_conv2d_timed);
the quantisation passes are not in either number."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srgd_amd import _lib  # noqa: E402
from tools.bench_conv import SHAPES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=25)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default="gpurun_out/bench_conv_fp8.json")
    args = ap.parse_args()
    lib = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = []
    for name, c0, c1, cout, hw, ks in SHAPES:
        if ks != 3 or (args.only and args.only not in name):
            continue
        B = max(1, args.batch * (256 * 256) // (hw * hw) // (8 if hw <= 64 else 1)) if hw < 256 else args.batch
        B = args.batch      # the engine launches every layer with the same tile count
        g = torch.Generator(device="cuda").manual_seed(0)
        x0 = torch.randn(B, hw, hw, c0, device="cuda", generator=g).to(torch.bfloat16)
        x1 = torch.randn(B, hw, hw, c1, device="cuda", generator=g).to(torch.bfloat16) if c1 else None
        w = (torch.randn(cout, c0 + c1, 3, 3) / (3 * (c0 + c1) ** 0.5)).float().contiguous()
        bias = torch.randn(cout).float()
        out = torch.empty(B, hw, hw, cout, device="cuda", dtype=torch.bfloat16)
        part = torch.zeros(B * 8 * (hw * hw // 32) * 2 + 64, device="cuda")
        flops = 2.0 * B * hw * hw * cout * 9 * (c0 + c1)
        ms, slots = C.c_float(), C.c_int()
        _lib.check(lib.srgd_k_conv3x3_mxfp8(C.c_void_p(x0.data_ptr()), C.c_void_p(x1.data_ptr() if c1 else 0), c0, c1, B, hw, hw,
                                            C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), cout, C.c_void_p(out.data_ptr()),
                                            C.c_void_p(part.data_ptr()), 8, args.iters, C.byref(ms), C.byref(slots), st), name)
        fp8 = flops / (ms.value * 1e-3) / 1e12
        _lib.check(lib.srgd_k_conv2d_timed(C.c_void_p(x0.data_ptr()), C.c_void_p(x1.data_ptr() if c1 else 0), c0, c1, B, hw, hw,
                                           3, 1, 1, 0, C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), cout,
                                           C.c_void_p(out.data_ptr()), C.c_void_p(0), C.c_void_p(part.data_ptr()), 8, 1, 0,
                                           args.iters, C.byref(ms), C.byref(slots), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0),
                                           st), name)
        bf = flops / (ms.value * 1e-3) / 1e12
        rows.append(dict(shape=name, batch=B, mxfp8_tflops=round(fp8, 1), bf16_tflops=round(bf, 1)))
        print(f"{name:28s} B={B:4d}  mxfp8 {fp8:7.1f} TF   bf16 {bf:7.1f} TF   x{fp8 / bf:.2f}", flush=True)
        del x0, x1, out
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
