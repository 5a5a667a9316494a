"""Micro-benchmark + cross-check of the convolution kernels on production shapes (GPU box only).

    python tools/bench_conv.py [--batch 25] [--iters 20] [--shapes big|all]

For each shape: runs the generic implicit-GEMM kernel (impl 1) and the engine's choice (impl 0, the
conv3x3_bf16 fast path where eligible) on the same bf16 data, checks they agree, and prints
algorithmic TFLOP/s from HIP-event timing (srgd_k_conv2d_timed)."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srgd_amd import _lib  # noqa: E402

# (name, C0, C1, Cout, HW, KS) at dim 128
SHAPES = [
    ("3x3 128->128 @256", 128, 0, 128, 256, 3),
    ("3x3 128+128->128 @256", 128, 128, 128, 256, 3),
    ("3x3 1024->1024 @32", 1024, 0, 1024, 32, 3),
    ("3x3 1024+512->1024 @32", 1024, 512, 1024, 32, 3),
    ("3x3 512+256->512 @64", 512, 256, 512, 64, 3),
    ("3x3 256+128->256 @128", 256, 128, 256, 128, 3),
    ("3x3 512->512 @64", 512, 0, 512, 64, 3),
    ("3x3 256->256 @128", 256, 0, 256, 128, 3),
    ("3x3 128->128 @128", 128, 0, 128, 128, 3),
    ("3x3 256->256 @64", 256, 0, 256, 64, 3),
    ("3x3 512->512 @32", 512, 0, 512, 32, 3),
    ("3x3 512->1024 @32", 512, 0, 1024, 32, 3),
    ("1x1 128->384 @256", 128, 0, 384, 256, 1),
    ("1x1 128+128->128 @256", 128, 128, 128, 256, 1),
    ("1x1 128->128 @256", 128, 0, 128, 256, 1),
    ("1x1 1024->2048 @32", 1024, 0, 2048, 32, 1),
    ("1x1 256+128->256 @128", 256, 128, 256, 128, 1),
    ("1x1 256->384 @128", 256, 0, 384, 128, 1),
    ("1x1 128->256 @128", 128, 0, 256, 128, 1),
    ("1x1 512+256->512 @64", 512, 256, 512, 64, 1),
    ("1x1 1024+512->1024 @32", 1024, 512, 1024, 32, 1),
    ("1x1 256->512 @128", 256, 0, 512, 128, 1),
    ("1x1 512->1024 @64", 512, 0, 1024, 64, 1),
    ("1x1 512->384 @64", 512, 0, 384, 64, 1),
    ("1x1 128->512 @64", 128, 0, 512, 64, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=25)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--shapes", default="all")
    ap.add_argument("--stats", type=int, default=1)
    ap.add_argument("--only", default="", help="substring filter on the shape name")
    ap.add_argument("--impls", default="1,0")
    ap.add_argument("--fp32", action="store_true",
                    help="fp32 tensors (is_bf16 = 0): impl 1 = exact-fp32 MFMA, 6 / 7 = the split-operand kernels (f16x3 mode)")
    args = ap.parse_args()
    lib = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = []
    for name, c0, c1, cout, hw, ks in SHAPES:
        if args.shapes == "big" and not name.startswith("3x3"):
            continue
        if args.only and args.only not in name:
            continue
        g = torch.Generator(device="cuda").manual_seed(0)
        B = args.batch
        adt = torch.float32 if args.fp32 else torch.bfloat16
        x0 = torch.randn(B, hw, hw, c0, device="cuda", generator=g).to(adt)
        x1 = torch.randn(B, hw, hw, c1, device="cuda", generator=g).to(adt) if c1 else None
        w = (torch.randn(cout, c0 + c1, ks, ks) / (ks * (c0 + c1) ** 0.5)).float().contiguous()
        bias = torch.randn(cout).float()
        outs, res = [], {}
        for impl in [int(v) for v in args.impls.split(',')]:
            out = torch.empty(B, hw, hw, cout, device="cuda", dtype=adt)
            groups = 8 if (args.stats and ks == 3) else 0
            part = torch.zeros(B * 8 * (hw * hw // 32) * 2, device="cuda") if groups else None
            ms = C.c_float()
            slots = C.c_int()
            coef = None
            if impl == 5:          # GroupNorm-in-staging: per-(sample, channel) scale | shift of the input, one allocation
                coef = torch.stack([1.0 + 0.1 * torch.randn(B, c0, device="cuda", generator=g), 0.1 * torch.randn(B, c0, device="cuda", generator=g)]).contiguous()
            rc = lib.srgd_k_conv2d_timed(C.c_void_p(x0.data_ptr()), C.c_void_p(x1.data_ptr() if c1 else 0), c0, c1, B, hw, hw,
                                         ks, 1, ks // 2, 0, C.c_void_p(w.data_ptr()), C.c_void_p(bias.data_ptr()), cout,
                                         C.c_void_p(out.data_ptr()), C.c_void_p(0),
                                         C.c_void_p(part.data_ptr() if groups else 0), groups, 0 if args.fp32 else 1, impl, args.iters,
                                         C.byref(ms), C.byref(slots), C.c_void_p(0), C.c_void_p(coef[0].data_ptr() if impl == 5 else 0),
                                         C.c_void_p(coef[1].data_ptr() if impl == 5 else 0), st)
            if rc != 0 and impl in (4, 6, 7):               # the MX pointwise kernel does not cover every shape: skip the shape
                print(f"{name:28s} impl {impl}: not eligible", flush=True)
                res = None
                break
            _lib.check(rc, name)
            torch.cuda.synchronize()
            flops = 2.0 * B * hw * hw * cout * ks * ks * (c0 + c1)
            res[impl] = flops / (ms.value * 1e-3) / 1e12
            s1 = part[:B * 8 * slots.value * 2].view(B, 8, slots.value, 2)[..., 0].sum(-1) if groups else None
            outs.append((out.float(), s1))
        if res is None:
            continue
        if len(outs) < 2 or 5 in res:
            print(name, res, flush=True)
            continue
        d = (outs[0][0] - outs[1][0]).abs().max().item()
        ref = outs[0][0].abs().max().item()
        ds = ((outs[0][1] - outs[1][1]).abs().max().item() / max(1.0, outs[0][1].abs().max().item())) if outs[0][1] is not None else 0.0
        ia, ib = [int(v) for v in args.impls.split(',')][:2]        # first = the reference arm (default: generic), second = the arm under test
        rows.append(dict(shape=name, impl_a=ia, impl_b=ib, generic_tflops=round(res[ia], 1), engine_tflops=round(res[ib], 1),
                         max_abs_diff=d, ref_max=ref, stats_rel_diff=ds))
        gb = (4.0 if args.fp32 else 2.0) * B * hw * hw * (c0 + c1 + cout) / 1e9           # in + out, once
        ms0 = 2.0 * B * hw * hw * cout * ks * ks * (c0 + c1) / (res[ib] * 1e12) * 1e3
        print(f"{name:28s} impl {ia} {res[ia]:7.1f} TF   impl {ib} {res[ib]:7.1f} TF ({ms0:.3f} ms, {gb / ms0:.2f} TB/s in+out)   |diff| {d:.3g} (max {ref:.3g})  stats {ds:.2g}", flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bench_conv.json", "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
