"""Phase stamps of la1_kernel's tile loop (GPU box): run with a library built by
    python tools/build_variant.py lastamps -DSRGD_LA_STAMPS=1
    SRGD_HIP_LIB=srgd_amd/variants/libsrgd_hip_lastamps.so python tools/la_stamps.py [--batch 125] [--hw 256]
Prints the `[la1 stamps]` line of the fused LinearAttention block (C = 128) on random data and the wall time of the whole block."""
import argparse
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srgd_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=125)
ap.add_argument("--hw", type=int, default=256)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
lib = _lib.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
Cc = 128
g = torch.Generator().manual_seed(1)
x = (torch.randn(a.batch, a.hw, a.hw, Cc, device="cuda") * 1.5).to(torch.bfloat16)
y = torch.empty_like(x)
hw = [(torch.randn(384, Cc, generator=g) / Cc ** 0.5).contiguous(), (1 + 0.1 * torch.randn(Cc, generator=g)).contiguous(),
      (torch.randn(Cc, 128, generator=g) / 128 ** 0.5).contiguous(), (0.1 * torch.randn(Cc, generator=g)).contiguous(),
      (1 + 0.1 * torch.randn(Cc, generator=g)).contiguous()]
for i in range(a.iters):
    t0 = time.time()
    _lib.check(lib.srgd_k_linattn_block_fused(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), a.batch, a.hw * a.hw, Cc,
                                              *[C.c_void_p(t.data_ptr()) for t in hw], st), "fused")
    torch.cuda.synchronize()
    print(f"block (pack + la1 + combine + la2, synchronous): {(time.time() - t0) * 1e3:.2f} ms", flush=True)
