"""Probe: does a producer->consumer chain of HBM-bound kernels run faster when the tensors fit the 256 MB Infinity Cache?
Ping-pongs srgd_k_rmsnorm (read x, write y, 128 channels bf16) between two buffers of a given size and reports GB/s."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srgd_amd import _lib  # noqa: E402

lib = _lib.lib()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.ones(128, device="cuda")
for tiles in (1, 2, 4, 8, 16, 32, 125):
    npix = tiles * 65536
    a = torch.randn(npix, 128, device="cuda").to(torch.bfloat16)
    b = torch.empty_like(a)
    def go(n):
        for i in range(n):
            src, dst = (a, b) if i % 2 == 0 else (b, a)
            _lib.check(lib.srgd_k_rmsnorm(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_void_p(0), C.c_void_p(g.data_ptr()),
                                          npix, 128, 1, st), "rms")
    go(4)
    torch.cuda.synchronize()
    n = max(8, 2000 // tiles)
    t0 = time.perf_counter()
    go(n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    mb = npix * 128 * 2 / 1e6
    print(f"{tiles:4d} tiles: tensor {mb:8.1f} MB, {dt * 1e6:8.1f} us per pass, {2 * mb / 1e3 / dt / 1e3:6.2f} TB/s (read + write)", flush=True)
