"""Practical bf16 MFMA ceiling of this MI355X box: hipBLASLt GEMMs through torch.matmul (not part of the product; a yardstick
for the roofline discussion in DESIGN.md)."""
import time
import torch

for (m, n, k) in [(8192, 8192, 8192), (16384, 8192, 4096), (65536, 1024, 9216), (131072, 128, 1152)]:
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(k, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        (a @ b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = 20
    for _ in range(it):
        (a @ b)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / it
    print(f"bf16 GEMM {m}x{n}x{k}: {2.0 * m * n * k / dt / 1e12:7.1f} TFLOP/s ({dt * 1e3:.3f} ms)", flush=True)
