import ctypes as C, torch, sys, os
sys.path.insert(0, os.getcwd())
from tests.test_kernels_gpu import DEV, L, ptr, stream
lib = L().lib()
g = torch.Generator().manual_seed(41)
cin, cout, B, N = 128, 384, 1, 512
x = torch.randn(B, N, cin, generator=g) * torch.logspace(-1, 1, N).view(1, N, 1)
w = torch.randn(cout, cin, generator=g) / cin ** 0.5
gain = 1 + 0.3 * torch.randn(cin, generator=g)
v = x.double(); xn = v / v.norm(dim=-1, keepdim=True) * gain.double() * cin ** 0.5
want = xn @ w.double().t()
dx = x.contiguous().to(DEV); out = torch.zeros(B, N, cout, device=DEV)
L().check(lib.srgd_k_conv1x1_split_rms(ptr(dx), cin, B, N, ptr(w.contiguous()), ptr(None), cout, ptr(gain.contiguous()), ptr(None), ptr(None), ptr(out), stream()), "x")
got = out.cpu().double()
err = (got - want).abs()
print("max", err.max().item())
pe = err[0].max(dim=1).values
bad = (pe > 1e-4).nonzero().flatten()
print("bad pixels", bad.numel(), bad[:40].tolist())
ce = err[0].max(dim=0).values
badc = (ce > 1e-4).nonzero().flatten()
print("bad channels", badc.numel(), badc[:40].tolist())
if bad.numel():
    p0 = bad[0].item()
    print("ratio got/want at first bad pixel", (got[0, p0, :8] / want[0, p0, :8]).tolist())
    print("norm", x[0, p0].norm().item(), "neighbour norms", [x[0, q].norm().item() for q in range(max(0,p0-2), p0+3)])
d = (got[0, 0] - want[0, 0]).abs()
print("pixel 0 bad channels:", (d > 1e-4).nonzero().flatten().tolist()[:200])
d = (got[0, 17] - want[0, 17]).abs()
print("pixel 17 bad channels:", (d > 1e-4).nonzero().flatten().tolist()[:200])
print("pixel 17 got/want", (got[0,17,:40]/want[0,17,:40]).tolist())
