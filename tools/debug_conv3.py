"""Debug helper (GPU box): conv3x3_bf16 (impl 2) against F.conv2d on one small shape; prints where the results differ."""
import ctypes as C, os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srgd_amd import _lib
lib = _lib.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, c0, cout, H, W = 2, 32, 128, 8, 32
g = torch.Generator().manual_seed(11)
x = torch.randn(B, c0, H, W, generator=g).bfloat16().float()
w = (torch.randn(cout, c0, 3, 3, generator=g) / (3 * c0 ** 0.5)).bfloat16().float()
b = torch.randn(cout, generator=g)
d = x.permute(0, 2, 3, 1).contiguous().cuda().bfloat16()
out = torch.full((B, H, W, cout), 777.0, device="cuda", dtype=torch.bfloat16)
part = torch.full((B * 8 * 64 * 2,), float("nan"), device="cuda")
ms, slots = C.c_float(), C.c_int()
rc = lib.srgd_k_conv2d_timed(C.c_void_p(d.data_ptr()), C.c_void_p(0), c0, 0, B, H, W, 3, 1, 1, 0, C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), cout,
                             C.c_void_p(out.data_ptr()), C.c_void_p(0), C.c_void_p(part.data_ptr()), 8, 1, 2, 0, C.byref(ms), C.byref(slots), C.c_void_p(0), C.c_void_p(0), C.c_void_p(0), st)
_lib.check(rc, "conv"); torch.cuda.synchronize()
got = out.float().cpu().permute(0, 3, 1, 2)
want = F.conv2d(x, w, b, padding=1)
bad = ~((got - want).abs() <= 0.07)
print("slots", slots.value, "bad", int(bad.sum()), "of", bad.numel(), "nan", int(torch.isnan(got).sum()), "unwritten", int((got == 777.0).sum()))
idx = bad.nonzero()
print("bad channels", sorted(set(idx[:, 1].tolist()))[:64])
print("bad ys", sorted(set(idx[:, 2].tolist())), "bad xs", sorted(set(idx[:, 3].tolist())))
print(idx[:10].tolist()); print(got[bad][:10], want[bad][:10])
p = part[:B * 8 * slots.value * 2].view(B, 8, slots.value, 2).cpu()
print("part finite", bool(torch.isfinite(p).all()), p[0, :, :, 0])
print("want s1", want.reshape(B, 8, -1).sum(-1)[0])
