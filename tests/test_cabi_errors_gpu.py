"""Error behaviour of the C ABI called directly (ctypes, no Python wrapper): every misuse returns non-zero with a message in
srgd_last_error() and leaves the engine usable - no aborts, no faults."""
import ctypes as C
import json
import os

import pytest
import torch

from srgd_amd import _lib
from srgd_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _cfg(dim=16, precision=0):
    cfg = _lib.UnetConfig()
    cfg.dim, cfg.n_stages, cfg.channels, cfg.groups, cfg.heads, cfg.dim_head = dim, 4, 3, 8, 4, 32
    cfg.sinus_dim, cfg.num_classes, cfg.precision, cfg.device = 32, 3, precision, 0
    for i, (m, f) in enumerate(zip((1, 2, 4, 8), (0, 0, 0, 1))):
        cfg.dim_mults[i], cfg.full_attn[i] = m, f
    return cfg


def _err(L, rc):
    assert rc != 0
    msg = L.srgd_last_error().decode()
    assert msg
    return msg


def test_misuse_returns_errors_and_engine_stays_usable():
    L = _lib.lib()
    assert _err(L, L.srgd_create(None, None))
    h = C.c_void_p()
    assert L.srgd_create(C.byref(_cfg()), C.byref(h)) == 0
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # weights: unknown key, wrong shape, finalize with tensors missing, forward before finalize
    one = torch.zeros(4)
    shp = (C.c_int64 * 1)(4)
    assert "Unexpected key" in _err(L, L.srgd_load_weight(h, b"nope.weight", C.c_void_p(one.data_ptr()), shp, 1))
    assert "size mismatch" in _err(L, L.srgd_load_weight(h, b"init_conv.bias", C.c_void_p(one.data_ptr()), shp, 1))
    assert "Missing key" in _err(L, L.srgd_finalize_weights(h))
    x = torch.zeros(1, 3, 128, 128, device="cuda")
    ls = (C.c_float * 1)(0.0)
    assert _err(L, L.srgd_unet_forward(h, C.c_void_p(x.data_ptr()), None, ls, -1, C.c_void_p(x.data_ptr()), 1, 128, 128, st))
    # load real weights
    with open(os.path.join(G, "schema_dim16.json")) as f:
        schema = {k: tuple(v) for k, v in json.load(f).items()}
    for k, v in synth_state_dict(schema, seed=0).items():
        t = v.float().contiguous()
        s = (C.c_int64 * max(1, t.dim()))(*t.shape)
        assert L.srgd_load_weight(h, k.encode(), C.c_void_p(t.data_ptr()), s, t.dim()) == 0, L.srgd_last_error()
    assert L.srgd_finalize_weights(h) == 0, L.srgd_last_error()
    assert "already finalized" in _err(L, L.srgd_load_weight(h, b"init_conv.bias", C.c_void_p(one.data_ptr()), shp, 1))
    # U-Net: bad spatial size (not divisible by 8), class label out of range
    bad = torch.zeros(1, 3, 100, 100, device="cuda")
    assert _err(L, L.srgd_unet_forward(h, C.c_void_p(bad.data_ptr()), None, ls, -1, C.c_void_p(bad.data_ptr()), 1, 100, 100, st))
    assert _err(L, L.srgd_unet_forward(h, C.c_void_p(x.data_ptr()), None, ls, 7, C.c_void_p(x.data_ptr()), 1, 128, 128, st))
    # sampler: step / end / exchange / EDM step without begin
    img = torch.zeros(1, 3, 256, 256, device="cuda")
    assert "begin first" in _err(L, L.srgd_sampler_step(h, 0, C.c_void_p(img.data_ptr()), C.c_void_p(img.data_ptr()), None, None, None,
                                                        1, 0, 1.0, 4, 0, st))
    assert _err(L, L.srgd_sampler_end(h, C.c_void_p(img.data_ptr()), C.c_void_p(img.data_ptr()), st))
    assert _err(L, L.srgd_edm_step(h, 0, C.c_void_p(img.data_ptr()), C.c_void_p(img.data_ptr()), None, C.c_void_p(img.data_ptr()),
                                   None, None, 1, 0, 1.0, 4, 0, st))
    # begin with bad geometry (reflect pad >= image), then a good one; step argument checks
    # a 100 x 300 image: the reference's canvas is 512 x 768, top pad 206 >= H -> F.pad(reflect) raises (model.py:3303)
    geo = _lib.SamplerGeometry(H=100, W=300, Hp=512, Wp=768, left=234, top=206, inner_l=128, inner_t=128, inner_r=640, inner_b=384,
                               tile=256, n_even=2, n_odd=1, n_images=1)
    tiles = (C.c_int32 * 4)(0, 0, 0, 256)
    sc = (_lib.StepScalars * 2)()
    lsn = (C.c_float * 2)(0.5, -0.5)
    cond = torch.zeros(1, 3, 100, 300, device="cuda")
    canvas = torch.zeros(1, 3, 512, 768, device="cuda")
    assert "Padding size" in _err(L, L.srgd_sampler_begin(h, C.byref(geo), C.c_void_p(cond.data_ptr()), C.c_void_p(canvas.data_ptr()),
                                                          tiles, tiles, 2, sc, lsn, 0, st))
    geo = _lib.SamplerGeometry(H=256, W=256, Hp=256, Wp=256, left=0, top=0, inner_l=0, inner_t=0, inner_r=256, inner_b=256,
                               tile=256, n_even=1, n_odd=1, n_images=1)
    outside = (C.c_int32 * 2)(8, 0)
    tile0 = (C.c_int32 * 2)(0, 0)
    cond = torch.rand(1, 3, 256, 256, device="cuda")
    assert "outside the canvas" in _err(L, L.srgd_sampler_begin(h, C.byref(geo), C.c_void_p(cond.data_ptr()), C.c_void_p(img.data_ptr()),
                                                                outside, tile0, 2, sc, lsn, 0, st))
    cc = torch.zeros(1, 3, 256, 256, device="cuda")
    assert L.srgd_sampler_begin(h, C.byref(geo), C.c_void_p(cond.data_ptr()), C.c_void_p(cc.data_ptr()), tile0, tile0, 2, sc, lsn,
                                0, st) == 0, L.srgd_last_error()
    p_img, p_cc = C.c_void_p(img.data_ptr()), C.c_void_p(cc.data_ptr())
    assert "out of range" in _err(L, L.srgd_sampler_step(h, 5, p_img, p_cc, None, None, None, 1, 0, 1.0, 4, 0, st))
    assert "passes" in _err(L, L.srgd_sampler_step(h, 0, p_img, p_cc, None, None, None, 3, 0, 1.0, 4, 0, st))
    assert "guidance_kind" in _err(L, L.srgd_sampler_step(h, 0, p_img, p_cc, None, None, None, 2, 0, 1.5, 4, 0, st))
    assert "sub_batch" in _err(L, L.srgd_sampler_step(h, 0, p_img, p_cc, None, None, None, 1, 0, 1.0, 0, 0, st))
    assert "tile range" in _err(L, L.srgd_sampler_step_tiles(h, 0, 1, 1, 1, p_img, p_cc, None, None, None, 1, 0, 1.0, 4, 0, st))
    assert "tile range" in _err(L, L.srgd_sampler_exchange_tiles(h, 0, 0, 2, p_img, p_img, 0, st))
    assert _err(L, L.srgd_edm_step(h, 0, p_img, p_cc, None, p_img, None, None, 1, 0, 1.0, 4, 0, st))      # DDPM run active, not EDM
    assert "srgd_edm_begin" in _err(L, L.srgd_edm_dpmpp_step(h, 0, p_img, p_cc, None, p_img, 1, 0, 1.0, 4, st))
    # ... and the engine still works: one valid step and the end of the run
    assert L.srgd_sampler_step(h, 0, p_img, p_cc, None, None, None, 1, 0, 1.0, 4, 1, st) == 0, L.srgd_last_error()
    out = torch.empty(1, 3, 256, 256, device="cuda")
    assert L.srgd_sampler_end(h, p_img, C.c_void_p(out.data_ptr()), st) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    # image helpers and the quantiser
    assert _err(L, L.srgd_image_resize_bicubic_u8(None, 4, 4, 16, 16, None, st))
    assert _err(L, L.srgd_image_unit_to_u8(None, 4, 4, None, st))
    assert _err(L, L.srgd_quantize_e4m3(None, None, 4, 1.0))
    assert L.srgd_destroy(h) == 0
