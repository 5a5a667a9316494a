"""CPU tests of the host side: config/CLI surface, state_dict schema, tile geometry helpers,
C-ABI library load + exported symbols (no compute without a GPU), loud failure off-GPU."""
import json
import logging
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(os.path.dirname(__file__), "golden")


def test_yaml_loads_unchanged_and_unknown_key_raises(tmp_path):
    from srgd_amd.config import load_config
    conf = load_config(os.path.join(ROOT, "conf", "conditional_continuous_linear_df8kost_dim128.yaml"))
    assert conf.model == "conditional_continuous" and conf.unet_dim == 128 and conf.noise_schedule == "linear"
    assert conf.ddpm_unet_dim_mults == "1,2,4,8" and conf.learned_sinusoidal_dim == 32 and conf.image_size == 256
    assert conf.full_attn == "False,False,False,True" and conf.num_classes == 3 and conf.load_strict is True
    bad = tmp_path / "bad.yaml"
    bad.write_text("unet_dim: 64\nnot_a_field: 1\n")
    with pytest.raises(TypeError):
        load_config(str(bad))


def test_state_dict_schema_is_the_reference_schema():
    from srgd_amd.config import load_config
    from srgd_amd.model import get_model
    conf = load_config(os.path.join(ROOT, "conf", "conditional_continuous_linear_df8kost_dim128.yaml"))
    for dim in (16, 128):
        conf.unet_dim = dim
        ema = get_model(conf, logging.getLogger("t"))
        sd = ema.module.state_dict()
        with open(os.path.join(G, f"schema_dim{dim}.json")) as f:
            ref = json.load(f)
        assert list(sd.keys()) == list(ref.keys())
        assert all(tuple(sd[k].shape) == tuple(ref[k]) for k in ref)
    assert sum(v.numel() for v in sd.values()) == 137_569_939


def test_checkpoint_roundtrip_strict(tmp_path):
    from srgd_amd.config import load_config
    from srgd_amd.model import get_model
    from srgd_amd.synth import synth_state_dict
    conf = load_config(os.path.join(ROOT, "conf", "conditional_continuous_linear_df8kost_dim128.yaml"))
    conf.unet_dim = 16
    with open(os.path.join(G, "schema_dim16.json")) as f:
        schema = {k: tuple(v) for k, v in json.load(f).items()}
    sd = synth_state_dict(schema, seed=3)
    path = tmp_path / "ckpt.pth"
    torch.save({"ema_model": sd}, path)
    conf.ckpt_path = str(path)
    ema = get_model(conf, logging.getLogger("t"))
    got = ema.module.state_dict()
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    del sd["model.final_conv.bias"]
    torch.save({"ema_model": sd}, path)
    with pytest.raises(RuntimeError):
        get_model(conf, logging.getLogger("t"))


def test_geometry_helpers_match_golden_tables():
    from srgd_amd.model import get_area, get_coord_and_pad, get_coords
    from tests.golden import cases as C
    with open(os.path.join(G, "geometry.json")) as f:
        geo = json.load(f)
    for (h, w) in C.GEOMETRY_SIZES:
        want = geo[f"{h}x{w}"]
        box, pad = get_coord_and_pad(h, w)
        assert list(box) == want["box"] and list(pad) == want["pad"]
        hp, wp = want["canvas"]
        even = get_coords(hp, wp, 256, 256, diff=0)
        odd = even if (hp <= 256 and wp <= 256) else get_coords(hp - 256, wp - 256, 256, 256, diff=128)
        assert len(even) == want["n_even"] and len(odd) == want["n_odd"]
        if not want["truncated"]:
            assert [list(c) for c in even] == want["even"] and [list(c) for c in odd] == want["odd"]
        inner, ipad = get_area(odd, hp, wp)
        assert list(inner) == want["inner"] and list(ipad) == want["inner_pad"]
    # ragged stride: last tile pulled back to the border
    assert get_coords(600, 256, 256, 200) == [(0, 256, 0, 256), (200, 456, 0, 256), (344, 600, 0, 256)]


def test_geometry_helpers_agree_with_the_pinned_oracle_on_random_sizes():
    # the product's int helpers (re-derived, math.ceil form) against the oracle's restatement - itself pinned 0.0 against the
    # reference's get_coord_and_pad / get_coords / get_area (oracle/pin_against_reference.py) - on sizes the tables do not hold:
    # one-pixel images, primes, sizes at and around tile multiples, ragged strides
    import random
    from oracle import srgd_oracle as O
    from srgd_amd.model import get_area, get_coord_and_pad, get_coords
    rng = random.Random(7)
    sizes = [(1, 1), (255, 257), (256, 512), (511, 513), (769, 1023), (2047, 31)] + \
            [(rng.randint(1, 3000), rng.randint(1, 3000)) for _ in range(300)]
    for (h, w) in sizes:
        assert get_coord_and_pad(h, w) == O.canvas_box_and_pad(h, w), (h, w)
        _, pad = get_coord_and_pad(h, w)
        hp, wp = h + pad[2] + pad[3], w + pad[0] + pad[1]
        even = get_coords(hp, wp, 256, 256, diff=0)
        odd = even if (hp <= 256 and wp <= 256) else get_coords(hp - 256, wp - 256, 256, 256, diff=128)
        oe, oo = O.sampling_grids(hp, wp)
        assert even == oe and odd == oo, (h, w)
        assert get_area(odd, hp, wp) == O.grid_bbox(oo, hp, wp), (h, w)
    for _ in range(200):                                  # ragged strides / shifts (tile_grid is the oracle's get_coords)
        hh, ww, stride, shift = rng.randint(256, 2000), rng.randint(256, 2000), rng.randint(1, 256), rng.choice((0, 128))
        assert get_coords(hh, ww, 256, stride, diff=shift) == O.tile_grid(hh, ww, 256, stride, shift), (hh, ww, stride, shift)


def test_schedule_scalars_bit_identical_to_oracle():
    from oracle import srgd_oracle as O
    from srgd_amd.model import _schedule
    for n in (10, 50):
        scalars, log_snrs = _schedule(n)
        steps = torch.linspace(1.0, 0.0, n + 1)
        for i in range(n):
            s = O.step_scalars(steps[i], steps[i + 1])
            assert scalars[i].alpha == float(s["alpha"]) and scalars[i].sigma == float(s["sigma"])
            assert scalars[i].alpha_next == float(s["alpha_next"]) and scalars[i].c == float(s["c"])
            assert scalars[i].noise_scale == float(s["var"].sqrt()) and log_snrs[i] == float(s["log_snr"])
            assert scalars[i].one_minus_c == float(1 - s["c"])


def test_cli_flags_are_the_reference_flags():
    from srgd_amd.inference import parse_args
    a = parse_args(["-c", "x.yaml", "-m", "w.pth", "--input_dir", "i", "--output_dir", "o"])
    assert (a.batch_size, a.num_sample_steps, a.interpolation, a.cond_scale, a.class_cond_scale) == (8, 250, "bicubic", 1.0, 1.0)
    assert (a.guidance_start_steps, a.class_guidance_start_steps, a.generation_start_steps) == (0, 0, 0)
    assert (a.start_index, a.end_index, a.test_label, a.amp, a.use_dpmpp_solver, a.seed, a.backend) == \
        (0, None, None, True, True, 71, "ddp")
    a = parse_args(["-c", "x", "-m", "w", "--input_dir", "i", "--output_dir", "o", "--no_amp", "--test_label", "0"])
    assert a.amp is False and a.test_label == 0
    # engine-only switch: the default precision is the one that meets the reference's 1e-3 bar at matrix-core speed (f16x3, round 6;
    # fp32 = exact-fp32 MFMA stays selectable); throughput modes are explicit opt-ins
    assert a.precision == "f16x3"
    assert parse_args(["-c", "x", "-m", "w", "--input_dir", "i", "--output_dir", "o", "--precision", "bf16"]).precision == "bf16"


def test_image_io_conventions():
    from PIL import Image
    from srgd_amd.inference import pil_to_unit_tensor, unit_tensor_to_pil
    t = torch.tensor([[[0.0, 0.999, 1.0]], [[0.5, 0.25, 0.0]], [[1.0 / 255, 254.9 / 255, 0.3]]])
    img = unit_tensor_to_pil(t)
    assert img.getpixel((1, 0)) == (254, 63, 254)          # truncation, like ToPILImage's mul(255).byte()
    back = pil_to_unit_tensor(img)
    assert back.shape == (3, 1, 3) and back.max() <= 1.0 and float(back[0, 0, 2]) == 1.0


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    from srgd_amd import _lib
    from srgd_amd.build import build
    build()
    lib = _lib.lib()                                      # binds every prototype; AttributeError if one is missing
    declared = set()
    for header in ("srgd_hip.h", "srgd_hip_kernels.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(srgd_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    for name in declared:
        assert hasattr(lib, name)
    assert b"gfx950" in lib.srgd_version()
    # ... and nothing else: the dynamic symbol table (nm -D) holds exactly the headers' functions (-fvisibility=hidden), so
    # the export count quoted in INTEGRATION.md / DESIGN.md cannot drift from the headers again
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if len(ln.split()) >= 3 and ln.split()[-2] in "TtWw"}
    exported = {n for n in exported if not n.startswith(("_init", "_fini", "__"))}
    assert exported == declared, exported ^ declared


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from srgd_amd._lib import SrgdHipError
    from srgd_amd.config import load_config
    from srgd_amd.model import get_model
    conf = load_config(os.path.join(ROOT, "conf", "conditional_continuous_linear_df8kost_dim128.yaml"))
    conf.unet_dim = 16
    sampler = get_model(conf, logging.getLogger("t")).module.eval()
    with pytest.raises(SrgdHipError):
        sampler.tiled_sample(condition_x=torch.rand(1, 3, 256, 256), num_sample_steps=2)
    with pytest.raises(SrgdHipError):
        sampler.model(torch.zeros(1, 3, 128, 128), torch.zeros(1))


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "srgd_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_e4m3_weight_rounding_matches_torch_float8():
    # the fp8-weight mode (SRGD_PRECISION_BF16_W8) rounds weights on the host; the rounding must be OCP e4m3 "fn"
    # round-to-nearest-even exactly as torch.float8_e4m3fn does it (saturating at 448 instead of NaN beyond)
    import ctypes as C
    from srgd_amd import _lib
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(50000, generator=g) * s for s in (1e-3, 0.1, 1.0, 30.0, 300.0)] +
                  [torch.tensor([0.0, -0.0, 448.0, 449.0, 463.9, 464.0, 500.0, -1000.0, 2.0 ** -9, 2.0 ** -10,
                                 1.5 * 2.0 ** -10, 0.99 * 2.0 ** -10, 0.0175, 0.017578125])]).contiguous()
    out = torch.empty_like(x)
    _lib.check(_lib.lib().srgd_quantize_e4m3(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.numel(), 1.0), "quantize")
    assert torch.equal(out, x.clamp(-448, 448).to(torch.float8_e4m3fn).float())
    # per-channel scale: result is scale * e4m3(x / scale)
    _lib.check(_lib.lib().srgd_quantize_e4m3(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.numel(), 0.37), "quantize")
    assert torch.equal(out, (x / 0.37).clamp(-448, 448).to(torch.float8_e4m3fn).float() * 0.37)


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    # `--gpus 2` under a launcher that started 4 ranks (or `--gpus 1` under 2) must fail before touching any GPU,
    # never print a line labelled with the wrong n_gpus (VERDICT r1 / ADVICE r1)
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and "{" not in r.stdout


def test_cli_rank_file_ranges_partition_the_selected_files():
    # inference.py under a launcher (RANK / WORLD_SIZE): contiguous per-rank slices of [start_index:end_index], the automatic
    # form of the reference's hand-set --start_index/--end_index per process (inference.py:36-37, :120)
    from srgd_amd.inference import rank_file_range
    for n in (0, 1, 7, 8, 64, 1000):
        for (s0, e0) in ((0, None), (3, None), (0, 5), (2, 50), (10, 4)):
            want = list(range(n))[s0:e0]
            for world in (1, 2, 3, 8):
                got = []
                sizes = []
                for r in range(world):
                    a, b = rank_file_range(n, s0, e0, r, world)
                    got += list(range(n))[a:b]
                    sizes.append(b - a)
                assert got == want, (n, s0, e0, world)
                assert max(sizes) - min(sizes) <= 1


def test_reflect_padding_larger_than_the_image_is_refused_like_upstream():
    # F.pad(mode="reflect") at model.py:3303 raises when a pad is >= the dimension it mirrors; the product raises the same
    # RuntimeError before touching the GPU (srgd_amd/model.py::_tiling), e.g. a 60-pixel-high image padded to 256 rows
    from srgd_amd.model import _tiling
    with pytest.raises(RuntimeError, match="Padding size should be less than the corresponding input dimension"):
        _tiling(60, 300, 256, 256)
    with pytest.raises(RuntimeError):
        _tiling(300, 100, 256, 256)
    (left, top, right, bottom), (hp, wp), even, odd, inner = _tiling(200, 200, 256, 256)     # 28-pixel pads: fine
    assert (hp, wp) == (256, 256) and len(even) == 1 and len(odd) == 1 and (right - left, bottom - top) == (200, 200)


def test_edm_step_tables_accept_more_call_steps_than_constructor_steps():
    # sample_org (model.py:2212-2306) never re-noises a ring, so a per-call step count beyond the constructor's
    # schedule must build its tables; the tiled loop (model.py:2457 -> :2187) indexes the constructor's schedule on odd steps
    # and fails mid-loop at the first odd step beyond it - the table marks those steps (nan) instead of failing up front
    import math
    from srgd_amd.model import ConditionalElucidatedDiffusionSR, ConditionalSRUnet
    unet = ConditionalSRUnet(dim=16, dim_mults=(1, 2, 4, 8), full_attn=(False, False, False, True), num_classes=3,
                             learned_sinusoidal_cond=True, pixel_shuffle_upsample=True)
    edm = ConditionalElucidatedDiffusionSR(unet, image_size=256, num_sample_steps=8)
    for n in (10, 12, 32):
        sigmas, noised, scalars, c_noise = edm._step_tables(n, True, with_ring=False)
        assert len(scalars) == n and len(c_noise) == 2 * n and all(s.ring_sigma == 0.0 for s in scalars)
        _, noised, scalars, _ = edm._step_tables(n, True)                  # tiled: no failure at table build either
        assert len(noised) == 9
        for i, s in enumerate(scalars):
            if i % 2 == 0:
                assert s.ring_sigma == 0.0
            elif i < 9:
                assert s.ring_sigma == pytest.approx(float(noised[i]))
            else:
                assert math.isnan(s.ring_sigma)                            # tiled_sample raises IndexError at this step
    _, _, scalars, _ = edm._step_tables(8, True)                            # in range: every odd step carries its sigma
    assert all(s.ring_sigma > 0 for s in scalars[1::2])


def test_step_lanes_rule():
    # srgd_amd.lanes: two concurrent halves only when the whole step is ONE launch of at most 100 samples (tiles x passes)
    from srgd_amd.lanes import lanes_wanted
    assert lanes_wanted(25, 1, 25, None) == 2 and lanes_wanted(16, 1, 25, None) == 2        # configs[1], one HR tile
    assert lanes_wanted(25, 2, 25, None) == 2                                                # configs[4]: 50 samples
    assert lanes_wanted(100, 1, 100, None) == 2 and lanes_wanted(125, 1, 125, None) == 1     # four / five HR tiles in lock-step
    assert lanes_wanted(75, 2, 75, None) == 1                                                # 150 samples
    assert lanes_wanted(25, 1, 4, None) == 1                                                 # several launches per step
    assert lanes_wanted(1, 1, 4, None) == 1 and lanes_wanted(1, 1, 4, 2) == 1                # a single tile cannot be split
    assert lanes_wanted(125, 1, 125, 2) == 2 and lanes_wanted(25, 1, 25, 1) == 1             # forced
    # automatic only in the precisions it was A/B-measured in: the parity modes keep one lane unless forced
    assert lanes_wanted(25, 1, 25, None, "fp32") == 1 and lanes_wanted(125, 1, 125, None, "fp32") == 1
    assert lanes_wanted(25, 1, 25, None, "f16x3") == 2 and lanes_wanted(125, 1, 125, None, "f16x3") == 2      # measured in round 6
    assert lanes_wanted(125, 2, 125, None, "f16x3") == 1 and lanes_wanted(125, 1, 64, None, "f16x3") == 1
    assert lanes_wanted(25, 1, 25, 2, "fp32") == 2 and lanes_wanted(25, 1, 25, None, "fp8") == 2
    import os
    from srgd_amd.lanes import lanes_setting_from_env
    keep = os.environ.get("SRGD_STEP_LANES")
    try:
        for v, want in (("0", 1), ("1", 1), ("2", 2), ("7", 2), ("", None), ("auto", None)):
            os.environ["SRGD_STEP_LANES"] = v
            assert lanes_setting_from_env() == want, v
    finally:
        if keep is None:
            os.environ.pop("SRGD_STEP_LANES", None)
        else:
            os.environ["SRGD_STEP_LANES"] = keep
