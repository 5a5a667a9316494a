"""Generate the committed golden fixtures by running the REFERENCE itself (build container only).

    python tests/golden/make_golden.py

Imports ``/root/reference/model.py`` through ``oracle/refshim.py`` (stubs for its un-vendored
imports), loads seeded synthetic weights (``srgd_amd.synth``; the published checkpoint is an LFS
pointer), and stores inputs' checksums + the reference's outputs.  The fixtures are data only.
Every case documents how its inputs are regenerated (seeds), so the GPU box - which has neither
the reference nor these scripts' inputs - can rebuild identical inputs with the same torch build.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import refshim                      # noqa: E402
from srgd_amd.synth import synth_state_dict     # noqa: E402
from tests.golden import cases as C             # noqa: E402


def main(only=None):
    ref = refshim.load_reference()
    assert ref is not None, "reference not present"
    rm, rc = ref
    torch.set_num_threads(8)
    if only:
        return make_samplers(rm, rc, only)

    # ---- G1 geometry -------------------------------------------------------------------
    geo = {}
    for (h, w) in C.GEOMETRY_SIZES:
        box, pad = rm.get_coord_and_pad(h, w)
        hp, wp = h + pad[2] + pad[3], w + pad[0] + pad[1]
        c0 = rm.get_coords(hp, wp, 256, 256, diff=0)
        c1 = (rm.get_coords(hp, wp, 256, 256, diff=0) if (hp <= 256 and wp <= 256)
              else rm.get_coords(hp - 256, wp - 256, 256, 256, diff=128))
        inner, ipad = rm.get_area(c1, hp, wp)
        big = len(c0) > 64
        geo[f"{h}x{w}"] = dict(box=list(box), pad=list(pad), canvas=[hp, wp], n_even=len(c0), n_odd=len(c1),
                               even=[list(c) for c in (c0[:3] + c0[-3:] if big else c0)],
                               odd=[list(c) for c in (c1[:3] + c1[-3:] if big else c1)],
                               truncated=big, inner=list(inner), inner_pad=list(ipad))
    with open(os.path.join(HERE, "geometry.json"), "w") as f:
        json.dump(geo, f, indent=1)

    # ---- G2 schedule scalars (fp32, bit patterns) -------------------------------------------
    sched = {}
    for n in C.SCHEDULE_STEPS:
        steps = torch.linspace(1.0, 0.0, n + 1)
        ls = torch.stack([rm.beta_linear_log_snr(steps[i]) for i in range(n + 1)])
        sched[f"log_snr_{n}"] = ls.numpy()
    np.savez(os.path.join(HERE, "schedule.npz"), **sched)

    # ---- schemas ---------------------------------------------------------------------------
    for dim in (16, 128):
        sampler, _ = refshim.build_reference_sampler(rm, rc, dim=dim)
        schema = {k: list(v.shape) for k, v in sampler.state_dict().items()}
        with open(os.path.join(HERE, f"schema_dim{dim}.json"), "w") as f:
            json.dump(schema, f, indent=0)

    # ---- G4 U-Net forward ----------------------------------------------------------------------
    out = {}
    for case in C.UNET_CASES:
        sampler, _ = refshim.build_reference_sampler(rm, rc, dim=case["dim"])
        schema = {k: tuple(v.shape) for k, v in sampler.state_dict().items()}
        sd = synth_state_dict(schema, seed=case["weight_seed"])
        sampler.load_state_dict(sd, strict=True)
        x, cnd, ls = C.unet_inputs(case)
        with torch.inference_mode():
            for mode in case["modes"]:
                label, c = C.unet_mode_args(mode, case, cnd)
                eps = sampler.model(x, ls, label, c)
                out[f"{case['name']}.{mode}"] = eps.numpy()
        out[f"{case['name']}.x_sum"] = np.float64(x.double().sum().item())
        out[f"{case['name']}.w_sum"] = np.float64(sum(v.double().abs().sum().item() for v in sd.values()))
    np.savez_compressed(os.path.join(HERE, "unet_eps.npz"), **out)

    make_samplers(rm, rc, None)


def make_samplers(rm, rc, only):
    # ---- G5-G7 tiled_sample ---------------------------------------------------------------------
    for case in C.SAMPLER_CASES:
        if only and case["name"] not in only:
            continue
        sampler, _ = refshim.build_reference_sampler(rm, rc, dim=case["dim"], num_sample_steps=case["steps"])
        schema = {k: tuple(v.shape) for k, v in sampler.state_dict().items()}
        sd = synth_state_dict(schema, seed=case["weight_seed"])
        sampler.load_state_dict(sd, strict=True)
        cond = C.sampler_condition(case)
        label = torch.tensor([case["label"]]) if case["label"] is not None else None
        torch.manual_seed(case["seed"])
        first = torch.randn(16)
        torch.manual_seed(case["seed"])
        with torch.inference_mode():
            img = sampler.tiled_sample(batch_size=case["batch_size"], condition_x=cond.clone(),
                                       class_label=label, cond_scale=case["cond_scale"],
                                       class_cond_scale=case["class_cond_scale"],
                                       num_sample_steps=case["steps"], **C.extra_kwargs(case))
        np.savez_compressed(os.path.join(HERE, f"sample_{case['name']}.npz"),
                            image=img.numpy(), cond_sum=np.float64(cond.double().sum().item()),
                            first_draw=first.numpy(),
                            w_sum=np.float64(sum(v.double().abs().sum().item() for v in sd.values())))
        print(case["name"], "done", tuple(img.shape), float(img.mean()))


def make_edm(only=None):
    """EDM wrapper fixtures (conf.model = 'conditional_elucidated', model.py:3593-3614)."""
    ref = refshim.load_reference()
    assert ref is not None, "reference not present"
    rm, rc = ref
    torch.set_num_threads(8)
    for case in C.EDM_CASES:
        if only and case["name"] not in only:
            continue
        sampler, _ = refshim.build_reference_sampler(rm, rc, dim=case["dim"],
                                                     num_sample_steps=case.get("ctor_steps", case["steps"]),
                                                     model="conditional_elucidated")
        schema = {k: tuple(v.shape) for k, v in sampler.state_dict().items()}
        assert all(k.startswith("net.") for k in schema)
        sd = synth_state_dict(schema, seed=case["weight_seed"])
        sampler.load_state_dict(sd, strict=True)
        cond = C.sampler_condition(case)
        label = torch.tensor([case["label"]]) if case["label"] is not None else None
        torch.manual_seed(case["seed"])
        with torch.inference_mode():
            img = sampler.tiled_sample(batch_size=case["batch_size"], condition_x=cond.clone(), class_label=label,
                                       cond_scale=case["cond_scale"], class_cond_scale=case["class_cond_scale"],
                                       num_sample_steps=case["steps"], **C.edm_extra_kwargs(case))
        np.savez_compressed(os.path.join(HERE, f"sample_edm_{case['name']}.npz"), image=img.numpy(),
                            cond_sum=np.float64(cond.double().sum().item()),
                            w_sum=np.float64(sum(v.double().abs().sum().item() for v in sd.values())))
        print("edm", case["name"], "done", tuple(img.shape), float(img.mean()))


def make_edm_untiled(only=None):
    """Un-tiled EDM fixtures: ``sample`` dispatching to sample_org / sample_using_dpmpp (model.py:2196-2209)."""
    ref = refshim.load_reference()
    assert ref is not None, "reference not present"
    rm, rc = ref
    torch.set_num_threads(8)
    for case in C.EDM_UNTILED_CASES:
        if only and case["name"] not in only:
            continue
        sampler, _ = refshim.build_reference_sampler(rm, rc, dim=case["dim"],
                                                     num_sample_steps=case.get("ctor_steps", case["steps"]),
                                                     model="conditional_elucidated")
        sampler.use_dpmpp_solver = case["dpmpp"]         # the constructor argument's attribute (model.py:2126), read at :2200
        schema = {k: tuple(v.shape) for k, v in sampler.state_dict().items()}
        sd = synth_state_dict(schema, seed=case["weight_seed"])
        sampler.load_state_dict(sd, strict=True)
        cond = C.sample_condition(case)
        label = torch.tensor([case["label"]])
        torch.manual_seed(case["seed"])
        with torch.inference_mode():
            img = sampler.sample(batch_size=case["batch"], condition_x=cond.clone(), class_label=label,
                                 cond_scale=case["cond_scale"], class_cond_scale=case["class_cond_scale"],
                                 num_sample_steps=case["steps"], **C.edm_extra_kwargs(case))
        np.savez_compressed(os.path.join(HERE, f"sample_edm_untiled_{case['name']}.npz"), image=img.numpy(),
                            cond_sum=np.float64(cond.double().sum().item()))
        print("edm un-tiled", case["name"], "done", tuple(img.shape), float(img.mean()))


def make_modules():
    """G3: per-module outputs of the reference U-Net's own sub-modules."""
    ref = refshim.load_reference()
    assert ref is not None, "reference not present"
    rm, rc = ref
    torch.set_num_threads(8)
    sampler, _ = refshim.build_reference_sampler(rm, rc, dim=C.MODULE_DIM)
    schema = {k: tuple(v.shape) for k, v in sampler.state_dict().items()}
    sampler.load_state_dict(synth_state_dict(schema, seed=0), strict=True)
    unet = sampler.model
    out = {}
    t = C.module_time_embedding()
    with torch.inference_mode():
        for name, (path, _) in C.MODULE_CASES.items():
            mod = unet.get_submodule(path)
            x = C.module_input(name)
            y = mod(x, t[:x.shape[0]]) if name.startswith("resnet") else mod(x)
            out[name] = y.numpy()
        ls = torch.tensor([-3.0, 2.5])
        out["time_mlp"] = unet.time_mlp(ls).numpy()
        out["class_mlp"] = unet.class_mlp(torch.tensor([1])).numpy()
    np.savez_compressed(os.path.join(HERE, f"modules_dim{C.MODULE_DIM}.npz"), **out)
    print("modules done", {k: v.shape for k, v in out.items()})


def make_sample():
    """un-tiled sample() fixtures."""
    ref = refshim.load_reference()
    assert ref is not None, "reference not present"
    rm, rc = ref
    torch.set_num_threads(8)
    for case in C.SAMPLE_CASES:
        sampler, _ = refshim.build_reference_sampler(rm, rc, dim=case["dim"], num_sample_steps=case["steps"])
        schema = {k: tuple(v.shape) for k, v in sampler.state_dict().items()}
        sd = synth_state_dict(schema, seed=case["weight_seed"])
        sampler.load_state_dict(sd, strict=True)
        cond = C.sample_condition(case)
        label = torch.tensor([case["label"]])
        torch.manual_seed(case["seed"])
        with torch.inference_mode():
            img = sampler.sample(batch_size=case["batch"], condition_x=cond.clone(), class_label=label,
                                 cond_scale=case["cond_scale"], class_cond_scale=case["class_cond_scale"],
                                 num_sample_steps=case["steps"], **C.sample_extra_kwargs(case))
        np.savez_compressed(os.path.join(HERE, f"sample_untiled_{case['name']}.npz"), image=img.numpy(),
                            cond_sum=np.float64(cond.double().sum().item()))
        print("sample", case["name"], "done", tuple(img.shape), float(img.mean()))


def make_long(cases=None):
    ref = refshim.load_reference()
    assert ref is not None, "reference not present"
    rm, rc = ref
    torch.set_num_threads(8)
    for case in (C.LONG_CASES if cases is None else cases):
        sampler, _ = refshim.build_reference_sampler(rm, rc, dim=case["dim"], num_sample_steps=case["steps"])
        schema = {k: tuple(v.shape) for k, v in sampler.state_dict().items()}
        sd = synth_state_dict(schema, seed=case["weight_seed"])
        sampler.load_state_dict(sd, strict=True)
        cond = C.sampler_condition(case)
        label = torch.tensor([case["label"]])
        torch.manual_seed(case["seed"])
        with torch.inference_mode():
            img = sampler.tiled_sample(batch_size=case["batch_size"], condition_x=cond.clone(), class_label=label,
                                       cond_scale=case["cond_scale"], class_cond_scale=case["class_cond_scale"],
                                       num_sample_steps=case["steps"])
        arrays = dict(cond_sum=np.float64(cond.double().sum().item()),
                      w_sum=np.float64(sum(v.double().abs().sum().item() for v in sd.values())))
        if img.shape[-1] > 512:
            # a 1024^2 fp32 image is 12.6 MB: store it as uint16 steps of 1/65535 (max rounding error 7.7e-6, two orders
            # inside the 1e-3 bar and below the reference's own thread-count noise) plus an fp64 checksum of the original
            arrays["image_u16"] = torch.round(img.clamp(0, 1) * 65535.0).to(torch.int32).numpy().astype(np.uint16)
            arrays["checksum"] = np.float64(img.double().sum().item())
        else:
            arrays["image"] = img.numpy()
        np.savez_compressed(os.path.join(HERE, f"sample_{case['name']}.npz"), **arrays)
        print("long", case["name"], "done", tuple(img.shape), float(img.mean()))


def make_full(cases=None, trace_steps=None):
    """BASELINE configs[1] (or, --config5-full, configs[4]) at full length, with the reference's own trajectory (with_images /
    with_x0_images)."""
    trace_steps = C.FULL_TRACE_STEPS if trace_steps is None else trace_steps
    ref = refshim.load_reference()
    assert ref is not None, "reference not present"
    rm, rc = ref
    torch.set_num_threads(int(os.environ.get("SRGD_GOLDEN_THREADS", "8")))
    for case in (C.FULL_CASES if cases is None else cases):
        sampler, _ = refshim.build_reference_sampler(rm, rc, dim=case["dim"], num_sample_steps=case["steps"])
        schema = {k: tuple(v.shape) for k, v in sampler.state_dict().items()}
        sd = synth_state_dict(schema, seed=case["weight_seed"])
        sampler.load_state_dict(sd, strict=True)
        cond = C.sampler_condition(case)
        label = torch.tensor([case["label"]])
        torch.manual_seed(case["seed"])
        with torch.inference_mode():
            img, xt_list, x0_list = sampler.tiled_sample(
                batch_size=case["batch_size"], condition_x=cond.clone(), class_label=label,
                cond_scale=case["cond_scale"], class_cond_scale=case["class_cond_scale"],
                num_sample_steps=case["steps"], with_images=True, with_x0_images=True)
        # entry 0 of both lists is the cropped start image (model.py:3319, :3323); entry i + 1 is the canvas after step i
        arrays = dict(cond_sum=np.float64(cond.double().sum().item()),
                      w_sum=np.float64(sum(v.double().abs().sum().item() for v in sd.values())),
                      image_u16=torch.round(img.clamp(0, 1) * 65535.0).to(torch.int32).numpy().astype(np.uint16),
                      checksum=np.float64(img.double().sum().item()),
                      xt_sum=np.array([t.double().sum().item() for t in xt_list[1:]]),
                      xt_abs=np.array([t.double().abs().sum().item() for t in xt_list[1:]]),
                      x0_sum=np.array([t.double().sum().item() for t in x0_list[1:]]),
                      x0_abs=np.array([t.double().abs().sum().item() for t in x0_list[1:]]),
                      trace_steps=np.array(trace_steps))
        for i in trace_steps:
            arrays[f"xt_{i}"] = C.trace_planes(xt_list[i + 1]).numpy().copy()
            arrays[f"x0_{i}"] = C.trace_planes(x0_list[i + 1]).numpy().copy()
        np.savez_compressed(os.path.join(HERE, f"sample_{case['name']}.npz"), **arrays)
        print("full", case["name"], "done", tuple(img.shape), float(img.mean()), flush=True)


if __name__ == "__main__":
    if "--config2-full" in sys.argv:
        make_full()
    elif "--config5-full" in sys.argv:
        make_full(C.FULL5_CASES, C.FULL5_TRACE_STEPS)
    elif "--sample-only" in sys.argv:
        make_sample()
    elif "--modules-only" in sys.argv:
        make_modules()
    elif "--config5-only" in sys.argv:
        make_long()
    elif "--config2-only" in sys.argv:
        make_long(C.WIDE_CASES)
    elif "--new-r2-only" in sys.argv:       # the fixtures added in round 2 (keeps the others byte-identical)
        make_edm(only={"dim16_300x300_ctor8_call5"})
        main(only={"dim16_300x300_lrcfg"})
    elif "--edm-only" in sys.argv:
        make_edm()
    elif "--edm-untiled-only" in sys.argv:
        make_edm_untiled()
    else:
        main()
        make_edm()
        make_modules()
        make_sample()
        make_edm_untiled()
