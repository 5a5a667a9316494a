"""Batch driver with the reference ``inference.py`` surface (CLI flags, function names, file naming,
skip-if-exists, per-image reseed) on top of the MI355X engine.

Differences that are deliberate and documented: torchvision/logzero are not needed
(the x4 bicubic resize and the uint8 conversions run as integer kernels on the GPU, bit-identical to what
``T.Resize`` on a PIL image, ``ToTensor`` and ``ToPILImage`` do; ``pil_to_unit_tensor`` / ``unit_tensor_to_pil`` are
the host-side equivalents kept for tests); ``--no_amp`` is accepted and, as upstream (whose sampler ignores ``amp`` and
always computes fp32), changes nothing: the default run reproduces the reference's fp32 numerics.
Engine-only switches: ``--precision {fp32,f16x3,bf16,bf16_w8,fp8,fp8_mixed}`` opts into a throughput mode, ``--device_noise`` switches from
the reference-compatible host noise stream to on-device Philox.
"""
from __future__ import annotations

import glob
import logging
import os
import random
from argparse import ArgumentParser

import numpy as np
import torch
from PIL import Image

from .config import load_config
from .model import get_model

logger = logging.getLogger("srgd_amd")


def parse_args(argv=None):
    p = ArgumentParser()
    p.add_argument("-c", "--conf", required=True, help="Path to config file")
    p.add_argument("-m", "--ckpt_path", type=str, required=True)
    p.add_argument("--input_dir", type=str, required=True)
    p.add_argument("--output_dir", type=str, required=True)
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("--num_sample_steps", type=int, default=250)
    p.add_argument("--interpolation", type=str, default="bicubic")
    p.add_argument("--cond_scale", type=float, default=1.0)
    p.add_argument("--class_cond_scale", type=float, default=1.0)
    p.add_argument("--guidance_start_steps", type=int, default=0)
    p.add_argument("--class_guidance_start_steps", type=int, default=0)
    p.add_argument("--generation_start_steps", type=int, default=0)
    p.add_argument("--start_index", type=int, default=0)
    p.add_argument("--end_index", type=int, default=None)
    p.add_argument("--test_label", type=int, default=None)
    p.add_argument("--no_amp", dest="amp", action="store_false")
    p.add_argument("--no_dpmpp_solver", dest="use_dpmpp_solver", action="store_false")
    p.add_argument("--seed", type=int, default=71)
    p.add_argument("--backend", type=str, default="ddp")
    # engine-only switches (absent upstream)
    p.add_argument("--precision", choices=["fp32", "f16x3", "f16mx2", "bf16", "bf16_w8", "fp8", "fp8_mixed"], default="f16x3",
                   help="f16x3 (default since round 6): fp32 tensors, every convolution product as three f16 MFMAs on (hi, lo) operand "
                        "pairs - within 1e-5 of the reference's CPU path on its fixtures (bar: 1e-3), 3x the speed of fp32; "
                        "fp32: exact-fp32 MFMA (the reference's arithmetic up to summation order); bf16 / bf16_w8 / fp8 / fp8_mixed: "
                        "throughput modes of the MI355X engine (explicit opt-in, not within 1e-3)")
    p.add_argument("--device_noise", action="store_true",
                   help="draw DDPM noise on the GPU (Philox) instead of replaying torch's CPU stream")
    p.add_argument("--lockstep", type=int, default=1,
                   help="sample up to N consecutive same-sized images together (their tiles share U-Net launches, "
                        "--batch_size tiles per image and launch); every image comes out bit-identical to its solo run")
    return p.parse_args(argv)


def seed_everything(seed):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def pil_to_unit_tensor(image: Image.Image) -> torch.Tensor:
    """ToTensor: HWC uint8 -> CHW float32 / 255."""
    arr = np.asarray(image.convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).to(torch.float32).div(255.0)


def unit_tensor_to_pil(t: torch.Tensor) -> Image.Image:
    """ToPILImage on a float tensor: mul(255) then byte() (truncation, not rounding)."""
    arr = t.detach().cpu().mul(255).to(torch.uint8).permute(1, 2, 0).contiguous().numpy()
    return Image.fromarray(arr, "RGB")


def upsample_bicubic_on_device(image: Image.Image, scale: int, device) -> torch.Tensor:
    """``T.Resize((h*scale, w*scale), BICUBIC)`` + ``ToTensor`` of the reference (inference.py:66-73) as integer HIP
    kernels: the LR image goes to the GPU as uint8 HWC, the [1,3,H,W] condition comes out resident in HBM, bit-identical
    to ``pil_to_unit_tensor(image.resize(..., Image.BICUBIC))``."""
    import ctypes as C

    from . import _lib
    arr = np.asarray(image.convert("RGB"), dtype=np.uint8)
    h, w, _ = arr.shape
    src = torch.from_numpy(arr.copy()).to(device)
    dst = torch.empty(1, 3, h * scale, w * scale, device=device, dtype=torch.float32)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().srgd_image_resize_bicubic_u8(C.c_void_p(src.data_ptr()), h, w, h * scale, w * scale,
                                                          C.c_void_p(dst.data_ptr()),
                                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "srgd_image_resize_bicubic_u8")
    return dst


def unit_tensor_to_u8_on_device(t: torch.Tensor) -> torch.Tensor:
    """``mul(255).byte()`` of ``ToPILImage`` (inference.py:93) as a HIP kernel: [3,H,W] fp32 in [0,1] -> [H,W,3] uint8, on the GPU."""
    import ctypes as C

    from . import _lib
    t = t.contiguous()
    _, h, w = t.shape
    out = torch.empty(h, w, 3, device=t.device, dtype=torch.uint8)
    with torch.cuda.device(t.device):
        _lib.check(_lib.lib().srgd_image_unit_to_u8(C.c_void_p(t.data_ptr()), h, w, C.c_void_p(out.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "srgd_image_unit_to_u8")
    return out


def unit_tensor_to_pil_on_device(t: torch.Tensor) -> Image.Image:
    """``ToPILImage`` on the GPU: only the uint8 HWC image crosses PCIe (1/4 of the bytes)."""
    return Image.fromarray(unit_tensor_to_u8_on_device(t).cpu().numpy(), "RGB")


def sr_target_image(image, sr_model, scale=4, batch_size=8, test_label=2, cond_scale=1.0, guidance_start_steps=0,
                    class_cond_scale=1.0, class_guidance_start_steps=0, generation_start_steps=0,
                    num_sample_steps=250, enable_amp=False, interpolation="bicubic", seed=71):
    width, height = image.size
    # the reference maps 'lanczos' to bicubic too (inference.py:66-69)
    condition_x = upsample_bicubic_on_device(image, scale, sr_model.device)
    label = torch.LongTensor([test_label]).to(sr_model.device) if test_label is not None else None
    seed_everything(seed)
    sr_model.device_noise_seed = seed
    with torch.inference_mode():
        output = sr_model.tiled_sample(batch_size=batch_size, condition_x=condition_x, class_label=label,
                                       cond_scale=cond_scale, guidance_start_steps=guidance_start_steps,
                                       class_cond_scale=class_cond_scale,
                                       class_guidance_start_steps=class_guidance_start_steps,
                                       generation_start_steps=generation_start_steps,
                                       num_sample_steps=num_sample_steps, amp=enable_amp)
    sr_img = unit_tensor_to_pil_on_device(output[0])
    assert sr_img.size == (width * 4, height * 4)
    return sr_img


def sr_target_images(images, sr_model, scale=4, batch_size=8, test_label=2, cond_scale=1.0, guidance_start_steps=0,
                     class_cond_scale=1.0, class_guidance_start_steps=0, generation_start_steps=0,
                     num_sample_steps=250, enable_amp=False, interpolation="bicubic", seed=71):
    """``sr_target_image`` for several same-sized images in lock-step (engine extension): one ``tiled_sample`` call on a
    ``[B,3,H,W]`` condition.  Each image is sampled exactly as it would be alone after the reference's per-image
    ``seed_everything(seed)`` (inference.py:73) - bit-identical outputs - while their tiles fill the U-Net launches."""
    assert len({im.size for im in images}) == 1, "lock-step images must have the same size"
    width, height = images[0].size
    condition_x = torch.cat([upsample_bicubic_on_device(im, scale, sr_model.device) for im in images], 0)
    label = torch.LongTensor([test_label]).to(sr_model.device) if test_label is not None else None
    seed_everything(seed)
    sr_model.device_noise_seed = seed
    with torch.inference_mode():
        output = sr_model.tiled_sample(batch_size=batch_size * len(images), condition_x=condition_x, class_label=label,
                                       cond_scale=cond_scale, guidance_start_steps=guidance_start_steps,
                                       class_cond_scale=class_cond_scale,
                                       class_guidance_start_steps=class_guidance_start_steps,
                                       generation_start_steps=generation_start_steps,
                                       num_sample_steps=num_sample_steps, amp=enable_amp)
    outs = [unit_tensor_to_pil_on_device(o) for o in output]
    assert all(o.size == (width * 4, height * 4) for o in outs)
    return outs


def try_open_image(image_path):
    try:
        return Image.open(image_path).convert("RGB")
    except (IOError, SyntaxError):
        return None


def batch_sr_target_images(input_dir, output_dir, sr_model, scale=4, batch_size=8, test_label=2, cond_scale=1.0,
                           guidance_start_steps=0, class_cond_scale=1.0, class_guidance_start_steps=0,
                           generation_start_steps=0, num_sample_steps=250, start_index=0, end_index=None,
                           enable_amp=False, interpolation="bicubic", seed=71, lockstep=1):
    print(f"save images at: {output_dir}")
    os.makedirs(output_dir, exist_ok=True)
    kw = dict(scale=scale, batch_size=batch_size, test_label=test_label, cond_scale=cond_scale,
              guidance_start_steps=guidance_start_steps, class_cond_scale=class_cond_scale,
              class_guidance_start_steps=class_guidance_start_steps, generation_start_steps=generation_start_steps,
              num_sample_steps=num_sample_steps, enable_amp=enable_amp, interpolation=interpolation, seed=seed)
    from concurrent.futures import ThreadPoolExecutor
    pending, saves = [], []                              # (image, save_path) of the current lock-step group; PNG writers

    with ThreadPoolExecutor(max_workers=2) as pool:      # PNG encoding overlaps the next group's sampling
        def flush():
            if not pending:
                return
            if len(pending) == 1:
                outs = [sr_target_image(pending[0][0], sr_model, **kw)]
            else:
                outs = sr_target_images([im for im, _ in pending], sr_model, **kw)
            for (_, path), sr in zip(pending, outs):
                saves.append(pool.submit(sr.save, path))
            pending.clear()

        for filename in sorted(glob.glob(f"{input_dir}/*"))[start_index:end_index]:
            save_path = os.path.join(output_dir, os.path.basename(filename).replace(".png", "_out.png"))
            if os.path.exists(save_path):
                print("skip")
                continue
            image = try_open_image(filename)
            if image is None:
                print("Invalid image or unable to open image:", filename)
                continue
            if pending and (len(pending) >= max(1, lockstep) or pending[0][0].size != image.size):
                flush()
            pending.append((image, save_path))
            if len(pending) >= max(1, lockstep):
                flush()
        flush()
        for f in saves:
            f.result()                                   # surface write errors


def rank_file_range(n_files, start_index, end_index, rank, world):
    """The reference parallelises by hand: one process per GPU with its own ``--start_index/--end_index`` (inference.py:36-37,
    :120).  Under a launcher that sets RANK / WORLD_SIZE (``torchrun --nproc-per-node 8 inference.py ...``) the selected range
    ``[start_index:end_index]`` is cut into ``world`` contiguous slices instead - contiguous, so that ``--lockstep`` groups of
    neighbouring same-sized images survive - and this returns rank's ``(start, end)``.  No collective is involved."""
    lo, hi, _ = slice(start_index, end_index).indices(n_files)
    n = max(0, hi - lo)
    per, rem = divmod(n, world)
    a = lo + rank * per + min(rank, rem)
    return a, a + per + (1 if rank < rem else 0)


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format="[%(levelname)s %(asctime)s] %(message)s")
    args = parse_args(argv)
    conf = load_config(args.conf)
    conf.num_sample_steps = args.num_sample_steps
    conf.ckpt_path = args.ckpt_path
    ema_model = get_model(conf, logger)
    if not torch.cuda.is_available():
        raise SystemExit("srgd_amd needs an MI355X: no GPU visible and there is no CPU fallback")
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:                                        # one process per GPU, each with its own slice of the input files
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        n_files = len(glob.glob(f"{args.input_dir}/*"))
        args.start_index, args.end_index = rank_file_range(n_files, args.start_index, args.end_index, rank, world)
        print(f"rank {rank}/{world}: files [{args.start_index}:{args.end_index}] of {n_files}")
    sr_model = ema_model.module.eval().to(torch.device("cuda", torch.cuda.current_device()))
    sr_model.noise_source = "device" if args.device_noise else "host"
    sr_model.precision = args.precision
    print(f"engine precision: {args.precision} (noise: {sr_model.noise_source})")
    if args.amp and args.precision in ("fp32", "f16x3", "f16mx2"):
        # callers of earlier builds got bf16 from amp=True; upstream's sampler ignores amp and so does this one
        print("note: amp is accepted and ignored as upstream (the sampler computes at fp32 accuracy: f16x3 ~3.4x, fp32 ~10x slower "
              "than the bf16 engine); pass --precision bf16 for the throughput mode")
    import time
    t_pack = time.perf_counter()
    unet = getattr(sr_model, "model", None) or sr_model.net
    unet.engine(args.precision)                         # pack + upload the weights now, not inside the first image
    from .lanes import AUTO_LANE_PRECISIONS
    if (sr_model.step_lanes is None and args.precision in AUTO_LANE_PRECISIONS) or (sr_model.step_lanes or 1) > 1:
        # small steps run as two concurrent halves through a second engine instance (srgd_amd.lanes: a second copy of the packed
        # weights, scratch and graphs): built here, not inside the first image's sampling loop
        unet.engine(args.precision, lane=1)
    torch.cuda.synchronize()
    print(f"engine ready: {len(sr_model.state_dict())} tensors packed and uploaded in {time.perf_counter() - t_pack:.2f} s")
    print(args)
    batch_sr_target_images(args.input_dir, args.output_dir, sr_model, scale=4, batch_size=args.batch_size,
                           test_label=args.test_label, cond_scale=args.cond_scale,
                           guidance_start_steps=args.guidance_start_steps, class_cond_scale=args.class_cond_scale,
                           class_guidance_start_steps=args.class_guidance_start_steps,
                           generation_start_steps=args.generation_start_steps,
                           num_sample_steps=args.num_sample_steps, start_index=args.start_index,
                           end_index=args.end_index, enable_amp=args.amp, interpolation=args.interpolation,
                           seed=args.seed, lockstep=args.lockstep)


if __name__ == "__main__":
    main()
