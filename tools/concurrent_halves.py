"""Experiment: is one HR tile's step (25 U-Net tiles, one launch per layer) faster as two concurrent half batches (13 + 12 tiles
on two HIP streams, so that one half's partial last wave of workgroups overlaps the other half's kernels)?  Emulated with two
engines and two images: A = one stream, two images one after the other, 25 tiles per launch; B = two streams, one image each,
13 tiles per launch (each stream runs its 13 + 12 back to back).  Same total work; prints HR tiles/s."""
import copy
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_sampler  # noqa: E402
from srgd_amd.synth import synthetic_lr_condition  # noqa: E402


def run_group(sampler, conds, stream, reps, tiles_per_launch):
    with torch.cuda.stream(stream):
        for _ in range(reps):
            for i in range(conds.shape[0]):
                sampler.device_noise_seed = 71
                sampler.tiled_sample(batch_size=tiles_per_launch, condition_x=conds[i:i + 1], class_label=torch.tensor([0], device="cuda"),
                                     num_sample_steps=50, precision="bf16")
        stream.synchronize()


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    s0, _ = build_sampler(128, dev, False, 0)
    s0.noise_source = "device"
    s1 = copy.deepcopy(s0)
    s1.noise_source = "device"
    conds = torch.cat([synthetic_lr_condition(i, 256, 256) for i in range(2)]).to(dev)
    for n_streams, tpl in ((1, 25), (2, 13), (2, 25), (1, 25), (2, 13), (2, 25), (1, 25), (2, 13), (2, 25), (1, 13)):
        samplers = [s0, s1][:n_streams]
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        per = 2 // n_streams
        groups = [conds[i * per:(i + 1) * per].contiguous() for i in range(n_streams)]
        ths = [threading.Thread(target=run_group, args=(samplers[i], groups[i], streams[i], 1, tpl)) for i in range(n_streams)]
        [t.start() for t in ths]; [t.join() for t in ths]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        ths = [threading.Thread(target=run_group, args=(samplers[i], groups[i], streams[i], reps, tpl)) for i in range(n_streams)]
        [t.start() for t in ths]; [t.join() for t in ths]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{n_streams} stream(s), {tpl} tiles per launch: {reps * 2 / dt:.4f} HR tiles/s", flush=True)


if __name__ == "__main__":
    main()
