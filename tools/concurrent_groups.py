"""Experiment: do two lock-step groups on two HIP streams (two engines, two host threads) overlap the MFMA-bound and the
HBM-bound kernels of one another?  Prints HR tiles/s for 1 stream x N images vs 2 streams x N/2 images."""
import copy
import logging
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_sampler  # noqa: E402
from srgd_amd.synth import synthetic_lr_condition  # noqa: E402


def run_group(sampler, conds, stream, reps, out):
    with torch.cuda.stream(stream):
        for _ in range(reps):
            sampler.device_noise_seed = 71
            out.append(sampler.tiled_sample(batch_size=25 * conds.shape[0], condition_x=conds, class_label=torch.tensor([0], device="cuda"),
                                            num_sample_steps=50, precision="bf16"))
        stream.synchronize()


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    s0, _ = build_sampler(128, dev, False, 0)
    s0.noise_source = "device"
    s1 = copy.deepcopy(s0)
    s1.noise_source = "device"
    conds = torch.cat([synthetic_lr_condition(i, 256, 256) for i in range(6)]).to(dev)
    for n_streams, per in ((1, 6), (2, 3), (1, 6), (2, 3)):
        samplers = [s0, s1][:n_streams]
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        groups = [conds[i * per:(i + 1) * per].contiguous() for i in range(n_streams)]
        # warm-up
        ths = [threading.Thread(target=run_group, args=(samplers[i], groups[i], streams[i], 1, [])) for i in range(n_streams)]
        [t.start() for t in ths]; [t.join() for t in ths]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 2
        ths = [threading.Thread(target=run_group, args=(samplers[i], groups[i], streams[i], reps, [])) for i in range(n_streams)]
        [t.start() for t in ths]; [t.join() for t in ths]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{n_streams} stream(s) x {per} images: {reps * n_streams * per / dt:.4f} HR tiles/s", flush=True)


if __name__ == "__main__":
    main()
