"""Headline benchmark: HR tiles/sec of the tiled CFG-DDPM sampling path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU over RCCL.  Under a launcher (torch.distributed.run sets WORLD_SIZE/RANK/LOCAL_RANK) this
process is one rank; started bare with --gpus N > 1 it spawns the N ranks itself (child processes of
``python -m torch.distributed.run``, started before this process touches the GPU) and relays rank 0's JSON line.
A WORLD_SIZE that disagrees with --gpus is an error, never a silent 1-GPU run.

One "step" = one HR tile = BASELINE.json config[1]: a 256x256 LR image -> x4 -> 1024x1024 through
``tiled_sample`` (canvas 1280^2, 25 tiles on even / 16 on odd DDPM steps, 50 steps,
class_cond_scale=1.0 -> 1,025 U-Net tile-forwards = 813.6 TFLOP), dim-128 U-Net, seeded synthetic
weights with the reference state_dict schema (the published checkpoint is an LFS pointer), synthetic
LR input already upsampled and resident in HBM when the clock starts.
The K HR tiles of a rank advance in lock-step groups of ``--images`` (default 5): their 256x256 U-Net tiles share
launches (125 / 80 tiles per launch), each image still sampled exactly as it would be alone.
Ranks shard independent images (weak scaling, no data-path collective); weights are broadcast from
rank 0 and the HR outputs gathered to rank 0 over RCCL inside the timed region.
``--workload canvas --lr_size 2048`` is the secondary mode (BASELINE configs[3]): one 8192^2 image whose tiles are
sharded over all ranks with a per-step tile all-gather (strong scaling).

Prints ONE JSON line (rank 0).  `roofline` is the dominant kernel, conv3x3_bf16_kernel (89% of the FLOPs):
algorithmic FLOPs / HIP-event time on the launch stream, collected in an extra profiled pass right
after the timed region; `cpu_baseline` is the CPU oracle (oracle/srgd_oracle.py, a port of the
reference's PyTorch path) timed on this box's host cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TFLOP_PER_HR_TILE = 813.6          # BASELINE.md section 3 (1,025 tile-forwards x 793.8 GFLOP)
TILE_FORWARDS_PER_HR_TILE = 1025
PMC_TRAFFIC_FILE = "r6_pmc_traffic.json"     # newest committed PMC summary (tools/pmc_traffic.py; bf16 and f16x3 passes of round 6, fp8 kernels from round 5)
# MI355X_MICROARCH.md: dense MFMA peaks of the dominant kernel's instruction (fp8 = block-scaled MX e4m3, 2x the bf16 rate)
# conv3x3_split (f16x3 mode): the f16 MFMA peak; the kernel issues THREE MFMAs per algorithmic product, so its algorithmic ceiling is
# a third of it (roofline.mfma_per_product, roofline.frac_of_issue_peak)
PEAK_TFLOPS = {"conv3x3_bf16": 2500.0, "conv3x3_mxfp8": 5000.0, "conv3x3_split": 2500.0, "conv_igemm:bf16": 2500.0, "conv_igemm:fp32": 157.3}
KERNEL_OF = {"conv3x3_bf16": "conv3x3_bf16_kernel", "conv3x3_mxfp8": "conv3x3_mxfp8_kernel", "conv_igemm": "conv_igemm_kernel",
             "conv3x3_split": "conv3x3_split_kernel"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5, help="HR tiles per GPU in the timed region")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--images", type=int, default=5,
                    help="HR tiles sampled in lock-step per tiled_sample call (their 256^2 U-Net tiles share launches)")
    ap.add_argument("--workload", choices=["tiles", "canvas"], default="tiles",
                    help="tiles: BASELINE configs[1] units, images sharded over ranks (weak scaling, the headline); "
                         "canvas: ONE --lr_size^2 image per step whose tiles are sharded over all ranks with a per-step "
                         "tile all-gather (configs[3] with --lr_size 2048; strong scaling, secondary)")
    ap.add_argument("--precision", choices=["bf16", "fp32", "f16x3", "f16mx2", "bf16_w8", "fp8", "fp8_mixed"], default="bf16",
                    help="f16x3: fp32 tensors, every convolution product as three f16 MFMAs on (hi, lo) operand pairs - meets the 1e-3 "
                         "parity bar like fp32; fp8: 3x3 convolutions on the block-scaled MX-fp8 matrix cores, e4m3 weights AND activations with a "
                         "scale per 32 channels (BASELINE configs[4] compute path; use with --ddpm_steps 100 "
                         "--class_cond_scale 2.0); fp8_mixed: fp8 below the top resolution, bf16 3x3 convolutions at 256x256 (53 dB vs bf16 "
                         "instead of 34 dB); bf16_w8: bf16 kernels with fp8-e4m3-rounded conv weights (numerics only)")
    ap.add_argument("--class_cond_scale", type=float, default=1.0,
                    help="!= 1: class guidance, two U-Net passes per step batched into one launch (configs[4] uses 2.0 "
                         "with --ddpm_steps 100); the headline metric is quoted at 1.0")
    ap.add_argument("--ddpm_steps", type=int, default=50)
    ap.add_argument("--lr_size", type=int, default=256)
    ap.add_argument("--sub_batch", type=int, default=0, help="tiles per U-Net launch (0: all tiles of a lock-step group)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_profile", action="store_true")
    ap.add_argument("--cpu_tile_forwards", type=int, default=128,
                    help="U-Net tile-forwards of the timed CPU-oracle sample over all worker processes (SURVEY 8(d): >= 41 = two "
                         "DDPM steps of one HR tile)")
    ap.add_argument("--cpu_budget_s", type=float, default=150.0,
                    help="stop the CPU sample early once this many seconds are spent (the count actually run is reported)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (this process has not touched the
    GPU: nothing above calls torch.cuda / HIP) and pass their output through.  Never re-execs the current process."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC; without it RCCL's intra-node transport
    # setup (and any cross-process device-memory sharing) fails with `hipIpcGetMemHandle: invalid argument`.  The image and
    # the GPU boxes export it already; it is only defaulted here so that a scrubbed environment still works
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def build_sampler(dim, device, use_dist, rank):
    from srgd_amd.config import load_config
    from srgd_amd.model import get_model
    from srgd_amd.synth import synth_state_dict
    conf = load_config(os.path.join(ROOT, "conf", "conditional_continuous_linear_df8kost_dim128.yaml"))
    conf.unet_dim = dim
    conf.num_sample_steps = 50
    sampler = get_model(conf, logging.getLogger("bench")).module
    schema = {k: tuple(v.shape) for k, v in sampler.state_dict().items()}
    if use_dist:
        # rank 0 owns the checkpoint; everyone else receives it over RCCL/xGMI as one flat buffer
        from srgd_amd.parallel import broadcast_state_dict
        sd = broadcast_state_dict(schema, synth_state_dict(schema, seed=0) if rank == 0 else None, src=0, device=device)
    else:
        sd = synth_state_dict(schema, seed=0)
    sampler.load_state_dict(sd, strict=True)
    return sampler.eval().to(device), sd


def cpu_worker(argv):
    """`bench.py --cpu_worker DIM THREADS MINIBATCHES [cpu,cpu,...]`: one of the k concurrent workers of the CPU baseline (the
    reference's own way to use a many-core host: N processes over disjoint file slices, inference.py:36-37).  Prints READY,
    waits for a line on stdin (so that all workers start together), runs MINIBATCHES U-Net forwards of 8 tiles, prints JSON."""
    dim, threads, n_mb = int(argv[0]), int(argv[1]), int(argv[2])
    if len(argv) > 3 and argv[3]:
        try:
            os.sched_setaffinity(0, {int(c) for c in argv[3].split(",")})
        except OSError:
            pass
    torch.set_num_threads(threads)
    from oracle import srgd_oracle as O
    from srgd_amd.synth import synth_state_dict
    with open(os.path.join(ROOT, "tests", "golden", f"schema_dim{dim}.json")) as f:
        schema = {k: tuple(v) for k, v in json.load(f).items()}
    usd = O.strip_model_prefix(synth_state_dict(schema, seed=0))
    cfg = O.UnetCfg(dim=dim)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(8, 3, 256, 256, generator=g)
    c = torch.rand(8, 3, 256, 256, generator=g) * 2 - 1
    ls, lab = torch.full((8,), 0.5), torch.tensor([0])
    with torch.inference_mode():
        O.unet_forward(usd, cfg, x[:2], ls[:2], lab, c[:2])          # warm-up (allocator, oneDNN primitive cache): discarded
        print("READY", flush=True)
        sys.stdin.readline()
        t0 = time.monotonic()                              # CLOCK_MONOTONIC: one clock for all worker processes of the host
        for _ in range(n_mb):
            O.unet_forward(usd, cfg, x, ls, lab, c)
        dt = time.monotonic() - t0
    print(json.dumps({"tile_forwards": 8 * n_mb, "seconds": dt, "t_start": t0, "t_end": t0 + dt}), flush=True)


def cpu_baseline_parallel(dim, threads_per_worker, minibatches, timeout_s=240.0):
    """k = physical cores / threads_per_worker concurrent worker processes (each pinned to its own cores) running the oracle on
    disjoint tile batches; the aggregate rate is all workers' tile-forwards over the window from the first worker's start to
    the last worker's end (one host-wide monotonic clock), so stragglers count against the baseline exactly once."""
    import select
    import subprocess
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    physical = cpus[:max(1, len(cpus) // 2)] if len(cpus) >= 2 * threads_per_worker else cpus     # first half = one thread per core
    k = max(1, len(physical) // threads_per_worker)
    try:                                                   # ~6 GB per worker (550 MB of weights + fp32 activations of 8 tiles)
        avail_gb = next(int(ln.split()[1]) for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")) / 1e6
        k = max(1, min(k, int(avail_gb * 0.5 // 6)))
    except (OSError, StopIteration, ValueError):
        pass
    k = min(k, 16)
    threads = min(threads_per_worker, max(1, len(physical) // k))
    procs = []
    for i in range(k):
        mine = ",".join(str(c) for c in physical[i * threads:(i + 1) * threads])
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu_worker", str(dim), str(threads),
                                       str(minibatches), mine], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, cwd=ROOT))
    res = []
    try:
        deadline = time.monotonic() + timeout_s

        def read_line(pr, what):                           # a line from a worker, or an error once the deadline has passed
            left = deadline - time.monotonic()
            if left <= 0 or not select.select([pr.stdout], [], [], left)[0]:
                raise RuntimeError(f"cpu worker timed out waiting for {what} (budget {timeout_s:.0f} s)")
            return pr.stdout.readline()

        for pr in procs:                                   # all workers warmed up and waiting
            line = read_line(pr, "READY")
            if "READY" not in line:
                raise RuntimeError(f"cpu worker did not come up: {line!r}")
        for pr in procs:
            pr.stdin.write("go\n")
            pr.stdin.flush()
        deadline = time.monotonic() + timeout_s
        for pr in procs:
            res.append(json.loads(read_line(pr, "its result")))
            pr.wait(timeout=30)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    total = sum(r["tile_forwards"] for r in res)
    window = max(r["t_end"] for r in res) - min(r["t_start"] for r in res)
    return {"workers": k, "threads_per_worker": threads, "tile_forwards": total, "seconds": window,
            "tile_forwards_per_s": total / window,
            "sum_of_worker_rates": sum(r["tile_forwards"] / r["seconds"] for r in res)}


def cpu_baseline(sd, dim, n_tile_forwards, budget_s):
    """The CPU oracle (port of the reference PyTorch path) on a bounded sample of the same workload: U-Net tile-forwards at
    the reference's default minibatch of 8 tiles (inference.py:27), extrapolated to HR tiles/s (1,025 tile-forwards each).
    Leg 1 (one process): one discarded warm-up call per thread setting and a short sweep over thread counts - oneDNN does not
    scale these convolutions past ~16 threads inside one process.  Leg 2 (the baseline's best showing, and the reported
    value): k = physical cores / 16 concurrent 16-thread worker PROCESSES on disjoint tile batches - the reference's own answer
    to a many-core host (one process per --start_index/--end_index slice, inference.py:36-37,120)."""
    from oracle import srgd_oracle as O
    usd = O.strip_model_prefix(sd)
    cfg = O.UnetCfg(dim=dim)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(8, 3, 256, 256, generator=g)
    c = torch.rand(8, 3, 256, 256, generator=g) * 2 - 1
    ls = torch.full((8,), 0.5)
    lab = torch.tensor([0])

    def fwd(b):
        O.unet_forward(usd, cfg, x[:b], ls[:b], lab, c[:b])

    logical = os.cpu_count() or 1
    default = torch.get_num_threads()
    cands = sorted({t for t in (8, 16, 32, 64) if 1 <= t <= logical} or {default})
    t_start = time.perf_counter()
    sweep = {}
    with torch.inference_mode():
        for t in cands:
            torch.set_num_threads(t)
            fwd(1)                                       # warm-up (allocator, oneDNN primitive cache): discarded
            t0 = time.perf_counter()
            fwd(2)
            sweep[t] = 2 / (time.perf_counter() - t0)
            if time.perf_counter() - t_start > 0.3 * budget_s:
                break
    torch.set_num_threads(default)
    best = max(sweep, key=sweep.get)
    del usd, x, c
    minibatches = max(1, -(-n_tile_forwards // 8 // 8))     # ~n_tile_forwards over all workers at k = 8, at least one minibatch each
    par = cpu_baseline_parallel(dim, min(16, best) if best >= 16 else best, minibatches)
    tf_per_s = par["tile_forwards_per_s"]
    return {"value": tf_per_s / TILE_FORWARDS_PER_HR_TILE, "unit": "HR tiles/s", "cores": par["workers"] * par["threads_per_worker"],
            "kind": "port", "workers": par["workers"], "threads_per_worker": par["threads_per_worker"],
            "sample": f"{par['tile_forwards']} U-Net tile-forwards (256x256, dim {dim}, minibatch 8; one warm-up call per worker discarded) "
                      f"by {par['workers']} concurrent worker processes x {par['threads_per_worker']} threads, each pinned to its own "
                      f"cores, in {par['seconds']:.1f} s = {tf_per_s:.3f} tile-forwards/s aggregate on {logical} logical CPUs; "
                      f"1 HR tile = {TILE_FORWARDS_PER_HR_TILE} tile-forwards",
            "single_process_thread_sweep_tile_forwards_per_s": {str(k): round(v, 4) for k, v in sweep.items()}}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu_worker":
        return cpu_worker(sys.argv[2:])
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    # SRGD_FORCE_DIST=1: take the distributed code path even at world size 1 - init_process_group("nccl"), the device-side
    # weight broadcast, the uint8 output gather, max-over-ranks, and (canvas workload) the per-step all_gather_into_tensor -
    # so that RCCL and every collective of the N > 1 run execute on a 1-GPU box (tests/test_bench_multirank_gpu.py)
    force_dist = os.environ.get("SRGD_FORCE_DIST", "0") == "1"
    if force_dist and "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s); refusing to mislabel the run")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # SRGD_DIST_BACKEND=gloo is a test hook: several ranks may then share one GPU (tests/test_bench_multirank_gpu.py runs the
    # N > 1 code path on a 1-GPU box that way); the driver's runs use nccl (= RCCL), one rank per GPU
    backend = os.environ.get("SRGD_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or force_dist:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    n_ranks = dist.get_world_size() if dist else 1           # what the JSON line reports: the communicator's size
    assert n_ranks == args.gpus

    from srgd_amd.synth import synthetic_lr_condition
    sampler, sd = build_sampler(args.dim, device, dist is not None, rank)
    sampler.noise_source = "device"
    sampler.precision = args.precision
    label = torch.tensor([0], device=device)
    total = args.warmup + args.steps
    # inputs resident in HBM before the clock starts (already x4-upsampled condition images)
    canvas_mode = args.workload == "canvas"
    if canvas_mode:
        args.images = 1
        if dist:
            from srgd_amd.parallel import shard_canvas
            shard_canvas(sampler, always_exchange=force_dist)
    # canvas mode: every rank works on the SAME image; tiles mode: item j of 0..world*total-1 belongs to rank j % world
    # (srgd_amd.parallel.shard_indices), so rank r's k-th image is item r + k*world
    conds = [synthetic_lr_condition(i if canvas_mode else rank + i * world, args.lr_size, args.lr_size).to(device)
             for i in range(total)]
    n_even = ((4 * args.lr_size + 255) // 256 + 1) ** 2 if args.lr_size * 4 > 256 else 1
    from srgd_amd.lanes import lanes_wanted

    def sample_local(idx):
        """The local images ``idx`` in ONE lock-step tiled_sample call (each image sampled exactly as it would be alone)."""
        sampler.device_noise_seed = 71
        return sampler.tiled_sample(batch_size=args.sub_batch or min(125, n_even * len(idx)),
                                    condition_x=torch.cat([conds[k] for k in idx], 0), class_label=label,
                                    class_cond_scale=args.class_cond_scale, num_sample_steps=args.ddpm_steps,
                                    precision=args.precision)

    def run(lo, hi, gather=False):
        """Local HR tiles [lo, hi) in lock-step groups of --images; with ``gather`` the HR outputs travel to rank 0 as uint8
        HWC images (3.1 MB each) and rank 0 gets them back in item order (srgd_amd.parallel.sample_images_sharded)."""
        if hi <= lo:
            return [], None
        if canvas_mode or not dist:
            return [sample_local(list(range(a, min(a + args.images, hi)))) for a in range(lo, hi, args.images)], None
        from srgd_amd.parallel import sample_images_sharded
        local, ordered = sample_images_sharded(lambda items: sample_local([lo + (j - rank) // world for j in items]),
                                               (hi - lo) * world, args.images, rank, world, dst=0, gather=gather)
        return [local], ordered

    run(0, args.warmup)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    shard = sampler.canvas_group if canvas_mode else None
    if shard:                                   # time every pack -> all-gather -> unpack of the timed region (HIP events)
        shard.exchange_ms()
        shard.timing = True
    t0 = time.perf_counter()
    outs, gathered = run(args.warmup, total, gather=True)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    exchange_ms = 0.0
    if shard:
        shard.timing = False
        exchange_ms = shard.exchange_ms()
        if dist:
            from srgd_amd.parallel import max_over_ranks
            exchange_ms = max_over_ranks(exchange_ms, device)
    if dist:
        from srgd_amd.parallel import max_over_ranks
        dt = max_over_ranks(dt, device)
    if dist and not canvas_mode and rank == 0:
        assert gathered is not None and len(gathered) == args.steps * world and all(g is not None for g in gathered)
    assert all(torch.isfinite(o).all() for o in outs)

    if rank == 0 and canvas_mode:
        from srgd_amd.model import get_coord_and_pad, get_coords
        h = 4 * args.lr_size
        _, pad = get_coord_and_pad(h, h)
        hp = h + pad[2] + pad[3]
        ne = len(get_coords(hp, hp, 256, 256))
        no = ne if hp <= 256 else len(get_coords(hp - 256, hp - 256, 256, 256, diff=128))
        tf = sum(ne if i % 2 == 0 else no for i in range(args.ddpm_steps))
        print(json.dumps({
            "metric": f"U-Net tile-forwards/sec on one {h}x{h} image ({args.ddpm_steps} steps, CFG=1.0)",
            "value": args.steps * tf / dt, "unit": "tile-forwards/s", "n_gpus": n_ranks, "rccl_ranks": n_ranks,
            "dist_backend": backend if dist else None, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic (seeded LR image, seeded weights)",
            "config": {"workload": f"one {args.lr_size}x{args.lr_size} LR image x4, canvas {hp}x{hp}, {ne}/{no} tiles per "
                                   f"even/odd step, tiles sharded over {world} rank(s), per-step tile all-gather",
                       "tile_forwards_per_step": tf, "parallelism": f"canvas-sharded x{world}"},
            "forced_dist": force_dist, "tile_allgathers": sampler.canvas_group.exchanges if sampler.canvas_group else 0,
            # per DDPM step of the timed region, max over ranks: pack + all-gather + unpack of the canvas tiles (HIP events on
            # the engine's stream); exchange_share = that / the wall time, i.e. what strong scaling loses to the collective
            # since round 5 the first half-slice's gather runs on a side stream under the second half's compute: these are the
            # EXPOSED part only (second half's gather + both unpacks on the compute stream) - not comparable with round 4's
            "exchange_ms": exchange_ms / max(1, args.steps * args.ddpm_steps),
            "exchange_exposed_ms": exchange_ms / max(1, args.steps * args.ddpm_steps),
            "exchange_share": exchange_ms / (1e3 * dt),
            "exchange_mb_per_step": (3 * 256 * 256 * 4 * ((ne + no) / 2.0) / 1e6) if sampler.canvas_group else 0.0,
            "hr_tile_equivalents_per_s": args.steps * tf / dt / TILE_FORWARDS_PER_HR_TILE,
            "tflops_effective": args.steps * tf / dt * 0.7938}), flush=True)
    elif rank == 0:
        tiles = args.steps * n_ranks
        value = tiles / dt
        n_step = n_even * min(args.images, args.steps)                 # samples of an even step of one lock-step group
        launch_limit = args.sub_batch or min(125, n_step)              # = sample_local's batch_size
        step_lanes = lanes_wanted(n_step, 1 if args.class_cond_scale == 1.0 else 2, launch_limit, sampler.step_lanes, args.precision)
        line = {
            "metric": "HR tiles/sec (256->1024 x4, 50 steps, CFG=1.0)", "value": value, "unit": "HR tiles/s",
            "n_gpus": n_ranks, "rccl_ranks": n_ranks, "dist_backend": backend if dist else None, "forced_dist": force_dist,
            "gathered_hr_tiles": len(gathered) if gathered else 0,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
            "data": "synthetic (seeded LR images, seeded weights with the reference state_dict schema)",
            "config": {"workload": f"{'BASELINE configs[1]' if (args.ddpm_steps == 50 and args.class_cond_scale == 1.0 and args.lr_size == 256 and args.precision == 'bf16') else 'BASELINE configs[4] (MX-fp8 3x3 convolutions: e4m3 weights + activations, E8M0 scale per 32 channels)' if args.precision == 'fp8' else 'BASELINE configs[4], mixed (MX-fp8 3x3 convolutions below the top resolution, bf16 at 256x256)' if args.precision == 'fp8_mixed' else 'BASELINE configs[4] numerics (fp8-rounded weights on bf16 kernels)' if args.precision == 'bf16_w8' else 'split-operand parity mode (fp32 tensors, three f16 MFMAs per product)' if args.precision == 'f16x3' else 'split-operand prototype (fp32 tensors; 3x3 convolutions: f16 leading term + both cross terms on MX-fp8 operands)' if args.precision == 'f16mx2' else 'variant'}: one {args.lr_size}x{args.lr_size} LR tile x4 SR per step, "
                                   f"{args.ddpm_steps} DDPM steps, class_cond_scale={args.class_cond_scale}, dim-{args.dim} U-Net, "
                                   f"{args.precision}, device Philox noise; {min(args.images, args.steps)} steps "
                                   f"(HR tiles) advance in lock-step so their U-Net tiles share launches",
                       "images_in_lockstep": min(args.images, args.steps),
                       # what the engine really launches: the step's samples split over the lanes (srgd_amd.lanes), each lane's part
                       # cut by the engine's balanced-launch rule (engine.hip: cdiv(n, cdiv(n, limit))) under the SAME limit
                       # sample_local passes as batch_size
                       "step_lanes": step_lanes,
                       "tiles_per_unet_launch": (lambda n, lim: -(-n // -(-n // lim)))(-(-n_step // step_lanes), launch_limit),
                       "tile_forwards_per_step": TILE_FORWARDS_PER_HR_TILE,
                       "parallelism": f"image-sharded x{world}"},
            "tflops_effective": value * TFLOP_PER_HR_TILE,
        }
        headline = args.lr_size == 256 and args.ddpm_steps == 50 and args.dim == 128 and args.class_cond_scale == 1.0
        if not headline:
            passes = 1 if args.class_cond_scale == 1.0 else 2
            line["metric"] = (f"HR tiles/sec ({args.lr_size}->{4 * args.lr_size} x4, {args.ddpm_steps} steps, "
                              f"CFG={args.class_cond_scale})")
            line["config"]["tile_forwards_per_step"] = None
            line["config"]["unet_passes_per_ddpm_step"] = passes
            line.pop("tflops_effective", None)
        if not args.no_profile:
            eng = sampler.model.engine(args.precision)
            # the per-kernel events of the profiled pass time one kernel at a time: one lane (with two concurrent lanes -
            # srgd_amd.lanes, small steps only - a kernel's bracket would include the other lane's work)
            lanes_setting, sampler.step_lanes = sampler.step_lanes, 1
            eng.profile_begin()
            sample_local(list(range(args.warmup, args.warmup + min(args.images, args.steps))))
            prof = eng.profile_end()
            sampler.step_lanes = lanes_setting
            # the dominant kernel: conv3x3_bf16_kernel in bf16 mode, conv3x3_mxfp8_kernel in fp8 mode, the generic implicit
            # GEMM in fp32 mode - whichever convolution family took the most time in the profiled pass
            fam = max(("conv3x3_bf16", "conv3x3_mxfp8", "conv3x3_split", "conv_igemm"), key=lambda k: prof["ms"].get(k, 0.0))
            conv_ms, n_launch, fl = prof["ms"][fam], prof["launches"][fam], prof["flops"][fam]
            achieved = fl / (conv_ms * 1e-3) / 1e12
            peak = PEAK_TFLOPS.get(fam) or PEAK_TFLOPS["conv_igemm:" + ("fp32" if args.precision == "fp32" else "bf16")]
            kname = KERNEL_OF[fam]
            # HBM bytes per launch come from separate rocprofv3 --pmc passes of this same command (FETCH_SIZE and WRITE_SIZE
            # cannot share a pass, and rocprofv3 cannot run inside the benchmark): the committed summary is quoted and labelled
            traffic, traffic_src = None, None
            pmc = os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)
            if os.path.exists(pmc) and args.precision in ("bf16", "fp8", "fp8_mixed", "f16x3"):
                t = json.load(open(pmc)).get(kname)
                if t:   # gfx950: FETCH_SIZE counts half of a wide coalesced read (MI355X_MICROARCH.md, HBM) -> x2
                    traffic = (2.0 * t["FETCH_SIZE"]["avg_kb"] + t["WRITE_SIZE"]["avg_kb"]) * 1024.0
                    traffic_src = f"profiles/{PMC_TRAFFIC_FILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
            line["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                                "frac": achieved / peak, "traffic": traffic, "traffic_unit": "bytes/launch",
                                "traffic_source": traffic_src, "traffic_measured_in_run": False, "kernel": kname, "launches": n_launch,
                                "avg_launch_ms": conv_ms / max(n_launch, 1),
                                "algorithmic_gflop_per_launch": fl / max(n_launch, 1) / 1e9,
                                "family_time_share": conv_ms / sum(prof["ms"].values())}
            if fam == "conv3x3_split" and args.precision == "f16mx2":
                # prototype mode: one f16 MFMA + two products on the scaled MX-fp8 MFMA at twice the f16 rate = two f16 MFMAs' worth
                line["roofline"]["kernel"] = "conv3x3_mx2_kernel"
                line["roofline"]["mfma_per_product"] = "1 f16 + 2 MX-fp8 (= 2 f16 MFMAs' worth of matrix time)"
                line["roofline"]["frac_of_issue_peak"] = 2.0 * achieved / peak
                line["roofline"]["traffic"] = line["roofline"]["traffic_source"] = None
            elif fam == "conv3x3_split":
                line["roofline"]["mfma_per_product"] = 3
                line["roofline"]["frac_of_issue_peak"] = 3.0 * achieved / peak
            conv_fams = ("conv_igemm", "conv3x3_bf16", "conv1x1_bf16", "conv3x3_mxfp8", "conv1x1_mxfp8", "conv3x3_split", "conv_igemm_split", "conv1x1_split")
            all_conv_ms = sum(prof["ms"].get(k, 0.0) for k in conv_fams)
            line["conv_all_tflops"] = sum(prof["flops"].get(k, 0.0) for k in conv_fams) / (all_conv_ms * 1e-3) / 1e12
            tot = sum(prof["ms"].values())
            line["kernel_time_share"] = {k: round(v / tot, 4) for k, v in prof["ms"].items() if v > 0}
            line["profiled_pass_ms"] = tot
            # SURVEY 8(d) secondary check: the HBM-bound kernel families, algorithmic bytes (every operand read once, every
            # result written once; summed by the engine per launch) / HIP-event time, against the 8 TB/s HBM3E peak
            hbm = {}
            for k in ("groupnorm_silu", "rmsnorm", "linear_attention", "conv1x1_bf16", "conv1x1_mxfp8", "conv1x1_split", "final_conv_ddpm_step", "quantize_mxfp8"):
                if prof["ms"].get(k, 0.0) > 0 and prof["bytes"].get(k, 0.0) > 0:
                    gbps = prof["bytes"][k] / (prof["ms"][k] * 1e-3) / 1e9
                    hbm[k] = {"achieved": round(gbps, 1), "unit": "GB/s", "peak": 8000.0, "frac": round(gbps / 8000.0, 4),
                              "launches": prof["launches"][k], "avg_launch_us": round(1e3 * prof["ms"][k] / prof["launches"][k], 2),
                              "algorithmic_mb_per_launch": round(prof["bytes"][k] / prof["launches"][k] / 1e6, 2)}
            line["hbm_kernels"] = hbm
        if not args.no_cpu_baseline and world == 1:          # reported once, at N = 1 (the other ranks would idle behind it)
            line["cpu_baseline"] = cpu_baseline(sd, args.dim, args.cpu_tile_forwards, args.cpu_budget_s)
        print(json.dumps(line), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
