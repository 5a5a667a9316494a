"""bench.py's N > 1 path (weight broadcast, per-rank images, output gather, max-over-ranks timing, one JSON line from rank 0)
run for real with two ranks.  The box has one GPU, so the ranks share cuda:0 and talk over gloo (SRGD_DIST_BACKEND test hook);
the driver's multi-GPU runs use the same code with nccl (= RCCL), one rank per GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_emits_one_valid_json_line():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SRGD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--images", "2", "--ddpm_steps", "4", "--dim", "16", "--no_cpu_baseline", "--no_profile"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "HR tiles/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"] + 1e-9


def _run_bare(extra, gpus=2):
    """`python bench.py --gpus N ...` with NO launcher: bench.py must start its N ranks itself."""
    env = dict(os.environ, SRGD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--ddpm_steps", "4",
           "--dim", "16", "--no_cpu_baseline", "--no_profile", *extra]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bare_gpus2_self_launches_two_ranks():
    d = _run_bare(["--images", "2"])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["parallelism"] == "image-sharded x2"


def test_bare_gpus2_canvas_workload_self_launches_two_ranks():
    d = _run_bare(["--workload", "canvas", "--lr_size", "128"])        # 512^2 image, canvas 768^2, 9/4 tiles over 2 ranks
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "strong" and d["value"] > 0


# ---- four real torch.distributed ranks (gloo, one shared GPU; the box admits six GPU processes): unequal slices through TorchComm
def test_bare_gpus4_image_sharding_gathers_every_rank():
    d = _run_bare(["--images", "2"], gpus=4)
    assert d["n_gpus"] == 4 and d["rccl_ranks"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    assert d["gathered_hr_tiles"] == 2 * 4 and d["config"]["parallelism"] == "image-sharded x4"


def test_bare_gpus4_canvas_workload_unequal_slices_and_exchange_timing():
    # 256^2 LR -> 1024^2: canvas 1280^2, 25 / 16 tiles over 4 ranks = slices 7, 7, 7, 4 (even steps) and 4 x 4 (odd steps):
    # all_gather_into_tensor with unequal fill through TorchComm itself, not the thread stand-in
    d = _run_bare(["--workload", "canvas", "--lr_size", "256"], gpus=4)
    assert d["n_gpus"] == 4 and d["rccl_ranks"] == 4 and d["scaling"] == "strong" and d["value"] > 0
    assert d["tile_allgathers"] == 3 * 4 * 2                               # two (half slices) per step of the 3 runs (warm-up + 2 timed)
    assert d["exchange_ms"] > 0 and 0 < d["exchange_share"] < 1            # timed with HIP events over the timed region
    assert abs(d["exchange_mb_per_step"] - 3 * 256 * 256 * 4 * 20.5 / 1e6) < 1e-6


# ---- RCCL itself, on the real GPU: SRGD_FORCE_DIST=1 takes the N > 1 code path at world size 1 with the nccl backend -
# init_process_group("nccl", device_id=...), the device-side weight broadcast, the uint8 gather, the max-over-ranks all-reduce,
# the barriers and (canvas workload) all_gather_into_tensor - the calls the driver's 8-GPU run makes first
def _run_forced(extra, timeout=600):
    env = dict(os.environ, SRGD_FORCE_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SRGD_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no_cpu_baseline", "--no_profile", *extra]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_forced_dist_world1_runs_the_image_sharded_path_on_rccl():
    d = _run_forced(["--steps", "2", "--warmup", "1", "--images", "2", "--ddpm_steps", "4", "--dim", "16"])
    assert d["dist_backend"] == "nccl" and d["rccl_ranks"] == 1 and d["n_gpus"] == 1 and d["forced_dist"] is True
    assert d["gathered_hr_tiles"] == 2 and d["value"] > 0 and d["scaling"] == "weak"


def test_forced_dist_world1_broadcasts_the_dim128_checkpoint_on_rccl():
    # the 550 MB flat fp32 weight buffer through dist.broadcast on the device, then load_state_dict + per-rank packing
    d = _run_forced(["--steps", "1", "--warmup", "0", "--images", "1", "--ddpm_steps", "2", "--dim", "128"])
    assert d["dist_backend"] == "nccl" and d["rccl_ranks"] == 1 and d["gathered_hr_tiles"] == 1 and d["value"] > 0


def test_forced_dist_world1_runs_the_canvas_all_gather_on_rccl():
    d = _run_forced(["--workload", "canvas", "--lr_size", "128", "--steps", "2", "--warmup", "1", "--ddpm_steps", "4",
                     "--dim", "16"])
    assert d["dist_backend"] == "nccl" and d["rccl_ranks"] == 1 and d["forced_dist"] is True and d["scaling"] == "strong"
    assert d["tile_allgathers"] == 3 * 4 * 2 and d["value"] > 0    # all_gather_into_tensor twice (half slices) per step of 3 runs
    assert d["exchange_ms"] > 0 and 0 < d["exchange_share"] < 1
