"""One canvas sharded over two ranks, run as two processes that share cuda:0 and
exchange tiles over gloo: every rank must return the image the single-process run returns, bit for bit."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from tests.golden import cases as C

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _sample(sampler, case, noise, amp, **kw):
    cond = C.sampler_condition(case).cuda()
    label = torch.tensor([case["label"]]).cuda()
    sampler.noise_source = noise
    sampler.device_noise_seed = 17
    torch.manual_seed(case["seed"])
    return sampler.tiled_sample(batch_size=4, condition_x=cond, class_label=label, num_sample_steps=case["steps"],
                                class_cond_scale=1.4, precision="bf16" if amp else "fp32", **kw)


def _case(name):
    return next(c for c in C.SAMPLER_CASES + C.WIDE_CASES if c["name"] == name)


def _run_all(sampler, case):
    out = {"host_fp32": _sample(sampler, case, "host", False).cpu(),
           "device_bf16": _sample(sampler, case, "device", True).cpu()}
    o, imgs, x0s = _sample(sampler, case, "host", False, with_images=True, with_x0_images=True)
    out["x0_last"] = x0s[-1]
    sampler.noise_source = "host"
    return out


def _worker(rank, world, port, out_dir, case_name):
    import datetime
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from srgd_amd.parallel import shard_canvas
    from tests.test_engine_gpu import build_sampler
    case = _case(case_name)
    sampler = shard_canvas(build_sampler(case["dim"]))
    torch.save(_run_all(sampler, case), os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


# dim 16 on a 768^2 canvas (9 / 4 tiles) and the dim-128 U-Net on BASELINE configs[1]'s 1280^2 canvas (25 / 16 tiles: slices of
# 13 + 12 and 8 + 8 tiles, CFG pairs batched per launch)
@pytest.mark.parametrize("case_name", ["dim16_300x500", "dim128_config2_1024_2steps"])
def test_two_ranks_sharing_a_canvas_equal_the_single_process_run(tmp_path, case_name):
    from tests.test_engine_gpu import build_sampler
    case = _case(case_name)
    sampler = build_sampler(case["dim"])
    assert sampler.canvas_group is None
    want = _run_all(sampler, case)
    # bounded wait: a rendezvous or collective that never completes must fail this test, not stall the suite
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), case_name)) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=300)
    hung = [pr for pr in procs if pr.is_alive()]
    for pr in hung:
        pr.terminate()
    assert not hung, "a rank did not finish within 300 s"
    assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
    for r in range(2):
        got = torch.load(tmp_path / f"r{r}.pt")
        for k in want:
            assert torch.equal(got[k], want[k]), (r, k)


# ---- the EDM wrapper's tiled loop over the same sharding (srgd_edm_step_tiles + parallel.sharded_edm_step)
def _run_all_edm(sampler):
    cond = C.synthetic_lr_condition(5, 75, 125).cuda()            # 300 x 500 -> 768^2 canvas, 9 / 4 tiles
    label = torch.tensor([2]).cuda()
    out = {}
    for noise, prec in (("host", "fp32"), ("device", "bf16")):
        sampler.noise_source = noise
        sampler.device_noise_seed = 23
        torch.manual_seed(5)
        o, imgs, x0s = sampler.tiled_sample(batch_size=4, condition_x=cond, class_label=label, num_sample_steps=5,
                                            class_cond_scale=1.4, precision=prec, with_images=True, with_x0_images=True)
        out[noise + "_" + prec] = o.cpu()
        out[noise + "_x0_last"] = x0s[-1]
    sampler.noise_source = "host"
    return out


def _edm_worker(rank, world, port, out_dir):
    import datetime
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from srgd_amd.parallel import shard_canvas
    from tests.test_engine_gpu import build_edm_sampler
    sampler = shard_canvas(build_edm_sampler(16))
    torch.save(_run_all_edm(sampler), os.path.join(out_dir, f"edm_r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sharing_an_edm_canvas_equal_the_single_process_run(tmp_path):
    from tests.test_engine_gpu import build_edm_sampler
    sampler = build_edm_sampler(16)
    assert sampler.canvas_group is None
    want = _run_all_edm(sampler)
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_edm_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=300)
    hung = [pr for pr in procs if pr.is_alive()]
    for pr in hung:
        pr.terminate()
    assert not hung, "a rank did not finish within 300 s"
    assert all(pr.exitcode == 0 for pr in procs), [pr.exitcode for pr in procs]
    for r in range(2):
        got = torch.load(tmp_path / f"edm_r{r}.pt")
        for k in want:
            assert torch.equal(got[k], want[k]), (r, k)
