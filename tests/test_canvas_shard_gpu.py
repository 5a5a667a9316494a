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
                                class_cond_scale=1.4, amp=amp, **kw)


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from srgd_amd.parallel import shard_canvas
    from tests.test_engine_gpu import build_sampler
    case = next(c for c in C.SAMPLER_CASES if c["name"] == "dim16_300x500")
    sampler = shard_canvas(build_sampler(case["dim"]))
    out = {"host_fp32": _sample(sampler, case, "host", False).cpu(),
           "device_bf16": _sample(sampler, case, "device", True).cpu()}
    o, imgs, x0s = _sample(sampler, case, "host", False, with_images=True, with_x0_images=True)
    out["x0_last"] = x0s[-1]
    torch.save(out, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sharing_a_canvas_equal_the_single_process_run(tmp_path):
    from tests.test_engine_gpu import build_sampler
    case = next(c for c in C.SAMPLER_CASES if c["name"] == "dim16_300x500")
    sampler = build_sampler(case["dim"])
    assert sampler.canvas_group is None
    want = {"host_fp32": _sample(sampler, case, "host", False).cpu(),
            "device_bf16": _sample(sampler, case, "device", True).cpu()}
    o, imgs, x0s = _sample(sampler, case, "host", False, with_images=True, with_x0_images=True)
    want["x0_last"] = x0s[-1]
    sampler.noise_source = "host"
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = torch.load(tmp_path / f"r{r}.pt")
        for k in want:
            assert torch.equal(got[k], want[k]), (r, k)
