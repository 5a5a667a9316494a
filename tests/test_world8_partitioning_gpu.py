"""BASELINE configs[2] / configs[3] on their real 8-rank partitioning.  The box has one GPU and its process
guard admits at most six processes on the card, so the eight ranks are eight THREADS of this process (`tests/thread_comm.py`:
the communicator interface of `srgd_amd.parallel` with barrier-and-copy collectives), each with its own sampler and engine; the
driver's 8-GPU runs execute the same Python (`srgd_amd.parallel`) on `TorchComm` with nccl (= RCCL), which
`tests/test_bench_multirank_gpu.py` exercises at world size 1 and `tests/test_canvas_shard_gpu.py` / `test_sharding_cpu.py` at
world 2-3 over gloo.  dim-16 U-Net, few steps: the property under test is the partitioning -

* configs[2]-shaped: 64 independent images, 8 per rank (item j -> rank j % 8), lock-step groups of 5 + 3, gathered to rank 0
  as uint8 HWC in item order: must equal the 1-rank run image for image;
* configs[3]-shaped: ONE canvas whose tiles are sliced over 8 ranks with a per-step tile all-gather - 1280^2 (25 / 16 tiles:
  slices of 4,4,4,4,4,4,1,0 and 2 x 8) and 1280 x 1792 (35 / 24 tiles: 5 x 7 + 0 and 3 x 8), i.e. short and EMPTY slices -
  DDPM and EDM, host-noise fp32 and device-noise bf16: every rank must return the single-process image bit for bit.
"""
import pytest
import torch

from tests.golden import cases as C
from tests.thread_comm import ThreadWorld

pytestmark = pytest.mark.gpu
WORLD = 8


# ---------------------------------------------------------------------------------------------------------------------
# configs[2]: 64 independent images over 8 ranks
# ---------------------------------------------------------------------------------------------------------------------
N_ITEMS, LOCKSTEP, LR = 64, 5, 96          # 96^2 LR -> 384^2 HR -> 768^2 canvas, 9 / 4 tiles per image


def _sample_items(sampler, items):
    cond = torch.cat([C.synthetic_lr_condition(100 + j, LR, LR) for j in items], 0).cuda()
    sampler.noise_source = "device"
    sampler.device_noise_seed = 71
    return sampler.tiled_sample(batch_size=45, condition_x=cond, class_label=torch.tensor([1]).cuda(), num_sample_steps=3,
                                precision="fp32")


def _images_worker(comm):
    from srgd_amd.parallel import sample_images_sharded
    from tests.test_engine_gpu import build_sampler
    sampler = build_sampler(16, fresh=True)        # one sampler + engine per rank (thread)
    groups = []

    def sample_group(items):
        groups.append(list(items))
        return _sample_items(sampler, items)

    local, ordered = sample_images_sharded(sample_group, N_ITEMS, LOCKSTEP, comm.rank, comm.world, dst=0, comm=comm)
    return {"groups": groups, "ordered": ordered}


def test_configs2_shape_64_images_over_8_ranks_equal_the_one_rank_run():
    from srgd_amd.inference import unit_tensor_to_u8_on_device
    from tests.test_engine_gpu import build_sampler
    sampler = build_sampler(16)
    # the 1-rank run: every image sampled alone (lock-step images are bit-identical to solo runs, so the grouping is free)
    want = [unit_tensor_to_u8_on_device(_sample_items(sampler, [j])[0]).cpu() for j in range(N_ITEMS)]
    assert len({w.numpy().tobytes() for w in want}) == N_ITEMS, "the images must differ for the order check to mean something"
    res = ThreadWorld(WORLD).run(_images_worker)
    for r in range(WORLD):
        got = res[r]
        assert got["groups"] == [[r + k * WORLD for k in range(5)], [r + k * WORLD for k in range(5, 8)]]
        if r:
            assert got["ordered"] is None
            continue
        assert len(got["ordered"]) == N_ITEMS
        for j in range(N_ITEMS):
            assert got["ordered"][j].shape == (4 * LR, 4 * LR, 3) and torch.equal(got["ordered"][j].cpu(), want[j]), j


# ---------------------------------------------------------------------------------------------------------------------
# configs[3]: one canvas over 8 ranks, DDPM and EDM
# ---------------------------------------------------------------------------------------------------------------------
CANVASES = {"1280x1280": (256, 256), "1280x1792": (250, 375)}     # LR sizes: x4 -> 1024^2 / 1000 x 1500 -> 25/16, 35/24 tiles


def _run_canvas(sampler, name, edm):
    cond = C.synthetic_lr_condition(7, *CANVASES[name]).cuda()    # the x4-upsampled condition, as inference.py hands it over
    label = torch.tensor([2]).cuda()
    out = {}
    steps = 3
    for noise, prec in (("host", "fp32"), ("device", "bf16")):
        sampler.noise_source = noise
        sampler.device_noise_seed = 29
        sampler.host_generator = torch.Generator().manual_seed(11)     # per rank: eight threads must not share the global one
        o, imgs, x0s = sampler.tiled_sample(batch_size=8, condition_x=cond, class_label=label, num_sample_steps=steps,
                                            class_cond_scale=1.3, precision=prec, with_images=True, with_x0_images=True)
        out[f"{noise}_{prec}"] = o.cpu()
        out[f"{noise}_{prec}_x0"] = x0s[-1]
    sampler.noise_source = "host"
    sampler.host_generator = None
    return out


def _canvas_worker(comm, name, edm):
    from srgd_amd.parallel import shard_canvas
    from tests.test_engine_gpu import build_edm_sampler, build_sampler
    sampler = shard_canvas(build_edm_sampler(16, fresh=True) if edm else build_sampler(16, fresh=True), comm=comm)
    res = _run_canvas(sampler, name, edm)
    res["exchanges"] = sampler.canvas_group.exchanges
    res["buffers"] = {k: (tuple(v[0].shape), tuple(v[1].shape)) for k, v in sampler.canvas_group._bufs.items()}
    return res


@pytest.mark.parametrize("edm", [False, True], ids=["ddpm", "edm"])
@pytest.mark.parametrize("name", list(CANVASES))
def test_configs3_shape_canvas_over_8_ranks_equals_the_single_process_run(name, edm):
    from srgd_amd.model import _tiling
    from tests.test_engine_gpu import build_edm_sampler, build_sampler
    sampler = build_edm_sampler(16) if edm else build_sampler(16)
    assert sampler.canvas_group is None
    hh, ww = (4 * v for v in CANVASES[name])
    _, (hp, wp), even, odd, _ = _tiling(hh, ww, 256, 256)
    assert f"{hp}x{wp}" == name and (len(even), len(odd)) == ((25, 16) if name == "1280x1280" else (35, 24))
    want = _run_canvas(sampler, name, edm)
    res = ThreadWorld(WORLD).run(_canvas_worker, name, edm)
    width = -(-len(even) // WORLD)
    for r in range(WORLD):
        got = res[r]
        for k in want:
            assert torch.equal(got[k], want[k]), (r, k)
        # two runs x 3 steps x (img + x_start) x two halves of every slice (round 5: the first half's all-gather runs on a side
        # stream under the second half's compute), through four preallocated buffer pairs (half, canvas) sized for the even grid
        h = (width + 1) // 2
        assert got["exchanges"] == 2 * 3 * 2 * 2
        assert sorted(got["buffers"].values()) == sorted([((h, 3, 256, 256), (WORLD * h, 3, 256, 256))] * 2 +
                                                         [((width - h, 3, 256, 256), (WORLD * (width - h), 3, 256, 256))] * 2)
