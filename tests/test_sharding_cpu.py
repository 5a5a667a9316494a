"""World-size-2 gloo test of the multi-GPU plumbing (weight broadcast, image sharding, output gather,
max-over-ranks timing) - the N>1 path of bench.py without GPUs."""
import json
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

G = os.path.join(os.path.dirname(__file__), "golden")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from srgd_amd.parallel import broadcast_state_dict, gather_outputs, max_over_ranks, shard_indices
    from srgd_amd.synth import synth_state_dict
    with open(os.path.join(G, "schema_dim16.json")) as f:
        schema = {k: tuple(v) for k, v in json.load(f).items()}
    sd = synth_state_dict(schema, seed=5) if rank == 0 else None
    got = broadcast_state_dict(schema, sd, src=0)
    want = synth_state_dict(schema, seed=5)
    ok = list(got.keys()) == list(want.keys()) and all(torch.equal(got[k], want[k]) for k in want)
    mine = shard_indices(7, rank, world)
    local = torch.stack([torch.full((3, 4, 4), float(i)) for i in mine[:3]])
    bucket = gather_outputs(local, dst=0)
    t = max_over_ranks(1.0 + rank, torch.device("cpu"))
    if rank == 0:
        flat = sorted(int(x[0, 0, 0]) for b in bucket for x in b)
        torch.save({"ok": ok, "gathered": flat, "tmax": t}, os.path.join(out_dir, "r0.pt"))
    else:
        torch.save({"ok": ok, "mine": mine, "tmax": t}, os.path.join(out_dir, "r1.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_shard_gather_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert r0["ok"] and r1["ok"]
    assert r1["mine"] == [1, 3, 5]
    assert r0["gathered"] == [0, 1, 2, 3, 4, 5]
    assert r0["tmax"] == 2.0 and r1["tmax"] == 2.0


def test_shard_indices_cover_everything_once():
    from srgd_amd.parallel import shard_indices
    for n in (0, 1, 7, 64):
        for world in (1, 2, 4, 8):
            allidx = sorted(i for r in range(world) for i in shard_indices(n, r, world))
            assert allidx == list(range(n))


# ---------------------------------------------------------------------------------------------
# one canvas over several ranks: the exchange logic of parallel.sharded_step
# ---------------------------------------------------------------------------------------------
class _FakeEngine:
    """CPU stand-in with the two engine calls sharded_step uses; a tile update depends on what the tile
    region holds (so stale neighbours are detected) and on the tile's global index (like its noise)."""

    def __init__(self, grids):
        self.grids = grids

    def sampler_step_tiles(self, step, first, count, do_ring, img, cond, x_start, nt, nc, passes, kind, scale, sb, seed):
        for t in range(first, first + count):
            y, x = self.grids[step & 1][t]
            reg = img[0, :, y:y + 256, x:x + 256]
            reg.copy_(reg * 0.5 + reg.mean() + (t + 1) * 0.01 + step)
            if x_start is not None:
                x_start[0, :, y:y + 256, x:x + 256] = reg * 2
        if (step & 1) and do_ring:
            img[0, :, :128] = step

    def edm_step_tiles(self, step, first, count, do_ring, img, cond, x_start, work, z, ring, passes, kind, scale, sb, seed):
        # two evaluations per tile (Heun): the second reads what the first left in the rank-local scratch canvas
        for t in range(first, first + count):
            y, x = self.grids[step & 1][t]
            reg = img[0, :, y:y + 256, x:x + 256]
            work[0, 0, :, y:y + 256, x:x + 256] = reg * 0.25 + (t + 1) * 0.02
            reg.copy_(reg * 0.5 + work[0, 0, :, y:y + 256, x:x + 256].mean() + step)
            if x_start is not None:
                x_start[0, :, y:y + 256, x:x + 256] = reg * 3
        if (step & 1) and do_ring:
            img[0, :, :128] = -step

    def sampler_exchange_tiles(self, parity, first, count, canvas, tiles, to_canvas):
        for j in range(count):
            y, x = self.grids[parity][first + j]
            if to_canvas:
                canvas[0, :, y:y + 256, x:x + 256] = tiles[j]
            else:
                tiles[j] = canvas[0, :, y:y + 256, x:x + 256]


def _canvas_grids():
    from srgd_amd.model import get_coords
    hp = wp = 1024
    even = [(a, c) for (a, _, c, _) in get_coords(hp, wp, 256, 256, diff=0)]
    odd = [(a, c) for (a, _, c, _) in get_coords(hp - 256, wp - 256, 256, 256, diff=128)]
    return hp, wp, (even, odd)


def _run_fake(group, steps=5, with_x0=True, edm=False, always_exchange=False, shard_out=None):
    from srgd_amd.parallel import CanvasShard, TorchComm, sharded_edm_step, sharded_step
    if group is not None:
        group = CanvasShard(group if hasattr(group, "all_gather_tiles") else TorchComm(group), always_exchange=always_exchange)
        if shard_out is not None:
            shard_out.append(group)
    hp, wp, grids = _canvas_grids()
    eng = _FakeEngine(grids)
    img = torch.linspace(-1, 1, 3 * hp * wp).reshape(1, 3, hp, wp).clone()
    xs = img.clone() if with_x0 else None
    work = torch.zeros(2, 1, 3, hp, wp)
    for i in range(steps):
        n = len(grids[i & 1])
        if edm and group is None:
            eng.edm_step_tiles(i, 0, n, True, img, None, xs, work, None, None, 1, 0, 1.0, n, 0)
        elif edm:
            sharded_edm_step(eng, group, i, n, img, None, xs, work, None, None, 1, 0, 1.0, n, 0)
        elif group is None:
            eng.sampler_step_tiles(i, 0, n, True, img, None, xs, None, None, 1, 0, 1.0, n, 0)
        else:
            sharded_step(eng, group, i, n, img, None, xs, None, None, 1, 0, 1.0, n, 0)
    return img, xs


def _canvas_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    img, xs = _run_fake(dist.group.WORLD)
    eimg, exs = _run_fake(dist.group.WORLD, edm=True)
    torch.save({"img": img, "xs": xs, "edm_img": eimg, "edm_xs": exs}, os.path.join(out_dir, f"canvas_r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_slices_partition_the_grid():
    from srgd_amd.parallel import tile_slices
    for n in (0, 1, 9, 16, 25, 1024, 1089):
        for world in (1, 2, 3, 8):
            sl = tile_slices(n, world)
            assert [t for r in sl for t in r] == list(range(n))
            w = len(sl[0])
            assert all(r.start == min(n, k * w) for k, r in enumerate(sl)) and all(len(r) <= w for r in sl)


# world 3: uneven slices; world 8 (BASELINE configs[3]'s rank count): 16 / 9 tiles over 8 ranks = slices of 2 with empty
# ranks on the odd grid (9 tiles -> widths 2,2,2,2,1,0,0,0)
@pytest.mark.parametrize("world", [3, 8])
def test_sharded_canvas_steps_equal_single_rank(tmp_path, world):
    port = _free_port()
    mp.spawn(_canvas_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    want_img, want_xs = _run_fake(None)
    want_eimg, want_exs = _run_fake(None, edm=True)
    for r in range(world):
        got = torch.load(tmp_path / f"canvas_r{r}.pt")
        assert torch.equal(got["img"], want_img), r
        assert torch.equal(got["xs"], want_xs), r
        assert torch.equal(got["edm_img"], want_eimg), r       # parallel.sharded_edm_step (EDM wrapper's Heun step)
        assert torch.equal(got["edm_xs"], want_exs), r


def _world1_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shards = []
    img, xs = _run_fake(dist.group.WORLD, always_exchange=True, shard_out=shards)
    quiet = []
    img2, _ = _run_fake(dist.group.WORLD, always_exchange=False, shard_out=quiet)
    bufs = next(iter(shards[0]._bufs.values()))
    torch.save({"img": img, "xs": xs, "img2": img2, "exchanges": shards[0].exchanges, "quiet": quiet[0].exchanges,
                "n_buffers": len(shards[0]._bufs), "packed": tuple(bufs[0].shape), "everyone": tuple(bufs[1].shape)},
               os.path.join(out_dir, "w1.pt"))
    dist.destroy_process_group()


def test_forced_exchange_at_world_size_one_is_a_no_op_on_the_result(tmp_path):
    # bench.py's SRGD_FORCE_DIST hook: pack -> all-gather -> unpack runs even with one rank (RCCL on a 1-GPU box) and must not
    # change anything; the exchange buffers are allocated once for the widest grid (16 tiles) and reused by all 5 steps
    mp.spawn(_world1_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    got = torch.load(tmp_path / "w1.pt")
    want_img, want_xs = _run_fake(None)
    assert torch.equal(got["img"], want_img) and torch.equal(got["xs"], want_xs) and torch.equal(got["img2"], want_img)
    assert got["exchanges"] == 5 * 2 and got["quiet"] == 0          # img + x_start canvases, five steps
    assert got["n_buffers"] == 1 and got["packed"] == (16, 3, 256, 256) and got["everyone"] == (16, 3, 256, 256)


# ---------------------------------------------------------------------------------------------
# independent images over ranks (BASELINE configs[2]): parallel.sample_images_sharded
# ---------------------------------------------------------------------------------------------
def _fake_image(j):
    return torch.full((1, 3, 8, 8), (j % 200) / 255.0 + 1e-4)


def _images_worker(rank, world, port, out_dir, n_items, lockstep):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from srgd_amd.parallel import sample_images_sharded
    groups = []

    def sample_group(items):
        groups.append(list(items))
        return torch.cat([_fake_image(j) for j in items], 0)

    local, ordered = sample_images_sharded(sample_group, n_items, lockstep, rank, world, dst=0)
    torch.save({"groups": groups, "local": local, "ordered": ordered}, os.path.join(out_dir, f"img_r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_64_images_over_8_ranks_come_back_in_item_order(tmp_path):
    # configs[2]'s partitioning: 64 independent tiles, 8 per rank (item j -> rank j % 8), lock-step groups of 5 (5 + 3)
    world, n = 8, 64
    mp.spawn(_images_worker, args=(world, _free_port(), str(tmp_path), n, 5), nprocs=world, join=True)
    for r in range(world):
        got = torch.load(tmp_path / f"img_r{r}.pt")
        assert got["groups"] == [[r + k * world for k in range(5)], [r + k * world for k in range(5, 8)]]
        assert got["local"].shape == (8, 3, 8, 8)
        if r == 0:
            assert len(got["ordered"]) == n
            for j, img in enumerate(got["ordered"]):
                assert img.dtype == torch.uint8 and img.shape == (8, 8, 3) and int(img[0, 0, 0]) == j % 200, j
        else:
            assert got["ordered"] is None


def test_thread_communicator_matches_the_gloo_path_at_world_8():
    # tests/thread_comm.py (ranks as threads of one process) is what the GPU suite runs the 8-rank partitionings on (a one-GPU
    # box admits at most six GPU processes).  Here, on CPU, the same stand-in engine goes through the same sharded steps on
    # that communicator: every rank must end with the single-rank canvases - i.e. the stand-in transport is a faithful
    # all-gather / gather / broadcast / all-reduce - and the helper collectives are checked directly.
    from tests.thread_comm import ThreadWorld
    want_img, want_xs = _run_fake(None)
    want_eimg, want_exs = _run_fake(None, edm=True)

    def rank_body(comm):
        img, xs = _run_fake(comm)
        eimg, exs = _run_fake(comm, edm=True)
        mine = torch.full((2, 3), float(comm.rank))
        bucket = comm.gather(mine, dst=0)
        flat = torch.arange(5.) if comm.rank == 0 else torch.zeros(5)
        comm.broadcast(flat, src=0)
        t = torch.tensor([float(comm.rank)], dtype=torch.float64)
        comm.all_reduce_max(t)
        return img, xs, eimg, exs, bucket, flat, t.item()

    res = ThreadWorld(8, timeout=120.0).run(rank_body)
    for r, (img, xs, eimg, exs, bucket, flat, tmax) in enumerate(res):
        assert torch.equal(img, want_img) and torch.equal(xs, want_xs), r
        assert torch.equal(eimg, want_eimg) and torch.equal(exs, want_exs), r
        assert torch.equal(flat, torch.arange(5.)) and tmax == 7.0
        assert (bucket is None) == (r != 0)
    assert [float(b[0, 0]) for b in res[0][4]] == [float(k) for k in range(8)]
