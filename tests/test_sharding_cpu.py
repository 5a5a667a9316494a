"""World-size-2 gloo test of the multi-GPU plumbing (weight broadcast, image sharding, output gather,
max-over-ranks timing) - the N>1 path of bench.py without GPUs."""
import json
import os
import socket

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


def _run_fake(group, steps=5, with_x0=True, edm=False):
    from srgd_amd.parallel import sharded_edm_step, sharded_step
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


def test_sharded_canvas_steps_equal_single_rank_world3(tmp_path):
    port = _free_port()
    mp.spawn(_canvas_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    want_img, want_xs = _run_fake(None)
    want_eimg, want_exs = _run_fake(None, edm=True)
    for r in range(3):
        got = torch.load(tmp_path / f"canvas_r{r}.pt")
        assert torch.equal(got["img"], want_img), r
        assert torch.equal(got["xs"], want_xs), r
        assert torch.equal(got["edm_img"], want_eimg), r       # parallel.sharded_edm_step (EDM wrapper's Heun step)
        assert torch.equal(got["edm_xs"], want_exs), r
