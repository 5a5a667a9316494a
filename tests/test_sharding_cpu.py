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
