"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on
ROCm; ``gloo`` in the CPU tests).  The path shards by independent images: there
is no data-path collective - only a one-time weight broadcast and a final gather of the HR outputs.
One huge image (config 4: 8192^2, 1089/1024 tiles per step) shards differently: every rank keeps the whole
canvas, runs a contiguous slice of each step's tiles, and the updated tiles are all-gathered after the step
(856 MB fp32 per step at 8448^2, ~3 ms over 7 xGMI links against ~130 ms of compute per rank and step).
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Optional, Sequence

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin image assignment (the reference's --start_index/--end_index manual sharding, automated)."""
    return list(range(rank, n_items, world))


class TorchComm:
    """The communicator the product runs on: a ``torch.distributed`` process group (``nccl`` = RCCL on the GPU box, ``gloo`` in
    the CPU tests).  The sharding code below talks to this small interface only - rank, world, backend and the four collectives
    the path needs - so that a test can stand eight ranks up as eight THREADS of one process (``tests/thread_comm.py``): a
    one-GPU box admits at most six processes on its card, and the 8-rank partitionings of BASELINE configs[2]/[3] have to be
    exercised on it."""

    def __init__(self, group=None):
        self.group = group if group is not None else dist.group.WORLD

    @property
    def rank(self) -> int:
        return dist.get_rank(self.group)

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group)

    @property
    def backend(self) -> str:
        return dist.get_backend(self.group)

    def all_gather_tiles(self, out: torch.Tensor, mine: torch.Tensor) -> None:
        """``out`` [world*width, ...] <- every rank's ``mine`` [width, ...] in rank order (both preallocated, contiguous)."""
        if self.backend == "nccl":
            dist.all_gather_into_tensor(out, mine, group=self.group)       # one RCCL all-gather, no staging copies
        else:
            dist.all_gather(list(out.split(mine.shape[0], 0)), mine, group=self.group)

    def gather(self, local: torch.Tensor, dst: int = 0) -> Optional[List[torch.Tensor]]:
        if self.backend != "nccl" and local.is_cuda:
            local = local.cpu()
        bucket = [torch.empty_like(local) for _ in range(self.world)] if self.rank == dst else None
        dist.gather(local, bucket, dst=dst, group=self.group)
        return bucket

    def broadcast(self, flat: torch.Tensor, src: int = 0) -> None:
        dist.broadcast(flat, src=src, group=self.group)

    def all_reduce_max(self, t: torch.Tensor) -> None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)

    def barrier(self) -> None:
        dist.barrier(group=self.group)


def _comm(comm=None):
    """``comm`` itself, or the WORLD group's communicator; None when no process group exists (single-process runs)."""
    if comm is not None:
        return comm
    return TorchComm() if dist.is_initialized() else None


def broadcast_state_dict(schema: Mapping[str, Sequence[int]], state_dict: Optional[Dict[str, torch.Tensor]],
                         src: int = 0, device: torch.device = torch.device("cpu"), comm=None) -> Dict[str, torch.Tensor]:
    """Rank ``src`` owns the checkpoint; every rank returns an identical CPU fp32 state_dict.  The tensors
    travel as ONE flat buffer (550 MB fp32 for the dim-128 model): a single broadcast instead of 280."""
    comm = _comm(comm)
    keys = list(schema.keys())
    sizes = [int(torch.Size(schema[k]).numel()) for k in keys]
    if comm.backend != "nccl":
        device = torch.device("cpu")                 # gloo (CPU tests, shared-GPU test hook): stage through host memory
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    if comm.rank == src:
        assert state_dict is not None
        flat.copy_(torch.cat([state_dict[k].reshape(-1).float() for k in keys]))
    comm.broadcast(flat, src=src)
    host = flat.cpu()
    out, o = {}, 0
    for k, n in zip(keys, sizes):
        out[k] = host[o:o + n].reshape(tuple(schema[k])).clone()
        o += n
    return out


def gather_outputs(local: torch.Tensor, dst: int = 0, comm=None) -> Optional[List[torch.Tensor]]:
    """Gather equally-shaped per-rank output stacks to ``dst`` (returns the list there, None elsewhere)."""
    return _comm(comm).gather(local, dst=dst)


def gather_outputs_u8(local: torch.Tensor, dst: int = 0, comm=None) -> Optional[List[torch.Tensor]]:
    """Gather the HR outputs the way they leave the pipeline (``ToPILImage``: uint8 HWC, inference.py:93): each rank converts
    its ``[n,3,H,W]`` fp32 stack on the GPU and 3.1 MB per 1024^2 tile travel instead of 12.6 MB."""
    if local.is_cuda:
        from .inference import unit_tensor_to_u8_on_device
        local = torch.stack([unit_tensor_to_u8_on_device(img) for img in local], 0)
    else:                                            # CPU tests (gloo): same arithmetic, torch ops
        local = local.mul(255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    return gather_outputs(local, dst=dst, comm=comm)


def sample_images_sharded(sample_group, n_items: int, lockstep: int, rank: int, world: int, dst: int = 0, gather: bool = True,
                          comm=None):
    """Independent images over ranks (BASELINE configs[2]; the reference's manual ``--start_index/--end_index`` sharding,
    inference.py:36-37,120, automated): item ``j`` of ``0..n_items-1`` belongs to rank ``j % world``; every rank samples its
    items in lock-step groups of ``lockstep`` through ``sample_group(list_of_item_ids) -> [k,3,H,W]`` and the HR outputs are
    gathered to ``dst`` as the uint8 HWC images the pipeline emits.  ``n_items`` must be a multiple of ``world`` (equal-sized
    gathers).  Returns ``(local_outputs, ordered)``: ``ordered`` is the list of all ``n_items`` uint8 images in item order on
    ``dst`` (None elsewhere, and None when ``gather`` is False or there is no process group).  ``comm``: the communicator
    (default: the WORLD process group)."""
    assert n_items % world == 0, "equal shares per rank (the gather moves equally-shaped stacks)"
    mine = shard_indices(n_items, rank, world)
    outs = []
    for a in range(0, len(mine), lockstep):
        outs.append(sample_group(mine[a:a + lockstep]))
    local = torch.cat(outs, 0)
    comm = _comm(comm)
    if not gather or comm is None:
        return local, None
    bucket = gather_outputs_u8(local, dst=dst, comm=comm)
    if bucket is None:
        return local, None
    ordered = [None] * n_items
    for r, stack in enumerate(bucket):
        for k, j in enumerate(shard_indices(n_items, r, world)):
            ordered[j] = stack[k]
    return local, ordered


def max_over_ranks(seconds: float, device: torch.device, comm=None) -> float:
    comm = _comm(comm)
    t = torch.tensor([seconds], dtype=torch.float64, device=device if comm.backend == "nccl" else "cpu")
    comm.all_reduce_max(t)
    return float(t.item())


# ---------------------------------------------------------------------------------------------
# one canvas over several GPUs
# ---------------------------------------------------------------------------------------------
def tile_slices(n_tiles: int, world: int) -> List[range]:
    """Contiguous equal-width slices (the last ones may be short or empty): rank r owns tiles
    ``[r*w, min(n, (r+1)*w))`` with ``w = ceil(n / world)``, so that the all-gathered buffer
    ``[world, w, ...]`` flattened holds tiles 0..n-1 in grid order in its first n entries."""
    w = -(-n_tiles // world)
    return [range(min(n_tiles, r * w), min(n_tiles, (r + 1) * w)) for r in range(world)]


class CanvasShard:
    """State of one canvas shared by the ranks of ``comm``: the communicator plus the exchange buffers of the per-step tile
    all-gather, allocated ONCE per (grid width, device) and reused by every step of every run (round 3: the per-step
    ``torch.zeros`` + receive-buffer allocations are gone, so a sharded step allocates nothing and the caching allocator
    never has to synchronise mid-run).

    Payload: the fp32 canvas tiles themselves (x_t is fp32 state in the reference, model.py:3381-3393); rounding them to
    bf16 for transport would make the sharded run differ from the single-GPU run, which is the property the sharded path is
    tested by (bit-identical at any world size).  Every tile of a step's grid changes in that step, so "only the changed
    tiles" is what already travels: 856 MB per step at 8448^2, ~3 ms of RCCL all-gather against ~110 ms of compute per rank.

    ``always_exchange``: run the pack -> all-gather -> unpack round trip even at world size 1 (bench.py's SRGD_FORCE_DIST
    hook: it puts ``all_gather_into_tensor`` on RCCL through its paces on a 1-GPU box; the result is unchanged)."""

    def __init__(self, comm, always_exchange: bool = False):
        self.comm = comm
        self.always_exchange = always_exchange
        self._bufs = {}
        self.exchanges = 0                      # tile all-gathers issued so far (tests / bench report it)
        # bench.py's canvas workload: time every exchange (pack -> all-gather -> unpack) with a pair of events on the stream the
        # engine runs on, so that the JSON line of an N-GPU run shows what the per-step collective really costs (DESIGN section 7
        # predicts ~3 ms per step for 856 MB at 8448^2 over 8 GPUs); off by default - a timed pair is two event records per exchange
        self.timing = False
        self._events = []

    def exchange_ms(self, reset: bool = True) -> float:
        """Sum of the timed exchanges so far, in ms (synchronises the device)."""
        if not self._events:
            return 0.0
        torch.cuda.synchronize()
        ms = float(sum(a.elapsed_time(b) for a, b in self._events))
        if reset:
            self._events = []
        return ms

    def buffers(self, width: int, world: int, device, tile: int = 256):
        """(packed [width,3,T,T], everyone [world*width,3,T,T]) views of buffers sized for the widest grid seen so far."""
        key = (str(device), tile)
        need = world * width
        have = self._bufs.get(key)
        if have is None or have[1].shape[0] < need or have[0].shape[0] < width:
            packed = torch.zeros(width, 3, tile, tile, device=device, dtype=torch.float32)
            everyone = torch.empty(need, 3, tile, tile, device=device, dtype=torch.float32)
            self._bufs[key] = have = (packed, everyone)
        return have[0][:width], have[1][:need]


def shard_canvas(sampler, group=None, always_exchange: bool = False, comm=None):
    """Make ``sampler.tiled_sample`` split every step's tile list over the ranks of ``group`` (default: WORLD; or of an explicit
    communicator ``comm``).
    All ranks must call tiled_sample with identical arguments (and, in host-noise mode, identical torch seeds);
    all of them return the full image.  Results are bit-identical to the single-GPU run: tiles are independent
    within a step (model.py:3361-3380) and the noise of a tile depends only on its index in the grid."""
    sampler.canvas_group = CanvasShard(comm if comm is not None else TorchComm(group), always_exchange)
    return sampler


def _exchange(eng, shard: CanvasShard, step: int, n_tiles: int, mine: range, width: int, canvases) -> None:
    world = shard.comm.world
    if world == 1 and not shard.always_exchange:
        return
    for canvas in canvases:
        if canvas is None:
            continue
        packed, everyone = shard.buffers(width, world, canvas.device)
        ev = None
        if shard.timing:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        eng.sampler_exchange_tiles(step & 1, mine.start, len(mine), canvas, packed, to_canvas=False)
        shard.comm.all_gather_tiles(everyone, packed)
        shard.exchanges += 1
        eng.sampler_exchange_tiles(step & 1, 0, n_tiles, canvas, everyone, to_canvas=True)
        if ev is not None:
            ev[1].record()
            shard._events.append(ev)


def sharded_step(eng, shard: CanvasShard, step: int, n_tiles: int, img, cond_canvas, x_start, noise_tiles, noise_canvas,
                 passes: int, kind: int, scale: float, sub_batch: int, seed: int) -> None:
    """One DDPM step of a canvas shared by the ranks of ``shard.comm``: my slice of the tiles, then exchange."""
    rank, world = shard.comm.rank, shard.comm.world
    sl = tile_slices(n_tiles, world)
    mine, width = sl[rank], len(sl[0])
    eng.sampler_step_tiles(step, mine.start, len(mine), True, img, cond_canvas, x_start, noise_tiles, noise_canvas,
                           passes, kind, scale, sub_batch, seed)
    _exchange(eng, shard, step, n_tiles, mine, width, (img, x_start))


def sharded_edm_step(eng, shard: CanvasShard, step: int, n_tiles: int, img, cond_canvas, x_start, work, noise_canvas,
                     ring_noise_canvas, passes: int, kind: int, scale: float, sub_batch: int, seed: int) -> None:
    """One EDM (Heun) step of a canvas shared by the ranks of ``shard.comm`` (reference model.py:2379-2463): my slice of the
    tiles (both network evaluations; the scratch canvases stay rank-local), the odd-step ring on every rank's own canvas, then
    the same tile exchange as the DDPM loop."""
    rank, world = shard.comm.rank, shard.comm.world
    sl = tile_slices(n_tiles, world)
    mine, width = sl[rank], len(sl[0])
    eng.edm_step_tiles(step, mine.start, len(mine), True, img, cond_canvas, x_start, work, noise_canvas, ring_noise_canvas,
                       passes, kind, scale, sub_batch, seed)
    _exchange(eng, shard, step, n_tiles, mine, width, (img, x_start))
