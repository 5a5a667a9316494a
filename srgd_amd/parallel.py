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
        # bench.py's canvas workload: time what every exchange leaves on the compute stream (second half's pack -> all-gather, the
        # wait for the side stream, the unpacking) with a pair of events, so that the JSON line of an N-GPU run shows what the
        # per-step collective really costs; off by default - a timed pair is two event records per exchange
        self.timing = False
        self._events = []
        self.overlap = True                     # first half of my slice all-gathered on a side stream under the second half's compute
        self._side = {}

    def exchange_ms(self, reset: bool = True) -> float:
        """Sum of the timed exchanges so far, in ms (synchronises the device)."""
        if not self._events:
            return 0.0
        torch.cuda.synchronize()
        ms = float(sum(a.elapsed_time(b) for a, b in self._events))
        if reset:
            self._events = []
        return ms

    def side_stream(self, device):
        """The stream the first half's exchange runs on while the second half computes (one per device, created once)."""
        key = str(device)
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device=device)
        return self._side[key]

    def _timed_begin(self):
        if not self.timing:
            return None
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        return ev

    def _timed_end(self, ev) -> None:
        if ev is not None:
            ev[1].record()
            self._events.append(ev)

    def buffers(self, width: int, world: int, device, tile: int = 256, slot=(0, 0)):
        """(packed [width,3,T,T], everyone [world*width,3,T,T]) views of buffers sized for the widest grid seen so far; one
        pair per ``slot`` = (half of the slice, canvas): two halves and two canvases can be in flight at once."""
        key = (str(device), tile, slot)
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


def _gather_part(eng, shard: CanvasShard, step: int, first: int, count: int, part: int, pw: int, canvases) -> list:
    """Pack my tiles [first, first + count) of every canvas and all-gather them, on the CURRENT stream.  Returns the gathered
    buffers (one per canvas, [world * pw, 3, T, T]; rank r's tiles in rows [r * pw, r * pw + its count))."""
    world = shard.comm.world
    out = []
    for ci, canvas in enumerate(canvases):
        if canvas is None:
            out.append(None)
            continue
        packed, everyone = shard.buffers(pw, world, canvas.device, slot=(part, ci))
        eng.sampler_exchange_tiles(step & 1, first, count, canvas, packed, to_canvas=False)
        shard.comm.all_gather_tiles(everyone, packed)
        shard.exchanges += 1
        out.append(everyone)
    return out


def _scatter_part(eng, shard: CanvasShard, step: int, n_tiles: int, w: int, off: int, pw: int, canvases, gathered) -> None:
    """Write a gathered part back: rank r's rows are tiles [r * w + off, r * w + off + pw) of the grid (clipped to n_tiles)."""
    world = shard.comm.world
    for canvas, everyone in zip(canvases, gathered):
        if canvas is None:
            continue
        if off == 0 and pw == w:                         # whole slices: the gathered buffer IS the grid in order - one launch
            eng.sampler_exchange_tiles(step & 1, 0, n_tiles, canvas, everyone, to_canvas=True)
            continue
        if hasattr(eng, "sampler_unpack_gathered"):      # half-slices: still ONE launch (row -> tile mapping in the kernel)
            eng.sampler_unpack_gathered(step & 1, world, w, off, pw, canvas, everyone)
            continue
        for r in range(world):                           # (engines without the strided unpack: one launch per rank)
            t0 = min(n_tiles, r * w + off)
            cnt = min(n_tiles, r * w + off + pw) - t0
            if cnt > 0:
                eng.sampler_exchange_tiles(step & 1, t0, cnt, canvas, everyone[r * pw:r * pw + cnt], to_canvas=True)


def _step_and_exchange(eng, shard: CanvasShard, step: int, n_tiles: int, canvases, run_tiles) -> None:
    """My slice of the step's tiles through ``run_tiles(first, count, do_ring)``, then the exchange.  Round 5: the slice runs
    as two halves (the engine already splits it into balanced launches), and the FIRST half's tiles are packed and all-gathered
    on a side stream while the second half computes; only the second half's gather and the unpacking are left on the compute
    stream.  Tiles of a step are disjoint canvas regions and the odd-step ring lies outside all of them, so the side stream reads
    finished tiles only.  Every rank splits at the same offset h = ceil(w / 2) of its slice, so the two gathers have the same
    shape everywhere (short and empty slices send zeros, as before)."""
    rank, world = shard.comm.rank, shard.comm.world
    sl = tile_slices(n_tiles, world)
    mine, w = sl[rank], len(sl[0])
    if world == 1 and not shard.always_exchange:
        run_tiles(mine.start, len(mine), True)
        return
    cuda = any(c is not None and c.is_cuda for c in canvases)
    h = (w + 1) // 2 if (shard.overlap and cuda and w >= 2) else w
    a0, a1 = mine.start, min(mine.stop, mine.start + h)
    if h == w:                                           # one part: compute, then pack -> all-gather -> unpack per canvas, in sequence
        run_tiles(a0, a1 - a0, True)                     # (through ONE buffer pair: a canvas is unpacked before the next one is packed)
        ev = shard._timed_begin()
        for canvas in canvases:
            g = _gather_part(eng, shard, step, a0, a1 - a0, 0, w, (canvas,))
            _scatter_part(eng, shard, step, n_tiles, w, 0, w, (canvas,), g)
        shard._timed_end(ev)
        return
    # streams and events of the CANVASES' device: the caller's current device may be another one, and a side stream
    # made on it would neither carry the pack / all-gather nor order against the engine's stream
    dev = next(c.device for c in canvases if c is not None and c.is_cuda)
    with torch.cuda.device(dev):
        main = torch.cuda.current_stream(dev)
        side = shard.side_stream(dev)
        run_tiles(a0, a1 - a0, False)
        first_done = torch.cuda.Event()
        first_done.record(main)
        with torch.cuda.stream(side):
            side.wait_event(first_done)
            g0 = _gather_part(eng, shard, step, a0, a1 - a0, 0, h, canvases)
            gathered = torch.cuda.Event()
            gathered.record(side)
        run_tiles(a1, mine.stop - a1, True)                  # overlaps the first half's exchange
        ev = shard._timed_begin()                            # what is left on the compute stream is the EXPOSED exchange time
        g1 = _gather_part(eng, shard, step, a1, mine.stop - a1, 1, w - h, canvases)
        main.wait_event(gathered)
        _scatter_part(eng, shard, step, n_tiles, w, 0, h, canvases, g0)
        _scatter_part(eng, shard, step, n_tiles, w, h, w - h, canvases, g1)
        shard._timed_end(ev)


def sharded_step(eng, shard: CanvasShard, step: int, n_tiles: int, img, cond_canvas, x_start, noise_tiles, noise_canvas,
                 passes: int, kind: int, scale: float, sub_batch: int, seed: int) -> None:
    """One DDPM step of a canvas shared by the ranks of ``shard.comm``: my slice of the tiles, then exchange."""
    _step_and_exchange(eng, shard, step, n_tiles, (img, x_start),
                       lambda first, count, ring: eng.sampler_step_tiles(step, first, count, ring, img, cond_canvas, x_start,
                                                                         noise_tiles, noise_canvas, passes, kind, scale,
                                                                         sub_batch, seed))


def sharded_edm_step(eng, shard: CanvasShard, step: int, n_tiles: int, img, cond_canvas, x_start, work, noise_canvas,
                     ring_noise_canvas, passes: int, kind: int, scale: float, sub_batch: int, seed: int) -> None:
    """One EDM (Heun) step of a canvas shared by the ranks of ``shard.comm`` (reference model.py:2379-2463): my slice of the
    tiles (both network evaluations; the scratch canvases stay rank-local), the odd-step ring on every rank's own canvas, then
    the same tile exchange as the DDPM loop."""
    _step_and_exchange(eng, shard, step, n_tiles, (img, x_start),
                       lambda first, count, ring: eng.edm_step_tiles(step, first, count, ring, img, cond_canvas, x_start, work,
                                                                     noise_canvas, ring_noise_canvas, passes, kind, scale,
                                                                     sub_batch, seed))
