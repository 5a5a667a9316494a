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


def broadcast_state_dict(schema: Mapping[str, Sequence[int]], state_dict: Optional[Dict[str, torch.Tensor]],
                         src: int = 0, device: torch.device = torch.device("cpu")) -> Dict[str, torch.Tensor]:
    """Rank ``src`` owns the checkpoint; every rank returns an identical CPU fp32 state_dict.  The tensors
    travel as ONE flat buffer (550 MB fp32 for the dim-128 model): a single broadcast instead of 280."""
    keys = list(schema.keys())
    sizes = [int(torch.Size(schema[k]).numel()) for k in keys]
    if dist.get_backend() != "nccl":
        device = torch.device("cpu")                 # gloo (CPU tests, shared-GPU test hook): stage through host memory
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    if dist.get_rank() == src:
        assert state_dict is not None
        flat.copy_(torch.cat([state_dict[k].reshape(-1).float() for k in keys]))
    dist.broadcast(flat, src=src)
    host = flat.cpu()
    out, o = {}, 0
    for k, n in zip(keys, sizes):
        out[k] = host[o:o + n].reshape(tuple(schema[k])).clone()
        o += n
    return out


def gather_outputs(local: torch.Tensor, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Gather equally-shaped per-rank output stacks to ``dst`` (returns the list there, None elsewhere)."""
    world = dist.get_world_size()
    if dist.get_backend() != "nccl" and local.is_cuda:
        local = local.cpu()
    bucket = [torch.empty_like(local) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(local, bucket, dst=dst)
    return bucket


def gather_outputs_u8(local: torch.Tensor, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Gather the HR outputs the way they leave the pipeline (``ToPILImage``: uint8 HWC, inference.py:93): each rank converts
    its ``[n,3,H,W]`` fp32 stack on the GPU and 3.1 MB per 1024^2 tile travel instead of 12.6 MB."""
    if local.is_cuda:
        from .inference import unit_tensor_to_u8_on_device
        local = torch.stack([unit_tensor_to_u8_on_device(img) for img in local], 0)
    else:                                            # CPU tests (gloo): same arithmetic, torch ops
        local = local.mul(255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    return gather_outputs(local, dst=dst)


def max_over_ranks(seconds: float, device: torch.device) -> float:
    t = torch.tensor([seconds], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
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


def shard_canvas(sampler, group=None):
    """Make ``sampler.tiled_sample`` split every step's tile list over the ranks of ``group`` (default: WORLD).
    All ranks must call tiled_sample with identical arguments (and, in host-noise mode, identical torch seeds);
    all of them return the full image.  Results are bit-identical to the single-GPU run: tiles are independent
    within a step (model.py:3361-3380) and the noise of a tile depends only on its index in the grid."""
    sampler.canvas_group = group if group is not None else dist.group.WORLD
    return sampler


def _all_gather_tiles(mine: torch.Tensor, group) -> torch.Tensor:
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, mine, group=group)       # one RCCL all-gather, no staging copies
    else:
        dist.all_gather(list(out.unbind(0)), mine, group=group)
    return out


def sharded_step(eng, group, step: int, n_tiles: int, img, cond_canvas, x_start, noise_tiles, noise_canvas,
                 passes: int, kind: int, scale: float, sub_batch: int, seed: int) -> None:
    """One DDPM step of a canvas shared by the ranks of ``group``: my slice of the tiles, then exchange."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sl = tile_slices(n_tiles, world)
    mine, width = sl[rank], len(sl[0])
    eng.sampler_step_tiles(step, mine.start, len(mine), True, img, cond_canvas, x_start, noise_tiles, noise_canvas,
                           passes, kind, scale, sub_batch, seed)
    if world == 1:
        return
    for canvas in (img, x_start):
        if canvas is None:
            continue
        packed = torch.zeros(width, 3, 256, 256, device=img.device, dtype=torch.float32)
        eng.sampler_exchange_tiles(step & 1, mine.start, len(mine), canvas, packed, to_canvas=False)
        everyone = _all_gather_tiles(packed, group).reshape(world * width, 3, 256, 256)
        eng.sampler_exchange_tiles(step & 1, 0, n_tiles, canvas, everyone, to_canvas=True)


def sharded_edm_step(eng, group, step: int, n_tiles: int, img, cond_canvas, x_start, work, noise_canvas, ring_noise_canvas,
                     passes: int, kind: int, scale: float, sub_batch: int, seed: int) -> None:
    """One EDM (Heun) step of a canvas shared by the ranks of ``group`` (reference model.py:2379-2463): my slice of the tiles
    (both network evaluations; the scratch canvases stay rank-local), the odd-step ring on every rank's own canvas, then the
    same tile exchange as the DDPM loop."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sl = tile_slices(n_tiles, world)
    mine, width = sl[rank], len(sl[0])
    eng.edm_step_tiles(step, mine.start, len(mine), True, img, cond_canvas, x_start, work, noise_canvas, ring_noise_canvas,
                       passes, kind, scale, sub_batch, seed)
    if world == 1:
        return
    for canvas in (img, x_start):
        if canvas is None:
            continue
        packed = torch.zeros(width, 3, 256, 256, device=img.device, dtype=torch.float32)
        eng.sampler_exchange_tiles(step & 1, mine.start, len(mine), canvas, packed, to_canvas=False)
        everyone = _all_gather_tiles(packed, group).reshape(world * width, 3, 256, 256)
        eng.sampler_exchange_tiles(step & 1, 0, n_tiles, canvas, everyone, to_canvas=True)
