"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on
ROCm; ``gloo`` in the CPU tests).  The path shards by independent images: there
is no data-path collective - only a one-time weight broadcast and a final gather of the HR outputs.
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
    bucket = [torch.empty_like(local) for _ in range(world)] if dist.get_rank() == dst else None
    dist.gather(local, bucket, dst=dst)
    return bucket


def max_over_ranks(seconds: float, device: torch.device) -> float:
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
