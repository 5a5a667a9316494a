"""Ranks as THREADS of one process: a stand-in for `srgd_amd.parallel.TorchComm` with the same interface.

Why it exists: a one-GPU box admits at most six processes on its card (the pool's process guard kills the run otherwise), and the
8-rank partitionings of BASELINE configs[2] / configs[3] have to be exercised there.  Eight threads of the pytest process each
own a sampler + engine (the C-ABI library is thread-safe per engine; hipGraph capture is thread-local) and meet in these
collectives, which copy tensors between the ranks' buffers after a `threading.Barrier` - what an all-gather / gather /
broadcast / all-reduce does, minus the transport.  The transport itself (RCCL) is exercised at world size 1 by
`tests/test_bench_multirank_gpu.py` and at N > 1 by the driver's multi-GPU runs; the property tested through this class is the
PARTITIONING: slices, gather order, buffer reuse, bit-identity with the single-rank run.  Test infrastructure only."""
import threading
from typing import List, Optional

import torch


class ThreadWorld:
    def __init__(self, world: int, timeout: float = 600.0):
        self.world = world
        self.barrier = threading.Barrier(world, timeout=timeout)
        self.slots: List[Optional[torch.Tensor]] = [None] * world
        self.errors: List[BaseException] = []

    def comm(self, rank: int) -> "ThreadComm":
        return ThreadComm(self, rank)

    def run(self, target, *args) -> list:
        """target(comm, *args) on `world` threads; returns the per-rank results; re-raises the first failure."""
        out = [None] * self.world

        def body(r):
            try:
                out[r] = target(self.comm(r), *args)
            except BaseException as e:          # noqa: BLE001 - reported to the test below
                self.errors.append(e)
                self.barrier.abort()            # wake the ranks waiting for this one

        threads = [threading.Thread(target=body, args=(r,), name=f"rank{r}") for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        real = [e for e in self.errors if not isinstance(e, threading.BrokenBarrierError)]
        if real or self.errors:
            raise (real or self.errors)[0]
        return out


class ThreadComm:
    backend = "threads"

    def __init__(self, w: ThreadWorld, rank: int):
        self.w, self.rank, self.world = w, rank, w.world

    def _publish(self, t: Optional[torch.Tensor]) -> List[Optional[torch.Tensor]]:
        if t is not None and t.is_cuda:
            torch.cuda.current_stream().synchronize()       # the producer's kernels are done before another rank reads
        self.w.slots[self.rank] = t
        self.w.barrier.wait()
        return list(self.w.slots)

    def _retire(self) -> None:
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()       # my reads of the others' buffers are done before they move on
        self.w.barrier.wait()

    def all_gather_tiles(self, out: torch.Tensor, mine: torch.Tensor) -> None:
        parts = self._publish(mine)
        n = mine.shape[0]
        for r, p in enumerate(parts):
            out[r * n:(r + 1) * n].copy_(p)
        self._retire()

    def gather(self, local: torch.Tensor, dst: int = 0):
        parts = self._publish(local)
        bucket = [p.clone() for p in parts] if self.rank == dst else None
        self._retire()
        return bucket

    def broadcast(self, flat: torch.Tensor, src: int = 0) -> None:
        parts = self._publish(flat)
        if self.rank != src:
            flat.copy_(parts[src])
        self._retire()

    def all_reduce_max(self, t: torch.Tensor) -> None:
        parts = self._publish(t.clone())
        t.copy_(torch.stack(parts, 0).max(0).values)
        self._retire()

    def barrier(self) -> None:
        self.w.barrier.wait()
