"""Two concurrent lanes for small steps.

One HR tile of BASELINE configs[1] is 25 (even grid) / 16 (odd grid) U-Net tiles per step.  As ONE launch per layer that is
1.56 "waves" of workgroups on the 32x32 layers and 3.1 on the 64x64 ones: the last, partial wave of every kernel leaves a
fifth of the chip idle, and nothing else is queued behind it because the next layer depends on this one.  Tiles of a step are
independent (disjoint canvas regions; the odd-step ring lies outside all of them - the property the sharded canvas of
``srgd_amd.parallel`` relies on), so the step is run as two halves on two HIP streams through two engines: while one half
drains the tail of a kernel, the other half's kernels fill the idle CUs.  Measured on one MI355X (profiles/r5/step_lanes_ab.txt,
`bench.py --images 1`, same box): one lane 1.231 HR tiles/s, two lanes 1.273 (+3.4 %), three / four lanes 1.184 / 1.183 (the
parts get too small to fill the chip on the shallow layers); configs[4] fp8 with one HR tile 0.4438 -> 0.4583 (+3.3 %).  With
more HR tiles in lock-step the gain shrinks - 50 tiles per step 1.329 -> 1.365 (+2.8 %), 75: 1.351 -> 1.366 (+1.1 %), 100:
1.361 -> 1.366 (+0.4 %), 125: 1.334 vs 1.331 - so steps of more than 100 samples keep one lane.  Results are bit-identical either way
(tests/test_engine_gpu.py::test_two_step_lanes_are_bitwise_identical_to_one).

The second engine is a second instance of the same C-ABI engine (its own scratch, graphs and packed weights); it is created
the first time a step wants two lanes."""
import os
from typing import Callable, Optional

import torch

MAX_SAMPLES_FOR_TWO_LANES = 100     # tiles x guidance passes of one launch (bf16 / fp8 modes)
MAX_LANES = 2                       # three and four lanes measured 4 % SLOWER than one (25 tiles per step)
# Automatic two-lane mode only where it was A/B-measured (profiles/r5/step_lanes_ab.txt: bf16 and fp8; round 6: f16x3).  fp32 (parity mode, 10x the kernel time per launch: the partial last wave is a rounding error there) keeps one lane
# unless forced: a second engine is a second copy of the packed weights, scratch and graphs (memory note).
AUTO_LANE_PRECISIONS = ("bf16", "bf16_w8", "fp8", "fp8_mixed", "f16x3", "f16mx2")
# f16x3 (profiles/r6/images_sweep_and_f16x3_lanes.txt, same box): one HR tile per step 0.3806 -> 0.4011 (+5.4 %), five in lock-step
# (125 tiles per launch) 0.4035 -> 0.4096 (+1.5 %): its 512-thread, one-workgroup-per-CU convolution kernels leave a longer partial
# last wave than the bf16 kernels, so every one-launch step runs as two lanes
MAX_SAMPLES_BY_PRECISION = {"f16x3": 125, "f16mx2": 125}


def lanes_wanted(n_tiles: int, passes: int, sub_batch: int, setting: Optional[int], precision: str = "bf16") -> int:
    """Number of concurrent lanes of a step.  ``setting``: None = automatic (two lanes when the whole step is ONE launch of at
    most 100 samples, in a precision of AUTO_LANE_PRECISIONS), an integer = forced (never more lanes than tiles)."""
    if n_tiles < 2:
        return 1
    if setting is not None:
        return max(1, min(int(setting), MAX_LANES, n_tiles))
    if precision not in AUTO_LANE_PRECISIONS:
        return 1
    one_launch = sub_batch >= n_tiles
    return 2 if (one_launch and n_tiles * passes <= MAX_SAMPLES_BY_PRECISION.get(precision, MAX_SAMPLES_FOR_TWO_LANES)) else 1


def lanes_setting_from_env() -> Optional[int]:
    """SRGD_STEP_LANES: 0 or 1 = one lane (off), 2 = two lanes forced, unset / anything else = automatic."""
    v = os.environ.get("SRGD_STEP_LANES", "").strip()
    if not v.isdigit():
        return None
    return max(1, min(int(v), MAX_LANES))


def lane_slices(n_tiles: int, lanes: int):
    """Contiguous, near-equal parts [(first, count)] of a step's tile list, larger parts first."""
    base, extra = divmod(n_tiles, lanes)
    out, first = [], 0
    for k in range(lanes):
        count = base + (1 if k < extra else 0)
        out.append((first, count))
        first += count
    return out


class StepLanes:
    """Runs ``call(engine, tile_first, tile_count, do_ring)`` for the parts of a step on their own streams and joins them."""

    def __init__(self, engines, device: torch.device):
        self.engines = tuple(engines)
        self.device = device
        self._side = [torch.cuda.Stream(device=device) for _ in self.engines[1:]]

    def run(self, n_tiles: int, call: Callable) -> None:
        parts = lane_slices(n_tiles, len(self.engines))
        main = torch.cuda.current_stream(self.device)
        for k, side in enumerate(self._side, start=1):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                call(self.engines[k], parts[k][0], parts[k][1], False)
        call(self.engines[0], parts[0][0], parts[0][1], True)     # the odd-step ring (outside every tile) rides with the first part
        for side in self._side:
            main.wait_stream(side)
