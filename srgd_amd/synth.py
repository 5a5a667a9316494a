"""Seeded synthetic weights with the reference ``state_dict`` schema.

The published checkpoint is a Git-LFS pointer, so parity tests and
``bench.py`` run on random weights that have exactly the reference key/shape schema
(SURVEY.md Appendix C).  Values depend only on ``(seed, key, shape)`` - not on module
construction order - so fixtures stay valid when host code is refactored.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Mapping, Sequence

import torch


def _gen(seed: int, key: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 63 - 1))
    return g


def synth_tensor(key: str, shape: Sequence[int], seed: int = 0) -> torch.Tensor:
    g = _gen(seed, key)
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "g" or key.endswith(".norm.weight"):            # RMSNorm gain / GroupNorm gamma
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if key.endswith(".norm.bias"):                             # GroupNorm beta
        return 0.1 * torch.randn(shape, generator=g)
    if leaf == "weights" or key.endswith("class_mlp.0.weight"):  # sinusoidal freqs / class embedding
        return torch.randn(shape, generator=g)
    if leaf == "bias":
        return 0.02 * torch.randn(shape, generator=g)
    if leaf == "weight" and len(shape) >= 2:                   # conv OIHW / linear [out,in]
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        bound = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * bound
    raise KeyError(f"no synthetic rule for {key} {shape}")


def synth_state_dict(schema: Mapping[str, Sequence[int]], seed: int = 0) -> Dict[str, torch.Tensor]:
    """schema: key -> shape (any iteration order)."""
    return {k: synth_tensor(k, schema[k], seed) for k in schema}


def synthetic_lr_condition(index: int, lr_h: int, lr_w: int, scale: int = 4, seed_base: int = 1234) -> torch.Tensor:
    """BASELINE.md section 4 synthetic input: seeded uint8 LR image -> PIL bicubic x``scale`` -> /255 float
    [1,3,H,W], i.e. what ``inference.py:71-73`` hands to ``tiled_sample`` (T.Resize on a PIL image is
    ``Image.resize(BICUBIC)``; ToTensor is /255)."""
    import numpy as np
    from PIL import Image
    g = torch.Generator().manual_seed(seed_base + index)
    lr = torch.randint(0, 256, (lr_h, lr_w, 3), dtype=torch.uint8, generator=g).numpy()
    hr = Image.fromarray(lr, "RGB").resize((lr_w * scale, lr_h * scale), Image.BICUBIC)
    arr = np.asarray(hr, dtype=np.uint8)
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255.0).unsqueeze(0).contiguous()
