"""Python handle on the HIP engine (thin: pointers and sizes in, status out).

PyTorch supplies device memory (``tensor.data_ptr()``) and the current HIP stream; every
arithmetic step of the hot path happens inside ``libsrgd_hip.so``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import SamplerGeometry, StepScalars, UnetConfig, check

_PRECISIONS = {"fp32": _lib.PRECISION_FP32, "bf16": _lib.PRECISION_BF16, "bf16_w8": _lib.PRECISION_BF16_W8,
               "fp8": _lib.PRECISION_FP8, "fp8_mixed": _lib.PRECISION_FP8_MIXED, "f16x3": _lib.PRECISION_F16X3, "f16mx2": _lib.PRECISION_F16MX2}


def _stream_ptr(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _dev_ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "engine tensors are contiguous fp32 on the GPU"
    return C.c_void_p(t.data_ptr())


class HipEngine:
    """One engine = one U-Net on one MI355X in one precision mode."""

    def __init__(self, *, dim: int, dim_mults: Sequence[int], full_attn: Sequence[bool], channels: int = 3,
                 groups: int = 8, heads: int = 4, dim_head: int = 32, sinus_dim: int = 32,
                 num_classes: Optional[int] = 3, precision: str = "fp32", device: Optional[torch.device] = None):
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
        if not torch.cuda.is_available():
            raise _lib.SrgdHipError("no MI355X visible: the HIP engine has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.precision = precision
        cfg = UnetConfig()
        cfg.dim = dim
        cfg.n_stages = len(dim_mults)
        for i, m in enumerate(dim_mults):
            cfg.dim_mults[i] = int(m)
            cfg.full_attn[i] = 1 if full_attn[i] else 0
        cfg.channels, cfg.groups, cfg.heads, cfg.dim_head = channels, groups, heads, dim_head
        cfg.sinus_dim = sinus_dim
        cfg.num_classes = int(num_classes or 0)
        cfg.precision = _PRECISIONS[precision]
        cfg.device = self.device.index
        self._L = _lib.lib()
        self._h = C.c_void_p()
        check(self._L.srgd_create(C.byref(cfg), C.byref(self._h)), "srgd_create")
        self.num_classes = cfg.num_classes
        self._keep: List[torch.Tensor] = []

    # ---------------------------------------------------------------- lifetime / weights
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.srgd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def weight_schema(self) -> Dict[str, Tuple[int, ...]]:
        out = {}
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        for i in range(self._L.srgd_num_weights(self._h)):
            check(self._L.srgd_weight_info(self._h, i, name, 256, shape, C.byref(nd)), "srgd_weight_info")
            out[name.value.decode()] = tuple(int(shape[k]) for k in range(nd.value))
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        """sd: U-Net state_dict (keys with or without the sampler's 'model.' prefix)."""
        known = self.weight_schema()
        for key, t in sd.items():
            bare = key[6:] if key.startswith("model.") else key
            if bare not in known:
                if strict:
                    raise _lib.SrgdHipError(f'Unexpected key(s) in state_dict: "{key}"')
                continue
            h = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
            shp = (C.c_int64 * max(1, h.dim()))(*h.shape)
            check(self._L.srgd_load_weight(self._h, bare.encode(), C.c_void_p(h.data_ptr()), shp, h.dim()),
                  "srgd_load_weight")
        check(self._L.srgd_finalize_weights(self._h), "srgd_finalize_weights")

    # ---------------------------------------------------------------- one U-Net evaluation
    def unet_forward(self, x: torch.Tensor, log_snr: torch.Tensor, class_id: int = -1,
                     cond: Optional[torch.Tensor] = None) -> torch.Tensor:
        b, c, h, w = x.shape
        x = x.contiguous().float()
        cond = None if cond is None else cond.contiguous().float()
        ls = log_snr.detach().to("cpu", torch.float32).contiguous()
        assert ls.numel() == b
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            check(self._L.srgd_unet_forward(self._h, _dev_ptr(x), _dev_ptr(cond),
                                            C.cast(C.c_void_p(ls.data_ptr()), C.POINTER(C.c_float)), int(class_id),
                                            _dev_ptr(out), b, h, w, _stream_ptr(self.device)), "srgd_unet_forward")
        return out

    # ---------------------------------------------------------------- tiled sampler
    def sampler_begin(self, geo: SamplerGeometry, cond01: torch.Tensor, cond_canvas: torch.Tensor,
                      tiles_even: Sequence[Tuple[int, int]], tiles_odd: Sequence[Tuple[int, int]],
                      scalars: Sequence[StepScalars], log_snr: Sequence[float], class_id: int) -> None:
        n = len(scalars)
        te = (C.c_int32 * (2 * len(tiles_even)))(*[v for yx in tiles_even for v in yx])
        to = (C.c_int32 * (2 * len(tiles_odd)))(*[v for yx in tiles_odd for v in yx])
        sc = (StepScalars * n)(*scalars)
        ls = (C.c_float * n)(*[float(v) for v in log_snr])
        with torch.cuda.device(self.device):
            check(self._L.srgd_sampler_begin(self._h, C.byref(geo), _dev_ptr(cond01), _dev_ptr(cond_canvas), te, to, n,
                                             sc, ls, int(class_id), _stream_ptr(self.device)), "srgd_sampler_begin")

    def sampler_step(self, step: int, img: torch.Tensor, cond_canvas: torch.Tensor, x_start: Optional[torch.Tensor],
                     noise_tiles: Optional[torch.Tensor], noise_canvas: Optional[torch.Tensor], passes: int,
                     guidance_kind: int, guidance_scale: float, sub_batch: int, seed: int = 0) -> None:
        with torch.cuda.device(self.device):
            check(self._L.srgd_sampler_step(self._h, step, _dev_ptr(img), _dev_ptr(cond_canvas), _dev_ptr(x_start),
                                            _dev_ptr(noise_tiles), _dev_ptr(noise_canvas), passes, guidance_kind,
                                            float(guidance_scale), int(sub_batch), int(seed) & (2 ** 64 - 1),
                                            _stream_ptr(self.device)), "srgd_sampler_step")

    def sampler_step_tiles(self, step: int, tile_first: int, tile_count: int, do_ring: bool, img: torch.Tensor,
                           cond_canvas: torch.Tensor, x_start: Optional[torch.Tensor],
                           noise_tiles: Optional[torch.Tensor], noise_canvas: Optional[torch.Tensor], passes: int,
                           guidance_kind: int, guidance_scale: float, sub_batch: int, seed: int = 0) -> None:
        with torch.cuda.device(self.device):
            check(self._L.srgd_sampler_step_tiles(self._h, step, int(tile_first), int(tile_count), int(bool(do_ring)),
                                                  _dev_ptr(img), _dev_ptr(cond_canvas), _dev_ptr(x_start),
                                                  _dev_ptr(noise_tiles), _dev_ptr(noise_canvas), passes, guidance_kind,
                                                  float(guidance_scale), int(sub_batch), int(seed) & (2 ** 64 - 1),
                                                  _stream_ptr(self.device)), "srgd_sampler_step_tiles")

    def sampler_unpack_gathered(self, parity: int, world: int, slice_w: int, part_off: int, part_w: int, canvas: torch.Tensor,
                                gathered: torch.Tensor) -> None:
        """One launch: rows of an all-gathered buffer [world * part_w, 3, T, T] back into the canvas (srgd_hip.h)."""
        assert gathered.is_contiguous() and gathered.dtype == torch.float32 and gathered.shape[0] == world * part_w
        with torch.cuda.device(self.device):
            check(self._L.srgd_sampler_unpack_gathered(self._h, int(parity), int(world), int(slice_w), int(part_off), int(part_w),
                                                       _dev_ptr(canvas), _dev_ptr(gathered), _stream_ptr(self.device)),
                  "srgd_sampler_unpack_gathered")

    def sampler_exchange_tiles(self, parity: int, tile_first: int, tile_count: int, canvas: torch.Tensor,
                               tiles: torch.Tensor, to_canvas: bool) -> None:
        """Pack (canvas -> tiles) or unpack (tiles -> canvas) tiles [tile_first, tile_first+tile_count) of a grid."""
        assert tiles.is_contiguous() and tiles.dtype == torch.float32 and tiles.numel() >= tile_count * 3 * 256 * 256
        with torch.cuda.device(self.device):
            check(self._L.srgd_sampler_exchange_tiles(self._h, int(parity), int(tile_first), int(tile_count),
                                                      _dev_ptr(canvas), _dev_ptr(tiles), int(bool(to_canvas)),
                                                      _stream_ptr(self.device)), "srgd_sampler_exchange_tiles")

    def edm_begin(self, geo: SamplerGeometry, cond01: torch.Tensor, cond_canvas: torch.Tensor,
                  tiles_even: Sequence[Tuple[int, int]], tiles_odd: Sequence[Tuple[int, int]], scalars, c_noise: Sequence[float],
                  class_id: int) -> None:
        from ._lib import EdmScalars
        n = len(scalars)
        assert len(c_noise) == 2 * n
        te = (C.c_int32 * (2 * len(tiles_even)))(*[v for yx in tiles_even for v in yx])
        to = (C.c_int32 * (2 * len(tiles_odd)))(*[v for yx in tiles_odd for v in yx])
        sc = (EdmScalars * n)(*scalars)
        cn = (C.c_float * (2 * n))(*[float(v) for v in c_noise])
        with torch.cuda.device(self.device):
            check(self._L.srgd_edm_begin(self._h, C.byref(geo), _dev_ptr(cond01), _dev_ptr(cond_canvas), te, to, n, sc, cn,
                                         int(class_id), _stream_ptr(self.device)), "srgd_edm_begin")

    def edm_step(self, step: int, img: torch.Tensor, cond_canvas: torch.Tensor, x_start: Optional[torch.Tensor],
                 work: torch.Tensor, noise_canvas: Optional[torch.Tensor], ring_noise_canvas: Optional[torch.Tensor],
                 passes: int, guidance_kind: int, guidance_scale: float, sub_batch: int, seed: int = 0) -> None:
        assert work.is_contiguous() and work.dtype == torch.float32 and work.numel() >= 2 * img.numel()
        with torch.cuda.device(self.device):
            check(self._L.srgd_edm_step(self._h, step, _dev_ptr(img), _dev_ptr(cond_canvas), _dev_ptr(x_start), _dev_ptr(work),
                                        _dev_ptr(noise_canvas), _dev_ptr(ring_noise_canvas), passes, guidance_kind,
                                        float(guidance_scale), int(sub_batch), int(seed) & (2 ** 64 - 1),
                                        _stream_ptr(self.device)), "srgd_edm_step")

    def edm_step_tiles(self, step: int, tile_first: int, tile_count: int, do_ring: bool, img: torch.Tensor,
                       cond_canvas: torch.Tensor, x_start: Optional[torch.Tensor], work: torch.Tensor,
                       noise_canvas: Optional[torch.Tensor], ring_noise_canvas: Optional[torch.Tensor], passes: int,
                       guidance_kind: int, guidance_scale: float, sub_batch: int, seed: int = 0) -> None:
        assert work.is_contiguous() and work.dtype == torch.float32 and work.numel() >= 2 * img.numel()
        with torch.cuda.device(self.device):
            check(self._L.srgd_edm_step_tiles(self._h, step, int(tile_first), int(tile_count), int(bool(do_ring)), _dev_ptr(img),
                                              _dev_ptr(cond_canvas), _dev_ptr(x_start), _dev_ptr(work), _dev_ptr(noise_canvas),
                                              _dev_ptr(ring_noise_canvas), passes, guidance_kind, float(guidance_scale),
                                              int(sub_batch), int(seed) & (2 ** 64 - 1), _stream_ptr(self.device)),
                  "srgd_edm_step_tiles")

    def edm_dpmpp_step(self, step: int, img: torch.Tensor, cond_canvas: torch.Tensor, x_start: Optional[torch.Tensor],
                       old_denoised: torch.Tensor, passes: int, guidance_kind: int, guidance_scale: float,
                       sub_batch: int) -> None:
        assert old_denoised.is_contiguous() and old_denoised.dtype == torch.float32 and old_denoised.numel() == img.numel()
        with torch.cuda.device(self.device):
            check(self._L.srgd_edm_dpmpp_step(self._h, step, _dev_ptr(img), _dev_ptr(cond_canvas), _dev_ptr(x_start),
                                              _dev_ptr(old_denoised), passes, guidance_kind, float(guidance_scale),
                                              int(sub_batch), _stream_ptr(self.device)), "srgd_edm_dpmpp_step")

    def sampler_q_start(self, cond01: torch.Tensor, noise_canvas: Optional[torch.Tensor], alpha: float, sigma: float,
                        img: torch.Tensor, seed: int = 0) -> None:
        with torch.cuda.device(self.device):
            check(self._L.srgd_sampler_q_start(self._h, _dev_ptr(cond01), _dev_ptr(noise_canvas), float(alpha), float(sigma),
                                               _dev_ptr(img), int(seed) & (2 ** 64 - 1), _stream_ptr(self.device)),
                  "srgd_sampler_q_start")

    def sampler_end(self, img: torch.Tensor, out01: torch.Tensor) -> None:
        with torch.cuda.device(self.device):
            check(self._L.srgd_sampler_end(self._h, _dev_ptr(img), _dev_ptr(out01), _stream_ptr(self.device)),
                  "srgd_sampler_end")

    def randn_(self, dst: torch.Tensor, seed: int, stream_id: int) -> torch.Tensor:
        with torch.cuda.device(self.device):
            check(self._L.srgd_randn(self._h, _dev_ptr(dst), dst.numel(), int(seed) & (2 ** 64 - 1), int(stream_id),
                                     _stream_ptr(self.device)), "srgd_randn")
        return dst

    # ---------------------------------------------------------------- measurement
    def profile_begin(self) -> None:
        check(self._L.srgd_profile_begin(self._h), "srgd_profile_begin")

    def profile_end(self) -> Dict[str, object]:
        n = self._L.srgd_profile_num_families()
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        fl = (C.c_double * n)()
        check(self._L.srgd_profile_end(self._h, ms, cnt, fl, n), "srgd_profile_end")
        by = (C.c_double * n)()
        check(self._L.srgd_profile_bytes(self._h, by, n), "srgd_profile_bytes")
        names = [self._L.srgd_profile_family_name(i).decode() for i in range(n)]
        return {"ms": {names[i]: ms[i] for i in range(n)}, "launches": {names[i]: int(cnt[i]) for i in range(n)},
                "flops": {names[i]: fl[i] for i in range(n)}, "bytes": {names[i]: by[i] for i in range(n)}}

    def bytes_in_use(self) -> int:
        return int(self._L.srgd_device_bytes_in_use(self._h))
