"""Host-side mirror of the reference's ``model.py`` surface for the ONE shipped sampling path
(``model: conditional_continuous``), backed by the MI355X engine (``libsrgd_hip.so``).

Kept identical to the reference so that callers and checkpoints switch over unchanged:
  * ``get_model(conf, logger)`` (reference model.py:3500-3666) -> object with ``.module``
  * ``ConditionalSRUnet(...)`` constructor signature (model.py:537-556) and ``state_dict`` keys/shapes
     - the published ``.pth`` loads with ``strict=True``
  * ``ConditionalContinuousTimeGaussianDiffusionSR.tiled_sample(...)`` signature, return value and
    error behaviour (model.py:3288-3413)
  * the pure-int tiling helpers ``get_coord_and_pad / get_coords / get_area`` (model.py:116-179)

The modules below hold parameters only; they have no PyTorch implementation of the arithmetic.
All compute runs in hand-written gfx950 kernels; calling this path without a GPU raises.
"""
from __future__ import annotations

import copy
import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from ._lib import SamplerGeometry, StepScalars
from .engine import HipEngine
from .lanes import StepLanes, lanes_setting_from_env, lanes_wanted

__all__ = ["get_coord_and_pad", "get_coords", "get_area", "beta_linear_log_snr", "ConditionalSRUnet",
           "ConditionalContinuousTimeGaussianDiffusionSR", "ConditionalElucidatedDiffusionSR", "ModelEma", "get_model"]


# ---------------------------------------------------------------------------------------------
# tiling geometry (pure ints; reference model.py:116-179)
# ---------------------------------------------------------------------------------------------
def _host_randn(generator, *shape):
    """Host-noise draw: ``torch.randn`` on the CPU generator the reference draws from (global unless one is given)."""
    return torch.randn(*shape, generator=generator)


def get_coord_and_pad(height: int, width: int, tile_size: int = 256):
    """Canvas size and placement of an ``height x width`` image: one tile if it fits, otherwise the
    size rounded up to whole tiles plus one extra tile (half a tile of margin per side)."""
    fits = height <= tile_size and width <= tile_size
    canvas_h = tile_size if fits else tile_size * (math.ceil(height / tile_size) + 1)
    canvas_w = tile_size if fits else tile_size * (math.ceil(width / tile_size) + 1)
    left, top = (canvas_w - width) // 2, (canvas_h - height) // 2
    coord = (left, top, left + width, top + height)
    pad = (left, canvas_w - width - left, top, canvas_h - height - top)
    return coord, pad


def _axis_starts(extent: int, tile_size: int, tile_stride: int) -> List[int]:
    starts = list(range(0, extent - tile_size + 1, tile_stride))
    if (extent - tile_size) % tile_stride:
        starts.append(extent - tile_size)      # last tile is pulled back to end at the border
    return starts


def get_coords(h: int, w: int, tile_size: int, tile_stride: int, diff: int = 0):
    """Row-major list of tile boxes ``(hs, he, ws, we)``, each shifted by ``diff``."""
    return [(y + diff, y + diff + tile_size, x + diff, x + diff + tile_size)
            for y in _axis_starts(h, tile_size, tile_stride) for x in _axis_starts(w, tile_size, tile_stride)]


def get_area(coords, height: int, width: int):
    """Bounding box ``(left, top, right, bottom)`` of a tile list and its margins inside the canvas."""
    top = min([height] + [c[0] for c in coords])
    bottom = max([0] + [c[1] for c in coords])
    left = min([width] + [c[2] for c in coords])
    right = max([0] + [c[3] for c in coords])
    return (left, top, right, bottom), (left, width - right, top, height - bottom)


# ---------------------------------------------------------------------------------------------
# schedule scalars: host-side fp32 torch ops in the reference's own order, so the numbers handed
# to the kernels are bit-identical to what the reference computes (model.py:2629-2633, :3127-3134)
# ---------------------------------------------------------------------------------------------
def beta_linear_log_snr(t: torch.Tensor) -> torch.Tensor:
    return -torch.log(torch.special.expm1(1e-4 + 10 * (t ** 2)).clamp(min=1e-20))


def _schedule(num_steps: int) -> Tuple[List[StepScalars], List[float]]:
    times = torch.linspace(1.0, 0.0, num_steps + 1)
    scalars, log_snrs = [], []
    for i in range(num_steps):
        ls, ls_next = beta_linear_log_snr(times[i]), beta_linear_log_snr(times[i + 1])
        c = -torch.special.expm1(ls - ls_next)
        alpha, sigma = ls.sigmoid().sqrt(), (-ls).sigmoid().sqrt()
        alpha_next = ls_next.sigmoid().sqrt()
        var = (-ls_next).sigmoid() * c
        s = StepScalars()
        s.alpha, s.sigma, s.alpha_next, s.c = float(alpha), float(sigma), float(alpha_next), float(c)
        s.one_minus_c = float(1 - c)
        s.noise_scale = float(var.sqrt())
        s.sigma_next = float((-ls_next).sigmoid().sqrt())
        scalars.append(s)
        log_snrs.append(float(ls))
    return scalars, log_snrs


# ---------------------------------------------------------------------------------------------
# parameter containers (names/shapes = the reference state_dict; no forward)
# ---------------------------------------------------------------------------------------------
class _Params(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError("parameter container: the arithmetic of this layer lives in libsrgd_hip.so")


class _Gain(_Params):                                   # RMSNorm.g [1,C,1,1]
    def __init__(self, c):
        super().__init__()
        self.g = nn.Parameter(torch.ones(1, c, 1, 1))


class _FourierFreqs(_Params):                           # RandomOrLearnedSinusoidalPosEmb.weights [half]
    def __init__(self, dim, frozen):
        super().__init__()
        self.weights = nn.Parameter(torch.randn(dim // 2), requires_grad=not frozen)


class _ConvNormAct(_Params):                            # Block: proj + norm
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.proj = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm = nn.GroupNorm(groups, cout)


class _Residual(_Params):                               # ResnetBlock
    def __init__(self, cin, cout, time_dim, groups):
        super().__init__()
        self.mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_dim, cout * 2))
        self.block1 = _ConvNormAct(cin, cout, groups)
        self.block2 = _ConvNormAct(cout, cout, groups)
        self.res_conv = nn.Conv2d(cin, cout, 1) if cin != cout else nn.Identity()


class _LinearAttn(_Params):
    def __init__(self, c, heads, dim_head):
        super().__init__()
        self.norm = _Gain(c)
        self.to_qkv = nn.Conv2d(c, heads * dim_head * 3, 1, bias=False)
        self.to_out = nn.Sequential(nn.Conv2d(heads * dim_head, c, 1), _Gain(c))


class _SoftmaxAttn(_Params):
    def __init__(self, c, heads, dim_head):
        super().__init__()
        self.norm = _Gain(c)
        self.to_qkv = nn.Conv2d(c, heads * dim_head * 3, 1, bias=False)
        self.to_out = nn.Conv2d(heads * dim_head, c, 1)


class _ShuffleUp(_Params):                              # PixelShuffleUpsample: conv1x1 -> SiLU -> PixelShuffle(2)
    def __init__(self, cin, cout):
        super().__init__()
        conv = nn.Conv2d(cin, cout * 4, 1)
        # ICNR-style start: the four sub-pixel filters of each output channel begin identical
        base = torch.empty(cout, cin, 1, 1)
        nn.init.kaiming_uniform_(base)
        with torch.no_grad():
            conv.weight.copy_(base.repeat_interleave(4, dim=0))
            conv.bias.zero_()
        self.net = nn.Sequential(conv, nn.SiLU(), nn.PixelShuffle(2))


def _single_class_id(class_label) -> int:
    """The samplers condition every tile on ONE class (the reference passes a ``[1]`` label that broadcasts over the
    tile batch, model.py:694).  A ``[B]`` label with differing entries (legal for the reference's un-tiled ``sample``)
    is refused rather than silently collapsed to its first element."""
    if class_label is None:
        return -1
    flat = class_label.reshape(-1)
    if flat.numel() > 1 and not bool((flat == flat[0]).all()):
        raise NotImplementedError("per-image class labels: this engine conditions one run on one class "
                                  "(pass a [1] label, or equal labels)")
    return int(flat[0])


def _as_tuple(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


class ConditionalSRUnet(nn.Module):
    """Class-conditional SR U-Net (6-channel input: noisy | LR condition; 3-channel eps output)."""

    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=3,
                 self_condition=True, resnet_block_groups=8, learned_variance=False,
                 learned_sinusoidal_cond=False, random_fourier_features=False, learned_sinusoidal_dim=16,
                 attn_dim_head=32, attn_heads=4, full_attn=(False, False, False, True), flash_attn=False,
                 pixel_shuffle_upsample=True, num_classes=None):
        super().__init__()
        unsupported = []
        if init_dim not in (None, dim): unsupported.append("init_dim != dim")
        if out_dim not in (None, channels): unsupported.append("out_dim != channels")
        if not self_condition: unsupported.append("self_condition=False")
        if learned_variance: unsupported.append("learned_variance=True")
        if not (learned_sinusoidal_cond or random_fourier_features): unsupported.append("fixed sinusoidal time embedding")
        if not pixel_shuffle_upsample: unsupported.append("pixel_shuffle_upsample=False")
        if unsupported:
            raise NotImplementedError("outside the shipped Real-SRGD configuration: " + ", ".join(unsupported))
        n = len(dim_mults)
        heads, dim_heads, full = _as_tuple(attn_heads, n), _as_tuple(attn_dim_head, n), _as_tuple(full_attn, n)
        assert len(full) == n
        if len(set(heads)) != 1 or len(set(dim_heads)) != 1:
            raise NotImplementedError("per-stage attn_heads / attn_dim_head")
        self.dim, self.dim_mults, self.channels = dim, tuple(dim_mults), channels
        self.self_condition = self_condition
        self.num_classes = num_classes
        self.groups, self.heads, self.dim_head = resnet_block_groups, heads[0], dim_heads[0]
        self.full_attn = tuple(bool(f) for f in full)
        self.sinus_dim = learned_sinusoidal_dim
        self.random_or_learned_sinusoidal_cond = True
        self.out_dim = channels
        self.flash_attn = flash_attn          # accepted; attention is a HIP kernel either way
        self.precision = "fp32"               # precision of forward(); the sampler picks its own

        time_dim = dim * 4
        dims = [dim] + [dim * m for m in dim_mults]
        self.init_conv = nn.Conv2d(channels * 2, dim, 7, padding=3)
        self.time_mlp = nn.Sequential(_FourierFreqs(learned_sinusoidal_dim, random_fourier_features),
                                      nn.Linear(learned_sinusoidal_dim + 1, time_dim), nn.GELU(),
                                      nn.Linear(time_dim, time_dim))
        if num_classes is not None:
            self.class_mlp = nn.Sequential(nn.Embedding(num_classes, dim), nn.Linear(dim, time_dim), nn.GELU(),
                                           nn.Linear(time_dim, time_dim))
        mk_res = lambda i, o: _Residual(i, o, time_dim, resnet_block_groups)
        mk_attn = lambda c, f: (_SoftmaxAttn if f else _LinearAttn)(c, self.heads, self.dim_head)
        self.downs, self.ups = nn.ModuleList(), nn.ModuleList()
        for s in range(n):
            cin, cout = dims[s], dims[s + 1]
            down = (nn.Sequential(nn.Identity(), nn.Conv2d(cin * 4, cout, 1)) if s < n - 1     # space-to-depth + 1x1
                    else nn.Conv2d(cin, cout, 3, padding=1))
            self.downs.append(nn.ModuleList([mk_res(cin, cin), mk_res(cin, cin), mk_attn(cin, self.full_attn[s]), down]))
        self.mid_block1 = mk_res(dims[-1], dims[-1])
        self.mid_attn = _SoftmaxAttn(dims[-1], self.heads, self.dim_head)
        self.mid_block2 = mk_res(dims[-1], dims[-1])
        for u in range(n):
            s = n - 1 - u
            cin, cout = dims[s], dims[s + 1]
            up = _ShuffleUp(cout, cin) if u < n - 1 else nn.Conv2d(cout, cin, 3, padding=1)
            self.ups.append(nn.ModuleList([mk_res(cout + cin, cout), mk_res(cout + cin, cout),
                                           mk_attn(cout, self.full_attn[s]), up]))
        self.final_res_block = mk_res(dim * 2, dim)
        self.final_conv = nn.Conv2d(dim, channels, 1)

        self._engines = {}
        self._weights_version = 0
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._invalidate_engines())

    # ---- engine plumbing ---------------------------------------------------------------------
    @property
    def downsample_factor(self) -> int:
        return 2 ** (len(self.dim_mults) - 1)

    def _invalidate_engines(self):
        for eng in self._engines.values():
            eng.close()
        self._engines = {}
        self._weights_version += 1

    def _apply(self, fn, *a, **k):                  # .to()/.cuda()/.float() may move or change the parameters
        out = super()._apply(fn, *a, **k)
        self._invalidate_engines()
        return out

    def __deepcopy__(self, memo):
        engines, self._engines = self._engines, {}
        try:
            cls = self.__class__
            new = cls.__new__(cls)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                new.__dict__[k] = copy.deepcopy(v, memo)
        finally:
            self._engines = engines
        new._engines = {}
        return new

    def engine(self, precision: str = "fp32", lane: int = 0) -> HipEngine:
        """The C-ABI engine of this U-Net for one precision; ``lane`` > 0: a further instance (srgd_amd.lanes)."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise _lib.SrgdHipError("ConditionalSRUnet runs on MI355X only: move the model to the GPU "
                                    "(the CPU restatement under oracle/ is test infrastructure, not a fallback)")
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), precision) + ((lane,) if lane else ())
        if key not in self._engines:
            eng = HipEngine(dim=self.dim, dim_mults=self.dim_mults, full_attn=self.full_attn, channels=self.channels,
                            groups=self.groups, heads=self.heads, dim_head=self.dim_head, sinus_dim=self.sinus_dim,
                            num_classes=self.num_classes, precision=precision, device=torch.device("cuda", key[0]))
            eng.load_state_dict({k: v for k, v in self.state_dict().items()}, strict=True)
            self._engines[key] = eng
        return self._engines[key]

    # ---- reference forward signature (model.py:678) ---------------------------------------------
    def forward(self, x, time, class_label=None, x_self_cond=None):
        f = self.downsample_factor
        assert all(d % f == 0 for d in x.shape[-2:]), \
            f"your input dimensions {x.shape[-2:]} need to be divisible by {f}, given the unet"
        class_id = -1
        if class_label is not None:
            if class_label.numel() != 1:
                raise NotImplementedError("per-sample class labels (the reference passes one label of shape [1])")
            class_id = int(class_label.reshape(-1)[0])
        time = time.reshape(-1)
        if time.numel() == 1 and x.shape[0] > 1:
            time = time.expand(x.shape[0])
        return self.engine(self.precision).unet_forward(x, time, class_id, x_self_cond)


# ---------------------------------------------------------------------------------------------
# the sampler
# ---------------------------------------------------------------------------------------------
class ConditionalContinuousTimeGaussianDiffusionSR(nn.Module):
    def __init__(self, model, *, image_size, channels=3, noise_schedule="linear", num_sample_steps=500,
                 clip_sample_denoised=True, learned_schedule_net_hidden_dim=1024,
                 learned_noise_schedule_frac_gradient=1.0, min_snr_loss_weight=False, min_snr_gamma=5,
                 cond_drop_prob=0.0, class_cond_drop_prob=0.0, loss_type="l2"):
        super().__init__()
        assert model.random_or_learned_sinusoidal_cond
        if noise_schedule != "linear":
            raise NotImplementedError(f"noise_schedule={noise_schedule!r}: only the shipped 'linear' log-SNR schedule is built")
        if not clip_sample_denoised:
            raise NotImplementedError("clip_sample_denoised=False")
        self.model = model
        self.channels, self.image_size = channels, image_size
        self.log_snr = beta_linear_log_snr
        self.num_sample_steps = num_sample_steps
        self.clip_sample_denoised = clip_sample_denoised
        self.min_snr_loss_weight, self.min_snr_gamma = min_snr_loss_weight, min_snr_gamma
        self.cond_drop_prob, self.class_cond_drop_prob = cond_drop_prob, class_cond_drop_prob
        self.loss_type = loss_type
        # engine knobs (not part of the reference surface)
        self.noise_source = "host"     # "host": torch CPU generator in the reference's draw order; "device": Philox
        self.host_generator = None     # host-noise draws: None = torch's global CPU generator (what seed_everything seeds, as the
                                       # reference); a torch.Generator makes a run independent of other threads' draws
        self.device_noise_seed = 0
        self.max_tiles_per_launch = None   # None: use the caller's batch_size as the reference does
        self.step_lanes = lanes_setting_from_env()   # None: automatic - small steps run as two concurrent halves (srgd_amd.lanes)
        # engine precision: "fp32" (default: the reference's numerics - upstream ignores ``amp`` and always computes fp32,
        # SURVEY App. E), "bf16" (throughput mode), "bf16_w8" (bf16 kernels, fp8-e4m3-rounded conv weights),
        # "fp8" (MX-fp8 3x3 convolutions, BASELINE configs[4]), "fp8_mixed" (fp8 below the top resolution only: 53 dB vs
        # bf16 instead of 34 dB).  tiled_sample(precision=...) overrides it per call.
        self.precision = "fp32"
        # set by srgd_amd.parallel.shard_canvas: a torch.distributed group whose ranks share ONE canvas - each rank
        # runs a contiguous slice of every step's tiles and the updated tiles are all-gathered
        self.canvas_group = None

    def set_seed(self, seed):
        torch.cuda.manual_seed(seed)
        self.device_noise_seed = int(seed)

    @property
    def device(self):
        return next(self.model.parameters()).device

    @torch.inference_mode()
    def tiled_sample(self, batch_size=4, tile_size=256, tile_stride=256, condition_x=None, class_label=None,
                     cond_scale=1.0, guidance_start_steps=0, class_cond_scale=1.0, class_guidance_start_steps=0,
                     generation_start_steps=0, num_sample_steps=None, with_images=False, with_x0_images=False,
                     start_white_noise=True, amp=False, precision=None):
        """Tiled CFG-DDPM sampling (reference model.py:3288-3413).

        ``amp`` is accepted and ignored exactly as in the reference (which always computes fp32); the engine
        precision is ``precision`` (engine-only keyword) or, if None, ``self.precision`` - "fp32" by default, i.e.
        the reference's numerics.  ``condition_x`` is ``[1,3,H,W]`` as in the reference, or
        ``[B,3,H,W]``: B same-sized images sampled in lock-step, each exactly as the reference would
        sample it on its own after ``seed_everything(seed)`` (inference.py:73) - i.e. all B see the
        same noise stream - with every U-Net launch spanning tiles of all images (fills the GPU
        better than 25/16 tiles of one image do).  ``batch_size`` counts tiles across all images."""
        num_sample_steps = self.num_sample_steps if num_sample_steps is None else num_sample_steps
        if cond_scale != 1.0 and class_cond_scale != 1.0:
            raise NotImplementedError("Currently, you cannot specify both cond_scale and class_cond_scale at the same time.")
        if tile_size != 256 or tile_stride != 256:
            raise NotImplementedError("tile_size/tile_stride other than 256 are unusable in the reference too "
                                      "(get_coord_and_pad is called without them, model.py:3301)")
        dev = self.device
        if dev.type != "cuda":
            raise _lib.SrgdHipError("tiled_sample runs on MI355X only (no CPU fallback)")
        batch, c, h, w = condition_x.shape
        if batch < 1 or c != 3:
            raise ValueError("condition_x must be [B,3,H,W] (B=1 in the reference, whose tile gather assumes batch 1)")
        f = self.model.downsample_factor
        assert tile_size % f == 0, f"your input dimensions need to be divisible by {f}, given the unet"
        eng = self.model.engine(precision or self.precision)
        class_id = _single_class_id(class_label)

        (left, top, right, bottom), pad = get_coord_and_pad(h, w)
        hp, wp = h + pad[2] + pad[3], w + pad[0] + pad[1]
        if max(pad[0], pad[1]) >= w or max(pad[2], pad[3]) >= h:
            raise RuntimeError("Padding size should be less than the corresponding input dimension "
                               f"(reflect pad {pad} of a {h}x{w} image)")
        coords0 = get_coords(hp, wp, tile_size, tile_size, diff=0)
        if hp <= tile_size and wp <= tile_size:
            coords1 = get_coords(hp, wp, tile_size, tile_stride, diff=0)
        else:
            coords1 = get_coords(hp - tile_size, wp - tile_size, tile_size, tile_stride, diff=tile_size // 2)
        (sl, st_, sr, sb), _ = get_area(coords1, hp, wp)
        geo = SamplerGeometry(H=h, W=w, Hp=hp, Wp=wp, left=left, top=top, inner_l=sl, inner_t=st_, inner_r=sr,
                              inner_b=sb, tile=tile_size, n_even=len(coords0), n_odd=len(coords1), n_images=batch)
        scalars, log_snrs = _schedule(num_sample_steps)

        cond01 = condition_x.to(dev, torch.float32).contiguous()
        cond_canvas = torch.empty(batch, 3, hp, wp, device=dev, dtype=torch.float32)
        eng.sampler_begin(geo, cond01, cond_canvas, [(a, c_) for (a, _, c_, _) in coords0],
                          [(a, c_) for (a, _, c_, _) in coords1], scalars, log_snrs, class_id)

        host_noise = self.noise_source == "host"
        if generation_start_steps > 0 or not start_white_noise:
            # start from the forward-diffused condition (model.py:3305-3308 / :3312-3315); same single draw
            t0 = (1.0 - torch.tensor(generation_start_steps / num_sample_steps)) if generation_start_steps > 0 \
                else torch.tensor(1.0)
            ls0 = beta_linear_log_snr(t0)
            img = torch.empty(batch, 3, hp, wp, device=dev)
            eng.sampler_q_start(cond01, _host_randn(self.host_generator, 1, 3, hp, wp).to(dev) if host_noise else None,
                                float(ls0.sigmoid().sqrt()), float((-ls0).sigmoid().sqrt()), img, self.device_noise_seed)
        elif host_noise:
            img = _host_randn(self.host_generator, 1, 3, hp, wp).to(dev).repeat(batch, 1, 1, 1)   # reference draw #1 (model.py:3311)
        else:
            img = eng.randn_(torch.empty(1, 3, hp, wp, device=dev), self.device_noise_seed, 0).repeat(batch, 1, 1, 1)
        x_start = img.clone() if with_x0_images else None
        image_list = [img[:, :, top:bottom, left:right].clone().cpu()] if with_images else None
        x0_image_list = [img[:, :, top:bottom, left:right].clone().cpu()] if with_x0_images else None

        sub_batch = self.max_tiles_per_launch or batch_size
        grids = (coords0, coords1)
        lanes = None
        for i in range(num_sample_steps):
            if i < generation_start_steps:
                continue
            cur_cond_scale = 1.0 if i < guidance_start_steps else cond_scale
            cur_class_scale = 1.0 if i < class_guidance_start_steps else class_cond_scale
            if cur_cond_scale != 1.0:
                passes, kind, scale = 2, 2, cur_cond_scale
            elif cur_class_scale != 1.0:
                passes, kind, scale = 2, 1, cur_class_scale
            else:
                passes, kind, scale = 1, 0, 1.0
            n_tiles = len(grids[i % 2])
            last = i == num_sample_steps - 1
            noise_tiles = noise_canvas = None
            if host_noise:
                # identical to the reference's per-minibatch randn_like draws (SURVEY Appendix D:
                # 16-element block property makes one contiguous draw equal to the minibatch draws)
                if not last:
                    noise_tiles = _host_randn(self.host_generator, n_tiles, 3, tile_size, tile_size).to(dev, non_blocking=True)
                if i % 2 == 1:
                    noise_canvas = _host_randn(self.host_generator, 1, 3, hp, wp).to(dev, non_blocking=True)
            n_step = n_tiles * batch
            n_lanes = lanes_wanted(n_step, passes, sub_batch, self.step_lanes, precision or self.precision) if self.canvas_group is None else 1
            if n_lanes > 1:
                if lanes is None or len(lanes.engines) != n_lanes:   # further engines: same run geometry, own copies of the condition canvas
                    more = [self.model.engine(precision or self.precision, lane=k) for k in range(1, n_lanes)]
                    for e_ in more:
                        e_.sampler_begin(geo, cond01, torch.empty_like(cond_canvas), [(a, c_) for (a, _, c_, _) in coords0],
                                         [(a, c_) for (a, _, c_, _) in coords1], scalars, log_snrs, class_id)
                    lanes = StepLanes([eng] + more, dev)
                lanes.run(n_step, lambda e_, first, count, ring: e_.sampler_step_tiles(
                    i, first, count, ring, img, cond_canvas, x_start, noise_tiles, noise_canvas, passes, kind, scale, sub_batch,
                    seed=self.device_noise_seed))
            elif self.canvas_group is None:
                eng.sampler_step(i, img, cond_canvas, x_start, noise_tiles, noise_canvas, passes, kind, scale, sub_batch,
                                 seed=self.device_noise_seed)
            else:
                from .parallel import sharded_step
                sharded_step(eng, self.canvas_group, i, n_tiles * batch, img, cond_canvas, x_start, noise_tiles,
                             noise_canvas, passes, kind, scale, sub_batch, self.device_noise_seed)
            if with_images:
                image_list.append(img.clone().cpu())
            if with_x0_images:
                x0_image_list.append(x_start.clone().cpu())

        out = torch.empty(batch, 3, h, w, device=dev, dtype=torch.float32)
        eng.sampler_end(img, out)
        if with_images:
            return (out, image_list, x0_image_list) if with_x0_images else (out, image_list)
        return out

    @torch.inference_mode()
    def sample(self, batch_size=16, condition_x=None, class_label=None, cond_scale=1.0, guidance_start_steps=0,
               class_cond_scale=1.0, class_guidance_start_steps=0, generation_start_steps=0, num_sample_steps=None,
               with_images=False, with_x0_images=False, x0=None, amp=False, precision=None):
        """Un-tiled sampling of a batch of ``image_size`` x ``image_size`` images (reference model.py:3417-3432,
        p_sample_loop :3191-3247): every image has its own noise (one ``randn`` over ``[B,3,S,S]`` per draw, host mode).

        Runs on the tiled machinery: the batch is laid out as one canvas of B stacked tiles (``[3, B*S, S]``, no padding, no
        ring re-noise), so one U-Net launch covers the whole batch.  Needs ``image_size == 256`` (the tile edge of the kernels;
        the shipped config).  ``amp`` / ``precision`` as in ``tiled_sample``."""
        num_sample_steps = self.num_sample_steps if num_sample_steps is None else num_sample_steps
        if cond_scale != 1.0 and class_cond_scale != 1.0:
            raise NotImplementedError("Currently, you cannot specify both cond_scale and class_cond_scale at the same time.")
        s_ = self.image_size
        if s_ != 256:
            raise NotImplementedError(f"sample(): image_size={s_}; this engine's un-tiled path needs image_size == 256")
        dev = self.device
        if dev.type != "cuda":
            raise _lib.SrgdHipError("sample runs on MI355X only (no CPU fallback)")
        b = int(batch_size)
        if tuple(condition_x.shape) != (b, self.channels, s_, s_):
            raise ValueError(f"condition_x must be [{b},{self.channels},{s_},{s_}] (model.py:3426 pairs it with the noise batch)")
        eng = self.model.engine(precision or self.precision)
        class_id = _single_class_id(class_label)
        to_canvas = lambda t: t.permute(1, 0, 2, 3).reshape(1, 3, b * s_, s_).contiguous()       # [B,3,S,S] -> [1,3,B*S,S]
        from_canvas = lambda t: t.reshape(3, b, s_, s_).permute(1, 0, 2, 3).contiguous()
        tiles = [(i * s_, 0) for i in range(b)]
        geo = SamplerGeometry(H=b * s_, W=s_, Hp=b * s_, Wp=s_, left=0, top=0, inner_l=0, inner_t=0, inner_r=s_, inner_b=b * s_,
                              tile=s_, n_even=b, n_odd=b, n_images=1)
        scalars, log_snrs = _schedule(num_sample_steps)
        cond01 = to_canvas(condition_x.to(dev, torch.float32))
        cond_canvas = torch.empty(1, 3, b * s_, s_, device=dev, dtype=torch.float32)
        eng.sampler_begin(geo, cond01, cond_canvas, tiles, tiles, scalars, log_snrs, class_id)
        host_noise = self.noise_source == "host"
        seed = self.device_noise_seed
        if generation_start_steps > 0:                                   # q_sample(condition, t_start) :3198-3201
            ls0 = beta_linear_log_snr(1.0 - torch.tensor(generation_start_steps / num_sample_steps))
            img = torch.empty(1, 3, b * s_, s_, device=dev)
            nz = to_canvas(_host_randn(self.host_generator, b, 3, s_, s_).to(dev)) if host_noise else None
            eng.sampler_q_start(cond01, nz, float(ls0.sigmoid().sqrt()), float((-ls0).sigmoid().sqrt()), img, seed)
        elif host_noise:
            img = to_canvas(_host_randn(self.host_generator, b, 3, s_, s_).to(dev))           # :3203
        else:
            img = eng.randn_(torch.empty(1, 3, b * s_, s_, device=dev), seed, 0)
        x_start = img.clone() if with_x0_images else None
        image_list = [from_canvas(img).cpu()] if with_images else None
        x0_image_list = [from_canvas(img).cpu()] if with_x0_images else None
        for i in range(num_sample_steps):
            if i < generation_start_steps:
                continue
            cur_cond_scale = 1.0 if i < guidance_start_steps else cond_scale
            cur_class_scale = 1.0 if i < class_guidance_start_steps else class_cond_scale
            if cur_cond_scale != 1.0:
                passes, kind, scale = 2, 2, cur_cond_scale
            elif cur_class_scale != 1.0:
                passes, kind, scale = 2, 1, cur_class_scale
            else:
                passes, kind, scale = 1, 0, 1.0
            last = i == num_sample_steps - 1
            noise_tiles = _host_randn(self.host_generator, b, 3, s_, s_).to(dev, non_blocking=True) if (host_noise and not last) else None
            # no ring re-noise in the un-tiled loop: run the step over all tiles with do_ring = False
            eng.sampler_step_tiles(i, 0, b, False, img, cond_canvas, x_start, noise_tiles, None, passes, kind, scale,
                                   self.max_tiles_per_launch or b, seed=seed)
            if with_images:
                image_list.append(from_canvas(img).cpu())
            if with_x0_images:
                x0_image_list.append(from_canvas(x_start).cpu())
        out = torch.empty(1, 3, b * s_, s_, device=dev, dtype=torch.float32)
        eng.sampler_end(img, out)
        out = from_canvas(out)
        if with_images:
            return (out, image_list, x0_image_list) if with_x0_images else (out, image_list)
        return out

    def forward(self, *args, **kwargs):
        raise NotImplementedError("training (p_losses) is not part of the inference-only release this engine mirrors")


# ---------------------------------------------------------------------------------------------
# EDM (Karras et al.) sampler over the same U-Net (reference model.py:2059-2475)
# ---------------------------------------------------------------------------------------------
def _tiling(h: int, w: int, tile_size: int, tile_stride: int):
    """Canvas + both tile grids of one image (reference model.py:2321-2323, :2360-2371 / :3301-3342)."""
    (left, top, right, bottom), pad = get_coord_and_pad(h, w)
    hp, wp = h + pad[2] + pad[3], w + pad[0] + pad[1]
    if max(pad[0], pad[1]) >= w or max(pad[2], pad[3]) >= h:
        raise RuntimeError("Padding size should be less than the corresponding input dimension "
                           f"(reflect pad {pad} of a {h}x{w} image)")
    coords0 = get_coords(hp, wp, tile_size, tile_size, diff=0)
    if hp <= tile_size and wp <= tile_size:
        coords1 = get_coords(hp, wp, tile_size, tile_stride, diff=0)
    else:
        coords1 = get_coords(hp - tile_size, wp - tile_size, tile_size, tile_stride, diff=tile_size // 2)
    inner, _ = get_area(coords1, hp, wp)
    return (left, top, right, bottom), (hp, wp), coords0, coords1, inner


class ConditionalElucidatedDiffusionSR(nn.Module):
    """``ConditionalElucidatedDiffusionSR`` of the reference (model.py:2059-2128 ctor, :2309-2475 ``tiled_sample``): Heun
    2nd-order EDM sampling, two U-Net evaluations per step, same tiling as the DDPM wrapper.

    The base class ``denoising_diffusion_pytorch.ElucidatedDiffusion`` (un-vendored, pinned 1.8.15) contributes the
    rho-schedule and the preconditioning coefficients; they are restated here from the published algorithm
    (``sample_schedule`` / ``c_in`` / ``c_skip`` / ``c_out`` / ``c_noise``) with torch fp32 ops, and handed to the engine as
    per-step scalars.  Inference only (``tiled_sample``, ``sample`` -> ``sample_org`` / ``sample_using_dpmpp``)."""

    def __init__(self, net, *, image_size, channels=3, num_sample_steps=32, sigma_min=0.002, sigma_max=80, sigma_data=0.5,
                 rho=7, P_mean=-1.2, P_std=1.2, S_churn=80, S_tmin=0.05, S_tmax=50, S_noise=1.003, cond_drop_prob=0.0,
                 class_cond_drop_prob=0.0, use_dpmpp_solver=False, loss_type="l2"):
        super().__init__()
        assert net.random_or_learned_sinusoidal_cond
        self.self_condition = net.self_condition
        self.net = net
        self.channels, self.image_size = channels, image_size
        self.sigma_min, self.sigma_max, self.sigma_data, self.rho = sigma_min, sigma_max, sigma_data, rho
        self.P_mean, self.P_std, self.num_sample_steps = P_mean, P_std, num_sample_steps
        self.S_churn, self.S_tmin, self.S_tmax, self.S_noise = S_churn, S_tmin, S_tmax, S_noise
        self.cond_drop_prob, self.class_cond_drop_prob = cond_drop_prob, class_cond_drop_prob
        self.use_dpmpp_solver, self.loss_type = use_dpmpp_solver, loss_type
        # engine knobs (not part of the reference surface)
        self.noise_source = "host"
        self.host_generator = None         # as in the DDPM wrapper: None = torch's global CPU generator
        self.device_noise_seed = 0
        self.max_tiles_per_launch = None
        self.step_lanes = lanes_setting_from_env()   # as in the DDPM wrapper
        self.precision = "fp32"            # as in the DDPM wrapper
        self.canvas_group = None           # set by srgd_amd.parallel.shard_canvas: tiles of every step split over its ranks

    def set_seed(self, seed):
        torch.cuda.manual_seed(seed)
        self.device_noise_seed = int(seed)

    @property
    def device(self):
        return next(self.net.parameters()).device

    # ---- the base class's published formulas (fp32 torch ops, as the reference evaluates them) ----
    def c_skip(self, sigma):
        return (self.sigma_data ** 2) / (sigma ** 2 + self.sigma_data ** 2)

    def c_out(self, sigma):
        return sigma * self.sigma_data * (self.sigma_data ** 2 + sigma ** 2) ** -0.5

    def c_in(self, sigma):
        return 1 * (sigma ** 2 + self.sigma_data ** 2) ** -0.5

    def c_noise(self, sigma):
        return torch.log(sigma.clamp(min=1e-20)) * 0.25

    def sample_schedule(self, num_sample_steps=None):
        n = self.num_sample_steps if num_sample_steps is None else num_sample_steps
        inv_rho = 1 / self.rho
        steps = torch.arange(n, dtype=torch.float32)
        sigmas = (self.sigma_max ** inv_rho + steps / (n - 1) * (self.sigma_min ** inv_rho - self.sigma_max ** inv_rho)) ** self.rho
        return torch.nn.functional.pad(sigmas, (0, 1), value=0.0)

    def _step_tables(self, n: int, clamp: bool, with_ring: bool = True):
        """Per-step scalars of the Heun loops.  ``with_ring``: the tiled loop re-noises the ring on odd steps from the
        CONSTRUCTOR's schedule (below); the un-tiled ``sample_org`` never does (model.py:2212-2306 has no get_noised_images
        call per step) and so runs with any per-call step count.  Odd steps beyond the constructor's schedule get
        ``ring_sigma = nan`` here; ``tiled_sample`` raises upstream's IndexError when the loop reaches the first of them."""
        from ._lib import EdmScalars
        sigmas = self.sample_schedule(n)
        # get_noised_images (model.py:2186-2189) is called without num_sample_steps at :2342 and :2457, so the
        # generation-start sigma and the odd-step ring sigmas come from the CONSTRUCTOR's schedule, not the per-call one
        noised_sigmas = self.sample_schedule(self.num_sample_steps)
        gammas = torch.where((sigmas >= self.S_tmin) & (sigmas <= self.S_tmax),
                             min(self.S_churn / n, math.sqrt(2) - 1), 0.0)                      # model.py:2333-2337
        scalars, c_noise = [], []
        for i in range(n):
            sigma, sigma_next, gamma = sigmas[i].item(), sigmas[i + 1].item(), gammas[i].item()
            sigma_hat = sigma + gamma * sigma                                                    # :2388
            sh, sn = torch.full((1,), sigma_hat), torch.full((1,), sigma_next)                   # :2137 fp32 tensors
            scalars.append(EdmScalars(
                s_noise=self.S_noise, hat_coef=math.sqrt(sigma_hat ** 2 - sigma ** 2), sigma_hat=sigma_hat,
                sigma_next=sigma_next, dt=sigma_next - sigma_hat, half_dt=0.5 * (sigma_next - sigma_hat),
                c_in_hat=float(self.c_in(sh)), c_skip_hat=float(self.c_skip(sh)), c_out_hat=float(self.c_out(sh)),
                c_in_next=float(self.c_in(sn)), c_skip_next=float(self.c_skip(sn)), c_out_next=float(self.c_out(sn)),
                # read on odd steps of the tiled loop only; upstream indexes its constructor-length schedule there
                ring_sigma=(float(noised_sigmas[i]) if i < len(noised_sigmas) else math.nan) if (with_ring and i % 2 == 1) else 0.0,
                clamp=1.0 if clamp else 0.0, dpm_gamma=0.0, pad1=0.0))
            c_noise += [float(self.c_noise(sh)), float(self.c_noise(sn))]
        return sigmas, noised_sigmas, scalars, c_noise

    @torch.inference_mode()
    def tiled_sample(self, batch_size=4, tile_size=256, tile_stride=256, condition_x=None, class_label=None,
                     cond_scale=1.0, guidance_start_steps=0, class_cond_scale=1.0, class_guidance_start_steps=0,
                     generation_start_steps=0, num_sample_steps=None, clamp=True, zero_init=False, with_images=False,
                     with_x0_images=False, start_white_noise=True, amp=False, precision=None):
        """Reference model.py:2309-2475 (``start_white_noise`` and ``amp`` are accepted and unused there too; ``precision``
        is the engine-only override of ``self.precision``)."""
        n = self.num_sample_steps if num_sample_steps is None else num_sample_steps
        if cond_scale != 1.0 and class_cond_scale != 1.0:
            raise NotImplementedError("Currently, you cannot specify both cond_scale and class_cond_scale at the same time.")
        if tile_size != 256 or tile_stride != 256:
            raise NotImplementedError("tile_size/tile_stride other than 256 are unusable in the reference too "
                                      "(get_coord_and_pad is called without them, model.py:2321)")
        dev = self.device
        if dev.type != "cuda":
            raise _lib.SrgdHipError("tiled_sample runs on MI355X only (no CPU fallback)")
        batch, c, h, w = condition_x.shape
        if batch < 1 or c != 3:
            raise ValueError("condition_x must be [B,3,H,W] (B=1 in the reference; B>1 = same-sized images in lock-step, "
                             "each sampled as it would be alone with the same seed)")
        eng = self.net.engine(precision or self.precision)
        class_id = _single_class_id(class_label)
        (left, top, right, bottom), (hp, wp), coords0, coords1, (sl, st_, sr, sb) = _tiling(h, w, tile_size, tile_stride)
        geo = SamplerGeometry(H=h, W=w, Hp=hp, Wp=wp, left=left, top=top, inner_l=sl, inner_t=st_, inner_r=sr,
                              inner_b=sb, tile=tile_size, n_even=len(coords0), n_odd=len(coords1), n_images=batch)
        sigmas, noised_sigmas, scalars, c_noise = self._step_tables(n, clamp)
        cond01 = condition_x.to(dev, torch.float32).contiguous()
        cond_canvas = torch.empty(batch, 3, hp, wp, device=dev, dtype=torch.float32)
        eng.edm_begin(geo, cond01, cond_canvas, [(a, c_) for (a, _, c_, _) in coords0],
                      [(a, c_) for (a, _, c_, _) in coords1], scalars, c_noise, class_id)
        host_noise = self.noise_source == "host"
        seed = self.device_noise_seed

        def canvas_noise(stream_id):
            if host_noise:
                return _host_randn(self.host_generator, 1, 3, hp, wp).to(dev, non_blocking=True)
            return eng.randn_(torch.empty(1, 3, hp, wp, device=dev), seed, stream_id)

        if generation_start_steps > 0:                                  # get_noised_images(condition, step) :2340, :2185
            img = torch.empty(batch, 3, hp, wp, device=dev)
            eng.sampler_q_start(cond01, canvas_noise(1), 1.0, float(noised_sigmas[generation_start_steps]), img, seed)
        elif zero_init:
            img = torch.zeros(batch, 3, hp, wp, device=dev)
        else:
            img = (canvas_noise(1) * float(sigmas[0])).repeat(batch, 1, 1, 1)   # :2346 (a tensor * tensor product upstream)
        x_start = img.clone() if with_x0_images else None
        image_list = [img[:, :, top:bottom, left:right].clone().cpu()] if with_images else None
        x0_image_list = [img[:, :, top:bottom, left:right].clone().cpu()] if with_x0_images else None
        work = torch.empty(2, batch, 3, hp, wp, device=dev, dtype=torch.float32)
        sub_batch = self.max_tiles_per_launch or batch_size
        lanes = None
        for i in range(n):
            if i < generation_start_steps:
                continue
            cur_cond_scale = 1.0 if i < guidance_start_steps else cond_scale
            cur_class_scale = 1.0 if i < class_guidance_start_steps else class_cond_scale
            if cur_cond_scale != 1.0:
                passes, kind, scale = 2, 2, cur_cond_scale
            elif cur_class_scale != 1.0:
                passes, kind, scale = 2, 1, cur_class_scale
            else:
                passes, kind, scale = 1, 0, 1.0
            if i % 2 == 1 and i >= len(noised_sigmas):
                # upstream fails here too, mid-loop: get_noised_images(images, i) indexes the constructor's schedule (:2457 -> :2187)
                raise IndexError(f"index {i} is out of bounds for dimension 0 with size {len(noised_sigmas)}")
            z = canvas_noise(None) if host_noise else None              # eps of the step (:2386), before the ring draw
            ring = canvas_noise(None) if (host_noise and i % 2 == 1) else None
            n_step = (len(coords1) if i % 2 else len(coords0)) * batch
            n_lanes = lanes_wanted(n_step, passes, sub_batch, self.step_lanes, precision or self.precision) if self.canvas_group is None else 1
            if n_lanes > 1:
                if lanes is None or len(lanes.engines) != n_lanes:   # further engines (srgd_amd.lanes): same run geometry
                    more = [self.net.engine(precision or self.precision, lane=k) for k in range(1, n_lanes)]
                    for e_ in more:
                        e_.edm_begin(geo, cond01, torch.empty_like(cond_canvas), [(a, c_) for (a, _, c_, _) in coords0],
                                     [(a, c_) for (a, _, c_, _) in coords1], scalars, c_noise, class_id)
                    lanes = StepLanes([eng] + more, dev)
                lanes.run(n_step, lambda e_, first, count, do_ring: e_.edm_step_tiles(
                    i, first, count, do_ring, img, cond_canvas, x_start, work, z, ring, passes, kind, scale, sub_batch, seed))
            elif self.canvas_group is None:
                eng.edm_step(i, img, cond_canvas, x_start, work, z, ring, passes, kind, scale, sub_batch, seed=seed)
            else:
                from .parallel import sharded_edm_step
                n_tiles = (len(coords1) if i % 2 else len(coords0)) * batch
                sharded_edm_step(eng, self.canvas_group, i, n_tiles, img, cond_canvas, x_start, work, z, ring, passes, kind,
                                 scale, sub_batch, seed)
            if with_images:
                image_list.append(img.clone().cpu())
            if with_x0_images:
                x0_image_list.append(x_start.clone().cpu())
        out = torch.empty(batch, 3, h, w, device=dev, dtype=torch.float32)
        eng.sampler_end(img, out)
        if with_images:
            return (out, image_list, x0_image_list) if with_x0_images else (out, image_list)
        return out

    def _dpmpp_tables(self, n: int, clamp: bool):
        """Per-step scalars of ``sample_using_dpmpp`` (model.py:2513-2541), evaluated with the reference's fp32 tensor ops:
        preconditioning at sigma_i, update coefficients sigma_fn(t_next)/sigma_fn(t) and expm1(-h), multistep weight gamma."""
        from ._lib import EdmScalars
        sigmas = self.sample_schedule(n)
        t_fn = lambda sigma: sigma.log().neg()
        sigma_fn = lambda t: t.neg().exp()
        scalars, c_noise = [], []
        for i in range(n):
            si = torch.full((1,), sigmas[i].item())                                             # :2137 fp32 tensor
            t, t_next = t_fn(sigmas[i]), t_fn(sigmas[i + 1])
            h = t_next - t
            scalars.append(EdmScalars(
                s_noise=0.0, hat_coef=0.0, sigma_hat=float(sigmas[i]), sigma_next=float(sigmas[i + 1]),
                dt=float(sigma_fn(t_next) / sigma_fn(t)), half_dt=float((-h).expm1()),
                c_in_hat=float(self.c_in(si)), c_skip_hat=float(self.c_skip(si)), c_out_hat=float(self.c_out(si)),
                c_in_next=0.0, c_skip_next=0.0, c_out_next=0.0, ring_sigma=0.0, clamp=1.0 if clamp else 0.0,
                dpm_gamma=0.0, pad1=0.0))                                    # dpm_gamma: set per call (depends on the start step)
            c_noise += [float(self.c_noise(si)), 0.0]
        return sigmas, scalars, c_noise

    @torch.inference_mode()
    def sample(self, batch_size=16, condition_x=None, class_label=None, cond_scale=1.0, guidance_start_steps=0,
               class_cond_scale=1.0, class_guidance_start_steps=0, generation_start_steps=0, num_sample_steps=None,
               clamp=True, with_images=False, with_x0_images=False, zero_init=False, precision=None):
        """Reference model.py:2196-2209: un-tiled sampling of a ``[B,3,256,256]`` batch - the Heun loop ``sample_org``
        (:2212-2306), or ``sample_using_dpmpp`` when the wrapper was built with ``use_dpmpp_solver``."""
        fn = self.sample_using_dpmpp if self.use_dpmpp_solver else self.sample_org
        return fn(batch_size, condition_x, class_label, cond_scale, guidance_start_steps, class_cond_scale,
                  class_guidance_start_steps, generation_start_steps, num_sample_steps, clamp, with_images, with_x0_images,
                  zero_init, precision=precision)

    def _untiled_setup(self, batch_size, condition_x, class_label, cond_scale, class_cond_scale, generation_start_steps,
                       zero_init, sigma0, precision):
        """Shared front end of the two un-tiled loops: the batch becomes one canvas of B stacked tiles (``[1,3,B*S,S]``, no
        padding, empty ring) exactly as in the DDPM wrapper's ``sample``; returns the engine, layout helpers and the start
        canvas (model.py:2219-2244 / :2490-2503)."""
        if cond_scale != 1.0 and class_cond_scale != 1.0:
            raise NotImplementedError("Currently, you cannot specify both cond_scale and class_cond_scale at the same time.")
        dev = self.device
        if dev.type != "cuda":
            raise _lib.SrgdHipError("sample runs on MI355X only (no CPU fallback)")
        b = int(batch_size)
        s_ = 256
        if tuple(condition_x.shape) != (b, self.channels, s_, s_):
            raise NotImplementedError(f"condition_x must be [{b},{self.channels},{s_},{s_}]: this engine's un-tiled path runs "
                                      "256 x 256 images (the tile edge of its kernels; the shipped image_size)")
        if generation_start_steps > 0 and b not in (1, s_):
            # get_noised_images (model.py:2191-2193) multiplies a [B] sigma vector into a [B,3,h,w] tensor
            raise RuntimeError(f"The size of tensor a ({b}) must match the size of tensor b ({s_}) at non-singleton dimension 3")
        eng = self.net.engine(precision or self.precision)
        to_canvas = lambda t: t.permute(1, 0, 2, 3).reshape(1, 3, b * s_, s_).contiguous()       # [B,3,S,S] -> [1,3,B*S,S]
        from_canvas = lambda t: t.reshape(3, b, s_, s_).permute(1, 0, 2, 3).contiguous()
        tiles = [(i * s_, 0) for i in range(b)]
        geo = SamplerGeometry(H=b * s_, W=s_, Hp=b * s_, Wp=s_, left=0, top=0, inner_l=0, inner_t=0, inner_r=s_, inner_b=b * s_,
                              tile=s_, n_even=b, n_odd=b, n_images=1)
        cond01 = to_canvas(condition_x.to(dev, torch.float32))
        host_noise = self.noise_source == "host"
        seed = self.device_noise_seed

        def batch_noise(stream_id):
            if host_noise:
                return to_canvas(_host_randn(self.host_generator, b, 3, s_, s_).to(dev, non_blocking=True))
            return eng.randn_(torch.empty(1, 3, b * s_, s_, device=dev), seed, stream_id)

        def start(eng_):
            if generation_start_steps > 0:                               # get_noised_images: the CONSTRUCTOR's schedule
                img = torch.empty(1, 3, b * s_, s_, device=dev)
                sig = float(self.sample_schedule(self.num_sample_steps)[generation_start_steps])
                eng_.sampler_q_start(cond01, batch_noise(1), 1.0, sig, img, seed)
                return img
            if zero_init:
                return torch.zeros(1, 3, b * s_, s_, device=dev)
            return batch_noise(1) * sigma0
        from types import SimpleNamespace
        return SimpleNamespace(eng=eng, b=b, s=s_, dev=dev, geo=geo, tiles=tiles, cond01=cond01, from_canvas=from_canvas,
                               batch_noise=batch_noise, start=start, host_noise=host_noise, seed=seed)

    @staticmethod
    def _guidance(i, cond_scale, guidance_start_steps, class_cond_scale, class_guidance_start_steps):
        cur_cond_scale = 1.0 if i < guidance_start_steps else cond_scale
        cur_class_scale = 1.0 if i < class_guidance_start_steps else class_cond_scale
        if cur_cond_scale != 1.0:
            return 2, 2, cur_cond_scale
        if cur_class_scale != 1.0:
            return 2, 1, cur_class_scale
        return 1, 0, 1.0

    @torch.inference_mode()
    def sample_org(self, batch_size=16, condition_x=None, class_label=None, cond_scale=1.0, guidance_start_steps=0,
                   class_cond_scale=1.0, class_guidance_start_steps=0, generation_start_steps=0, num_sample_steps=None,
                   clamp=True, with_images=False, with_x0_images=False, zero_init=False, precision=None):
        """Reference model.py:2212-2306 (stochastic Heun, two network evaluations per step) on the tiled EDM machinery:
        the same per-step arithmetic as ``tiled_sample`` with one tile per image, per-image noise and no ring."""
        n = self.num_sample_steps if num_sample_steps is None else num_sample_steps
        sigmas, _, scalars, c_noise = self._step_tables(n, clamp, with_ring=False)
        u = self._untiled_setup(batch_size, condition_x, class_label, cond_scale, class_cond_scale, generation_start_steps,
                                zero_init, float(sigmas[0]), precision)
        eng, b, s_, dev = u.eng, u.b, u.s, u.dev
        cond_canvas = torch.empty(1, 3, b * s_, s_, device=dev, dtype=torch.float32)
        eng.edm_begin(u.geo, u.cond01, cond_canvas, u.tiles, u.tiles, scalars, c_noise, _single_class_id(class_label))
        img = u.start(eng)
        x_start = img.clone() if with_x0_images else None
        image_list = [u.from_canvas(img).cpu()] if with_images else None
        x0_image_list = [u.from_canvas(img).cpu()] if with_x0_images else None
        work = torch.empty(2, 1, 3, b * s_, s_, device=dev, dtype=torch.float32)
        for i in range(n):
            if i < generation_start_steps:
                continue
            passes, kind, scale = self._guidance(i, cond_scale, guidance_start_steps, class_cond_scale,
                                                 class_guidance_start_steps)
            z = u.batch_noise(None) if u.host_noise else None           # eps of the step (:2269)
            # the ring of this geometry is empty (inner area = the whole canvas): no second draw on odd steps
            eng.edm_step(i, img, cond_canvas, x_start, work, z, None, passes, kind, scale, self.max_tiles_per_launch or b,
                         seed=u.seed)
            if with_images:
                image_list.append(u.from_canvas(img).cpu())
            if with_x0_images:
                x0_image_list.append(u.from_canvas(x_start).cpu())
        out = torch.empty(1, 3, b * s_, s_, device=dev, dtype=torch.float32)
        eng.sampler_end(img, out)
        out = u.from_canvas(out)
        if with_images:
            return (out, image_list, x0_image_list) if with_x0_images else (out, image_list)
        return out

    @torch.inference_mode()
    def sample_using_dpmpp(self, batch_size=16, condition_x=None, class_label=None, cond_scale=1.0, guidance_start_steps=0,
                           class_cond_scale=1.0, class_guidance_start_steps=0, generation_start_steps=0,
                           num_sample_steps=None, clamp=True, with_images=False, with_x0_images=False, zero_init=False,
                           precision=None):
        """Reference model.py:2479-2557: DPM-Solver++(2M) in t = -log sigma, one network evaluation per step
        (``srgd_edm_dpmpp_step``); deterministic after the start canvas."""
        n = self.num_sample_steps if num_sample_steps is None else num_sample_steps
        sigmas, scalars, c_noise = self._dpmpp_tables(n, clamp)
        t_fn = lambda sigma: sigma.log().neg()
        for i in range(n):                                               # :2533-2538; old_denoised is None on the first
            first = i == max(generation_start_steps, 0)                  # executed step, the multistep term is dropped on the last
            if not first and sigmas[i + 1] != 0 and i > 0:
                t, t_next = t_fn(sigmas[i]), t_fn(sigmas[i + 1])
                r = (t - t_fn(sigmas[i - 1])) / (t_next - t)
                scalars[i].dpm_gamma = float(-1 / (2 * r))
        u = self._untiled_setup(batch_size, condition_x, class_label, cond_scale, class_cond_scale, generation_start_steps,
                                zero_init, float(sigmas[0]), precision)
        eng, b, s_, dev = u.eng, u.b, u.s, u.dev
        cond_canvas = torch.empty(1, 3, b * s_, s_, device=dev, dtype=torch.float32)
        eng.edm_begin(u.geo, u.cond01, cond_canvas, u.tiles, u.tiles, scalars, c_noise, _single_class_id(class_label))
        img = u.start(eng)
        x_start = img.clone() if with_x0_images else None
        image_list = [u.from_canvas(img).cpu()] if with_images else None
        x0_image_list = [u.from_canvas(img).cpu()] if with_x0_images else None
        old = torch.zeros(1, 3, b * s_, s_, device=dev, dtype=torch.float32)
        for i in range(n):
            if i < generation_start_steps:
                continue
            passes, kind, scale = self._guidance(i, cond_scale, guidance_start_steps, class_cond_scale,
                                                 class_guidance_start_steps)
            eng.edm_dpmpp_step(i, img, cond_canvas, x_start, old, passes, kind, scale, self.max_tiles_per_launch or b)
            if with_images:
                image_list.append(u.from_canvas(img).cpu())
            if with_x0_images:
                x0_image_list.append(u.from_canvas(x_start).cpu())
        out = torch.empty(1, 3, b * s_, s_, device=dev, dtype=torch.float32)
        eng.sampler_end(img, out)
        out = u.from_canvas(out)
        if with_images:
            return (out, image_list, x0_image_list) if with_x0_images else (out, image_list)
        return out

    def forward(self, *args, **kwargs):
        raise NotImplementedError("training is not part of the inference-only release this engine mirrors")


# ---------------------------------------------------------------------------------------------
# factory (reference model.py:3500-3666)
# ---------------------------------------------------------------------------------------------
class ModelEma(nn.Module):
    """Stand-in for ``timm.utils.ModelEmaV2`` as the reference uses it at inference: a holder whose
    ``.module`` is an eval-mode copy that receives ``ckpt['ema_model']`` (model.py:3657-3662)."""

    def __init__(self, model, decay=0.9999, device=None):
        super().__init__()
        self.module = copy.deepcopy(model)
        self.module.eval()
        self.decay, self.device = decay, device


def _parse_bool_list(text: str) -> Tuple[bool, ...]:
    table = {"true": True, "false": False}
    try:
        return tuple(table[t.strip().lower()] for t in text.split(","))
    except KeyError as exc:
        raise ValueError(f"full_attn must be a comma list of True/False, got {text!r}") from exc


def get_model(conf, logger):
    dim_mults = tuple(int(t) for t in conf.ddpm_unet_dim_mults.split(","))
    full_attn = _parse_bool_list(conf.full_attn)
    if conf.model not in ("conditional_continuous", "conditional_elucidated"):
        raise NotImplementedError(
            f"model={conf.model!r}: this engine covers the shipped 'conditional_continuous' path and the "
            "'conditional_elucidated' (EDM) sampler over the same U-Net (SURVEY.md section 8; the other wrappers have "
            "no released config or weights)")
    unet = ConditionalSRUnet(dim=conf.unet_dim, dim_mults=dim_mults, full_attn=full_attn,
                             learned_variance=conf.learned_variance,
                             learned_sinusoidal_cond=conf.learned_sinusoidal_cond,
                             learned_sinusoidal_dim=conf.learned_sinusoidal_dim, flash_attn=conf.flash_attn,
                             pixel_shuffle_upsample=conf.pixel_shuffle_upsample, num_classes=conf.num_classes)
    logger.info(f"ConditionalSRUnet: channels=6 dim={conf.unet_dim} dim_mults={conf.ddpm_unet_dim_mults} "
                f"num_classes={conf.num_classes}")
    assert conf.learned_sinusoidal_cond
    if conf.model == "conditional_elucidated":                          # reference model.py:3593-3614
        model = ConditionalElucidatedDiffusionSR(
            net=unet, image_size=conf.image_size, num_sample_steps=conf.num_sample_steps, sigma_min=conf.sigma_min,
            sigma_max=conf.sigma_max, sigma_data=conf.sigma_data, rho=conf.rho, P_mean=conf.P_mean, P_std=conf.P_std,
            S_churn=conf.S_churn, S_tmin=conf.S_tmin, S_tmax=conf.S_tmax, S_noise=conf.S_noise,
            cond_drop_prob=conf.cond_drop_prob, class_cond_drop_prob=conf.class_cond_drop_prob,
            use_dpmpp_solver=conf.use_dpmpp_solver, loss_type=conf.loss_type)
        logger.info(f"ConditionalElucidatedDiffusionSR: image_size={conf.image_size} num_sample_steps={conf.num_sample_steps}")
        ema_model = ModelEma(model, decay=conf.ema_decay)
        if conf.ckpt_path:
            ckpt = torch.load(conf.ckpt_path, map_location="cpu", weights_only=True)
            check = ema_model.module.load_state_dict(ckpt["ema_model"], strict=conf.load_strict)
            logger.info(f"load ema_model weight from : {conf.ckpt_path}")
            logger.info(f"check: {check}")
        return ema_model
    conf.use_dpmpp_solver = False
    model = ConditionalContinuousTimeGaussianDiffusionSR(
        unet, image_size=conf.image_size, noise_schedule=conf.noise_schedule,
        num_sample_steps=conf.num_sample_steps, clip_sample_denoised=conf.clip_sample_denoised,
        learned_schedule_net_hidden_dim=conf.learned_schedule_net_hidden_dim,
        learned_noise_schedule_frac_gradient=conf.learned_noise_schedule_frac_gradient,
        min_snr_loss_weight=conf.min_snr_loss_weight, min_snr_gamma=conf.min_snr_gamma,
        cond_drop_prob=conf.cond_drop_prob, class_cond_drop_prob=conf.class_cond_drop_prob,
        loss_type=conf.loss_type)
    logger.info(f"ConditionalContinuousTimeGaussianDiffusionSR: image_size={conf.image_size} "
                f"num_sample_steps={conf.num_sample_steps}")
    ema_model = ModelEma(model, decay=conf.ema_decay)
    if conf.ckpt_path:
        ckpt = torch.load(conf.ckpt_path, map_location="cpu", weights_only=True)
        check = ema_model.module.load_state_dict(ckpt["ema_model"], strict=conf.load_strict)
        logger.info(f"load ema_model weight from : {conf.ckpt_path}")
        logger.info(f"check: {check}")
    return ema_model
