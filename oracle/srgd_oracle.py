"""CPU oracle for the Real-SRGD tiled CFG-DDPM sampling path (TEST INFRASTRUCTURE ONLY).

This file is the *checker*, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The shipped path
(``srgd_amd/``) never routes through it.

It is a from-scratch functional restatement (plain ``torch`` CPU fp32 ops on a flat
``state_dict``; no ``nn.Module`` graph, no einops, no third-party diffusion package) of
the reference algorithm; every function cites the reference lines it follows
(``/root/reference/model.py`` unless another file is named).

Pinning status: the reference ships no tests, golden vectors or fixtures for this path
(SURVEY.md section 4), so this oracle is pinned against *outputs of the reference itself*,
run in the build container by ``oracle/pin_against_reference.py`` (imports the reference
``model.py`` with throw-away stubs for its three un-vendored imports) and frozen as
``tests/golden/*.npz`` by ``tests/golden/make_golden.py``.
Third-party arithmetic not under /root/reference: ``denoising-diffusion-pytorch==1.8.15``
``attend.Attend(flash=False)`` (call site model.py:352) is restated as textbook softmax
attention with scale ``dim_head ** -0.5``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# configuration of the denoiser topology (model.py:537-556 constructor arguments)
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class UnetCfg:
    dim: int = 128
    dim_mults: Tuple[int, ...] = (1, 2, 4, 8)
    channels: int = 3
    groups: int = 8                     # resnet_block_groups
    sinus_dim: int = 32                 # learned_sinusoidal_dim
    heads: int = 4
    dim_head: int = 32
    full_attn: Tuple[bool, ...] = (False, False, False, True)
    num_classes: Optional[int] = 3

    @property
    def dims(self) -> List[int]:        # model.py:585  [init_dim, dim*m ...]
        return [self.dim] + [self.dim * m for m in self.dim_mults]

    @property
    def time_dim(self) -> int:          # model.py:592
        return self.dim * 4

    @property
    def downsample_factor(self) -> int:  # property of the un-vendored Unet base (model.py:679)
        return 2 ** (len(self.dim_mults) - 1)


# --------------------------------------------------------------------------------------
# tile geometry (pure ints)
# --------------------------------------------------------------------------------------
def canvas_box_and_pad(height: int, width: int, tile: int = 256):
    """model.py:116-135 get_coord_and_pad: the padded canvas and where the image sits in it.

    Returns ((left, top, right, bottom), (pad_l, pad_r, pad_t, pad_b)).
    """
    if height <= tile and width <= tile:
        canvas_h = canvas_w = tile
    else:
        canvas_h = (-(-height // tile)) * tile + tile
        canvas_w = (-(-width // tile)) * tile + tile
    left = (canvas_w - width) // 2
    top = (canvas_h - height) // 2
    box = (left, top, left + width, top + height)
    pad = (left, canvas_w - left - width, top, canvas_h - top - height)
    return box, pad


def tile_grid(h: int, w: int, tile: int, stride: int, shift: int = 0):
    """model.py:137-150 get_coords: row-major (hs, he, ws, we) tile boxes."""
    def starts(n):
        s = list(range(0, n - tile + 1, stride))
        if (n - tile) % stride != 0:
            s.append(n - tile)
        return s
    return [(y + shift, y + tile + shift, x + shift, x + tile + shift)
            for y in starts(h) for x in starts(w)]


def grid_bbox(coords, height: int, width: int):
    """model.py:152-179 get_area: bounding box of a grid and its distance to the canvas edge."""
    top = min([c[0] for c in coords] + [height])
    bottom = max([c[1] for c in coords] + [0])
    left = min([c[2] for c in coords] + [width])
    right = max([c[3] for c in coords] + [0])
    pad = (left, width - right, top, height - bottom)
    return (left, top, right, bottom), pad


def sampling_grids(canvas_h: int, canvas_w: int, tile: int = 256, stride: int = 256):
    """model.py:3328-3334: even-step grid over the full canvas, odd-step grid shifted by tile/2."""
    even = tile_grid(canvas_h, canvas_w, tile, tile, 0)
    if canvas_h <= tile and canvas_w <= tile:
        odd = tile_grid(canvas_h, canvas_w, tile, stride, 0)
    else:
        odd = tile_grid(canvas_h - tile, canvas_w - tile, tile, stride, tile // 2)
    return even, odd


# --------------------------------------------------------------------------------------
# continuous-time schedule (scalars)
# --------------------------------------------------------------------------------------
def log_snr_linear(t: Tensor) -> Tensor:
    """model.py:2629-2633: -log(expm1(1e-4 + 10 t^2)) with the log argument clamped at 1e-20."""
    return -torch.log(torch.special.expm1(1e-4 + 10 * (t ** 2)).clamp(min=1e-20))


def step_scalars(t: Tensor, t_next: Tensor) -> Dict[str, Tensor]:
    """model.py:3127-3134 and :3168 - all 0-dim fp32 tensors."""
    ls, ls_n = log_snr_linear(t), log_snr_linear(t_next)
    c = -torch.special.expm1(ls - ls_n)
    a2, a2n = ls.sigmoid(), ls_n.sigmoid()
    s2, s2n = (-ls).sigmoid(), (-ls_n).sigmoid()
    return dict(log_snr=ls, log_snr_next=ls_n, c=c, alpha=a2.sqrt(), sigma=s2.sqrt(),
                alpha_next=a2n.sqrt(), var=s2n * c)


# --------------------------------------------------------------------------------------
# U-Net building blocks (functional, NCHW fp32)
# --------------------------------------------------------------------------------------
def _w(sd: Dict[str, Tensor], key: str) -> Tensor:
    return sd[key]


def rms_norm(x: Tensor, g: Tensor) -> Tensor:
    """model.py:201-207: F.normalize over channels (eps 1e-12 on the norm) * g * sqrt(C)."""
    return F.normalize(x, dim=1) * g * (x.shape[1] ** 0.5)


def gn_block(sd, p: str, x: Tensor, groups: int, scale_shift=None) -> Tensor:
    """model.py:243-259 Block: conv3x3 -> GroupNorm -> (x*(scale+1)+shift) -> SiLU."""
    x = F.conv2d(x, _w(sd, p + ".proj.weight"), _w(sd, p + ".proj.bias"), padding=1)
    x = F.group_norm(x, groups, _w(sd, p + ".norm.weight"), _w(sd, p + ".norm.bias"), eps=1e-5)
    if scale_shift is not None:
        scale, shift = scale_shift
        x = x * (scale + 1) + shift
    return F.silu(x)


def resnet_block(sd, p: str, x: Tensor, t_emb: Tensor, groups: int) -> Tensor:
    """model.py:261-285 ResnetBlock; chunk(2): first half scale, second half shift (:279)."""
    e = F.linear(F.silu(t_emb), _w(sd, p + ".mlp.1.weight"), _w(sd, p + ".mlp.1.bias"))
    e = e[:, :, None, None]
    half = e.shape[1] // 2
    h = gn_block(sd, p + ".block1", x, groups, (e[:, :half], e[:, half:]))
    h = gn_block(sd, p + ".block2", h, groups)
    if (p + ".res_conv.weight") in sd:
        x = F.conv2d(x, _w(sd, p + ".res_conv.weight"), _w(sd, p + ".res_conv.bias"))
    return h + x


def linear_attention_core(qkv: Tensor, heads: int, dim_head: int) -> Tensor:
    """model.py:311-323 between to_qkv and to_out: qkv [b, 3*heads*dh, h, w] -> [b, heads*dh, h, w]."""
    b, _, hh, ww = qkv.shape
    n = hh * ww
    q, k, v = [z.reshape(b, heads, dim_head, n) for z in qkv.chunk(3, dim=1)]
    q = q.softmax(dim=2) * (dim_head ** -0.5)
    k = k.softmax(dim=3)
    ctx = torch.matmul(k, v.transpose(2, 3))            # [b,h,d,e] = sum_n k[d,n] v[e,n]
    out = torch.matmul(ctx.transpose(2, 3), q)          # [b,h,e,n] = sum_d ctx[d,e] q[d,n]
    return out.reshape(b, heads * dim_head, hh, ww)


def full_attention_core(qkv: Tensor, heads: int, dim_head: int) -> Tensor:
    """model.py:348-354 + Attend(flash=False): qkv [b, 3*heads*dh, h, w] -> [b, heads*dh, h, w]."""
    b, _, hh, ww = qkv.shape
    n = hh * ww
    q, k, v = [z.reshape(b, heads, dim_head, n).transpose(2, 3) for z in qkv.chunk(3, dim=1)]
    sim = torch.matmul(q, k.transpose(2, 3)) * (dim_head ** -0.5)
    out = torch.matmul(sim.softmax(dim=-1), v)          # [b,h,n,d]
    return out.transpose(2, 3).reshape(b, heads * dim_head, hh, ww)


def linear_attention(sd, p: str, x: Tensor, heads: int, dim_head: int) -> Tensor:
    """model.py:287-324: q softmax over d, k softmax over positions, context = k v^T."""
    x = rms_norm(x, _w(sd, p + ".norm.g"))
    out = linear_attention_core(F.conv2d(x, _w(sd, p + ".to_qkv.weight")), heads, dim_head)
    out = F.conv2d(out, _w(sd, p + ".to_out.0.weight"), _w(sd, p + ".to_out.0.bias"))
    return rms_norm(out, _w(sd, p + ".to_out.1.g"))


def full_attention(sd, p: str, x: Tensor, heads: int, dim_head: int) -> Tensor:
    """model.py:326-355 + Attend(flash=False): softmax(q k^T / sqrt(d)) v."""
    x = rms_norm(x, _w(sd, p + ".norm.g"))
    out = full_attention_core(F.conv2d(x, _w(sd, p + ".to_qkv.weight")), heads, dim_head)
    return F.conv2d(out, _w(sd, p + ".to_out.weight"), _w(sd, p + ".to_out.bias"))


def space_to_depth_conv(sd, p: str, x: Tensor) -> Tensor:
    """model.py:106-110 Downsample: 'b c (h 2)(w 2) -> b (c 2 2) h w' then conv1x1."""
    b, c, h, w = x.shape
    x = x.reshape(b, c, h // 2, 2, w // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, h // 2, w // 2)
    return F.conv2d(x, _w(sd, p + ".1.weight"), _w(sd, p + ".1.bias"))


def pixel_shuffle_up(sd, p: str, x: Tensor) -> Tensor:
    """model.py:70-98 PixelShuffleUpsample: conv1x1 -> SiLU -> PixelShuffle(2)."""
    x = F.conv2d(x, _w(sd, p + ".net.0.weight"), _w(sd, p + ".net.0.bias"))
    return F.pixel_shuffle(F.silu(x), 2)


def time_embedding(sd, log_snr: Tensor) -> Tensor:
    """model.py:223-238 + :603-608: [x, sin(2 pi x w), cos(2 pi x w)] -> Linear -> GELU(erf) -> Linear."""
    x = log_snr[:, None]
    freqs = x * _w(sd, "time_mlp.0.weights")[None, :] * 2 * math.pi
    f = torch.cat((x, freqs.sin(), freqs.cos()), dim=-1)
    h = F.gelu(F.linear(f, _w(sd, "time_mlp.1.weight"), _w(sd, "time_mlp.1.bias")))
    return F.linear(h, _w(sd, "time_mlp.3.weight"), _w(sd, "time_mlp.3.bias"))


def class_embedding(sd, label: Tensor) -> Tensor:
    """model.py:612-619: Embedding -> Linear -> GELU(erf) -> Linear."""
    e = F.embedding(label, _w(sd, "class_mlp.0.weight"))
    h = F.gelu(F.linear(e, _w(sd, "class_mlp.1.weight"), _w(sd, "class_mlp.1.bias")))
    return F.linear(h, _w(sd, "class_mlp.3.weight"), _w(sd, "class_mlp.3.bias"))


def unet_forward(sd: Dict[str, Tensor], cfg: UnetCfg, x: Tensor, log_snr: Tensor,
                 class_label: Optional[Tensor], cond: Optional[Tensor]) -> Tensor:
    """model.py:678-725 ConditionalSRUnet.forward (topology from :583-675)."""
    f = cfg.downsample_factor
    assert x.shape[-2] % f == 0 and x.shape[-1] % f == 0, \
        f"your input dimensions {tuple(x.shape[-2:])} need to be divisible by {f}, given the unet"
    if cond is None:
        cond = torch.zeros_like(x)
    x = torch.cat((x, cond), dim=1)                                     # :684 (noisy, condition)
    x = F.conv2d(x, _w(sd, "init_conv.weight"), _w(sd, "init_conv.bias"), padding=3)
    r = x
    t = time_embedding(sd, log_snr)
    if class_label is not None:
        t = t + class_embedding(sd, class_label)                         # :692-694, [1,512] broadcasts
    n_stage = len(cfg.dim_mults)
    attn = lambda full: full_attention if full else linear_attention
    skips: List[Tensor] = []
    for s in range(n_stage):
        p = f"downs.{s}"
        x = resnet_block(sd, p + ".0", x, t, cfg.groups); skips.append(x)
        x = resnet_block(sd, p + ".1", x, t, cfg.groups)
        x = attn(cfg.full_attn[s])(sd, p + ".2", x, cfg.heads, cfg.dim_head) + x
        skips.append(x)
        if s < n_stage - 1:
            x = space_to_depth_conv(sd, p + ".3", x)
        else:
            x = F.conv2d(x, _w(sd, p + ".3.weight"), _w(sd, p + ".3.bias"), padding=1)
    x = resnet_block(sd, "mid_block1", x, t, cfg.groups)
    x = full_attention(sd, "mid_attn", x, cfg.heads, cfg.dim_head) + x
    x = resnet_block(sd, "mid_block2", x, t, cfg.groups)
    for u in range(n_stage):
        p = f"ups.{u}"
        full = cfg.full_attn[n_stage - 1 - u]
        x = resnet_block(sd, p + ".0", torch.cat((x, skips.pop()), dim=1), t, cfg.groups)
        x = resnet_block(sd, p + ".1", torch.cat((x, skips.pop()), dim=1), t, cfg.groups)
        x = attn(full)(sd, p + ".2", x, cfg.heads, cfg.dim_head) + x
        if u < n_stage - 1:
            x = pixel_shuffle_up(sd, p + ".3", x)
        else:
            x = F.conv2d(x, _w(sd, p + ".3.weight"), _w(sd, p + ".3.bias"), padding=1)
    x = resnet_block(sd, "final_res_block", torch.cat((x, r), dim=1), t, cfg.groups)
    return F.conv2d(x, _w(sd, "final_conv.weight"), _w(sd, "final_conv.bias"))


def strip_model_prefix(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """The sampler's state_dict keys carry the 'model.' prefix."""
    return {(k[6:] if k.startswith("model.") else k): v for k, v in sd.items()}


# --------------------------------------------------------------------------------------
# sampler
# --------------------------------------------------------------------------------------
class NoiseSource:
    """Draws noise in the reference's order from torch's global CPU generator,
    and optionally records every draw so a device run can replay the identical stream."""

    def __init__(self, record: bool = False):
        self.record = record
        self.draws: List[Tensor] = []

    def randn(self, shape) -> Tensor:
        z = torch.randn(tuple(shape))
        if self.record:
            self.draws.append(z)
        return z


class ReplayNoise:
    def __init__(self, draws: Sequence[Tensor]):
        self.draws = list(draws)
        self.i = 0

    def randn(self, shape) -> Tensor:
        z = self.draws[self.i]
        self.i += 1
        assert tuple(z.shape) == tuple(shape), (tuple(z.shape), tuple(shape))
        return z


def predict_and_step(sd, cfg: UnetCfg, x: Tensor, t: Tensor, t_next: Tensor, cond: Tensor,
                     class_label: Optional[Tensor], cond_scale: float, class_cond_scale: float,
                     noise: "NoiseSource", trace: Optional[dict] = None):
    """model.py:3122-3188 p_mean_variance + p_sample for one minibatch of tiles."""
    s = step_scalars(t, t_next)
    ls = s["log_snr"].expand(x.shape[0])
    if cond_scale != 1.0 and class_cond_scale != 1.0:
        raise NotImplementedError(
            "Currently, you cannot specify both cond_scale and class_cond_scale at the same time.")
    if cond_scale != 1.0:                                               # :3147-3150
        a = unet_forward(sd, cfg, x, ls, class_label, cond)
        b = unet_forward(sd, cfg, x, ls, class_label, None)
        eps = b + (a - b) * cond_scale
    elif class_cond_scale != 1.0:                                       # :3151-3154
        a = unet_forward(sd, cfg, x, ls, class_label, cond)
        b = unet_forward(sd, cfg, x, ls, None, cond)
        eps = b + (a - b) * class_cond_scale
    else:                                                               # :3155-3156
        eps = unet_forward(sd, cfg, x, ls, class_label, cond)
    x0 = ((x - s["sigma"] * eps) / s["alpha"]).clamp(-1.0, 1.0)          # :3160-3163
    mean = s["alpha_next"] * (x * (1 - s["c"]) / s["alpha"] + s["c"] * x0)   # :3164
    if trace is not None:
        trace.setdefault("eps", []).append(eps.clone())
    if t_next == 0:                                                     # :3184-3185
        return mean, x0
    return mean + s["var"].sqrt() * noise.randn(x.shape), x0            # :3187-3188


def tiled_sample(sd, cfg: UnetCfg, condition_x: Tensor, class_label: Optional[Tensor] = None, *,
                 batch_size: int = 4, num_sample_steps: int = 50, cond_scale: float = 1.0,
                 guidance_start_steps: int = 0, class_cond_scale: float = 1.0,
                 class_guidance_start_steps: int = 0, tile: int = 256, generation_start_steps: int = 0,
                 start_white_noise: bool = True,
                 noise: Optional["NoiseSource"] = None, trace: Optional[dict] = None) -> Tensor:
    """model.py:3288-3413 tiled_sample.

    condition_x: [1,3,H,W] in [0,1].  Returns [1,3,H,W] in [0,1].
    """
    noise = noise or NoiseSource()
    cond = condition_x * 2 - 1                                           # :3296
    _, _, h, w = cond.shape
    (left, top, right, bottom), pad = canvas_box_and_pad(h, w)            # :3301 (tile fixed at 256)
    cond = F.pad(cond, pad, mode="reflect")                              # :3303
    if generation_start_steps > 0 or not start_white_noise:              # :3305-3308 / :3312-3315 q_sample start
        t0 = (1.0 - torch.tensor(generation_start_steps / num_sample_steps)) if generation_start_steps > 0 \
            else torch.tensor(1.0)
        ls0 = log_snr_linear(t0)
        img = cond * ls0.sigmoid().sqrt() + noise.randn(cond.shape) * (-ls0).sigmoid().sqrt()   # :3434-3442
    else:
        img = noise.randn(cond.shape)                                    # :3311
    steps = torch.linspace(1.0, 0.0, num_sample_steps + 1)               # :3325
    hp, wp = cond.shape[-2:]
    grids = sampling_grids(hp, wp, tile, tile)
    (il, it, ir, ib), ipad = grid_bbox(grids[1], hp, wp)                  # :3337
    cond = F.pad(cond[:, :, it:ib, il:ir], ipad, mode="constant", value=0.0)   # :3341-3342
    x_start = img.clone()
    for i in range(num_sample_steps):
        if i < generation_start_steps:                                   # :3347-3348
            continue
        cs = cond_scale if i >= guidance_start_steps else 1.0            # :3349-3356
        ccs = class_cond_scale if i >= class_guidance_start_steps else 1.0
        t, t_next = steps[i], steps[i + 1]
        boxes = grids[i % 2]
        for j in range(0, len(boxes), batch_size):                       # :3364-3390 minibatching
            chunk = boxes[j:j + batch_size]
            xb = torch.cat([img[:, :, a:b, c:d] for (a, b, c, d) in chunk], dim=0)
            cb = torch.cat([cond[:, :, a:b, c:d] for (a, b, c, d) in chunk], dim=0)
            out, x0 = predict_and_step(sd, cfg, xb, t, t_next, cb, class_label, cs, ccs, noise, trace)
            for k, (a, b, c, d) in enumerate(chunk):
                img[:, :, a:b, c:d] = out[k]
                x_start[:, :, a:b, c:d] = x0[k]
        if i % 2 == 1:                                                   # :3392-3396 ring re-noise
            inner = img[:, :, it:ib, il:ir].clone()
            sigma = (-log_snr_linear(t_next)).sigmoid().sqrt()
            img = noise.randn(cond.shape) * sigma                         # q_sample(0, t') :3434-3442
            img[:, :, it:ib, il:ir] = inner
        if trace is not None:
            trace.setdefault("img", []).append(img.clone())
            trace.setdefault("x_start", []).append(x_start.clone())
    out = img[:, :, top:bottom, left:right].clamp(-1.0, 1.0)             # :3403-3404
    return (out + 1) * 0.5                                               # :3405


# --------------------------------------------------------------------------------------
# EDM (Karras et al.) sampler over the same U-Net: ConditionalElucidatedDiffusionSR.tiled_sample
# (model.py:2309-2475, preconditioned_network_forward :2132-2183, get_noised_images :2185-2194).
# The schedule and the preconditioning coefficients live in the un-vendored base class
# ``denoising_diffusion_pytorch.ElucidatedDiffusion`` (pinned 1.8.15); they are RESTATED here from the
# published algorithm (parity unpinned at that boundary, like Attend) - anchored on the call sites above:
#   c_in = (sigma^2 + sigma_data^2)^-1/2, c_skip = sigma_data^2 / (sigma^2 + sigma_data^2),
#   c_out = sigma * sigma_data * (sigma^2 + sigma_data^2)^-1/2, c_noise = log(clamp(sigma, 1e-20)) / 4,
#   sigmas_i = (sigma_max^(1/rho) + i/(N-1) * (sigma_min^(1/rho) - sigma_max^(1/rho)))^rho, then a trailing 0.
# --------------------------------------------------------------------------------------
@dataclass
class EdmCfg:
    sigma_min: float = 0.002
    sigma_max: float = 80.0
    sigma_data: float = 0.5
    rho: float = 7.0
    S_churn: float = 80.0
    S_tmin: float = 0.05
    S_tmax: float = 50.0
    S_noise: float = 1.003
    num_sample_steps: int = 32      # the CONSTRUCTOR's step count: get_noised_images (model.py:2186-2189) falls back to it


def edm_sigmas(e: EdmCfg, n: int) -> Tensor:
    inv_rho = 1 / e.rho
    steps = torch.arange(n, dtype=torch.float32)
    sig = (e.sigma_max ** inv_rho + steps / (n - 1) * (e.sigma_min ** inv_rho - e.sigma_max ** inv_rho)) ** e.rho
    return F.pad(sig, (0, 1), value=0.0)


def edm_gammas(e: EdmCfg, sigmas: Tensor, n: int) -> Tensor:            # model.py:2333-2337
    return torch.where((sigmas >= e.S_tmin) & (sigmas <= e.S_tmax), min(e.S_churn / n, math.sqrt(2) - 1), 0.0)


def edm_precond_coeffs(e: EdmCfg, sigma: Tensor) -> Dict[str, Tensor]:
    sd2 = e.sigma_data ** 2
    return {"c_in": 1 * (sigma ** 2 + sd2) ** -0.5, "c_skip": sd2 / (sigma ** 2 + sd2),
            "c_out": sigma * e.sigma_data * (sd2 + sigma ** 2) ** -0.5,
            "c_noise": torch.log(sigma.clamp(min=1e-20)) * 0.25}


def edm_denoise(sd, cfg: UnetCfg, e: EdmCfg, x: Tensor, sigma: float, cond: Tensor, class_label: Optional[Tensor],
                cond_scale: float, class_cond_scale: float, clamp: bool) -> Tensor:
    """preconditioned_network_forward, model.py:2132-2183."""
    s = torch.full((x.shape[0],), sigma)
    c = edm_precond_coeffs(e, s)
    pad = lambda v: v.reshape(-1, 1, 1, 1)
    if cond_scale != 1.0 and class_cond_scale != 1.0:
        raise NotImplementedError("Currently, you cannot specify both cond_scale and class_cond_scale at the same time.")
    xin = pad(c["c_in"]) * x
    out = pad(c["c_skip"]) * x + pad(c["c_out"]) * unet_forward(sd, cfg, xin, c["c_noise"], class_label, cond)
    if cond_scale != 1.0:
        null = pad(c["c_skip"]) * x + pad(c["c_out"]) * unet_forward(sd, cfg, xin, c["c_noise"], class_label, None)
        out = null + (out - null) * cond_scale
    if class_cond_scale != 1.0:
        null = pad(c["c_skip"]) * x + pad(c["c_out"]) * unet_forward(sd, cfg, xin, c["c_noise"], None, cond)
        out = null + (out - null) * class_cond_scale
    return out.clamp(-1.0, 1.0) if clamp else out


def edm_tiled_sample(sd, cfg: UnetCfg, e: EdmCfg, condition_x: Tensor, class_label: Optional[Tensor] = None, *,
                     batch_size: int = 4, num_sample_steps: int = 32, cond_scale: float = 1.0,
                     guidance_start_steps: int = 0, class_cond_scale: float = 1.0, class_guidance_start_steps: int = 0,
                     generation_start_steps: int = 0, clamp: bool = True, zero_init: bool = False, tile: int = 256,
                     noise: Optional["NoiseSource"] = None) -> Tensor:
    """model.py:2309-2475.  condition_x [1,3,H,W] in [0,1] -> [1,3,H,W] in [0,1]."""
    noise = noise or NoiseSource()
    n = num_sample_steps
    cond = condition_x * 2 - 1
    _, _, h, w = cond.shape
    (left, top, right, bottom), pad = canvas_box_and_pad(h, w)
    cond = F.pad(cond, pad, mode="reflect")
    shape = cond.shape
    sigmas = edm_sigmas(e, n)
    gammas = edm_gammas(e, sigmas, n)
    # get_noised_images is called WITHOUT num_sample_steps (:2342, :2457): its sigmas are the constructor's schedule
    noised_sigmas = edm_sigmas(e, e.num_sample_steps)
    if generation_start_steps > 0:                                       # get_noised_images(cond, step) :2185-2194
        img = cond + noised_sigmas[generation_start_steps] * noise.randn(shape)
    elif zero_init:
        img = torch.zeros(shape)
    else:
        img = sigmas[0] * noise.randn(shape)
    hp, wp = shape[-2:]
    grids = sampling_grids(hp, wp, tile, tile)
    (il, it, ir, ib), ipad = grid_bbox(grids[1], hp, wp)
    cond = F.pad(cond[:, :, it:ib, il:ir], ipad, mode="constant", value=0.0)
    for i in range(n):
        if i < generation_start_steps:
            continue
        cs = cond_scale if i >= guidance_start_steps else 1.0
        ccs = class_cond_scale if i >= class_guidance_start_steps else 1.0
        sigma, sigma_next, gamma = sigmas[i].item(), sigmas[i + 1].item(), gammas[i].item()
        eps = e.S_noise * noise.randn(shape)                             # :2386
        sigma_hat = sigma + gamma * sigma
        img_hat = img + math.sqrt(sigma_hat ** 2 - sigma ** 2) * eps     # :2389
        boxes = grids[i % 2]
        for j in range(0, len(boxes), batch_size):
            chunk = boxes[j:j + batch_size]
            xb = torch.cat([img_hat[:, :, a:b, c:d] for (a, b, c, d) in chunk], dim=0)
            cb = torch.cat([cond[:, :, a:b, c:d] for (a, b, c, d) in chunk], dim=0)
            out = edm_denoise(sd, cfg, e, xb, sigma_hat, cb, class_label, cs, ccs, clamp)
            d = (xb - out) / sigma_hat                                   # :2406
            nxt = xb + (sigma_next - sigma_hat) * d                      # :2407
            if sigma_next != 0:                                          # :2409-2414 Heun correction
                out2 = edm_denoise(sd, cfg, e, nxt, sigma_next, cb, class_label, cs, ccs, clamp)
                d2 = (nxt - out2) / sigma_next
                nxt = xb + 0.5 * (sigma_next - sigma_hat) * (d + d2)
            for k, (a, b, c, d_) in enumerate(chunk):
                img[:, :, a:b, c:d_] = nxt[k]
        if i % 2 == 1:                                                   # :2448-2452
            inner = img[:, :, it:ib, il:ir].clone()
            img = torch.zeros(shape) + noised_sigmas[i] * noise.randn(shape)    # get_noised_images(0, i)
            img[:, :, it:ib, il:ir] = inner
    out = img[:, :, top:bottom, left:right].clamp(-1.0, 1.0)
    return (out + 1) * 0.5


# --------------------------------------------------------------------------------------
# un-tiled sampling: ConditionalContinuousTimeGaussianDiffusionSR.sample / p_sample_loop (model.py:3191-3247, :3417-3432)
# --------------------------------------------------------------------------------------
def sample(sd, cfg: UnetCfg, condition_x: Tensor, class_label: Optional[Tensor] = None, *, num_sample_steps: int = 50,
           cond_scale: float = 1.0, guidance_start_steps: int = 0, class_cond_scale: float = 1.0,
           class_guidance_start_steps: int = 0, generation_start_steps: int = 0,
           noise: Optional["NoiseSource"] = None) -> Tensor:
    """condition_x: [B,3,S,S] in [0,1] (S = image_size); every image of the batch has its own noise (one randn over the
    whole batch tensor per draw).  Returns [B,3,S,S] in [0,1]."""
    noise = noise or NoiseSource()
    cond = condition_x * 2 - 1                                           # :3424
    if generation_start_steps > 0:                                       # :3198-3201 q_sample(condition, t_start)
        ls0 = log_snr_linear(1.0 - torch.tensor(generation_start_steps / num_sample_steps))
        img = cond * ls0.sigmoid().sqrt() + noise.randn(cond.shape) * (-ls0).sigmoid().sqrt()
    else:
        img = noise.randn(cond.shape)                                    # :3203
    steps = torch.linspace(1.0, 0.0, num_sample_steps + 1)
    for i in range(num_sample_steps):
        if i < generation_start_steps:
            continue
        cs = cond_scale if i >= guidance_start_steps else 1.0
        ccs = class_cond_scale if i >= class_guidance_start_steps else 1.0
        img, _ = predict_and_step(sd, cfg, img, steps[i], steps[i + 1], cond, class_label, cs, ccs, noise)
    return (img.clamp(-1.0, 1.0) + 1) * 0.5                              # :3239-3240


# --------------------------------------------------------------------------------------
# un-tiled EDM sampling: ConditionalElucidatedDiffusionSR.sample -> sample_org (model.py:2196-2306, Heun) or
# sample_using_dpmpp (model.py:2479-2557, DPM-Solver++(2M)) when the wrapper was built with use_dpmpp_solver
# --------------------------------------------------------------------------------------
def _edm_start(e: EdmCfg, cond: Tensor, sigmas: Tensor, generation_start_steps: int, zero_init: bool, noise) -> Tensor:
    if generation_start_steps > 0:                                       # get_noised_images, ctor schedule (:2186-2194)
        return cond + edm_sigmas(e, e.num_sample_steps)[generation_start_steps] * noise.randn(cond.shape)
    if zero_init:
        return torch.zeros(cond.shape)
    return sigmas[0] * noise.randn(cond.shape)


def edm_sample(sd, cfg: UnetCfg, e: EdmCfg, condition_x: Tensor, class_label: Optional[Tensor] = None, *,
               num_sample_steps: int = 32, cond_scale: float = 1.0, guidance_start_steps: int = 0,
               class_cond_scale: float = 1.0, class_guidance_start_steps: int = 0, generation_start_steps: int = 0,
               clamp: bool = True, zero_init: bool = False, noise: Optional["NoiseSource"] = None) -> Tensor:
    """sample_org, model.py:2212-2306.  condition_x [B,3,h,w] in [0,1]; one randn over the batch tensor per draw."""
    noise = noise or NoiseSource()
    n = num_sample_steps
    cond = condition_x * 2 - 1                                           # :2223
    sigmas = edm_sigmas(e, n)
    gammas = edm_gammas(e, sigmas, n)
    img = _edm_start(e, cond, sigmas, generation_start_steps, zero_init, noise)
    for i in range(n):
        if i < generation_start_steps:
            continue
        cs = cond_scale if i >= guidance_start_steps else 1.0
        ccs = class_cond_scale if i >= class_guidance_start_steps else 1.0
        sigma, sigma_next, gamma = sigmas[i].item(), sigmas[i + 1].item(), gammas[i].item()
        eps = e.S_noise * noise.randn(cond.shape)                        # :2269
        sigma_hat = sigma + gamma * sigma
        img_hat = img + math.sqrt(sigma_hat ** 2 - sigma ** 2) * eps     # :2272
        out = edm_denoise(sd, cfg, e, img_hat, sigma_hat, cond, class_label, cs, ccs, clamp)
        d = (img_hat - out) / sigma_hat                                  # :2276
        nxt = img_hat + (sigma_next - sigma_hat) * d                     # :2278
        if sigma_next != 0:                                              # :2282-2286
            out2 = edm_denoise(sd, cfg, e, nxt, sigma_next, cond, class_label, cs, ccs, clamp)
            d2 = (nxt - out2) / sigma_next
            nxt = img_hat + 0.5 * (sigma_next - sigma_hat) * (d + d2)
        img = nxt
    return (img.clamp(-1.0, 1.0) + 1) * 0.5                              # :2298, :2306


def edm_sample_dpmpp(sd, cfg: UnetCfg, e: EdmCfg, condition_x: Tensor, class_label: Optional[Tensor] = None, *,
                     num_sample_steps: int = 32, cond_scale: float = 1.0, guidance_start_steps: int = 0,
                     class_cond_scale: float = 1.0, class_guidance_start_steps: int = 0, generation_start_steps: int = 0,
                     clamp: bool = True, zero_init: bool = False, noise: Optional["NoiseSource"] = None) -> Tensor:
    """sample_using_dpmpp, model.py:2479-2557: one U-Net evaluation per step, a two-step multistep update in t = -log sigma."""
    noise = noise or NoiseSource()
    n = num_sample_steps
    cond = condition_x * 2 - 1                                           # :2494
    sigmas = edm_sigmas(e, n)
    img = _edm_start(e, cond, sigmas, generation_start_steps, zero_init, noise)
    t_of = lambda s: s.log().neg()                                       # :2513-2514
    sigma_of = lambda t: t.neg().exp()
    old = None
    for i in range(n):
        if i < generation_start_steps:
            continue
        cs = cond_scale if i >= guidance_start_steps else 1.0
        ccs = class_cond_scale if i >= class_guidance_start_steps else 1.0
        den = edm_denoise(sd, cfg, e, img, sigmas[i].item(), cond, class_label, cs, ccs, clamp)   # :2528
        t, t_next = t_of(sigmas[i]), t_of(sigmas[i + 1])
        h = t_next - t
        if old is None or sigmas[i + 1] == 0:                            # :2533
            den_d = den
        else:
            r = (t - t_of(sigmas[i - 1])) / h                            # :2536-2537
            g = -1 / (2 * r)
            den_d = (1 - g) * den + g * old                              # :2539
        img = (sigma_of(t_next) / sigma_of(t)) * img - (-h).expm1() * den_d   # :2541
        old = den
    return (img.clamp(-1.0, 1.0) + 1) * 0.5                              # :2549, :2557
