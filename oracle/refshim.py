"""Import the reference ``model.py``/``config.py`` in the BUILD CONTAINER ONLY (oracle tooling).

``/root/reference/model.py`` needs three packages that are neither installed nor installable
here: ``denoising_diffusion_pytorch`` (only ``Unet.downsample_factor``
and ``attend.Attend`` are used on the hot path), ``timm.utils.ModelEmaV2`` (a deepcopy shell)
and, for ``inference.py``, ``torchvision``/``logzero`` (not needed here).  This helper writes
minimal stand-ins for the first two into a temporary directory, puts it and the reference on
``sys.path`` and returns the imported reference modules.  Nothing from the reference is copied
into this repository, and nothing here travels to the GPU box (``/root/reference`` is absent
there; callers must treat ``load_reference() is None`` as "not available").
"""
from __future__ import annotations

import importlib
import os
import sys
import tempfile
import textwrap

REFERENCE_DIR = "/root/reference"

_DDP_INIT = '''
import torch.nn as nn

class Unet(nn.Module):
    """Stand-in: the reference overwrites every sub-module of the stock U-Net (model.py:583-675);
    only this property survives (used by the assert at model.py:679)."""
    def __init__(self, *args, **kwargs):
        super().__init__()
    @property
    def downsample_factor(self):
        return 2 ** (len(self.downs) - 1)

class GaussianDiffusion(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()

class ElucidatedDiffusion(nn.Module):
    """Published algorithm of denoising-diffusion-pytorch==1.8.15 elucidated_diffusion.ElucidatedDiffusion, reduced to
    what the reference's ConditionalElucidatedDiffusionSR.tiled_sample calls (model.py:2132-2194, :2309-2475):
    the Karras preconditioning coefficients and the rho-schedule."""
    def __init__(self, net, *, image_size, channels=3, num_sample_steps=32, sigma_min=0.002, sigma_max=80,
                 sigma_data=0.5, rho=7, P_mean=-1.2, P_std=1.2, S_churn=80, S_tmin=0.05, S_tmax=50, S_noise=1.003):
        super().__init__()
        self.net = net
        self.channels, self.image_size = channels, image_size
        self.sigma_min, self.sigma_max, self.sigma_data, self.rho = sigma_min, sigma_max, sigma_data, rho
        self.P_mean, self.P_std, self.num_sample_steps = P_mean, P_std, num_sample_steps
        self.S_churn, self.S_tmin, self.S_tmax, self.S_noise = S_churn, S_tmin, S_tmax, S_noise
    @property
    def device(self):
        return next(self.net.parameters()).device
    def c_skip(self, sigma):
        return (self.sigma_data ** 2) / (sigma ** 2 + self.sigma_data ** 2)
    def c_out(self, sigma):
        return sigma * self.sigma_data * (self.sigma_data ** 2 + sigma ** 2) ** -0.5
    def c_in(self, sigma):
        return 1 * (sigma ** 2 + self.sigma_data ** 2) ** -0.5
    def c_noise(self, sigma):
        import torch
        return torch.log(sigma.clamp(min=1e-20)) * 0.25
    def sample_schedule(self, num_sample_steps=None):
        import torch
        import torch.nn.functional as F
        num_sample_steps = self.num_sample_steps if num_sample_steps is None else num_sample_steps
        N = num_sample_steps
        inv_rho = 1 / self.rho
        steps = torch.arange(num_sample_steps, device=self.device, dtype=torch.float32)
        sigmas = (self.sigma_max ** inv_rho + steps / (N - 1) * (self.sigma_min ** inv_rho - self.sigma_max ** inv_rho)) ** self.rho
        return F.pad(sigmas, (0, 1), value=0.)
'''

_DDP_ATTEND = '''
import torch
import torch.nn as nn

class Attend(nn.Module):
    """Published algorithm of denoising-diffusion-pytorch==1.8.15 attend.Attend with flash=False,
    dropout 0: softmax(q k^T * d^-0.5) v on [b, h, n, d] tensors."""
    def __init__(self, dropout=0.0, flash=False, scale=None):
        super().__init__()
        assert not flash
        self.scale = scale
    def forward(self, q, k, v):
        scale = self.scale if self.scale is not None else q.shape[-1] ** -0.5
        sim = torch.einsum("b h i d, b h j d -> b h i j", q, k) * scale
        attn = sim.softmax(dim=-1)
        return torch.einsum("b h i j, b h j d -> b h i d", attn, v)
'''

_TIMM_UTILS = '''
from copy import deepcopy
import torch.nn as nn

class ModelEmaV2(nn.Module):
    def __init__(self, model, decay=0.9999, device=None):
        super().__init__()
        self.module = deepcopy(model)
        self.module.eval()
        self.decay = decay
        self.device = device
'''


def load_reference():
    """Returns (model_module, config_module) of the reference, or None when it is not present."""
    if not os.path.isfile(os.path.join(REFERENCE_DIR, "model.py")):
        return None
    shim = tempfile.mkdtemp(prefix="srgd_refshim_")
    os.makedirs(os.path.join(shim, "denoising_diffusion_pytorch"))
    os.makedirs(os.path.join(shim, "timm"))
    with open(os.path.join(shim, "denoising_diffusion_pytorch", "__init__.py"), "w") as f:
        f.write(textwrap.dedent(_DDP_INIT))
    with open(os.path.join(shim, "denoising_diffusion_pytorch", "attend.py"), "w") as f:
        f.write(textwrap.dedent(_DDP_ATTEND))
    with open(os.path.join(shim, "timm", "__init__.py"), "w") as f:
        f.write("")
    with open(os.path.join(shim, "timm", "utils.py"), "w") as f:
        f.write(textwrap.dedent(_TIMM_UTILS))
    # stubs first, then the reference itself; never cached as bytecode into /root/reference
    sys.dont_write_bytecode = True
    sys.path.insert(0, REFERENCE_DIR)
    sys.path.insert(0, shim)
    saved = {k: sys.modules.pop(k) for k in ("model", "config") if k in sys.modules}
    try:
        ref_model = importlib.import_module("model")
        ref_config = importlib.import_module("config")
    finally:
        sys.path.remove(REFERENCE_DIR)
        for k in ("model", "config"):
            m = sys.modules.pop(k, None)
            if k in saved:
                sys.modules[k] = saved[k]
    ref_model.tqdm = lambda it, **kw: it      # quiet progress bars
    return ref_model, ref_config


class _Logger:
    def info(self, *a, **k):
        pass


def build_reference_sampler(ref_model, ref_config, *, dim=None, dim_mults=None, num_sample_steps=50, model=None,
                            yaml_path=os.path.join(REFERENCE_DIR, "conf",
                                                   "conditional_continuous_linear_df8kost_dim128.yaml")):
    """get_model(conf) exactly as inference.py:147-156 does, optionally with a smaller width; ``model`` overrides
    conf.model (e.g. 'conditional_elucidated' for the EDM wrapper, model.py:3593-3614)."""
    conf = ref_config.load_config(yaml_path)
    conf.num_sample_steps = num_sample_steps
    conf.ckpt_path = ""
    if model is not None:
        conf.model = model
    if dim is not None:
        conf.unet_dim = dim
    if dim_mults is not None:
        conf.ddpm_unet_dim_mults = ",".join(str(m) for m in dim_mults)
        conf.full_attn = ",".join(["False"] * (len(dim_mults) - 1) + ["True"])
    ema = ref_model.get_model(conf, _Logger())
    return ema.module.eval(), conf
