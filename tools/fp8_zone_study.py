"""fp8 mode: which U-Net zones cost the PSNR?  (GPU box)  Zones: 0-3 down stages, 4 middle, 5-8 up stages, 9 final block.
Runs BASELINE configs[4] (1024^2, 100 steps, class CFG 2.0) in bf16 and in fp8 with selected zones kept on the bf16 3x3
kernel (SRGD_FP8_BF16_ZONES bit mask), same device noise; prints PSNR vs bf16 and the time per run."""
import json, logging, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srgd_amd.config import load_config
from srgd_amd.model import get_model
from srgd_amd.synth import synth_state_dict, synthetic_lr_condition

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
conf = load_config(os.path.join(ROOT, "conf", "conditional_continuous_linear_df8kost_dim128.yaml"))
sampler = get_model(conf, logging.getLogger("z")).module
schema = {k: tuple(v.shape) for k, v in sampler.state_dict().items()}
sampler.load_state_dict(synth_state_dict(schema, seed=0), strict=True)
sampler = sampler.eval().cuda()
sampler.noise_source = "device"
cond = torch.cat([synthetic_lr_condition(i, 256, 256) for i in range(5)]).cuda()
label = torch.tensor([0]).cuda()
steps = int(os.environ.get("STEPS", "100"))

def run(prec):
    sampler.device_noise_seed = 71
    sampler.tiled_sample(batch_size=125, condition_x=cond, class_label=label, class_cond_scale=2.0, num_sample_steps=4, precision=prec)
    torch.cuda.synchronize(); t0 = time.time()
    sampler.device_noise_seed = 71
    out = sampler.tiled_sample(batch_size=125, condition_x=cond, class_label=label, class_cond_scale=2.0, num_sample_steps=steps, precision=prec)
    torch.cuda.synchronize()
    return out.cpu(), time.time() - t0

ref, t_bf = run("bf16")
print(f"bf16: {t_bf:.2f} s for 5 HR tiles")
rows = []
extra = [(f"mask {int(m, 0)}", int(m, 0)) for m in os.environ.get("MASKS", "").split(",") if m]
for name, mask in extra or [("fp8 everywhere", 0), ("final block bf16", 1 << 9), ("final + up3 bf16", (1 << 9) | (1 << 8)), ("down0 + final + up3 bf16", 1 | (1 << 9) | (1 << 8)),
                   ("all 256^2 + 128^2 zones bf16", 1 | 2 | (1 << 7) | (1 << 8) | (1 << 9)), ("only middle+32^2/64^2 fp8", 1 | 2 | (1 << 7) | (1 << 8) | (1 << 9))][:len(extra) or 5]:
    os.environ["SRGD_FP8_BF16_ZONES"] = str(mask)
    sampler.model._invalidate_engines()
    out, t = run("fp8")
    mse = float(((out - ref) ** 2).mean())
    psnr = 10 * np.log10(1.0 / max(mse, 1e-20))
    rows.append(dict(zones_bf16=name, mask=mask, psnr_db_vs_bf16=round(psnr, 2), seconds_5_hr_tiles=round(t, 2), speedup_vs_bf16=round(t_bf / t, 3)))
    print(rows[-1], flush=True)
os.environ.pop("SRGD_FP8_BF16_ZONES", None)
json.dump(dict(bf16_seconds=t_bf, rows=rows), open(os.path.join(ROOT, "gpurun_out", "fp8_zone_study_extra.json" if extra else "fp8_zone_study.json"), "w"), indent=1)
