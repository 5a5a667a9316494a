"""Configuration object of the sampling path - same surface as the reference ``config.py``:
``load_config(path) -> Config`` accepts the shipped ``conf/*.yaml`` unchanged, every YAML key must be
a ``Config`` field (an unknown key raises ``TypeError`` exactly as a dataclass constructor does,
reference config.py:191-194), and values are stored untyped as PyYAML parsed them.

Only the fields in ``INFERENCE_FIELDS`` influence this engine; the rest are the reference's
training/validation knobs (config.py:123-188), kept so that its YAML files load and so that code
written against ``conf.<field>`` keeps working.
"""
from __future__ import annotations

import dataclasses

import yaml

# fields the sampler / factory read (reference model.py:3500-3514, :3634-3650, :3656-3664)
INFERENCE_FIELDS = dict(
    model="continuous", noise_schedule="linear", num_sample_steps=32, clip_sample_denoised=True,
    image_size=128, unet_dim=64, ddpm_unet_dim_mults="1,2,4,8", full_attn="False,False,False,True",
    learned_variance=False, learned_sinusoidal_cond=True, learned_sinusoidal_dim=32, flash_attn=False,
    pixel_shuffle_upsample=True, num_classes=3, ckpt_path="", load_strict=True, ema_decay=0.995,
    cond_scale=1.0, class_cond_scale=1.0, test_label=0, guidance_start_steps=0,
    class_guidance_start_steps=0, generation_start_steps=0, use_dpmpp_solver=True, seed=71, amp=False,
    cond_drop_prob=0.1, class_cond_drop_prob=0.1, loss_type="l2", min_snr_loss_weight=False,
    min_snr_gamma=5, learned_schedule_net_hidden_dim=1024, learned_noise_schedule_frac_gradient=1.0,
)

# everything else the reference dataclass declares (not used by this engine)
_OTHER_FIELDS = dict(
    save_dir="srgd", prefix="conditional_continuous_linear", base_dir="./input/",
    dataset_name="cropped_df2kost_400x400_overlap200", conditional_task_type="realsr_denoise_sr",
    objective="pred_noise", beta_schedule="linear", timesteps=1000, sampling_timesteps=250,
    offset_noise_strength=0.0, sigma_min=0.002, sigma_max=80, sigma_data=0.5, rho=7, P_mean=-1.2, P_std=1.2,
    S_churn=80, S_tmin=0.05, S_tmax=50, S_noise=1.003, val_num_sample_steps=32, n_fold=10, train_fold="0",
    skip_sample=False, skip_val=False, validation_ratio=0.5, val_realsrv3=False, val_drealsr=False,
    val_realsrv3_scale=4, val_drealsr_scale=4, crop_size=256, hr_image_size=256, lr_image_size=128,
    crop_rate=2, scale_size=256, crop_size_limit=False, batch_size=32, sample_size=16, hflip=False,
    rotate=False, interpolation="BICUBIC", shuffle=True, torch_compile=False, amp_dtype="float16",
    ema_device="cuda", optimizer="adamw", lr=1e-4, min_lr=1e-4, weight_decay=0.0, momentum=0.9,
    nesterov=False, amsgrad=False, madgrad_decoupled_decay=True, epochs=300, warmup_epochs=0,
    warmup_lr_init=1e-6, plateau_mode="min", factor=0.1, patience=4, plateau_eps=1e-8, scheduler="cosine",
    cosine_interval_type="step", train_preprocess="randomcrop", valid_preprocess="centercrop",
    train_trans_mode="realesrgan", valid_trans_mode="simple", usm_sharpener=False, blur_prob=0.5,
    advance_blur_prob=0.5, gaussian_blur_prob=0.5, sinc_blur_prob=0.5, sinc_blur_factor_min=0.9,
    sinc_blur_factor_max=1.1, image_compression_prob=0.5, quality_lower=50, quality_upper=100,
    noise_prob=0.5, gauss_noise_prob=0.5, iso_noise_prob=0.5, multiplicative_noise_prob=0.5, train=True,
    test=False, debug=False, save_validation_sample=False, save_validation_hr_sample=False,
    save_every_epoch=False, test_target="best_loss", num_workers=4, device="cuda", pin_memory=True,
    model_dir="models", log_dir="logs", print_freq=0,
)

Config = dataclasses.make_dataclass(
    "Config",
    [(name, type(default), dataclasses.field(default=default))
     for name, default in {**INFERENCE_FIELDS, **_OTHER_FIELDS}.items()],
)
Config.__doc__ = "Flat run configuration (field names and defaults of the reference Config)."


def load_config(config_file: str) -> "Config":
    with open(config_file, "r") as fp:
        opts = yaml.safe_load(fp) or {}
    return Config(**opts)
