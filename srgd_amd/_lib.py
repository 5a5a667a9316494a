"""ctypes binding of ``libsrgd_hip.so`` (C ABI: include/srgd_hip.h).

There is deliberately no fallback: if the library is missing or a call fails, the product
path raises.  The CPU oracle under ``oracle/`` is test infrastructure and is never imported here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRGD_HIP_LIB") or os.path.join(_HERE, "libsrgd_hip.so")   # override: A/B builds only

MAX_STAGES = 8
PRECISION_FP32 = 0
PRECISION_BF16 = 1
PRECISION_BF16_W8 = 2    # bf16 kernels, conv weights rounded through fp8 e4m3 (per-output-channel scale)
PRECISION_FP8 = 3        # 3x3 convolutions on the block-scaled MX-fp8 matrix cores (e4m3 + E8M0 per 32 channels), rest bf16
PRECISION_F16X3 = 5      # fp32 tensors, convolutions as three f16 MFMAs per product on (hi, lo) operand pairs: <= 1e-3 parity like fp32
PRECISION_F16MX2 = 6     # prototype: F16X3 with the 3x3 convolutions' cross terms on MX-fp8 operands (conv3x3_mx2.hip)
PRECISION_FP8_MIXED = 4  # the same below the top resolution; the 256x256-resolution zones keep bf16 3x3 convolutions


class UnetConfig(C.Structure):
    _fields_ = [("dim", C.c_int32), ("n_stages", C.c_int32),
                ("dim_mults", C.c_int32 * MAX_STAGES), ("full_attn", C.c_int32 * MAX_STAGES),
                ("channels", C.c_int32), ("groups", C.c_int32), ("heads", C.c_int32),
                ("dim_head", C.c_int32), ("sinus_dim", C.c_int32), ("num_classes", C.c_int32),
                ("precision", C.c_int32), ("device", C.c_int32)]


class StepScalars(C.Structure):
    _fields_ = [("alpha", C.c_float), ("sigma", C.c_float), ("alpha_next", C.c_float), ("c", C.c_float),
                ("one_minus_c", C.c_float), ("noise_scale", C.c_float), ("sigma_next", C.c_float),
                ("reserved", C.c_float)]


class EdmScalars(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("s_noise", "hat_coef", "sigma_hat", "sigma_next", "dt", "half_dt", "c_in_hat",
                                         "c_skip_hat", "c_out_hat", "c_in_next", "c_skip_next", "c_out_next", "ring_sigma",
                                         "clamp", "dpm_gamma", "pad1")]


class SamplerGeometry(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("Hp", C.c_int32), ("Wp", C.c_int32),
                ("left", C.c_int32), ("top", C.c_int32),
                ("inner_l", C.c_int32), ("inner_t", C.c_int32), ("inner_r", C.c_int32), ("inner_b", C.c_int32),
                ("tile", C.c_int32), ("n_even", C.c_int32), ("n_odd", C.c_int32), ("n_images", C.c_int32)]


# name -> (restype, argtypes); every symbol include/srgd_hip.h declares
PROTOTYPES = {
    "srgd_last_error": (C.c_char_p, []),
    "srgd_version": (C.c_char_p, []),
    "srgd_create": (C.c_int, [C.POINTER(UnetConfig), C.POINTER(C.c_void_p)]),
    "srgd_destroy": (C.c_int, [C.c_void_p]),
    "srgd_num_weights": (C.c_int, [C.c_void_p]),
    "srgd_weight_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int)]),
    "srgd_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "srgd_finalize_weights": (C.c_int, [C.c_void_p]),
    "srgd_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "srgd_sampler_begin": (C.c_int, [C.c_void_p, C.POINTER(SamplerGeometry), C.c_void_p, C.c_void_p,
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.POINTER(StepScalars),
                                     C.POINTER(C.c_float), C.c_int, C.c_void_p]),
    "srgd_sampler_step": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_float, C.c_int, C.c_uint64, C.c_void_p]),
    "srgd_sampler_step_tiles": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_uint64,
                                          C.c_void_p]),
    "srgd_sampler_exchange_tiles": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_void_p]),
    "srgd_sampler_unpack_gathered": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_void_p]),
    "srgd_edm_begin": (C.c_int, [C.c_void_p, C.POINTER(SamplerGeometry), C.c_void_p, C.c_void_p, C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int32), C.c_int, C.POINTER(EdmScalars), C.POINTER(C.c_float), C.c_int,
                                 C.c_void_p]),
    "srgd_edm_step": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int, C.c_int, C.c_float, C.c_int, C.c_uint64, C.c_void_p]),
    "srgd_edm_step_tiles": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_uint64, C.c_void_p]),
    "srgd_edm_dpmpp_step": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_float, C.c_int, C.c_void_p]),
    "srgd_sampler_q_start": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_uint64,
                                       C.c_void_p]),
    "srgd_sampler_end": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srgd_quantize_e4m3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float]),
    "srgd_image_resize_bicubic_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "srgd_image_unit_to_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "srgd_randn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_void_p]),
    "srgd_profile_begin": (C.c_int, [C.c_void_p]),
    "srgd_profile_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                   C.c_int]),
    "srgd_profile_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int]),
    "srgd_profile_num_families": (C.c_int, []),
    "srgd_profile_family_name": (C.c_char_p, [C.c_int]),
    "srgd_device_bytes_in_use": (C.c_int64, [C.c_void_p]),
    # include/srgd_hip_kernels.h
    "srgd_k_conv2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int, C.c_int, C.c_void_p]),
    "srgd_k_conv2d_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                      C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srgd_k_groupnorm_silu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "srgd_k_rmsnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                 C.c_void_p]),
    "srgd_k_linear_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "srgd_k_full_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "srgd_k_quant_mxfp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "srgd_k_conv3x3_mxfp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float),
                                       C.POINTER(C.c_int), C.c_void_p]),
    "srgd_k_conv1x1_split_rms": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "srgd_k_linattn_block_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


class SrgdHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load the engine library (once).  Raises if it has not been built - no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SrgdHipError(
                f"{LIB_PATH} is missing: build it with `python -m srgd_amd.build` (hipcc, gfx950). "
                "The MI355X engine is the only implementation of this path; there is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)            # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().srgd_last_error().decode(errors="replace")
        raise SrgdHipError(f"{what}: {msg}" if what else msg)
