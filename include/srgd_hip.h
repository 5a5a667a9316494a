/* libsrgd_hip - C ABI of the MI355X (gfx950) engine for the Real-SRGD sampling hot path.
 *
 * The reference (yahoojapan/srgd) is pure Python/PyTorch: it has no FFI layer of its own, so
 * this header IS the boundary a maintainer binds (ctypes stub: INTEGRATION.md).  Each entry
 * point names the reference interface it stands behind (file:line in /root/reference).
 *
 * Conventions: plain C, no torch types.  Device pointers are raw HBM addresses owned by the
 * caller (e.g. tensor.data_ptr() of PyTorch-ROCm tensors); "host" pointers are ordinary CPU
 * memory.  Image tensors are fp32 NCHW exactly as the reference holds them.  Every call
 * returns 0 on success, <0 on error (message: srgd_last_error()).  Calls that take a
 * `stream` (a hipStream_t passed as void*) are asynchronous on it, never synchronise and never
 * allocate after the first call with a given shape.  One engine per device, not thread-safe.
 */
#ifndef SRGD_HIP_H
#define SRGD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only the entry points declared here are exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef struct srgd_engine srgd_engine;

#define SRGD_MAX_STAGES 8
#define SRGD_PRECISION_FP32 0 /* fp32 activations, exact-fp32 MFMA: parity mode (<=1e-3 vs reference) */
#define SRGD_PRECISION_BF16 1 /* bf16 activations/weights, fp32 accumulate: throughput mode */
#define SRGD_PRECISION_BF16_W8 2 /* BF16 mode with every convolution weight (3x3, 1x1, attention to_qkv / to_out, resamplers,
                                  * input / output conv) rounded through fp8 e4m3 with one scale per output channel
                                  * (max|w| -> 448) when the weights are packed: the NUMERICS of fp8 weights (BASELINE
                                  * configs[4]'s parity-vs-bf16 check) on the bf16 MFMA kernels */
#define SRGD_PRECISION_FP8 3 /* BASELINE configs[4] compute path: every 3x3 convolution (Block.proj model.py:246 and the
                              * last-stage resamplers, 89 % of the FLOPs) runs on the block-scaled MX matrix cores
                              * (v_mfma_scale_f32_16x16x128_f8f6f4): OCP e4m3 activations and weights, one E8M0 scale per
                              * 32 input channels (activations: per pixel; weights: per output channel and tap), fp32
                              * accumulate, bf16 out.  Pointwise layers, attention, norms and the sampler stay as in
                              * SRGD_PRECISION_BF16.  Needs channel counts that are multiples of 128 (the shipped dim-128
                              * U-Net); other 3x3 layers fall back to the bf16 kernels. */
#define SRGD_PRECISION_FP8_MIXED 4 /* SRGD_PRECISION_FP8 below the top resolution only: the first down stage, the last up stage
                                    * and the final ResnetBlock (everything at the tile's own 256x256 resolution) keep their 3x3
                                    * convolutions on the bf16 kernel.  Measured on BASELINE configs[4]: 53.5 dB vs the bf16
                                    * engine (all-fp8: 36.6 dB) at 1.16x its throughput (all-fp8: 1.32x). */

#define SRGD_PRECISION_F16X3 5 /* split-operand precision: fp32 activations, statistics, attention and sampler arithmetic exactly as
                                * SRGD_PRECISION_FP32; the convolutions (Block.proj model.py:246, res_conv :271, to_qkv / to_out, the
                                * resamplers) contract on the 16-bit matrix cores with every operand carried as an f16 (hi, lo) pair -
                                * x*w ~= x_hi*w_hi + x_lo*w_hi + x_hi*w_lo, fp32 accumulate, weights pre-scaled by a power of two
                                * per layer: 2^-22 per product instead of fp32's 2^-24 (plain bf16: 2^-9).  Meets the 1e-3 parity
                                * bar like SRGD_PRECISION_FP32 at three 16-bit MFMAs per product instead of the 1/16-rate fp32
                                * MFMA.  Layers whose input channels are not a multiple of 32 stay on the exact-fp32 kernel. */
#define SRGD_PRECISION_F16MX2 6 /* PROTOTYPE: SRGD_PRECISION_F16X3 with the 3x3 convolutions' two cross terms (x_lo.w_hi, x_hi.w_lo) on MX-fp8
                                 * operands - v_mfma_scale_f32_16x16x128_f8f6f4, two taps per instruction - next to the exact f16
                                 * leading term (conv3x3_mx2.hip): two MFMAs' worth of matrix work per product instead of three,
                                 * ~2^-15 of a product instead of 2^-22 (1.3e-4 on the reference's configs[0] fixture in CPU emulation;
                                 * bar 1e-3).  The blocks at the tile's own resolution (final block, last up stage, first down stage) keep
                                 * the three-MFMA arithmetic - that is where the error is made: within 1.4-1.8x of SRGD_PRECISION_F16X3
                                 * on every reference fixture, +7 % throughput.  Everything else as SRGD_PRECISION_F16X3. */

/* Constructor arguments of ConditionalSRUnet (model.py:537-556) as get_model passes them
 * (model.py:3504-3514).  Unsupported combinations are rejected by srgd_create. */
typedef struct srgd_unet_config {
  int32_t dim;                         /* unet_dim */
  int32_t n_stages;                    /* len(dim_mults) */
  int32_t dim_mults[SRGD_MAX_STAGES];
  int32_t full_attn[SRGD_MAX_STAGES];  /* 0 linear attention, 1 softmax attention */
  int32_t channels;                    /* 3 */
  int32_t groups;                      /* resnet_block_groups = 8 */
  int32_t heads;                       /* attn_heads = 4 */
  int32_t dim_head;                    /* attn_dim_head = 32 */
  int32_t sinus_dim;                   /* learned_sinusoidal_dim (even) */
  int32_t num_classes;                 /* 0: no class embedding */
  int32_t precision;                   /* SRGD_PRECISION_* */
  int32_t device;                      /* HIP device ordinal */
} srgd_unet_config;

const char* srgd_last_error(void);
const char* srgd_version(void);

/* ---- lifetime -------------------------------------------------------------------------------
 * replaces: ConditionalSRUnet.__init__ (model.py:537-675) + `.to(device)` (inference.py:155-156) */
int srgd_create(const srgd_unet_config* cfg, srgd_engine** out);
int srgd_destroy(srgd_engine* e);

/* ---- weights --------------------------------------------------------------------------------
 * replaces: load_state_dict(ckpt['ema_model'], strict) (model.py:3659-3662).  Names are the
 * U-Net's state_dict keys (an optional leading "model." is accepted); data is host fp32 in the
 * PyTorch layout (conv OIHW, linear [out,in]).  The engine packs its own device copy. */
int srgd_num_weights(const srgd_engine* e);
int srgd_weight_info(const srgd_engine* e, int index, char* name, size_t name_cap, int64_t shape[4], int* ndim);
int srgd_load_weight(srgd_engine* e, const char* name, const float* host_data, const int64_t* shape, int ndim);
int srgd_finalize_weights(srgd_engine* e); /* fails if any tensor is missing (strict=True) */

/* ---- one U-Net evaluation -------------------------------------------------------------------
 * replaces: ConditionalSRUnet.forward(x, time, class_label, x_self_cond) (model.py:678-725).
 * x, cond (nullable -> zeros), eps_out: device fp32 [B,3,H,W]; log_snr: host [B];
 * class_id < 0 means class_label=None.  H, W must be divisible by 2^(n_stages-1). */
int srgd_unet_forward(srgd_engine* e, const float* x, const float* cond, const float* log_snr_host, int class_id,
                      float* eps_out, int B, int H, int W, void* stream);

/* ---- tiled sampler --------------------------------------------------------------------------
 * replaces: ConditionalContinuousTimeGaussianDiffusionSR.tiled_sample (model.py:3288-3413) and,
 * per step, p_sample / p_mean_variance (model.py:3122-3188) and q_sample (model.py:3434-3447). */
typedef struct srgd_step_scalars { /* fp32 values of model.py:3127-3134,:3168 for one step */
  float alpha, sigma, alpha_next, c, one_minus_c;
  float noise_scale; /* sqrt(sigma_next^2 * c) */
  float sigma_next;  /* sqrt(sigmoid(-log_snr(t'))) for the odd-step ring (model.py:3395) */
  float reserved;
} srgd_step_scalars;

typedef struct srgd_sampler_geometry { /* ints of get_coord_and_pad / get_coords / get_area (model.py:116-179) */
  int32_t H, W;                 /* image size (x4 bicubic of the LR input) */
  int32_t Hp, Wp;               /* padded canvas */
  int32_t left, top;            /* crop box origin of the image inside the canvas */
  int32_t inner_l, inner_t, inner_r, inner_b; /* bounding box of the shifted (odd-step) grid */
  int32_t tile;                 /* 256 */
  int32_t n_even, n_odd;        /* tiles per grid (of ONE image) */
  int32_t n_images;             /* >= 1: same-sized images sampled in lock-step (one U-Net batch spans all of them);
                                 * every image sees the noise stream the reference gives it after its own reseed */
} srgd_sampler_geometry;

/* Prepares a run: cond canvas = zero outside the inner box, reflect-padded (2*cond01-1) inside
 * (model.py:3296-3303,:3337-3342); uploads both tile grids ([n][2] = (y, x) canvas offsets) and
 * the schedule; computes the conditioning table for every step x {label, no label}.
 * cond01: device fp32 [n_images,3,H,W] in [0,1]; cond_canvas: device fp32 [n_images,3,Hp,Wp] (written). */
int srgd_sampler_begin(srgd_engine* e, const srgd_sampler_geometry* g, const float* cond01, float* cond_canvas,
                       const int32_t* tiles_even_host, const int32_t* tiles_odd_host, int n_steps,
                       const srgd_step_scalars* scalars_host, const float* log_snr_host, int class_id, void* stream);

/* One denoising step over every tile of grid (step % 2), `sub_batch` tiles per U-Net launch
 * (the reference's --batch_size; results do not depend on it).
 *   passes = 1: eps = unet(label, cond).                       (model.py:3155-3156)
 *   passes = 2, guidance_kind 1: class guidance  (label vs None)   (model.py:3151-3154)
 *   passes = 2, guidance_kind 2: condition guidance (cond vs zeros) (model.py:3147-3150)
 * img / x_start (nullable): device fp32 canvases [n_images,3,Hp,Wp], updated in place; sub_batch counts tiles over
 * all images (image-major order).
 * noise_tiles: device fp32 [n_tiles_of_this_grid,3,tile,tile] in reference draw order, or NULL;
 * noise_canvas: device fp32 [3,Hp,Wp] for the odd-step ring, or NULL.  When a needed noise
 * pointer is NULL the engine draws it on the device (Philox, `seed`).  The last step adds none.  The noise of
 * one image is shared by all n_images (each image of the reference is sampled after reseeding with the same seed,
 * inference.py:73, so same-sized images see identical draws). */
int srgd_sampler_step(srgd_engine* e, int step, float* img, const float* cond_canvas, float* x_start,
                      const float* noise_tiles, const float* noise_canvas, int passes, int guidance_kind,
                      float guidance_scale, int sub_batch, uint64_t seed, void* stream);

/* Same as srgd_sampler_step restricted to tiles [tile_first, tile_first+tile_count) of the step's grid (image-major
 * index; tile_count < 0 = to the end); the odd-step ring re-noise runs only when do_ring != 0.  This is the
 * per-rank unit when ONE canvas is sharded over several GPUs: every rank holds
 * the whole canvas, updates its own tiles, then the tiles are exchanged (srgd_sampler_exchange_tiles + an
 * all-gather).  noise_tiles still points at the noise of the WHOLE grid (indexed by the global tile number). */
int srgd_sampler_step_tiles(srgd_engine* e, int step, int tile_first, int tile_count, int do_ring, float* img,
                            const float* cond_canvas, float* x_start, const float* noise_tiles,
                            const float* noise_canvas, int passes, int guidance_kind, float guidance_scale,
                            int sub_batch, uint64_t seed, void* stream);

/* Copies tiles [tile_first, tile_first+tile_count) of grid `parity` between a canvas [n_images,3,Hp,Wp] and a
 * packed device buffer [tile_count,3,tile,tile]: to_canvas = 0 packs (canvas -> tiles), 1 unpacks. */
int srgd_sampler_exchange_tiles(srgd_engine* e, int parity, int tile_first, int tile_count, float* canvas, float* tiles,
                                int to_canvas, void* stream);

/* Unpacks an all-gathered buffer [world * part_w, 3, tile, tile] into the canvas in ONE launch: ranks own contiguous slices of
 * slice_w tiles of grid `parity`, each contributed tiles [part_off, part_off + part_w) of its slice, so row j of `gathered` is tile
 * (j / part_w) * slice_w + part_off + j % part_w; rows past the end of the grid (short and empty slices) are skipped.  Equivalent to
 * `world` calls of srgd_sampler_exchange_tiles(to_canvas = 1). */
int srgd_sampler_unpack_gathered(srgd_engine* e, int parity, int world, int slice_w, int part_off, int part_w, float* canvas,
                                 const float* gathered, void* stream);

/* Optional start from the forward-diffused condition instead of white noise (generation_start_steps > 0 or
 * start_white_noise=False): img = reflect_pad(2*cond01-1) * alpha + noise * sigma over the whole canvas
 * (q_sample, model.py:3305-3308, :3312-3315, :3434-3442).  noise_canvas: device [3,Hp,Wp] or NULL (device RNG). */
int srgd_sampler_q_start(srgd_engine* e, const float* cond01, const float* noise_canvas, float alpha, float sigma,
                         float* img, uint64_t seed, void* stream);

/* Crop, clamp to [-1,1], map to [0,1] (model.py:3403-3405).  out01: device fp32 [3,H,W]. */
int srgd_sampler_end(srgd_engine* e, const float* img, float* out01, void* stream);

/* ---- EDM (Karras) sampler over the same U-Net: ConditionalElucidatedDiffusionSR.tiled_sample (model.py:2309-2475) ----
 * Per step i the host supplies fp32 scalars computed with the reference's own ops (schedule and preconditioning of the
 * un-vendored base class denoising_diffusion_pytorch.ElucidatedDiffusion, restated in srgd_amd/model.py). */
typedef struct srgd_edm_scalars {
  float s_noise, hat_coef;                      /* img_hat = img + hat_coef * (s_noise * z)        (model.py:2386-2389) */
  float sigma_hat, sigma_next;                  /* sigma_next == 0 on the last step: no Heun correction */
  float dt, half_dt;                            /* sigma_next - sigma_hat and half of it            (model.py:2407,:2414) */
  float c_in_hat, c_skip_hat, c_out_hat;        /* preconditioning at sigma_hat                     (model.py:2140-2149) */
  float c_in_next, c_skip_next, c_out_next;     /* ... at sigma_next */
  float ring_sigma;                             /* sigmas[i]: odd-step ring = ring_sigma * z'       (model.py:2448-2452) */
  float clamp;                                  /* != 0: clamp the denoised prediction to [-1, 1]   (model.py:2180) */
  float dpm_gamma, pad1;                        /* srgd_edm_dpmpp_step only: multistep weight, see there */
} srgd_edm_scalars;

/* As srgd_sampler_begin; c_noise_host: [2*n_steps] = c_noise(sigma_hat_i), c_noise(sigma_next_i) - the "time" input of
 * the two network evaluations of step i (the second is unused on the last step). */
int srgd_edm_begin(srgd_engine* e, const srgd_sampler_geometry* g, const float* cond01, float* cond_canvas,
                   const int32_t* tiles_even_host, const int32_t* tiles_odd_host, int n_steps,
                   const srgd_edm_scalars* scalars_host, const float* c_noise_host, int class_id, void* stream);

/* One EDM step over every tile of grid (step % 2): Euler evaluation at sigma_hat, Heun correction at sigma_next, scatter,
 * odd-step ring re-noise.  img / x_start (nullable) as in srgd_sampler_step; work: device fp32 scratch of two canvases
 * [2][n_images,3,Hp,Wp]; noise_canvas: the step's z [3,Hp,Wp] (device) or NULL = drawn on the device (Philox, `seed`);
 * ring_noise_canvas: z' of an odd step or NULL.  passes / guidance_* as in srgd_sampler_step.  Finish with srgd_sampler_end. */
int srgd_edm_step(srgd_engine* e, int step, float* img, const float* cond_canvas, float* x_start, float* work,
                  const float* noise_canvas, const float* ring_noise_canvas, int passes, int guidance_kind,
                  float guidance_scale, int sub_batch, uint64_t seed, void* stream);

/* srgd_edm_step restricted to tiles [tile_first, tile_first + tile_count) of the step's grid (tile_count < 0: to the end) - the
 * per-rank unit when one canvas is shared by several GPUs, as srgd_sampler_step_tiles is for the DDPM loop (reference tile loop
 * model.py:2397-2452); do_ring = 0 skips the odd-step ring re-noise.  `work` is touched at those tiles only; exchange the updated
 * tiles of img / x_start with srgd_sampler_exchange_tiles. */
int srgd_edm_step_tiles(srgd_engine* e, int step, int tile_first, int tile_count, int do_ring, float* img,
                        const float* cond_canvas, float* x_start, float* work, const float* noise_canvas,
                        const float* ring_noise_canvas, int passes, int guidance_kind, float guidance_scale, int sub_batch,
                        uint64_t seed, void* stream);

/* One DPM-Solver++(2M) step of the EDM wrapper's un-tiled loop (reference sample_using_dpmpp, model.py:2517-2542) over
 * every tile of grid (step % 2), one network evaluation, no noise, no ring:
 *   den   = clamp(c_skip_hat * img + c_out_hat * net(c_in_hat * img, c_noise[2*step]))          (:2528, preconditioning :2140-2183)
 *   den_d = (1 - dpm_gamma) * den + dpm_gamma * old_denoised                                      (:2533-2539)
 *   img   = dt * img - half_dt * den_d ;  old_denoised = den ;  x_start (nullable) = den_d         (:2541-2547)
 * with the step's srgd_edm_scalars read as: *_hat / clamp at sigma_i, dt = sigma_fn(t_next) / sigma_fn(t),
 * half_dt = expm1(-h), dpm_gamma = -1 / (2 r), or 0 on the first step and on the last (sigma_next == 0) - the host computes
 * them with the reference's fp32 tensor ops.  old_denoised: device fp32 canvas [n_images,3,Hp,Wp], zero before the first step.
 * Run between srgd_edm_begin and srgd_sampler_end like srgd_edm_step. */
int srgd_edm_dpmpp_step(srgd_engine* e, int step, float* img, const float* cond_canvas, float* x_start, float* old_denoised,
                        int passes, int guidance_kind, float guidance_scale, int sub_batch, void* stream);

/* Host helper (no GPU): out[i] = e4m3(in[i] / scale) * scale, round-to-nearest-even, saturating at +-448 (the OCP
 * "fn" variant torch.float8_e4m3fn implements) - the weight rounding of SRGD_PRECISION_BF16_W8, exposed for tests. */
int srgd_quantize_e4m3(const float* in, float* out, size_t n, float scale);

/* ---- image front / back end (reference inference.py:66-73, :93) ------------------------------------------------
 * Bicubic resize of an 8-bit RGB image, bit-exact with Pillow's Image.resize((out_w, out_h), BICUBIC) - what
 * torchvision's T.Resize does for the PIL input of sr_target_image - followed by ToTensor:
 * src_hwc: device uint8 [h][w][3]; dst01_chw: device fp32 [3][out_h][out_w] = resized / 255.
 * Builds Pillow's fixed-point coefficient tables on the host; synchronises `stream` before returning. */
int srgd_image_resize_bicubic_u8(const uint8_t* src_hwc, int h, int w, int out_h, int out_w, float* dst01_chw,
                                 void* stream);
/* ToPILImage of a float image: img01_chw device fp32 [3][h][w] in [0,1] -> dst_hwc device uint8 [h][w][3] =
 * trunc(img * 255) (torchvision: pic.mul(255).byte()).  Asynchronous on `stream`. */
int srgd_image_unit_to_u8(const float* img01_chw, int h, int w, uint8_t* dst_hwc, void* stream);

/* Fills dst[n] with N(0,1) draws of the engine's counter-based generator (initial canvas noise
 * in throughput mode; the parity mode uploads torch's CPU stream instead). */
int srgd_randn(srgd_engine* e, float* dst, size_t n, uint64_t seed, uint64_t stream_id, void* stream);

/* ---- measurement ----------------------------------------------------------------------------
 * Between begin and end every kernel family is bracketed by HIP events on its launch stream;
 * end synchronises and returns, per family, accumulated milliseconds, launch counts and (for the two
 * convolution families) the algorithmic FLOPs issued (2*M*Cout*K).  Family names: srgd_profile_family_name(i). */
int srgd_profile_begin(srgd_engine* e);
int srgd_profile_end(srgd_engine* e, double* ms, int64_t* launches, double* flops, int n_families);
/* Algorithmic HBM bytes per family of the interval the last srgd_profile_begin/end pair bracketed: every operand read once
 * and every result written once (the roofline numerator of the HBM-bound kernels; bench.py divides by the event time). */
int srgd_profile_bytes(const srgd_engine* e, double* bytes, int n_families);
int srgd_profile_num_families(void);
const char* srgd_profile_family_name(int i);
int64_t srgd_device_bytes_in_use(const srgd_engine* e);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SRGD_HIP_H */
