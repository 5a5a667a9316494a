// Launchers of the hand-written gfx950 kernels (one translation unit per kernel family).
// All tensors are NHWC in the engine's activation type (fp32 or bf16, flag `is_bf16`);
// all accumulation and normalisation arithmetic is fp32.  Every launcher is asynchronous
// on `st`, performs no allocation and no synchronisation (hipGraph-capturable), and
// returns 0 or -1 with the message available from srgd_last_error().
#pragma once
#include <vector>

#include "common.hpp"

namespace srgd {

// ---------------------------------------------------------------- conv_igemm.hip
enum { CONV_PLAIN = 0, CONV_PIXEL_SHUFFLE_SILU = 1 };
struct ConvArgs {
  const void* in0;        // NHWC [B,Hin,Win,C0]
  const void* in1;        // NHWC [B,Hin,Win,C1] or null (channel concat (in0, in1))
  int C0, C1;
  int ps0, ps1;           // elements between consecutive input pixels of each source (normally C0 / C1; smaller for
                          // the overlapping-window view the 7x7 input conv uses)
  int B, Hin, Win, Hout, Wout;
  int KH, KW, stride, pad;
  const void* w;          // packed [KH*KW][CoutPad][C0+C1], activation type
  const float* bias;      // [Cout] or null
  int Cout, CoutPad;
  void* out;              // NHWC [B,Hout,Wout,Cout]; pixel-shuffle mode: [B,2Hout,2Wout,Cout/4]
  const void* residual;   // optional, same shape as out (plain mode)
  int mode;
  float* gn_partial;      // optional [B][groups][Hout*Wout/128][2]
  int groups;
  // optional fused tail of a ResnetBlock (model.py:283-285) for the 1x1 res_conv, bf16 only:
  //   out = (acc + bias) + silu(gn_res_a[b][c] * gn_res_src + gn_res_b[b][c])      (out may alias gn_res_src)
  const void* gn_res_src;
  const float* gn_res_a;
  const float* gn_res_b;
  // optional MX-fp8 twin of the (bf16) output, written by the same epilogue (fp8 mode: the tensor feeds a 3x3 convolution
  // on the MX matrix cores): e4m3 [.., C] and E8M0 [.., C/32] with C = Cout (pixel-shuffle mode: Cout / 4).  Supported by
  // conv1x1_bf16 and conv3x3_mxfp8; identical to quant_mxfp8 of the stored bf16 values.
  void* out_q = nullptr;
  void* out_s = nullptr;
  // optional, conv1x1_bf16 with the GroupNorm tail and Cout == 128 only (the last ResnetBlock of the U-Net): the block's
  // output is consumed by the 1x1 output convolution alone (model.py:776-777), so the epilogue applies that convolution to
  // the bf16 values it would have stored and writes eps4[pixel] = (e0, e1, e2, 0) instead of `out` (16 B instead of 256 B
  // per pixel, and no second pass reading them back).  fin_w: [3][128] fp32, fin_b: [3].
  float* eps4 = nullptr;
  const float* fin_w = nullptr;
  const float* fin_b = nullptr;
  // conv1x1_split only (f16x3 mode), the RMSNorms around the attention projections (model.py:201-207):
  //   rms_in: the convolution reads RMSNorm(in) - the caller folded gain * sqrt(Cin) into the weights, the kernel supplies the
  //   per-pixel 1 / max(||x||, 1e-12);  rms_out_g ([Cout], gain * sqrt(Cout); Cout == 128, residual set): out = RMSNorm(conv) * g + residual
  bool rms_in = false;
  const float* rms_out_g = nullptr;
};
int conv_igemm(const ConvArgs& a, bool is_bf16, hipStream_t st);
int conv_tile_m();
int conv_tile_n();
void pack_conv_weights(const float* src_oihw, const float* bias_in, int kind, int Cin, int Cout, int CoutPad, int KS,
                       bool to_bf16, std::vector<unsigned char>& packed_out, std::vector<float>& bias_out);
const char* last_error();

// ---------------------------------------------------------------- conv1x1_bf16.hip
// bf16 streaming GEMM for the pointwise layers (1x1 incl. two-source K, residual add, fused GroupNorm tail,
// pixel-shuffle + SiLU; 2x2/stride-2 space-to-depth gather).  Same ConvArgs (a.w ignored); weights from pack_conv1x1_bf16
// which takes the generic path's fp32 [tap][Cout][Cin] order (pack_conv_weights with to_bf16 = false).
bool conv1x1_bf16_eligible(const ConvArgs& a);
void pack_conv1x1_bf16(const float* src_tap_o_i, int taps, int Cin, int Cout, std::vector<unsigned short>& out,
                       unsigned short (*to_bf16)(float));
int conv1x1_bf16(const ConvArgs& a, const void* packed_w, hipStream_t st);

// ---------------------------------------------------------------- conv3x3_bf16.hip
// Fast path for the 3x3/s1/p1 bf16 convolutions (halo patch in LDS, LDS-DMA staging).  Takes the same
// ConvArgs (a.w is ignored) plus weights packed by pack_conv3x3_bf16.
bool conv3x3_bf16_eligible(const ConvArgs& a);
int conv3x3_bf16_stats_slots(const ConvArgs& a);
void pack_conv3x3_bf16(const float* src_oihw, int Cin, int Cout, std::vector<unsigned short>& out,
                       unsigned short (*to_bf16)(float));
// gn_in_a/gn_in_b (nullable): [B][Cin] fp32 coefficients of y = silu(a*x + b) applied to the INPUT while it is
// staged (the producer's GroupNorm+SiLU, fused; single source, Cin <= 1024).
int conv3x3_bf16(const ConvArgs& a, const void* packed_w, const float* gn_in_a, const float* gn_in_b,
                 hipStream_t st);
unsigned short f32_to_bf16_host(float f);

// ---------------------------------------------------------------- conv3x3_split.hip / conv_igemm.hip (split-operand precision)
// fp32 tensors in HBM, contraction as three 16-bit MFMAs per product on (hi, lo) operand pairs (f16 halves, or bf16 halves for
// the numerics comparison).  Weights are split on the host after scaling by split_weight_scale (a power of two; 1 for bf16).
float split_weight_scale(const float* w, size_t n, bool f16);
void split_halves_host(float v, bool f16, unsigned short* hi, unsigned short* lo);
bool conv3x3_split_eligible(const ConvArgs& a);          // same shapes and GroupNorm slot layout as conv3x3_bf16
void pack_conv3x3_split(const float* src_oihw, int Cin, int Cout, bool f16, float scale, std::vector<unsigned short>& out);
// gn_in_a / gn_in_b (nullable, f16 halves only): [B][Cin] fp32 coefficients of y = silu(a*x + b) applied to the INPUT while it is
// staged (the producer's GroupNorm + SiLU, fused; single source)
// form: 0 = the engine's choice (SRGD_SPLIT3_WG, default 1), 1 = the 512-thread kernel (one workgroup per CU), 2 = the 256-thread
// kernel (two workgroups per CU; f16 halves only)
int conv3x3_split(const ConvArgs& a, const void* packed_w, float w_inv_scale, bool f16, hipStream_t st,
                  const float* gn_in_a = nullptr, const float* gn_in_b = nullptr, int form = 0);
// conv3x3_mx2.hip: prototype of a TWO-MFMA split arithmetic (f16 leading term, both cross terms on MX-fp8 operands); conv3x3_split's shapes
void pack_conv3x3_mx2(const float* src_oihw, int Cin, int Cout, float scale, std::vector<unsigned char>& out);
int conv3x3_mx2(const ConvArgs& a, const void* packed_w, float w_inv_scale, hipStream_t st, const float* gn_in_a = nullptr,
                const float* gn_in_b = nullptr);     // gn_in_*: [B][C0] fp32, one allocation (shift behind scale), one source
// generic implicit GEMM of that mode (a.w ignored; every epilogue of the fp32 conv_igemm); weights from pack_conv_weights_split,
// which takes pack_conv_weights' fp32 [tap][CoutPad][Cin] order
bool conv_igemm_split_eligible(const ConvArgs& a);
void pack_conv_weights_split(const float* src_tap_o_i, int taps, int Cin, int CoutPad, bool f16, float scale,
                             std::vector<unsigned short>& out);
int conv_igemm_split(const ConvArgs& a, const void* packed_w, float w_inv_scale, bool f16, hipStream_t st);
// conv1x1_split.hip: the pointwise layers of that mode as a streaming kernel (3-deep LDS-DMA ring of fp32 pixel rows + split weights; plain,
// + residual, SiLU + PixelShuffle epilogues; 1x1 incl. two sources, 2x2 / stride-2 gather); f16 halves; weights from
// pack_conv1x1_split, which takes pack_conv_weights' fp32 [tap][Cout][Cin] order
bool conv1x1_split_eligible(const ConvArgs& a);
void pack_conv1x1_split(const float* src_tap_o_i, int taps, int Cin, int Cout, float scale, std::vector<unsigned short>& out);
int conv1x1_split(const ConvArgs& a, const void* packed_w, float w_inv_scale, hipStream_t st);

// ---------------------------------------------------------------- conv3x3_mxfp8.hip / quant_mxfp8.hip
// Block-scaled MX-fp8 path (v_mfma_scale_f32_16x16x128_f8f6f4, BASELINE configs[4]): the 3x3 convolutions take e4m3
// activations [B,H,W,C] + E8M0 scales [B,H,W,C/32] (one per 32 channels) and e4m3 weights with one scale per
// (output channel, tap, 32 input channels); fp32 accumulate, bf16 out, same epilogue as conv3x3_bf16.
bool conv3x3_mxfp8_eligible(const ConvArgs& a);
int conv3x3_mxfp8_stats_slots(const ConvArgs& a);
void pack_conv3x3_mxfp8(const float* src_oihw, int Cin, int Cout, std::vector<unsigned char>& out);
int conv3x3_mxfp8(const ConvArgs& a, const void* q0, const void* s0, const void* q1, const void* s1, const void* packed_w,
                  hipStream_t st);
// conv1x1_mxfp8.hip: the pointwise layers on the same matrix cores (same ConvArgs as conv1x1_bf16 incl. every epilogue;
// inputs as MX-fp8 twins; weights from pack_conv1x1_mxfp8, which takes the generic path's fp32 [tap][Cout][Cin] order)
bool conv1x1_mxfp8_eligible(const ConvArgs& a);
void pack_conv1x1_mxfp8(const float* src_tap_o_i, int taps, int Cin, int Cout, std::vector<unsigned char>& out);
int conv1x1_mxfp8(const ConvArgs& a, const void* q0, const void* s0, const void* q1, const void* s1, const void* packed_w,
                  hipStream_t st);
unsigned char e4m3_encode(float x);
int mx_block_exponent(float amax);
float round_through_e4m3(float x);
// bf16 [npix][C] -> e4m3 [npix][C] + E8M0 [npix][C/32]; the second form applies y = silu(coefA*x + coefB) first
int quant_mxfp8(const void* x_bf16, void* q, void* s, long npix, int C, hipStream_t st);
int gn_apply_silu_mxfp8(const void* x_bf16, void* q, void* s, const float* coefA, const float* coefB, int B, int hw, int C,
                        hipStream_t st);

// ---------------------------------------------------------------- norm_act.hip
struct GnFinalizeArgs {
  const float* partial;   // [B][groups][nslots][2]
  int nslots;
  int B, C, groups;
  int hw;                 // pixels per sample
  const float* gamma;     // [C]
  const float* beta;      // [C]
  const float* ss_table;  // conditioning table or null: row r holds scale at ss_offset, shift at ss_offset + C
  const int* ss_rows;     // [B] table row of each batch entry
  const int* step_ptr;    // optional device step counter: row += *step_ptr * step_mul
  int step_mul;
  int ss_stride, ss_offset;
  float eps;
  float* coefA;           // [B][C]   y = silu(coefA * x + coefB)
  float* coefB;
};
int gn_finalize(const GnFinalizeArgs& a, hipStream_t st);
// out_q / out_s (nullable, bf16 only): MX-fp8 twin of y written alongside (see ConvArgs::out_q)
int gn_apply_silu(const void* x, void* y, const void* residual, const float* coefA, const float* coefB,
                  int B, int hw, int C, bool is_bf16, hipStream_t st, void* out_q = nullptr, void* out_s = nullptr);
int rms_norm(const void* x, void* y, const void* residual, const float* g, long npix, int C,
             bool is_bf16, hipStream_t st);

// ---------------------------------------------------------------- attention.hip
// qkv: [B, N, 3*heads*dh] (q | k | v, each (head, d)); out: [B, N, heads*dh]
size_t linear_attention_workspace(int B, int N, int heads, int dh);
int linear_attention(const void* qkv, void* out, int B, int N, int heads, int dh, float* ws,
                     bool is_bf16, hipStream_t st);
int full_attention(const void* qkv, void* out, int B, int N, int heads, int dh, bool is_bf16,
                   hipStream_t st);
// merge per-chunk (max, sum, context) partials [bh][nch][...] -> normalised context * scale, [bh][32][32]
int linear_attention_combine(const float* pm, const float* pl, const float* pctx, int bh, int nch, float scale,
                             float* ctxn, hipStream_t st);

// ---------------------------------------------------------------- linattn_fused.hip
// Whole LinearAttention block + residual (model.py:306-324, :703) in two kernels; bf16, C = 128 or 256, 4 heads x 32.
bool linattn_fused_eligible(int C, int heads, int dh, int N, bool is_bf16);
size_t linattn_fused_workspace(int B, int N);
void linattn_fused_pack(const float* to_qkv, const float* norm_g, const float* to_out, int C,
                        std::vector<unsigned short>& wkv_img, std::vector<unsigned short>& wq,
                        std::vector<unsigned short>& wout);
int linattn_fused(const void* x, void* y, int B, int N, int C, const void* wkv_img, const void* wq, const void* wout,
                  const float* bout, const float* g2_scaled, float* ws, hipStream_t st, void* y_q = nullptr,
                  void* y_s = nullptr);          // y_q / y_s: optional MX-fp8 twin of y (see ConvArgs::out_q)
// linattn_fused256.hip: the 32-pixel-tile kernels (C = 256) behind the entry points above
bool linattn_fused256_eligible(int C, int heads, int dh, int N, bool is_bf16);
void linattn_fused256_pack(const float* to_qkv, const float* norm_g, const float* to_out, int C, std::vector<unsigned short>& wkv,
                           std::vector<unsigned short>& wq, std::vector<unsigned short>& wout);
int linattn_fused256(const void* x, void* y, int B, int N, int C, const void* wkv, const void* wq, const void* wout,
                     const float* bout, const float* g2_scaled, float* pm, float* pl, float* pctx, float* ctxn, float* rinv,
                     int strip, hipStream_t st, void* y_q, void* y_s);

// ---------------------------------------------------------------- cond.hip
// feat[r] = [x, sin(2 pi x w_i), cos(2 pi x w_i)]   (reference model.py:233-238)
int time_features(const float* log_snr, const float* w, int half, int rows, float* feat, hipStream_t st);
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU_IN = 2 };
// y[r][o] = act_out(sum_i W[o][i] * act_in(x[r][i]) + b[o]) (+ add[r*add_stride + o])
int linear_rows(const float* x, int x_stride, const float* W, const float* b, float* y, int y_stride,
                int rows, int in_f, int out_f, int act, const float* add, int add_stride, hipStream_t st);

// ---------------------------------------------------------------- sampler.hip
struct StepScalars {       // fp32 values computed by the host exactly as the reference does
  float alpha, sigma, alpha_next, c, one_minus_c, noise_scale;   // noise_scale = sqrt(sigma_next^2 * c), 0 on the last step
  float sigma_next;        // for the ring re-noise (q_sample(0, t'))
  float pad;
};
struct EdmScalars {        // == srgd_edm_scalars (include/srgd_hip.h); fp32 values computed by the host as the reference does
  float s_noise, hat_coef;           // img_hat = img + hat_coef * (s_noise * z)            (model.py:2386-2389)
  float sigma_hat, sigma_next;       // sigma_next == 0 on the last step (no Heun correction)
  float dt, half_dt;                 // sigma_next - sigma_hat, half of it
  float c_in_hat, c_skip_hat, c_out_hat;
  float c_in_next, c_skip_next, c_out_next;
  float ring_sigma;                  // sigmas[i]: odd-step ring = ring_sigma * z'            (model.py:2448-2452)
  float clamp;                       // != 0: clamp the denoised prediction to [-1, 1]
  float dpm_gamma, pad1;             // srgd_edm_dpmpp_step: multistep weight (dt / half_dt hold its two update coefficients)
};
struct TileBatch {
  const int* tile_yx;      // device [n_images * n_local][3] = (y, x, image) of each tile, image-major
  int first;               // first tile of this sub-batch in tile_yx
  int ntiles;              // tiles in this sub-batch
  int Hp, Wp;              // canvas size
  int tile;                // tile edge (256)
  int n_local;             // tiles per image (noise is indexed by the tile's position inside its image)
};
// 7x7 input convolution (model.py:583) on MFMA: gather the 6 input planes of every tile into a zero-haloed NHWC image
// [entries][H+6][W+8][8 ch] (ch 6,7 = 0; position px holds input column px-3), after which output pixel (y,x), tap row
// dy reads ONE contiguous 64-element run (8 pixels x 8 ch) -> a 7x1 implicit GEMM with 64 "channels" whose pixel
// stride is 8 elements (conv_igemm with ps0 = 8).  The 8th pixel and channels 6,7 carry zero weights.
int init_gather_from_canvas(const float* img, const float* cond, const TileBatch& tb, int passes, int use_cond_mask,
                            void* padded, bool is_bf16, hipStream_t st);
// EDM variant (model.py:2140-2146): the gathered input is c_in * x with, for edm_pass 0, x = img + hat_coef*(s_noise*z)
// (z: one noise canvas [3][Hp][Wp] shared by all images) and c_in = c_in_hat; for edm_pass 1, x = img (the Euler
// result canvas) and c_in = c_in_next.  Scalars are read on the device: sc[*step_ptr].
int init_gather_from_canvas_edm(const float* img, const float* z, const float* cond, const TileBatch& tb, int passes,
                                int use_cond_mask, const EdmScalars* sc, const int* step_ptr, int edm_pass, void* padded,
                                bool is_bf16, hipStream_t st);
int init_gather_from_nchw(const float* x, const float* cond, int B, int H, int W, void* padded, bool is_bf16,
                          hipStream_t st);
// 1x1 output conv (model.py:675) -> eps; NCHW fp32 out (U-Net-only entry point)
int final_conv_to_nchw(const void* act, int B, int H, int W, int C, const float* w /*[3][C]*/,
                       const float* bias, float* out, bool is_bf16, hipStream_t st);
// 1x1 output conv + guidance combine + DDPM posterior step (model.py:3147-3168, :3184-3188),
// scattering straight into the canvases.  noise: [ntiles][3][tile][tile] fp32 or null (last step).
struct FinalStepArgs {
  const void* act;         // NHWC [ntiles*passes, tile, tile, C]
  int C, passes;
  float guidance;          // eps = null + (cond - null) * guidance when passes == 2
  const float* w;          // [3][C]
  const float* bias;       // [3]
  float* img;              // canvas, updated in place
  float* x_start;          // optional canvas
  const float* noise;
  const float* eps4 = nullptr;   // when set: the output convolution was already applied (ConvArgs::eps4); act / w / bias unused
  const StepScalars* sc;   // device pointer
  const int* step_ptr;     // optional device step counter indexing sc
};
int final_step(const FinalStepArgs& a, const TileBatch& tb, bool is_bf16, hipStream_t st);
// EDM: 1x1 output conv + guidance combine + Karras preconditioning + Euler step (edm_pass 0) or Heun correction
// (edm_pass 1), model.py:2403-2425.  `a.img` is the image canvas, `a.noise` the z canvas [3][Hp][Wp] (indexed by
// position, not by tile), `work` = two canvases [2][n_images*3*Hp*Wp]: the Euler result and the slope d.
// On the last step (sigma_next == 0) pass 0 writes the canvas directly.  a.sc / a.x_start as in final_step.
int final_step_edm(const FinalStepArgs& a, const EdmScalars* sc, float* work, size_t canvas_elems, int edm_pass,
                   const TileBatch& tb, bool is_bf16, hipStream_t st);

// canvas kernels (model.py:3296-3303, :3337-3342, :3392-3396, :3403-3405)
// `planes` = 3 * n_images; the noise canvas [3][Hp][Wp] is shared by every image (index taken modulo 3*Hp*Wp)
int canvas_prepare_cond(const float* cond01 /*[planes][H][W]*/, int planes, int H, int W, int pad_l, int pad_t, int Hp,
                        int Wp, int il, int it, int ir, int ib, float* cond_canvas, hipStream_t st);
int canvas_q_start(const float* cond01, int planes, int H, int W, int pad_l, int pad_t, int Hp, int Wp,
                   const float* noise, float alpha, float sigma, float* img, hipStream_t st);
// sigma = sigma_base[*step_ptr * sigma_stride] (a float field of the per-step scalar records of either sampler)
int canvas_ring_renoise(float* img, int planes, const float* noise /*[3][Hp][Wp]*/, int Hp, int Wp, int il, int it,
                        int ir, int ib, const float* sigma_base, int sigma_stride, const int* step_ptr, hipStream_t st);
int canvas_finish(const float* img, int planes, int Hp, int Wp, int left, int top, int H, int W, float* out01,
                  hipStream_t st);
// copies the tiles [tb.first, tb.first+tb.ntiles) between a canvas and a packed [ntiles][3][tile][tile] buffer
// (the exchange unit when one canvas is sharded over ranks: SURVEY section 8(e), config 4)
int canvas_unpack_gathered(float* canvas, const float* gathered, const TileBatch& tb, int w, int off, int pw, int n_grid, hipStream_t st);
int canvas_exchange_tiles(float* canvas, float* tiles, const TileBatch& tb, bool to_canvas, hipStream_t st);
// counter-based Gaussian noise (Philox4x32-10 + Box-Muller), throughput mode only
int philox_normal(float* dst, size_t n, uint64_t seed, uint64_t stream_id, const int* step_ptr, hipStream_t st);

}  // namespace srgd
