/* libsrgd_hip - kernel-level C ABI: each hand-written gfx950 kernel family of the Real-SRGD hot path,
 * callable on raw device pointers.  The engine (srgd_hip.h) is built from exactly these launchers;
 * they are exported so that every operator can be parity-tested in isolation against the CPU oracle
 * (tests/test_kernels_gpu.py) and bound separately by a host that wants only one of them.
 *
 * Layout: activations are NHWC in the activation type selected by `is_bf16` (0: fp32, 1: bf16);
 * statistics, coefficients and weights handed in as fp32.  All pointers except `*_host` are device
 * pointers.  Calls are asynchronous on `stream` unless stated; return 0 or <0 (srgd_last_error()).
 */
#ifndef SRGD_HIP_KERNELS_H
#define SRGD_HIP_KERNELS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only the entry points declared here are exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* Convolution as implicit GEMM on MFMA.  replaces: nn.Conv2d call sites of the U-Net - Block.proj
 * (model.py:246), res_conv (:271), to_qkv/to_out (:300,:303,:341,:342), Downsample (:106-110, kind 1:
 * pixel-unshuffle + 1x1 == 2x2/stride-2), PixelShuffleUpsample (:70-98, kind 2: 1x1 + SiLU + PixelShuffle),
 * and torch.cat on the skip path (:713,:716,:722: two channel-concatenated sources in0|in1).
 * weight_oihw_host / bias_host: PyTorch-layout fp32 on the HOST (packed + uploaded inside; this entry
 * point synchronises).  gn_partial: optional device [B][groups][nslots][2] receiving partial (sum, sum of squares)
 * of the output for a following GroupNorm.  nslots depends on the kernel the call selects (srgd_k_conv2d_timed reports it):
 * the generic kernel writes Hout*Wout/128 slots, the 3x3 fast paths one per contributing WAVE of a 256-pixel patch (4, or 8 per
 * 128-channel tile of a group that spans whole tiles).  Size the buffer for the largest:
 *     nslots_capacity = (Hout*Wout / 32) * max(1, (Cout / groups) / 64). */
int srgd_k_conv2d(const void* in0, const void* in1, int C0, int C1, int B, int Hin, int Win, int KS, int stride,
                  int pad, int kind, const float* weight_oihw_host, const float* bias_host, int Cout, void* out,
                  const void* residual, float* gn_partial, int groups, int is_bf16, void* stream);
/* Same, with implementation choice and timing: impl 0 = what the engine would pick, 1 = generic implicit-GEMM
 * kernel (conv_igemm.hip), 2 = the 3x3 bf16 halo/LDS-DMA kernel (conv3x3_bf16.hip; error if not eligible),
 * 3 = the pointwise bf16 streaming GEMM (conv1x1_bf16.hip; error if not eligible), 4 = the pointwise layer on the block-scaled
 * MX-fp8 matrix cores (conv1x1_mxfp8.hip, fp8 mode: the bf16 sources are quantised inside with srgd_k_quant_mxfp8's rule, the
 * weights per (output channel, tap, 32 input channels); C0, C1, Cout % 128 == 0; error if not eligible),
 * 5 = impl 2 with the PRODUCER's GroupNorm + SiLU applied while the input is staged (Block.forward model.py:250-259 between
 * two convolutions): conv(silu(gn_tail_a[b][c] * in0 + gn_tail_b[b][c])), zero padding applied after the activation; one source,
 * gn_tail_a / gn_tail_b = device fp32 [B][C0] in ONE allocation (shift behind scale), gn_tail_src must be NULL.
 * 6 / 7 = the split-operand kernels of SRGD_PRECISION_F16X3 (is_bf16 = 0: fp32 tensors; every product as three f16 MFMAs on (hi, lo)
 * operand pairs): 6 = the 3x3 halo-patch kernel (conv3x3_split.hip; conv3x3_bf16's shapes), 7 = the generic implicit GEMM
 * (conv_igemm.hip; C0, C1 % 32 == 0); 8 / 9 = the same two kernels with bf16 halves (numerics comparison only); 10 = the streaming
 * pointwise kernel of that mode (conv1x1_split.hip: 1x1 incl. two sources, 2x2 / stride-2 gather, + residual, SiLU + PixelShuffle;
 * C0, C1 % 32 == 0, Cout % 128 == 0, Hout * Wout % 256 == 0); 11 = impl 6 with the producer's GroupNorm + SiLU applied while the input
 * is staged, as impl 5 (gn_tail_a / gn_tail_b = device fp32 [B][C0], 16-byte aligned; one source; gn_tail_src NULL); 12 / 13 = impl
 * 6 / 11 on the 256-thread form of that kernel (two workgroups per CU) instead of the engine's default (512 threads, one per CU).
 * 14 / 15 = the two-MFMA prototype of SRGD_PRECISION_F16MX2 (conv3x3_mx2.hip: f16 leading term, both cross terms on MX-fp8 operands;
 * conv3x3_split's shapes), 15 with the producer's GroupNorm + SiLU applied while the input is staged (as impl 5: gn_tail_a / gn_tail_b in
 * ONE allocation, one source).
 * gn_tail_src (nullable, NHWC like out): out = silu(gn_tail_a[b][c] * gn_tail_src + gn_tail_b[b][c]) + conv(in) -
 * the second GroupNorm+SiLU of a ResnetBlock and its residual add folded into the 1x1 res_conv (model.py:250-259,:285);
 * gn_tail_a / gn_tail_b: device fp32 [B][Cout].  May alias out.
 * iters > 0: the launch is repeated `iters` times between two HIP events on `stream`, *avg_ms = mean duration.
 * *stats_slots (nullable) = slots per (sample, group) written to gn_partial - pass it to srgd_k_groupnorm_silu. */
int srgd_k_conv2d_timed(const void* in0, const void* in1, int C0, int C1, int B, int Hin, int Win, int KS, int stride,
                        int pad, int kind, const float* weight_oihw_host, const float* bias_host, int Cout, void* out,
                        const void* residual, float* gn_partial, int groups, int is_bf16, int impl, int iters,
                        float* avg_ms, int* stats_slots, const void* gn_tail_src, const float* gn_tail_a,
                        const float* gn_tail_b, void* stream);

/* bf16 NHWC [npix][C] -> OCP MX-fp8: q [npix][C] e4m3 bytes + s [npix][C/32] E8M0 bytes (x ~ q * 2^(s-127); one scale per 32
 * consecutive channels = floor(log2 max|x|) - 8, + 1 when the block maximum would exceed 448 after scaling; elements clamped to
 * +-448, round-to-nearest-even).  C % 32 == 0.
 * The activation format of the fp8 mode's 3x3 convolutions (SRGD_PRECISION_FP8). */
int srgd_k_quant_mxfp8(const void* x_bf16, void* q, void* s, int64_t npix, int C, void* stream);
/* 3x3 / stride 1 / pad 1 convolution on v_mfma_scale_f32_16x16x128_f8f6f4.  replaces: Block.proj (model.py:246) in fp8 mode.
 * in0 / in1: bf16 NHWC sources (channel concat; in1 nullable), quantised to MX-fp8 inside (srgd_k_quant_mxfp8); weights /
 * bias: PyTorch-layout fp32 on the HOST, quantised per (output channel, tap, 32 input channels).  C0, C1, Cout % 128 == 0,
 * H % 8 == 0, W % 32 == 0.  out: bf16 NHWC.  gn_partial / stats_slots as in srgd_k_conv2d_timed; iters > 0 times the
 * convolution kernel alone (*avg_ms).  Synchronises. */
int srgd_k_conv3x3_mxfp8(const void* in0, const void* in1, int C0, int C1, int B, int H, int W,
                         const float* weight_oihw_host, const float* bias_host, int Cout, void* out, float* gn_partial,
                         int groups, int iters, float* avg_ms, int* stats_slots, void* stream);

/* GroupNorm (from the conv's partial statistics) -> x*(scale+1)+shift -> SiLU (+ residual).
 * replaces: Block.forward after the conv (model.py:250-259) and the ResnetBlock residual add (:285).
 * gamma, beta: device [C]; scale_shift: device [B][2C] (scale | shift) or NULL; in place if y == x.
 * gn_partial: [B][groups][nslots][2] as written by the conv (nslots from srgd_k_conv2d_timed / srgd_k_conv3x3_mxfp8). */
int srgd_k_groupnorm_silu(const void* x, void* y, const void* residual, const float* gn_partial, int B, int hw,
                          int C, int groups, const float* gamma, const float* beta, const float* scale_shift,
                          int nslots, int is_bf16, void* stream);

/* RMSNorm over channels * g * sqrt(C) (+ residual).  replaces: RMSNorm.forward (model.py:206-207). */
int srgd_k_rmsnorm(const void* x, void* y, const void* residual, const float* g, int64_t npix, int C, int is_bf16,
                   void* stream);

/* qkv: [B,N,3*heads*32] -> out [B,N,heads*32].  replaces: LinearAttention.forward core (model.py:312-323)
 * and Attention.forward core + Attend (model.py:349-354). */
int srgd_k_linear_attention(const void* qkv, void* out, int B, int N, int heads, int is_bf16, void* stream);
int srgd_k_full_attention(const void* qkv, void* out, int B, int N, int heads, int is_bf16, void* stream);

/* The whole LinearAttention block plus its residual, fused (bf16, C = 128, 4 heads x 32, N % 128 == 0):
 * y = RMSNorm(to_out(linear_attention(to_qkv(RMSNorm(x))))) + x.   replaces: LinearAttention.forward
 * (model.py:306-324) and the `attn(x) + x` at model.py:703/:718.  x, y: device bf16 [B,N,C]; weights host fp32 in
 * PyTorch layout (to_qkv [384,C], norm.g [C], to_out.0.weight [C,128], to_out.0.bias [C], to_out.1.g [C]). */
int srgd_k_linattn_block_fused(const void* x, void* y, int B, int N, int C, const float* to_qkv_host,
                               const float* norm_g_host, const float* to_out_w_host, const float* to_out_b_host,
                               const float* out_g_host, void* stream);

/* f16x3 mode: a pointwise projection with an RMSNorm folded into its kernel (conv1x1_split.hip), fp32 NHWC tensors, N % 256 == 0.
 *   pre_norm_g_host  != NULL:  out = W . RMSNorm_g(x)                      replaces: self.norm(x) -> self.to_qkv (model.py:311-312, :348-349)
 *   post_norm_g_host != NULL:  out = RMSNorm_g(W . x + bias) + residual    replaces: LinearAttention.to_out (Conv2d, RMSNorm; model.py:300-303)
 *                              and the block's `attn(x) + x` (:703); Cout == 128, residual required.
 * RMSNorm_g(v) = v / max(||v||_2, 1e-12) * g * sqrt(C) over the channels of a pixel (model.py:201-207).  Weights [Cout][Cin], gains and
 * bias fp32 on the HOST (the gain of the pre-norm is folded into the weights before they are split); synchronises. */
int srgd_k_conv1x1_split_rms(const void* x, int Cin, int B, int N, const float* weight_oi_host, const float* bias_host, int Cout,
                             const float* pre_norm_g_host, const float* post_norm_g_host, const void* residual, void* out,
                             void* stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
