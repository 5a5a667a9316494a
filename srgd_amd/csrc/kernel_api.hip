// Kernel-level C ABI (include/srgd_hip_kernels.h): thin wrappers that let each kernel family be
// driven - and parity-tested - in isolation.  Convenience allocations here are per call; the
// engine itself (engine.hip) never allocates on its hot path.
#include <cmath>
#include <vector>

#include "../../include/srgd_hip_kernels.h"
#include "kernels.hpp"

using namespace srgd;

namespace {
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t n) { SRGD_HIP(hipMalloc(&p, n)); return 0; }
};
__global__ void iota_rows(int* r, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) r[i] = i;
}
}  // namespace

extern "C" {

int srgd_k_conv2d_timed(const void* in0, const void* in1, int C0, int C1, int B, int Hin, int Win, int KS, int stride,
                        int pad, int kind, const float* weight_oihw_host, const float* bias_host, int Cout, void* out,
                        const void* residual, float* gn_partial, int groups, int is_bf16, int impl, int iters,
                        float* avg_ms, int* stats_slots, const void* gn_tail_src, const float* gn_tail_a,
                        const float* gn_tail_b, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int Cin = C0 + C1;
  const int CoutPad = cdiv(Cout, conv_tile_n()) * conv_tile_n();
  std::vector<unsigned char> packed;
  std::vector<float> bias;
  pack_conv_weights(weight_oihw_host, bias_host, kind, Cin, Cout, CoutPad, KS, is_bf16 != 0, packed, bias);
  DevBuf dw, db, dw3, dw1;
  SRGD_TRY(dw.alloc(packed.size()));
  SRGD_HIP(hipMemcpy(dw.p, packed.data(), packed.size(), hipMemcpyHostToDevice));
  if (bias_host) {
    SRGD_TRY(db.alloc(bias.size() * 4));
    SRGD_HIP(hipMemcpy(db.p, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  }
  ConvArgs a;
  a.in0 = in0; a.in1 = in1; a.C0 = C0; a.C1 = C1; a.ps0 = C0; a.ps1 = C1; a.B = B; a.Hin = Hin; a.Win = Win;
  a.Hout = (Hin + 2 * pad - KS) / stride + 1;
  a.Wout = (Win + 2 * pad - KS) / stride + 1;
  a.KH = KS; a.KW = KS; a.stride = stride; a.pad = pad; a.w = dw.p; a.bias = (const float*)db.p; a.Cout = Cout; a.CoutPad = CoutPad;
  a.out = out; a.residual = residual; a.mode = kind == 2 ? CONV_PIXEL_SHUFFLE_SILU : CONV_PLAIN;
  a.gn_partial = gn_partial; a.groups = groups;
  a.gn_res_src = gn_tail_src; a.gn_res_a = gn_tail_src ? gn_tail_a : nullptr; a.gn_res_b = gn_tail_src ? gn_tail_b : nullptr;
  if (gn_tail_src && (!gn_tail_a || !gn_tail_b)) SRGD_FAIL("srgd_k_conv2d: gn_tail_src needs gn_tail_a and gn_tail_b");
  // impl 5: conv3x3_bf16 with the producer's GroupNorm + SiLU applied while the input is staged (GNIN): gn_tail_a / gn_tail_b
  // are then the [B][Cin] scale / shift arrays of the INPUT (one allocation, shift behind scale), not a tail operand
  const bool gnin = impl == 5;
  if (gnin) {
    if (!gn_tail_a || !gn_tail_b || C1) SRGD_FAIL("srgd_k_conv2d: impl 5 (GroupNorm-in-staging) needs one source and gn_tail_a / gn_tail_b");
    a.gn_res_src = nullptr; a.gn_res_a = nullptr; a.gn_res_b = nullptr;
  }
  const bool fast = (impl == 0 || impl == 2 || gnin) && is_bf16 && kind == 0 && conv3x3_bf16_eligible(a);
  if (gnin && !fast) SRGD_FAIL("srgd_k_conv2d: the conv3x3_bf16 fast path does not cover this shape");
  if (impl == 2 && !fast) SRGD_FAIL("srgd_k_conv2d: the conv3x3_bf16 fast path does not cover this shape");
  const bool fast1 = !fast && (impl == 0 || impl == 3) && is_bf16 && conv1x1_bf16_eligible(a);
  if (impl == 3 && !fast1) SRGD_FAIL("srgd_k_conv2d: the conv1x1_bf16 fast path does not cover this shape");
  if (fast1) {
    std::vector<unsigned char> f32p;
    std::vector<float> unused;
    pack_conv_weights(weight_oihw_host, bias_host, kind, Cin, Cout, CoutPad, KS, false, f32p, unused);
    std::vector<unsigned short> p1;
    pack_conv1x1_bf16(reinterpret_cast<const float*>(f32p.data()), KS * KS, Cin, Cout, p1, f32_to_bf16_host);
    SRGD_TRY(dw1.alloc(p1.size() * 2));
    SRGD_HIP(hipMemcpy(dw1.p, p1.data(), p1.size() * 2, hipMemcpyHostToDevice));
  }
  // impl 4: the pointwise layer on the MX matrix cores (conv1x1_mxfp8.hip); the bf16 sources are quantised here
  const bool fastq1 = impl == 4;
  DevBuf dwq1, q0, s0, q1, s1;
  if (fastq1) {
    if (!is_bf16 || !conv1x1_mxfp8_eligible(a)) SRGD_FAIL("srgd_k_conv2d: the conv1x1_mxfp8 path does not cover this shape");
    std::vector<unsigned char> f32p, pq1;
    std::vector<float> unused;
    pack_conv_weights(weight_oihw_host, bias_host, kind, Cin, Cout, CoutPad, KS, false, f32p, unused);
    pack_conv1x1_mxfp8(reinterpret_cast<const float*>(f32p.data()), KS * KS, Cin, Cout, pq1);
    SRGD_TRY(dwq1.alloc(pq1.size()));
    SRGD_HIP(hipMemcpy(dwq1.p, pq1.data(), pq1.size(), hipMemcpyHostToDevice));
    const size_t npix = (size_t)B * Hin * Win;
    SRGD_TRY(q0.alloc(npix * C0)); SRGD_TRY(s0.alloc(npix * (C0 / 32)));
    SRGD_TRY(quant_mxfp8(in0, q0.p, s0.p, (long)npix, C0, st));
    if (C1) {
      SRGD_TRY(q1.alloc(npix * C1)); SRGD_TRY(s1.alloc(npix * (C1 / 32)));
      SRGD_TRY(quant_mxfp8(in1, q1.p, s1.p, (long)npix, C1, st));
    }
  }
  if (fast) {
    std::vector<unsigned short> p3;
    pack_conv3x3_bf16(weight_oihw_host, Cin, Cout, p3, f32_to_bf16_host);
    SRGD_TRY(dw3.alloc(p3.size() * 2));
    SRGD_HIP(hipMemcpy(dw3.p, p3.data(), p3.size() * 2, hipMemcpyHostToDevice));
  }
  // impl 6 / 7: the split-operand kernels (fp32 tensors; f16 (hi, lo) operand pairs): 6 = conv3x3_split, 7 = conv_igemm_split;
  // impl 8 / 9: the same two kernels with bf16 halves (numerics comparison only - the engine uses f16)
  // impl 10: the streaming pointwise kernel of that mode (conv1x1_split.hip; f16 halves)
  // impl 11: impl 6 with the producer's GroupNorm + SiLU applied while the input is staged (gn_tail_a / gn_tail_b = [B][C0] scale / shift)
  const bool split_gnin = impl == 11 || impl == 13;
  if (split_gnin) {
    if (!gn_tail_a || !gn_tail_b || C1) SRGD_FAIL("srgd_k_conv2d: impl 11 (GroupNorm-in-staging) needs one source and gn_tail_a / gn_tail_b");
    a.gn_res_src = nullptr; a.gn_res_a = nullptr; a.gn_res_b = nullptr;
  }
  // impl 12 / 13: impl 6 / 11 on the 256-thread form of the kernel (two workgroups per CU) instead of the engine's default
  const bool split_form2 = impl == 12 || impl == 13;
  const bool split3 = impl == 6 || impl == 8 || impl == 12 || split_gnin, splitg = impl == 7 || impl == 9, split1 = impl == 10, split_f16 = impl == 6 || impl == 7 || impl == 12 || split1 || split_gnin;
  DevBuf dws;
  float ws_inv = 1.f;
  if (split3 || splitg || split1) {
    if (is_bf16) SRGD_FAIL("srgd_k_conv2d: the split-operand kernels take fp32 tensors (is_bf16 = 0)");
    if (split3 ? !conv3x3_split_eligible(a) || kind != 0 : split1 ? !conv1x1_split_eligible(a) : !conv_igemm_split_eligible(a))
      SRGD_FAIL("srgd_k_conv2d: the split-operand kernel does not cover this shape");
    const float scale = split_weight_scale(weight_oihw_host, (size_t)Cout * Cin * KS * KS, split_f16);
    ws_inv = 1.0f / scale;
    std::vector<unsigned short> ps;
    if (split3) pack_conv3x3_split(weight_oihw_host, Cin, Cout, split_f16, scale, ps);
    else if (split1) pack_conv1x1_split(reinterpret_cast<const float*>(packed.data()), KS * KS, Cin, Cout, scale, ps);
    else pack_conv_weights_split(reinterpret_cast<const float*>(packed.data()), KS * KS, Cin, CoutPad, split_f16, scale, ps);
    SRGD_TRY(dws.alloc(ps.size() * 2));
    SRGD_HIP(hipMemcpy(dws.p, ps.data(), ps.size() * 2, hipMemcpyHostToDevice));
  }
  // impl 14: the two-MFMA split arithmetic prototype (conv3x3_mx2.hip: f16 leading term + both cross terms on MX-fp8 operands)
  const bool mx2 = impl == 14 || impl == 15;      // 15: with the producer's GroupNorm + SiLU applied while the input is staged (as impl 11)
  if (impl == 15 && (!gn_tail_a || !gn_tail_b || C1)) SRGD_FAIL("srgd_k_conv2d: impl 15 (GroupNorm-in-staging) needs one source and gn_tail_a / gn_tail_b");
  if (impl == 15) { a.gn_res_src = nullptr; a.gn_res_a = nullptr; a.gn_res_b = nullptr; }
  DevBuf dwm;
  if (mx2) {
    if (is_bf16 || kind != 0 || !conv3x3_split_eligible(a)) SRGD_FAIL("srgd_k_conv2d: impl 14 takes fp32 tensors and conv3x3_split's shapes");
    const float scale = split_weight_scale(weight_oihw_host, (size_t)Cout * Cin * KS * KS, true);
    ws_inv = 1.0f / scale;
    std::vector<unsigned char> pm;
    pack_conv3x3_mx2(weight_oihw_host, Cin, Cout, scale, pm);
    SRGD_TRY(dwm.alloc(pm.size()));
    SRGD_HIP(hipMemcpy(dwm.p, pm.data(), pm.size(), hipMemcpyHostToDevice));
  }
  if (stats_slots) *stats_slots = (fast || split3 || mx2) ? conv3x3_bf16_stats_slots(a) : (a.Hout * a.Wout) / conv_tile_m();
  auto run = [&]() -> int {
    if (mx2) return conv3x3_mx2(a, dwm.p, ws_inv, st, impl == 15 ? gn_tail_a : nullptr, impl == 15 ? gn_tail_b : nullptr);
    if (split3) return conv3x3_split(a, dws.p, ws_inv, split_f16, st, split_gnin ? gn_tail_a : nullptr, split_gnin ? gn_tail_b : nullptr, split_form2 ? 2 : 0);
    if (split1) return conv1x1_split(a, dws.p, ws_inv, st);
    if (splitg) return conv_igemm_split(a, dws.p, ws_inv, split_f16, st);
    if (fastq1) return conv1x1_mxfp8(a, q0.p, s0.p, q1.p, s1.p, dwq1.p, st);
    return fast ? conv3x3_bf16(a, dw3.p, gnin ? gn_tail_a : nullptr, gnin ? gn_tail_b : nullptr, st) : fast1 ? conv1x1_bf16(a, dw1.p, st) : conv_igemm(a, is_bf16 != 0, st);
  };
  SRGD_TRY(run());
  SRGD_HIP(hipStreamSynchronize(st));
  if (iters > 0 && avg_ms) {
    hipEvent_t e0, e1;
    SRGD_HIP(hipEventCreate(&e0));
    SRGD_HIP(hipEventCreate(&e1));
    SRGD_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) SRGD_TRY(run());
    SRGD_HIP(hipEventRecord(e1, st));
    SRGD_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    SRGD_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  return 0;
}

int srgd_k_conv2d(const void* in0, const void* in1, int C0, int C1, int B, int Hin, int Win, int KS, int stride,
                  int pad, int kind, const float* weight_oihw_host, const float* bias_host, int Cout, void* out,
                  const void* residual, float* gn_partial, int groups, int is_bf16, void* stream) {
  return srgd_k_conv2d_timed(in0, in1, C0, C1, B, Hin, Win, KS, stride, pad, kind, weight_oihw_host, bias_host, Cout,
                             out, residual, gn_partial, groups, is_bf16, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                             stream);
}

int srgd_k_groupnorm_silu(const void* x, void* y, const void* residual, const float* gn_partial, int B, int hw,
                          int C, int groups, const float* gamma, const float* beta, const float* scale_shift,
                          int nslots, int is_bf16, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (nslots <= 0) SRGD_FAIL("groupnorm: nslots must be the slot count the producing conv reported");
  DevBuf cA, cB, rows;
  SRGD_TRY(cA.alloc((size_t)B * C * 4));
  SRGD_TRY(cB.alloc((size_t)B * C * 4));
  SRGD_TRY(rows.alloc((size_t)B * 4));
  hipLaunchKernelGGL(iota_rows, dim3(cdiv(B, 256)), dim3(256), 0, st, (int*)rows.p, B);
  GnFinalizeArgs f;
  f.partial = gn_partial; f.nslots = nslots; f.B = B; f.C = C; f.groups = groups; f.hw = hw;
  f.gamma = gamma; f.beta = beta; f.ss_table = scale_shift; f.ss_rows = (const int*)rows.p; f.step_ptr = nullptr;
  f.step_mul = 0; f.ss_stride = 2 * C; f.ss_offset = 0; f.eps = 1e-5f; f.coefA = (float*)cA.p; f.coefB = (float*)cB.p;
  SRGD_TRY(gn_finalize(f, st));
  SRGD_TRY(gn_apply_silu(x, y, residual, (const float*)cA.p, (const float*)cB.p, B, hw, C, is_bf16 != 0, st));
  SRGD_HIP(hipStreamSynchronize(st));
  return 0;
}

int srgd_k_rmsnorm(const void* x, void* y, const void* residual, const float* g, int64_t npix, int C, int is_bf16,
                   void* stream) {
  return rms_norm(x, y, residual, g, (long)npix, C, is_bf16 != 0, (hipStream_t)stream);
}

int srgd_k_linear_attention(const void* qkv, void* out, int B, int N, int heads, int is_bf16, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  DevBuf ws;
  SRGD_TRY(ws.alloc(linear_attention_workspace(B, N, heads, 32)));
  SRGD_TRY(linear_attention(qkv, out, B, N, heads, 32, (float*)ws.p, is_bf16 != 0, st));
  SRGD_HIP(hipStreamSynchronize(st));
  return 0;
}

int srgd_k_linattn_block_fused(const void* x, void* y, int B, int N, int C, const float* to_qkv_host,
                               const float* norm_g_host, const float* to_out_w_host, const float* to_out_b_host,
                               const float* out_g_host, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!linattn_fused_eligible(C, 4, 32, N, true)) SRGD_FAIL("linattn_block_fused: needs C = 128 or 256, N % 64 == 0");
  std::vector<unsigned short> wkv, wq, wo;
  linattn_fused_pack(to_qkv_host, norm_g_host, to_out_w_host, C, wkv, wq, wo);
  std::vector<float> g2(C);
  for (int c = 0; c < C; ++c) g2[c] = out_g_host[c] * sqrtf((float)C);
  DevBuf dkv, dq, dout, db, dg, ws;
  SRGD_TRY(dkv.alloc(wkv.size() * 2)); SRGD_TRY(dq.alloc(wq.size() * 2)); SRGD_TRY(dout.alloc(wo.size() * 2));
  SRGD_TRY(db.alloc(C * 4)); SRGD_TRY(dg.alloc(C * 4)); SRGD_TRY(ws.alloc(linattn_fused_workspace(B, N)));
  SRGD_HIP(hipMemcpy(dkv.p, wkv.data(), wkv.size() * 2, hipMemcpyHostToDevice));
  SRGD_HIP(hipMemcpy(dq.p, wq.data(), wq.size() * 2, hipMemcpyHostToDevice));
  SRGD_HIP(hipMemcpy(dout.p, wo.data(), wo.size() * 2, hipMemcpyHostToDevice));
  SRGD_HIP(hipMemcpy(db.p, to_out_b_host, C * 4, hipMemcpyHostToDevice));
  SRGD_HIP(hipMemcpy(dg.p, g2.data(), C * 4, hipMemcpyHostToDevice));
  SRGD_TRY(linattn_fused(x, y, B, N, C, dkv.p, dq.p, dout.p, (const float*)db.p, (const float*)dg.p, (float*)ws.p, st));
  SRGD_HIP(hipStreamSynchronize(st));
  return 0;
}

int srgd_k_conv1x1_split_rms(const void* x, int Cin, int B, int N, const float* weight_oi_host, const float* bias_host, int Cout,
                             const float* pre_norm_g_host, const float* post_norm_g_host, const void* residual, void* out,
                             void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!x || !weight_oi_host || !out || Cin <= 0 || Cout <= 0) SRGD_FAIL("srgd_k_conv1x1_split_rms: null argument");
  if (pre_norm_g_host && post_norm_g_host) SRGD_FAIL("srgd_k_conv1x1_split_rms: one RMSNorm per call");
  std::vector<float> wf((size_t)Cout * Cin);
  const float rc = sqrtf((float)Cin);
  for (int o = 0; o < Cout; ++o)
    for (int c = 0; c < Cin; ++c)
      wf[(size_t)o * Cin + c] = weight_oi_host[(size_t)o * Cin + c] * (pre_norm_g_host ? pre_norm_g_host[c] * rc : 1.0f);
  const float scale = split_weight_scale(wf.data(), wf.size(), true);
  std::vector<unsigned short> ps;
  ConvArgs a{};
  a.in0 = x; a.C0 = Cin; a.ps0 = Cin; a.B = B; a.Hin = 1; a.Win = N; a.Hout = 1; a.Wout = N; a.KH = a.KW = 1; a.stride = 1;
  a.Cout = Cout; a.CoutPad = Cout; a.out = out; a.mode = CONV_PLAIN; a.residual = residual;
  a.rms_in = pre_norm_g_host != nullptr;
  DevBuf dw, db, dg;
  if (bias_host) {
    SRGD_TRY(db.alloc((size_t)Cout * 4));
    SRGD_HIP(hipMemcpy(db.p, bias_host, (size_t)Cout * 4, hipMemcpyHostToDevice));
    a.bias = (const float*)db.p;
  }
  if (post_norm_g_host) {
    std::vector<float> g2(Cout);
    for (int c = 0; c < Cout; ++c) g2[c] = post_norm_g_host[c] * sqrtf((float)Cout);
    SRGD_TRY(dg.alloc((size_t)Cout * 4));
    SRGD_HIP(hipMemcpy(dg.p, g2.data(), (size_t)Cout * 4, hipMemcpyHostToDevice));
    a.rms_out_g = (const float*)dg.p;
  }
  if (!conv1x1_split_eligible(a)) SRGD_FAIL("srgd_k_conv1x1_split_rms: needs Cin % 32 == 0, Cout % 128 == 0 (== 128 with a post-norm, and a residual), N % 256 == 0");
  pack_conv1x1_split(wf.data(), 1, Cin, Cout, scale, ps);
  SRGD_TRY(dw.alloc(ps.size() * 2));
  SRGD_HIP(hipMemcpy(dw.p, ps.data(), ps.size() * 2, hipMemcpyHostToDevice));
  SRGD_TRY(conv1x1_split(a, dw.p, 1.0f / scale, st));
  SRGD_HIP(hipStreamSynchronize(st));
  return 0;
}

int srgd_k_quant_mxfp8(const void* x_bf16, void* q, void* s, int64_t npix, int C, void* stream) {
  if (!x_bf16 || !q || !s) SRGD_FAIL("srgd_k_quant_mxfp8: null argument");
  return quant_mxfp8(x_bf16, q, s, (long)npix, C, (hipStream_t)stream);
}

int srgd_k_conv3x3_mxfp8(const void* in0, const void* in1, int C0, int C1, int B, int H, int W,
                         const float* weight_oihw_host, const float* bias_host, int Cout, void* out, float* gn_partial,
                         int groups, int iters, float* avg_ms, int* stats_slots, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!in0 || !weight_oihw_host || !out) SRGD_FAIL("srgd_k_conv3x3_mxfp8: null argument");
  ConvArgs a{};
  a.C0 = C0; a.C1 = in1 ? C1 : 0; a.ps0 = C0; a.ps1 = a.C1; a.B = B; a.Hin = H; a.Win = W; a.Hout = H; a.Wout = W;
  a.KH = a.KW = 3; a.stride = 1; a.pad = 1; a.Cout = Cout; a.CoutPad = Cout; a.out = out; a.mode = CONV_PLAIN;
  a.gn_partial = gn_partial; a.groups = groups;
  if (!conv3x3_mxfp8_eligible(a)) SRGD_FAIL("srgd_k_conv3x3_mxfp8: needs C0, C1, Cout % 128 == 0, H % 8 == 0, W % 32 == 0");
  std::vector<unsigned char> pw;
  pack_conv3x3_mxfp8(weight_oihw_host, a.C0 + a.C1, Cout, pw);
  const size_t npix = (size_t)B * H * W;
  DevBuf dw, db, q0, s0, q1, s1;
  SRGD_TRY(dw.alloc(pw.size()));
  SRGD_HIP(hipMemcpy(dw.p, pw.data(), pw.size(), hipMemcpyHostToDevice));
  if (bias_host) {
    SRGD_TRY(db.alloc((size_t)Cout * 4));
    SRGD_HIP(hipMemcpy(db.p, bias_host, (size_t)Cout * 4, hipMemcpyHostToDevice));
    a.bias = (const float*)db.p;
  }
  SRGD_TRY(q0.alloc(npix * C0)); SRGD_TRY(s0.alloc(npix * (C0 / 32)));
  SRGD_TRY(quant_mxfp8(in0, q0.p, s0.p, (long)npix, C0, st));
  if (a.C1) {
    SRGD_TRY(q1.alloc(npix * a.C1)); SRGD_TRY(s1.alloc(npix * (a.C1 / 32)));
    SRGD_TRY(quant_mxfp8(in1, q1.p, s1.p, (long)npix, a.C1, st));
  }
  if (stats_slots) *stats_slots = conv3x3_mxfp8_stats_slots(a);
  SRGD_TRY(conv3x3_mxfp8(a, q0.p, s0.p, q1.p, s1.p, dw.p, st));
  if (iters > 0 && avg_ms) {
    hipEvent_t e0, e1;
    SRGD_HIP(hipEventCreate(&e0)); SRGD_HIP(hipEventCreate(&e1));
    SRGD_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) SRGD_TRY(conv3x3_mxfp8(a, q0.p, s0.p, q1.p, s1.p, dw.p, st));
    SRGD_HIP(hipEventRecord(e1, st));
    SRGD_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    SRGD_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  SRGD_HIP(hipStreamSynchronize(st));
  return 0;
}

int srgd_k_full_attention(const void* qkv, void* out, int B, int N, int heads, int is_bf16, void* stream) {
  return full_attention(qkv, out, B, N, heads, 32, is_bf16 != 0, (hipStream_t)stream);
}

}  // extern "C"
