// bf16 NHWC activations -> OCP MX-fp8 for the block-scaled 3x3 convolution (conv3x3_mxfp8.hip):
//   q [npix][C]     e4m3 ("fn": bias 7, max 448, no infinity), one byte per element, channels contiguous
//   s [npix][C/32]  E8M0 shared exponent per 32 consecutive channels: x ~ q * 2^(s - 127)
// Scale rule: s - 127 = floor(log2(max |x| over the block)) - 8 (OCP Microscaling recipe, e4m3: emax = 8), plus one when the block
// maximum's mantissa exceeds 1.75 - the smallest power of two that does not saturate it (mx_quant8, common.hpp); elements scaled
// by 2^-(s-127), clamped to +-448 and rounded to nearest-even.
// Optionally fused in front: y = silu(a[b][c] * x + b[b][c]) - the GroupNorm-apply + SiLU between the two convolutions of a
// ResnetBlock (reference Block.forward model.py:250-259), so that pass writes 1 byte per element instead of 2.
// HBM-bound: 2 B read + 1.03 B written per element; 4 lanes share a scale block (16-byte loads, 8-byte stores).
#include "kernels.hpp"

namespace srgd {
namespace {

// One grid row (blockIdx.y) per sample for the GroupNorm variant: 32-bit index arithmetic and block-uniform coefficient rows
// (a flat 64-bit index costs a 64-bit division + remainder per 16-byte vector and makes the pass VALU-bound).
// HOIST (round 4, as gn_apply_kernel): the grid stride is a multiple of the vectors per pixel, so a thread's channels - and its 16
// coefficients - are the same in every trip: loaded once instead of four 16-byte loads per vector.
// Non-temporal loads of x and stores of q (as gn_apply in norm_act.hip: the same tensors, streamed once): configs[4] fp8
// 0.4575 -> 0.4608 HR tiles/s (+0.7 %), same box (profiles/r5/nt_policy/r5_qnt.json).
__device__ __forceinline__ bf16x8 ld_x(const bf16x8* p) { return __builtin_nontemporal_load(p); }

template <bool GN, bool HOIST>
__global__ __launch_bounds__(256) void quant_mxfp8_kernel(const bf16* __restrict__ x, unsigned char* __restrict__ q,
                                                           unsigned char* __restrict__ s, int vec_per_sample, int C,
                                                           const float* __restrict__ cA, const float* __restrict__ cB) {
  const int vec_per_pixel = C >> 3;
  const int b = blockIdx.y;
  const size_t base = (size_t)b * vec_per_sample;
  const bf16x8* xs = reinterpret_cast<const bf16x8*>(x) + base;
  uint2* qs = reinterpret_cast<uint2*>(q) + base;
  unsigned char* ss = s + (base >> 2);
  const float* pa = GN ? cA + (size_t)b * C : nullptr;
  const float* pb = GN ? cB + (size_t)b * C : nullptr;
  const bool pow2 = (vec_per_pixel & (vec_per_pixel - 1)) == 0;
  f32x4 ha0 = {0.f, 0.f, 0.f, 0.f}, ha1 = ha0, hb0 = ha0, hb1 = ha0;
  if (GN && HOIST) {
    const int c = (int)((blockIdx.x * 256u + threadIdx.x) & (unsigned)(vec_per_pixel - 1)) * 8;
    ha0 = *reinterpret_cast<const f32x4*>(pa + c); ha1 = *reinterpret_cast<const f32x4*>(pa + c + 4);
    hb0 = *reinterpret_cast<const f32x4*>(pb + c); hb1 = *reinterpret_cast<const f32x4*>(pb + c + 4);
  }
  auto finish = [&](unsigned i, const bf16x8& v) __attribute__((always_inline)) {
    float y[8];
    if (GN) {
      const int c = HOIST ? 0 : (int)(pow2 ? (i & (unsigned)(vec_per_pixel - 1)) : (i % (unsigned)vec_per_pixel)) * 8;
      const f32x4 a0 = HOIST ? ha0 : *reinterpret_cast<const f32x4*>(pa + c), a1 = HOIST ? ha1 : *reinterpret_cast<const f32x4*>(pa + c + 4);
      const f32x4 b0 = HOIST ? hb0 : *reinterpret_cast<const f32x4*>(pb + c), b1 = HOIST ? hb1 : *reinterpret_cast<const f32x4*>(pb + c + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y[j] = silu<false>(a0[j] * (float)v[j] + b0[j]);
        y[4 + j] = silu<false>(a1[j] * (float)v[4 + j] + b1[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = (float)v[j];
    }
    int sb;
    const uint2 w = mx_quant8(y, &sb);
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(u32x2_t{w.x, w.y}, reinterpret_cast<u32x2_t*>(&qs[i]));
    if ((threadIdx.x & 3) == 0) ss[i >> 2] = (unsigned char)sb;
  };
  // four vectors per trip, loads first (norm_act.hip: one load per trip leaves half of HBM's latency-bandwidth product unused);
  // the four lanes of a 32-channel block stay together: vec_per_sample and the stride are multiples of 4
  const unsigned stride = gridDim.x * 256u, n = (unsigned)vec_per_sample;
  unsigned i = blockIdx.x * 256u + threadIdx.x;
  for (; i + 3u * stride < n; i += 4u * stride) {
    const bf16x8 v0 = ld_x(&xs[i]), v1 = ld_x(&xs[i + stride]), v2 = ld_x(&xs[i + 2u * stride]), v3 = ld_x(&xs[i + 3u * stride]);
    finish(i, v0);
    finish(i + stride, v1);
    finish(i + 2u * stride, v2);
    finish(i + 3u * stride, v3);
  }
  for (; i < n; i += stride) finish(i, ld_x(&xs[i]));
}

}  // namespace

// B samples of hw pixels; the plain variant (no coefficients) treats the whole tensor as ceil(npix / 65536)-pixel "samples"
static int launch_quant(const void* x, void* q, void* s, int B, long hw, int C, const float* cA, const float* cB, hipStream_t st) {
  if (C % 32 != 0) SRGD_FAIL("quant_mxfp8: C must be a multiple of 32");
  const long vps = hw * (C / 8);
  if (vps >= (1L << 31) || B > 65535 || B < 1) SRGD_FAIL("quant_mxfp8: tensor too large for the 32-bit vector index");
  const int gx = (int)std::max<long>(1, std::min<long>((vps + 255) / 256, (256L * 64 + B - 1) / B));
  const int vpp = C / 8;
  const bool hoist = (vpp & (vpp - 1)) == 0 && vpp <= 256;       // the grid stride gx * 256 is then a multiple of vpp
  if (cA && hoist)
    hipLaunchKernelGGL((quant_mxfp8_kernel<true, true>), dim3(gx, B), dim3(256), 0, st, (const bf16*)x, (unsigned char*)q,
                       (unsigned char*)s, (int)vps, C, cA, cB);
  else if (cA)
    hipLaunchKernelGGL((quant_mxfp8_kernel<true, false>), dim3(gx, B), dim3(256), 0, st, (const bf16*)x, (unsigned char*)q,
                       (unsigned char*)s, (int)vps, C, cA, cB);
  else
    hipLaunchKernelGGL((quant_mxfp8_kernel<false, false>), dim3(gx, B), dim3(256), 0, st, (const bf16*)x, (unsigned char*)q,
                       (unsigned char*)s, (int)vps, C, nullptr, nullptr);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int quant_mxfp8(const void* x_bf16, void* q, void* s, long npix, int C, hipStream_t st) {
  if (npix <= 0) return 0;
  // split into equal "samples" so that the per-sample vector count stays a 32-bit int and the grid's y extent < 65536; the
  // split must keep every sample's vector count a multiple of 4 (one scale byte per 4 vectors): any whole number of pixels is
  if (C % 32 != 0) SRGD_FAIL("quant_mxfp8: C must be a multiple of 32");
  long per = npix;
  int B = 1;
  for (int d : {4096, 1024, 256, 64, 16, 4, 2})
    if (npix % d == 0 && npix / d >= 1024) { B = d; per = npix / d; break; }
  return launch_quant(x_bf16, q, s, B, per, C, nullptr, nullptr, st);
}

int gn_apply_silu_mxfp8(const void* x_bf16, void* q, void* s, const float* coefA, const float* coefB, int B, int hw, int C,
                        hipStream_t st) {
  if (!coefA || !coefB) SRGD_FAIL("gn_apply_silu_mxfp8: null coefficients");
  return launch_quant(x_bf16, q, s, B, hw, C, coefA, coefB, st);
}

}  // namespace srgd
