// bf16 NHWC activations -> OCP MX-fp8 for the block-scaled 3x3 convolution (conv3x3_mxfp8.hip):
//   q [npix][C]     e4m3 ("fn": bias 7, max 448, no infinity), one byte per element, channels contiguous
//   s [npix][C/32]  E8M0 shared exponent per 32 consecutive channels: x ~ q * 2^(s - 127)
// Scale rule (OCP Microscaling spec, e4m3: emax = 8): s - 127 = floor(log2(max |x| over the block)) - 8, elements scaled by
// 2^-(s-127), clamped to +-448 (a block maximum with mantissa > 1.75 would otherwise overflow) and rounded to nearest-even.
// Optionally fused in front: y = silu(a[b][c] * x + b[b][c]) - the GroupNorm-apply + SiLU between the two convolutions of a
// ResnetBlock (reference Block.forward model.py:250-259), so that pass writes 1 byte per element instead of 2.
// HBM-bound: 2 B read + 1.03 B written per element; 4 lanes share a scale block (16-byte loads, 8-byte stores).
#include "kernels.hpp"

namespace srgd {
namespace {

template <bool GN>
__global__ __launch_bounds__(256) void quant_mxfp8_kernel(const bf16* __restrict__ x, unsigned char* __restrict__ q,
                                                           unsigned char* __restrict__ s, long nvec, int C,
                                                           int vec_per_sample, const float* __restrict__ cA,
                                                           const float* __restrict__ cB) {
  const int vec_per_pixel = C >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    const bf16x8 v = reinterpret_cast<const bf16x8*>(x)[i];
    float y[8];
    if (GN) {
      const int b = (int)(i / vec_per_sample);
      const int c = (int)(i % vec_per_pixel) * 8;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(cA + (size_t)b * C + c), a1 = *reinterpret_cast<const f32x4*>(cA + (size_t)b * C + c + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(cB + (size_t)b * C + c), b1 = *reinterpret_cast<const f32x4*>(cB + (size_t)b * C + c + 4);
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
    reinterpret_cast<uint2*>(q)[i] = w;
    if ((threadIdx.x & 3) == 0) s[i >> 2] = (unsigned char)sb;
  }
}

}  // namespace

static int launch_quant(const void* x, void* q, void* s, long npix, int C, int hw, const float* cA, const float* cB,
                        hipStream_t st) {
  if (C % 32 != 0) SRGD_FAIL("quant_mxfp8: C must be a multiple of 32");
  const long nvec = npix * (C / 8);
  const int grid = (int)std::min<long>((nvec + 255) / 256, 256L * 64);
  if (cA)
    hipLaunchKernelGGL((quant_mxfp8_kernel<true>), dim3(grid), dim3(256), 0, st, (const bf16*)x, (unsigned char*)q,
                       (unsigned char*)s, nvec, C, hw * (C / 8), cA, cB);
  else
    hipLaunchKernelGGL((quant_mxfp8_kernel<false>), dim3(grid), dim3(256), 0, st, (const bf16*)x, (unsigned char*)q,
                       (unsigned char*)s, nvec, C, hw * (C / 8), nullptr, nullptr);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int quant_mxfp8(const void* x_bf16, void* q, void* s, long npix, int C, hipStream_t st) {
  return launch_quant(x_bf16, q, s, npix, C, 1, nullptr, nullptr, st);
}

int gn_apply_silu_mxfp8(const void* x_bf16, void* q, void* s, const float* coefA, const float* coefB, int B, int hw, int C,
                        hipStream_t st) {
  if (!coefA || !coefB) SRGD_FAIL("gn_apply_silu_mxfp8: null coefficients");
  return launch_quant(x_bf16, q, s, (long)B * hw, C, hw, coefA, coefB, st);
}

}  // namespace srgd
