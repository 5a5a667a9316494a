// Image front/back end of the sampling path on the GPU (reference inference.py:66-73, :93):
//   * x4 (any integer or fractional factor) bicubic upsample of an 8-bit RGB image, bit-exact with Pillow's
//     Image.resize(BICUBIC) - which is what torchvision's T.Resize does for a PIL input - followed by ToTensor (/255);
//   * ToPILImage of the sampler's output: mul(255) and truncation to uint8.
// Pillow's algorithm (src/libImaging/Resample.c, pinned version 12.2.0 in this image; restated from its published
// source): per output coordinate a window [xmin, xmin+n) of the input (support 2.0 for bicubic, a = -0.5, clipped to
// the image) with double-precision weights renormalised to sum 1, converted to 22-bit fixed point (round half away from
// zero); horizontal pass, rounded and clipped to 8 bits, then vertical pass; accumulators start at 1 << 21 and the
// result is clip8(acc >> 22).  The tables are built on the host exactly as Pillow builds them; the passes are integer
// kernels (HBM-bound byte work, one thread per output sample; no MFMA shape to be had here).
#include <cmath>
#include <vector>

#include "../../include/srgd_hip.h"
#include "kernels.hpp"

namespace srgd {
namespace {

constexpr int PREC_BITS = 32 - 8 - 2;
constexpr int KSIZE_MAX = 64;

double bicubic_weight(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

// bounds: [out][2] = (first input index, count); kk: [out][ksize] fixed-point weights
int build_coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& kk, int* ksize_out) {
  const double scale = (double)in_size / (double)out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  if (ksize > KSIZE_MAX) SRGD_FAIL("image resize: reduction factor too large for this build");
  bounds.assign((size_t)out_size * 2, 0);
  kk.assign((size_t)out_size * ksize, 0);
  std::vector<double> w(ksize);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      w[x] = bicubic_weight((x + xmin - center + 0.5) * ss);
      ww += w[x];
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) w[x] /= ww;
      kk[(size_t)xx * ksize + x] = w[x] < 0 ? (int)(-0.5 + w[x] * (1 << PREC_BITS)) : (int)(0.5 + w[x] * (1 << PREC_BITS));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  *ksize_out = ksize;
  return 0;
}

__device__ __forceinline__ int clip8(int acc) {
  const int v = acc >> PREC_BITS;               // arithmetic shift, as Pillow's clip8 lookup index
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// src [h][w][3] u8 -> tmp [h][out_w][3] u8
__global__ void resample_h_kernel(const unsigned char* __restrict__ src, int h, int w, int out_w,
                                  const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                  unsigned char* __restrict__ tmp) {
  const long n = (long)h * out_w * 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % 3);
    const long t = i / 3;
    const int xx = (int)(t % out_w), y = (int)(t / out_w);
    const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
    int acc = 1 << (PREC_BITS - 1);
    for (int x = 0; x < cnt; ++x) acc += (int)src[((long)y * w + xmin + x) * 3 + c] * kk[xx * ksize + x];
    tmp[i] = (unsigned char)clip8(acc);
  }
}

// tmp [h][out_w][3] u8 -> dst [3][out_h][out_w] fp32 = u8 / 255 (ToTensor)
__global__ void resample_v_unit_kernel(const unsigned char* __restrict__ tmp, int h, int out_h, int out_w,
                                       const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                       float* __restrict__ dst) {
  const long plane = (long)out_h * out_w;
  const long n = plane * 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i / plane);
    const long r = i - (long)c * plane;
    const int yy = (int)(r / out_w), x = (int)(r - (long)yy * out_w);
    const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
    int acc = 1 << (PREC_BITS - 1);
    for (int y = 0; y < cnt; ++y) acc += (int)tmp[((long)(ymin + y) * out_w + x) * 3 + c] * kk[yy * ksize + y];
    dst[i] = __fdiv_rn((float)clip8(acc), 255.0f);
  }
}

// img [3][h][w] fp32 in [0,1] -> dst [h][w][3] u8 : mul(255).byte()
__global__ void unit_to_u8_kernel(const float* __restrict__ img, int h, int w, unsigned char* __restrict__ dst) {
  const long plane = (long)h * w;
  const long n = plane * 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % 3);
    const long px = i / 3;
    const float v = __fmul_rn(img[(long)c * plane + px], 255.0f);
    dst[i] = (unsigned char)(int)v;              // truncation toward zero, as Tensor.byte()
  }
}

int grid1d(long n) { return (int)std::min<long>((n + 255) / 256, 256L * 32); }

}  // namespace
}  // namespace srgd

using namespace srgd;

extern "C" {

int srgd_image_resize_bicubic_u8(const uint8_t* src_hwc, int h, int w, int out_h, int out_w, float* dst01_chw,
                                 void* stream) {
  if (!src_hwc || !dst01_chw) SRGD_FAIL("srgd_image_resize_bicubic_u8: null argument");
  if (h < 1 || w < 1 || out_h < 1 || out_w < 1) SRGD_FAIL("srgd_image_resize_bicubic_u8: bad size");
  hipStream_t st = (hipStream_t)stream;
  std::vector<int> bw, kw, bh, kh;
  int ksw = 0, ksh = 0;
  SRGD_TRY(build_coeffs(w, out_w, bw, kw, &ksw));
  SRGD_TRY(build_coeffs(h, out_h, bh, kh, &ksh));
  // one scratch allocation: [tmp u8 | tables]; released after the stream has drained (once per image, off the hot loop)
  const size_t tmp_bytes = ((size_t)h * out_w * 3 + 255) & ~(size_t)255;
  const size_t tab_ints = bw.size() + kw.size() + bh.size() + kh.size();
  char* scratch = nullptr;
  SRGD_HIP(hipMalloc((void**)&scratch, tmp_bytes + tab_ints * 4));
  int* d_bw = reinterpret_cast<int*>(scratch + tmp_bytes);
  int* d_kw = d_bw + bw.size();
  int* d_bh = d_kw + kw.size();
  int* d_kh = d_bh + bh.size();
  hipError_t err = hipSuccess;
  auto up = [&](int* d, const std::vector<int>& v) {
    if (err == hipSuccess) err = hipMemcpyAsync(d, v.data(), v.size() * 4, hipMemcpyHostToDevice, st);
  };
  up(d_bw, bw); up(d_kw, kw); up(d_bh, bh); up(d_kh, kh);
  if (err == hipSuccess) {
    unsigned char* tmp = reinterpret_cast<unsigned char*>(scratch);
    hipLaunchKernelGGL(resample_h_kernel, dim3(grid1d((long)h * out_w * 3)), dim3(256), 0, st, src_hwc, h, w, out_w, d_bw,
                       d_kw, ksw, tmp);
    hipLaunchKernelGGL(resample_v_unit_kernel, dim3(grid1d((long)out_h * out_w * 3)), dim3(256), 0, st, tmp, h, out_h,
                       out_w, d_bh, d_kh, ksh, dst01_chw);
    err = hipGetLastError();
  }
  const hipError_t serr = hipStreamSynchronize(st);      // host tables and the scratch must outlive the copies / kernels
  (void)hipFree(scratch);
  if (err != hipSuccess) SRGD_FAIL(std::string("srgd_image_resize_bicubic_u8: ") + hipGetErrorString(err));
  if (serr != hipSuccess) SRGD_FAIL(std::string("srgd_image_resize_bicubic_u8: ") + hipGetErrorString(serr));
  return 0;
}

int srgd_image_unit_to_u8(const float* img01_chw, int h, int w, uint8_t* dst_hwc, void* stream) {
  if (!img01_chw || !dst_hwc) SRGD_FAIL("srgd_image_unit_to_u8: null argument");
  if (h < 1 || w < 1) SRGD_FAIL("srgd_image_unit_to_u8: bad size");
  hipLaunchKernelGGL(unit_to_u8_kernel, dim3(grid1d((long)h * w * 3)), dim3(256), 0, (hipStream_t)stream, img01_chw, h, w,
                     dst_hwc);
  SRGD_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
