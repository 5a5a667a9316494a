// Conditioning path (reference model.py:223-238, :603-619, :264-267, :277-279): learned
// sinusoidal features of log-SNR and small dense layers.  Inputs are one scalar per
// (step, pass), so the whole table for a sampling run is computed once, ahead of the loop.
#include "kernels.hpp"

namespace srgd {
namespace {

__global__ void time_features_kernel(const float* __restrict__ log_snr, const float* __restrict__ w, int half,
                                     int rows, float* __restrict__ feat) {
  const int r = blockIdx.x;
  const int i = threadIdx.x;
  if (r >= rows) return;
  const float x = log_snr[r];
  float* f = feat + (size_t)r * (2 * half + 1);
  if (i == 0) f[0] = x;
  if (i < half) {
    // same association as the reference: ((x * w) * 2) * pi, all fp32
    const float fr = ((x * w[i]) * 2.0f) * 3.14159265358979323846f;
    f[1 + i] = sinf(fr);
    f[1 + half + i] = cosf(fr);
  }
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// one wave per output feature of one row
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x, int x_stride,
                                                          const float* __restrict__ W, const float* __restrict__ b,
                                                          float* __restrict__ y, int y_stride, int in_f, int out_f,
                                                          int act, const float* __restrict__ add, int add_stride) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int r = blockIdx.y;
  const int lane = threadIdx.x & 63;
  if (o >= out_f) return;
  const float* xr = x + (size_t)r * x_stride;
  const float* wr = W + (size_t)o * in_f;
  float s = 0.f;
  for (int i = lane; i < in_f; i += 64) {
    float v = xr[i];
    if (act == ACT_SILU_IN) v = v / (1.0f + expf(-v));
    s += wr[i] * v;
  }
  s = wave_sum(s);
  if (lane == 0) {
    s += b ? b[o] : 0.f;
    if (act == ACT_GELU) s = gelu_erf(s);
    if (add) s += add[(size_t)r * add_stride + o];
    y[(size_t)r * y_stride + o] = s;
  }
}

}  // namespace

int time_features(const float* log_snr, const float* w, int half, int rows, float* feat, hipStream_t st) {
  if (half > 1024) SRGD_FAIL("time_features: learned_sinusoidal_dim too large");
  const int threads = ((std::max(half, 1) + 63) / 64) * 64;
  hipLaunchKernelGGL(time_features_kernel, dim3(rows), dim3(threads), 0, st, log_snr, w, half, rows, feat);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int linear_rows(const float* x, int x_stride, const float* W, const float* b, float* y, int y_stride, int rows,
                int in_f, int out_f, int act, const float* add, int add_stride, hipStream_t st) {
  dim3 g(cdiv(out_f, 4), rows);
  hipLaunchKernelGGL(linear_rows_kernel, g, dim3(256), 0, st, x, x_stride, W, b, y, y_stride, in_f, out_f, act, add,
                     add_stride);
  SRGD_HIP(hipGetLastError());
  return 0;
}

}  // namespace srgd
