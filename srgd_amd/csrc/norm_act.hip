// GroupNorm (finalize + apply + time scale/shift + SiLU + residual) and RMSNorm for gfx950.
// Memory-bound element-wise kernels: 16-byte vector accesses along the NHWC channel axis.
#include "kernels.hpp"

// gn_apply streams tensors far larger than the caches (0.5-4 GB per launch) once: non-temporal loads of x / residual and
// non-temporal stores of y.  Round 5, same box, alternating: 5.27 -> 5.40 (loads) -> 5.58 TB/s (loads + stores), GroupNorm share of
// a step 5.1 -> 4.84 %, +0.2 % end to end (profiles/r5/nt_policy/r5_gnnt.json).
namespace srgd {
namespace {

template <typename V> __device__ __forceinline__ V ld_stream(const V* p) {
  V r;
  r.v = __builtin_nontemporal_load(&p->v);
  return r;
}
template <typename V> __device__ __forceinline__ void st_stream(V* p, const V& x) {
  __builtin_nontemporal_store(x.v, &p->v);
}

// One wave per (sample, group): sums the per-tile partials the conv epilogue wrote (fixed
// order, fp64) and folds GroupNorm's affine and the ResnetBlock's (scale+1, shift)
// (reference model.py:250-257) into per-(sample, channel) coefficients y = A*x + B.
__global__ __launch_bounds__(64) void gn_finalize_kernel(GnFinalizeArgs a) {
  const int b = blockIdx.x / a.groups, g = blockIdx.x - b * a.groups;
  const int lane = threadIdx.x;
  const float* p = a.partial + (size_t)(b * a.groups + g) * a.nslots * 2;
  double s1 = 0.0, s2 = 0.0;
  for (int i = lane; i < a.nslots; i += 64) {
    s1 += (double)p[2 * i];
    s2 += (double)p[2 * i + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  const int cpg = a.C / a.groups;
  const double n = (double)a.hw * (double)cpg;
  const double mean = s1 / n;
  double var = s2 / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)a.eps));
  const float fmean = (float)mean;
  const float* ss = nullptr;
  if (a.ss_table) {
    int row = a.ss_rows[b];
    if (a.step_ptr) row += (*a.step_ptr) * a.step_mul;
    ss = a.ss_table + (size_t)row * a.ss_stride + a.ss_offset;
  }
  for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
    float A = rstd * a.gamma[c];
    float B = a.beta[c] - fmean * A;
    if (ss) {
      const float sc = ss[c] + 1.0f, sh = ss[a.C + c];
      A *= sc;
      B = B * sc + sh;
    }
    a.coefA[(size_t)b * a.C + c] = A;
    a.coefB[(size_t)b * a.C + c] = B;
  }
}

// One grid row (blockIdx.y) per sample: the vector index inside the sample is a 32-bit int, its channel offset a 32-bit
// remainder (a mask when C / N is a power of two) and the coefficient rows are block-uniform.  (The first version ran one flat
// 64-bit index over the batch and paid a 64-bit division + remainder per 16-byte vector: VALU-bound at 5.1 TB/s.)
// HOIST (round 4): when the grid stride is a multiple of the vectors per pixel (C / N a power of two <= 256: every production
// layer), a thread meets the SAME channels in every trip - its 2 x N coefficients are loaded once, not per vector: the loop used
// to issue four 16-byte coefficient loads (L1 hits, but full vector-memory instructions) next to each 16-byte data load.
template <typename T, bool PRECISE, bool HOIST>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                        const T* __restrict__ res,
                                                        const float* __restrict__ cA,
                                                        const float* __restrict__ cB, int vec_per_sample,
                                                        int vec_per_pixel, int C, unsigned char* __restrict__ oq,
                                                        unsigned char* __restrict__ os) {
  constexpr int N = Vec16<T>::N;
  const int b = blockIdx.y;
  const size_t base = (size_t)b * vec_per_sample;
  const Vec16<T>* xs = reinterpret_cast<const Vec16<T>*>(x) + base;
  const Vec16<T>* rs = res ? reinterpret_cast<const Vec16<T>*>(res) + base : nullptr;
  Vec16<T>* ys = reinterpret_cast<Vec16<T>*>(y) + base;
  const float* pa0 = cA + (size_t)b * C;
  const float* pb0 = cB + (size_t)b * C;
  const bool pow2 = (vec_per_pixel & (vec_per_pixel - 1)) == 0;
  // one 16-byte vector: coefficients of its channels, y = silu(a x + b) (+ residual), store (+ MX-fp8 twin)
  float hca[N], hcb[N];
  if (HOIST) {
    const int c = (int)((blockIdx.x * 256u + threadIdx.x) & (unsigned)(vec_per_pixel - 1)) * N;
#pragma unroll
    for (int j = 0; j < N; j += 4) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(pa0 + c + j), b4 = *reinterpret_cast<const f32x4*>(pb0 + c + j);
#pragma unroll
      for (int k = 0; k < 4; ++k) { hca[j + k] = a4[k]; hcb[j + k] = b4[k]; }
    }
  }
  auto finish = [&](unsigned i, const Vec16<T>& v, const Vec16<T>& r) __attribute__((always_inline)) {
    float ca[N], cb[N];
    if (HOIST) {
#pragma unroll
      for (int j = 0; j < N; ++j) { ca[j] = hca[j]; cb[j] = hcb[j]; }
    } else {
      const int c = (int)(pow2 ? (i & (unsigned)(vec_per_pixel - 1)) : (i % (unsigned)vec_per_pixel)) * N;
#pragma unroll
      for (int j = 0; j < N; j += 4) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(pa0 + c + j), b4 = *reinterpret_cast<const f32x4*>(pb0 + c + j);
#pragma unroll
        for (int k = 0; k < 4; ++k) { ca[j + k] = a4[k]; cb[j + k] = b4[k]; }
      }
    }
    Vec16<T> o;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float t = silu<PRECISE>(ca[j] * v.get(j) + cb[j]);
      if (rs) t += r.get(j);
      o.set(j, t);
    }
    st_stream(&ys[i], o);
    if constexpr (sizeof(T) == 2) {
      if (oq) mx_store_twin(o.v, oq, os, (base + i) * 8, threadIdx.x & 3);      // MX-fp8 twin of the stored bf16 values
    }
  };
  // Round 3: four vectors per trip with all their loads issued first.  A thread walks ~30 vectors of its sample; one load per
  // trip left ~32 KiB in flight per CU (32 waves x 1 KiB), about half of what HBM's latency-bandwidth product asks for.
  const unsigned stride = gridDim.x * 256u, n = (unsigned)vec_per_sample;
  unsigned i = blockIdx.x * 256u + threadIdx.x;
  for (; i + 3u * stride < n; i += 4u * stride) {
    const Vec16<T> v0 = ld_stream(&xs[i]), v1 = ld_stream(&xs[i + stride]), v2 = ld_stream(&xs[i + 2u * stride]), v3 = ld_stream(&xs[i + 3u * stride]);
    Vec16<T> r0, r1, r2, r3;
    if (rs) { r0 = ld_stream(&rs[i]); r1 = ld_stream(&rs[i + stride]); r2 = ld_stream(&rs[i + 2u * stride]); r3 = ld_stream(&rs[i + 3u * stride]); }
    finish(i, v0, r0);
    finish(i + stride, v1, r1);
    finish(i + 2u * stride, v2, r2);
    finish(i + 3u * stride, v3, r3);
  }
  for (; i < n; i += stride) {
    const Vec16<T> v = xs[i];
    Vec16<T> r;
    if (rs) r = rs[i];
    finish(i, v, r);
  }
}

// RMSNorm (model.py:201-207): x / max(||x||_2, 1e-12) * g * sqrt(C)  (+ residual).
// L lanes cooperate on one pixel (L = largest power of two <= min(64, C/vec)).
template <typename T>
__global__ __launch_bounds__(256) void rms_norm_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                        const T* __restrict__ res, const float* __restrict__ g,
                                                        long npix, int C, int L, float sqrtC) {
  constexpr int N = Vec16<T>::N;
  const int vpp = C / N;
  const int sub = threadIdx.x % L;
  const int pix_per_block = 256 / L;
  for (long p = (long)blockIdx.x * pix_per_block + threadIdx.x / L; p < npix; p += (long)gridDim.x * pix_per_block) {
    const Vec16<T>* xp = reinterpret_cast<const Vec16<T>*>(x) + p * vpp;
    float ss = 0.f;
    for (int v = sub; v < vpp; v += L) {
      Vec16<T> t = xp[v];
#pragma unroll
      for (int j = 0; j < N; ++j) ss += t.get(j) * t.get(j);
    }
    for (int o = L >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = sqrtC / fmaxf(sqrtf(ss), 1e-12f);
    for (int v = sub; v < vpp; v += L) {
      Vec16<T> t = xp[v], o, r;
      if (res) r = reinterpret_cast<const Vec16<T>*>(res)[p * vpp + v];
#pragma unroll
      for (int j = 0; j < N; ++j) {
        float q = t.get(j) * inv * g[v * N + j];
        if (res) q += r.get(j);
        o.set(j, q);
      }
      reinterpret_cast<Vec16<T>*>(y)[p * vpp + v] = o;
    }
  }
}

}  // namespace

int gn_finalize(const GnFinalizeArgs& a, hipStream_t st) {
  if (a.C % a.groups != 0) SRGD_FAIL("gn_finalize: C % groups != 0");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(a.B * a.groups), dim3(64), 0, st, a);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int gn_apply_silu(const void* x, void* y, const void* residual, const float* coefA, const float* coefB, int B,
                  int hw, int C, bool is_bf16, hipStream_t st, void* out_q, void* out_s) {
  const int N = is_bf16 ? 8 : 4;
  if (C % N != 0) SRGD_FAIL("gn_apply: C must be a multiple of the 16-byte vector width");
  if (out_q && (!is_bf16 || !out_s || C % 32 != 0)) SRGD_FAIL("gn_apply: the MX-fp8 twin needs bf16 activations and C % 32 == 0");
  const int vps = (int)((long)hw * C / N), vpp = C / N;
  if ((long)hw * C / N >= (1L << 31) || B > 65535) SRGD_FAIL("gn_apply: sample too large for the 32-bit vector index");
  // ~64 blocks per CU over the whole launch, at least one block per sample
  const int gx = (int)std::max<long>(1, std::min<long>((vps + 255) / 256, (256L * 64 + B - 1) / B));
  // the grid stride gx * 256 is a multiple of the vectors per pixel whenever those are a power of two <= 256
  const bool hoist = (vpp & (vpp - 1)) == 0 && vpp <= 256;
#define K_GN_GO(T_, P_, H_, ...) hipLaunchKernelGGL((gn_apply_kernel<T_, P_, H_>), dim3(gx, B), dim3(256), 0, st, __VA_ARGS__)
  if (is_bf16) {
    if (hoist) K_GN_GO(bf16, false, true, (const bf16*)x, (bf16*)y, (const bf16*)residual, coefA, coefB, vps, vpp, C, (unsigned char*)out_q, (unsigned char*)out_s);
    else K_GN_GO(bf16, false, false, (const bf16*)x, (bf16*)y, (const bf16*)residual, coefA, coefB, vps, vpp, C, (unsigned char*)out_q, (unsigned char*)out_s);
  } else {
    if (hoist) K_GN_GO(float, true, true, (const float*)x, (float*)y, (const float*)residual, coefA, coefB, vps, vpp, C, nullptr, nullptr);
    else K_GN_GO(float, true, false, (const float*)x, (float*)y, (const float*)residual, coefA, coefB, vps, vpp, C, nullptr, nullptr);
  }
#undef K_GN_GO
  SRGD_HIP(hipGetLastError());
  return 0;
}

int rms_norm(const void* x, void* y, const void* residual, const float* g, long npix, int C, bool is_bf16,
             hipStream_t st) {
  const int N = is_bf16 ? 8 : 4;
  if (C % N != 0) SRGD_FAIL("rms_norm: C must be a multiple of the 16-byte vector width");
  int L = 1;
  while (L * 2 <= 64 && L * 2 <= C / N) L *= 2;
  const long blocks = (npix + (256 / L) - 1) / (256 / L);
  const int grid = (int)std::min<long>(blocks, 256L * 32);
  const float sqrtC = sqrtf((float)C);
  if (is_bf16)
    hipLaunchKernelGGL((rms_norm_kernel<bf16>), dim3(grid), dim3(256), 0, st, (const bf16*)x, (bf16*)y,
                       (const bf16*)residual, g, npix, C, L, sqrtC);
  else
    hipLaunchKernelGGL((rms_norm_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)x, (float*)y,
                       (const float*)residual, g, npix, C, L, sqrtC);
  SRGD_HIP(hipGetLastError());
  return 0;
}

}  // namespace srgd
