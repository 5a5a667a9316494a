// Sampler-side kernels for gfx950: the gather that feeds the 7x7 input convolution straight out of
// the canvases (fuses the tile gather model.py:3368-3369 and torch.cat model.py:684), the
// 1x1 output convolution fused with guidance + the DDPM posterior step and the tile scatter
// (model.py:3147-3168, :3184-3188, :3379-3380), canvas preparation / ring re-noise / crop,
// and a counter-based Gaussian generator for the throughput mode.
#include "kernels.hpp"

namespace srgd {
namespace {

// ------------------------------------------------------------------ 7x7 input conv
struct InitSrc {
  const float* x;          // noisy input planes
  const float* cond;       // condition planes (may be null)
  int canvas;              // 1: tiles of a canvas, 0: plain NCHW batch
  const int* tile_yx;
  int first, ntiles;       // canvas mode: entries = passes * ntiles
  int use_cond_mask;       // bit p: pass p sees the condition
  int H, W;                // tile / image size
  int row_stride;          // canvas width or W
  long plane_stride;       // canvas plane or H*W
  // EDM (canvas mode only): x = (x + hat_coef * (s_noise * z)) * c_in, scalars read on the device
  const EdmScalars* edm;   // null: plain gather
  const int* step_ptr;
  const float* z;          // noise canvas [3][Hp][Wp] (edm_pass 0) or null
  int edm_pass;
};

// Gather for the MFMA route of the 7x7 input conv (see kernels.hpp): one thread per padded position.
template <typename T>
__global__ __launch_bounds__(256) void init_gather_kernel(InitSrc s, int entries, T* __restrict__ padded) {
  const int Wq = s.W + 8, Hq = s.H + 6;
  const long n = (long)entries * Hq * Wq;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int px = (int)(i % Wq);
    const long t = i / Wq;
    const int py = (int)(t % Hq);
    const int entry = (int)(t / Hq);
    const int y = py - 3, x = px - 3;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (y >= 0 && y < s.H && x >= 0 && x < s.W) {
      long origin, zorigin = 0;
      bool use_cond = s.cond != nullptr;
      if (s.canvas) {
        const int pass = entry / s.ntiles, tt = entry - pass * s.ntiles;
        const int* tyx = s.tile_yx + 3 * (s.first + tt);
        zorigin = (long)tyx[0] * s.row_stride + tyx[1];
        origin = (long)tyx[2] * 3 * s.plane_stride + zorigin;
        use_cond = use_cond && ((s.use_cond_mask >> pass) & 1);
      } else {
        origin = (long)entry * 3 * s.plane_stride;
      }
      const long o = origin + (long)y * s.row_stride + x;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        v[c] = s.x[o + c * s.plane_stride];
        if (use_cond) v[3 + c] = s.cond[o + c * s.plane_stride];
      }
      if (s.edm) {
        const EdmScalars sc = s.edm[s.step_ptr ? *s.step_ptr : 0];
        const long oz = zorigin + (long)y * s.row_stride + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float xv = v[c];
          if (s.z) xv = xv + sc.hat_coef * (sc.s_noise * s.z[oz + c * s.plane_stride]);   // pass 0 of the Heun step only
          v[c] = (s.edm_pass == 1 ? sc.c_in_next : sc.c_in_hat) * xv;
        }
      }
    }
    T* dst = padded + i * 8;
    if constexpr (sizeof(T) == 2) {
      bf16x8 o8;
#pragma unroll
      for (int c = 0; c < 8; ++c) o8[c] = (bf16)v[c];
      *reinterpret_cast<bf16x8*>(dst) = o8;
    } else {
      *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  }
}

// ------------------------------------------------------------------ 1x1 output conv (+ DDPM step)
template <typename T>
__device__ __forceinline__ void out_conv3(const T* __restrict__ a, int C, const float* __restrict__ w,
                                          const float* __restrict__ bias, float e[3]) {
  constexpr int N = Vec16<T>::N;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int v = 0; v < C / N; ++v) {
    Vec16<T> t = reinterpret_cast<const Vec16<T>*>(a)[v];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float x = t.get(j);
      s0 += x * w[v * N + j];
      s1 += x * w[C + v * N + j];
      s2 += x * w[2 * C + v * N + j];
    }
  }
  e[0] = s0 + bias[0];
  e[1] = s1 + bias[1];
  e[2] = s2 + bias[2];
}

template <typename T>
__global__ __launch_bounds__(256) void final_conv_nchw_kernel(const T* __restrict__ act, long npix, int hw, int C,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ bias,
                                                              float* __restrict__ out) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  float e[3];
  out_conv3<T>(act + p * C, C, w, bias, e);
  const long b = p / hw, r = p - b * hw;
#pragma unroll
  for (int c = 0; c < 3; ++c) out[(b * 3 + c) * hw + r] = e[c];
}

// 16 lanes share one pixel's channel vector (one 16-byte load each, a wave reads 4 pixels = a contiguous run), partial
// dot products are combined with xor-shuffles inside the 16-lane group; returns the 3 outputs in every lane of the group.
template <typename T>
__device__ __forceinline__ void out_conv3_coop(const T* __restrict__ a, int C, const float* __restrict__ w,
                                               const float* __restrict__ bias, int lane16, float e[3]) {
  constexpr int N = Vec16<T>::N;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int v = lane16; v < C / N; v += 16) {
    Vec16<T> t = reinterpret_cast<const Vec16<T>*>(a)[v];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const float x = t.get(j);
      s0 += x * w[v * N + j];
      s1 += x * w[C + v * N + j];
      s2 += x * w[2 * C + v * N + j];
    }
  }
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) {
    s0 += __shfl_xor(s0, m, 64);
    s1 += __shfl_xor(s1, m, 64);
    s2 += __shfl_xor(s2, m, 64);
  }
  e[0] = s0 + bias[0];
  e[1] = s1 + bias[1];
  e[2] = s2 + bias[2];
}

// One block = 256 consecutive pixels of one tile.  Phase 1: 16 groups of 16 lanes walk the pixels (coalesced reads of
// the activation rows), eps (after the guidance combine) goes to LDS; phase 2: one thread per pixel does the DDPM
// update on the three canvas planes (coalesced along x).
template <typename T>
__global__ __launch_bounds__(256) void final_step_kernel(FinalStepArgs a, TileBatch tb) {
  __shared__ float eps[3][256];
  const int tile = tb.tile;
  const long per_tile = (long)tile * tile;
  const long p0 = (long)blockIdx.x * 256;                 // per_tile % 256 == 0: the block stays inside one tile
  const int lane16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const T* act = reinterpret_cast<const T*>(a.act);
  if (a.eps4) {                              // the output convolution ran in the last ResnetBlock's epilogue (ConvArgs::eps4)
    f32x4 e = *reinterpret_cast<const f32x4*>(a.eps4 + (p0 + threadIdx.x) * 4);
    if (a.passes == 2) {
      const f32x4 n = *reinterpret_cast<const f32x4*>(a.eps4 + (p0 + threadIdx.x + per_tile * tb.ntiles) * 4);
#pragma unroll
      for (int c = 0; c < 3; ++c) e[c] = n[c] + (e[c] - n[c]) * a.guidance;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) eps[c][threadIdx.x] = e[c];
  } else {
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int q = i * 16 + grp;
    float e[3];
    out_conv3_coop<T>(act + (p0 + q) * a.C, a.C, a.w, a.bias, lane16, e);
    if (a.passes == 2) {
      float n[3];
      out_conv3_coop<T>(act + (p0 + q + per_tile * tb.ntiles) * a.C, a.C, a.w, a.bias, lane16, n);
#pragma unroll
      for (int c = 0; c < 3; ++c) e[c] = n[c] + (e[c] - n[c]) * a.guidance;     // model.py:3150 / :3154
    }
    if (lane16 < 3) eps[lane16][q] = lane16 == 0 ? e[0] : (lane16 == 1 ? e[1] : e[2]);
  }
  }
  __syncthreads();
  const long p = p0 + threadIdx.x;
  const int t = (int)(p / per_tile);
  const int r = (int)(p - t * per_tile);
  const int y = r / tile, x = r - y * tile;
  const StepScalars sc = a.sc[a.step_ptr ? *a.step_ptr : 0];
  const int* tyx = tb.tile_yx + 3 * (tb.first + t);
  const int ty = tyx[0], tx = tyx[1];
  const long plane = (long)tb.Hp * tb.Wp;
  const long o = (long)tyx[2] * 3 * plane + (long)(ty + y) * tb.Wp + tx + x;
  const int tl = (tb.first + t) % tb.n_local;       // tile index inside its image: selects the noise tile
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float xt = a.img[c * plane + o];
    float x0 = (xt - sc.sigma * eps[c][threadIdx.x]) / sc.alpha;                // model.py:3160
    x0 = fminf(fmaxf(x0, -1.0f), 1.0f);                                        // :3163
    float mean = sc.alpha_next * (xt * sc.one_minus_c / sc.alpha + sc.c * x0);  // :3164
    if (a.noise) mean += sc.noise_scale * a.noise[((long)tl * 3 + c) * per_tile + r];   // :3187-3188
    a.img[c * plane + o] = mean;
    if (a.x_start) a.x_start[c * plane + o] = x0;
  }
}

// EDM epilogue (model.py:2140-2183 preconditioning + guidance, :2403-2425 Euler / Heun): same two phases as final_step_kernel.
template <typename T>
__global__ __launch_bounds__(256) void final_step_edm_kernel(FinalStepArgs a, const EdmScalars* __restrict__ scp,
                                                             float* __restrict__ work, size_t canvas_elems, int edm_pass,
                                                             TileBatch tb) {
  __shared__ float eps[3][256];
  const int tile = tb.tile;
  const long per_tile = (long)tile * tile;
  const long p0 = (long)blockIdx.x * 256;
  const int lane16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const T* act = reinterpret_cast<const T*>(a.act);
  if (a.eps4) {
    f32x4 e = *reinterpret_cast<const f32x4*>(a.eps4 + (p0 + threadIdx.x) * 4);
    if (a.passes == 2) {
      const f32x4 n = *reinterpret_cast<const f32x4*>(a.eps4 + (p0 + threadIdx.x + per_tile * tb.ntiles) * 4);
#pragma unroll
      for (int c = 0; c < 3; ++c) e[c] = n[c] + (e[c] - n[c]) * a.guidance;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) eps[c][threadIdx.x] = e[c];
  } else {
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int q = i * 16 + grp;
    float e[3];
    out_conv3_coop<T>(act + (p0 + q) * a.C, a.C, a.w, a.bias, lane16, e);
    if (a.passes == 2) {
      float n[3];
      out_conv3_coop<T>(act + (p0 + q + per_tile * tb.ntiles) * a.C, a.C, a.w, a.bias, lane16, n);
      // guidance on the preconditioned outputs (model.py:2162, :2176) == guidance on the raw outputs: the
      // preconditioning is the same affine map for both
#pragma unroll
      for (int c = 0; c < 3; ++c) e[c] = n[c] + (e[c] - n[c]) * a.guidance;
    }
    if (lane16 < 3) eps[lane16][q] = lane16 == 0 ? e[0] : (lane16 == 1 ? e[1] : e[2]);
  }
  }
  __syncthreads();
  const long p = p0 + threadIdx.x;
  const int t = (int)(p / per_tile);
  const int r = (int)(p - t * per_tile);
  const int y = r / tile, x = r - y * tile;
  const EdmScalars sc = scp[a.step_ptr ? *a.step_ptr : 0];
  const int* tyx = tb.tile_yx + 3 * (tb.first + t);
  const long plane = (long)tb.Hp * tb.Wp;
  const long oz = (long)(tyx[0] + y) * tb.Wp + tyx[1] + x;
  const long o = (long)tyx[2] * 3 * plane + oz;
  float* nxt_c = work;
  float* d_c = work + canvas_elems;
  const bool last = sc.sigma_next == 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float net = eps[c][threadIdx.x];
    if (edm_pass == 2) {                      // DPM-Solver++(2M) update of the un-tiled loop (model.py:2528-2547)
      const float xi = a.img[c * plane + o];
      float den = sc.c_skip_hat * xi + sc.c_out_hat * net;
      if (sc.clamp != 0.0f) den = fminf(fmaxf(den, -1.0f), 1.0f);
      const float den_d = (1.0f - sc.dpm_gamma) * den + sc.dpm_gamma * nxt_c[c * plane + o];     // :2539 (gamma 0: den)
      a.img[c * plane + o] = sc.dt * xi - sc.half_dt * den_d;                                    // :2541
      nxt_c[c * plane + o] = den;                                                                // old_denoised :2542
      if (a.x_start) a.x_start[c * plane + o] = den_d;                                           // :2547
      continue;
    }
    const float xh = a.img[c * plane + o] + sc.hat_coef * (sc.s_noise * a.noise[c * plane + oz]);     // :2389
    if (edm_pass == 0) {
      float den = sc.c_skip_hat * xh + sc.c_out_hat * net;                     // :2149
      if (sc.clamp != 0.0f) den = fminf(fmaxf(den, -1.0f), 1.0f);              // :2180-2181
      const float d = (xh - den) / sc.sigma_hat;                               // :2406
      const float nx = xh + sc.dt * d;                                         // :2407
      if (last) {
        a.img[c * plane + o] = nx;
        if (a.x_start) a.x_start[c * plane + o] = d;                           // :2423-2424
      } else {
        nxt_c[c * plane + o] = nx;
        d_c[c * plane + o] = d;
      }
    } else {
      const float nx = nxt_c[c * plane + o], d = d_c[c * plane + o];
      float den = sc.c_skip_next * nx + sc.c_out_next * net;
      if (sc.clamp != 0.0f) den = fminf(fmaxf(den, -1.0f), 1.0f);
      const float d2 = (nx - den) / sc.sigma_next;                             // :2413
      a.img[c * plane + o] = xh + sc.half_dt * (d + d2);                       // :2414
      if (a.x_start) a.x_start[c * plane + o] = d2;                            // :2421-2422
    }
  }
}

// ------------------------------------------------------------------ canvas kernels
__global__ void canvas_prepare_cond_kernel(const float* __restrict__ c01, int planes, int H, int W, int pad_l, int pad_t,
                                           int Hp, int Wp, int il, int it, int ir, int ib,
                                           float* __restrict__ canvas) {
  const long n = (long)planes * Hp * Wp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i / ((long)Hp * Wp));
    const long rem = i - (long)c * Hp * Wp;
    const int Y = (int)(rem / Wp), X = (int)(rem - (long)Y * Wp);
    float v = 0.f;
    if (Y >= it && Y < ib && X >= il && X < ir) {
      int y = Y - pad_t, x = X - pad_l;
      if (y < 0) y = -y;
      if (y >= H) y = 2 * (H - 1) - y;
      if (x < 0) x = -x;
      if (x >= W) x = 2 * (W - 1) - x;
      v = c01[((long)c * H + y) * W + x] * 2.0f - 1.0f;
    }
    canvas[i] = v;
  }
}

// q_sample start of a run (model.py:3305-3308, :3312-3315): img = reflect_pad(2*cond-1) * alpha + noise * sigma over
// the WHOLE canvas (the condition is not yet zeroed outside the inner box at that point of the reference).
__global__ void canvas_q_start_kernel(const float* __restrict__ c01, int planes, int H, int W, int pad_l, int pad_t,
                                      int Hp, int Wp, const float* __restrict__ noise, float alpha, float sigma,
                                      float* __restrict__ img) {
  const long n = (long)planes * Hp * Wp;
  const long nmod = 3L * Hp * Wp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i / ((long)Hp * Wp));
    const long rem = i - (long)c * Hp * Wp;
    const int Y = (int)(rem / Wp), X = (int)(rem - (long)Y * Wp);
    int y = Y - pad_t, x = X - pad_l;
    if (y < 0) y = -y;
    if (y >= H) y = 2 * (H - 1) - y;
    if (x < 0) x = -x;
    if (x >= W) x = 2 * (W - 1) - x;
    const float v = c01[((long)c * H + y) * W + x] * 2.0f - 1.0f;
    img[i] = v * alpha + noise[i % nmod] * sigma;
  }
}

__global__ void canvas_ring_renoise_kernel(float* __restrict__ img, int planes, const float* __restrict__ noise, int Hp,
                                           int Wp, int il, int it, int ir, int ib,
                                           const float* __restrict__ sigma_base, int sigma_stride,
                                           const int* __restrict__ step_ptr) {
  const float sigma = sigma_base[(long)(step_ptr ? *step_ptr : 0) * sigma_stride];
  const long n = (long)planes * Hp * Wp;
  const long nmod = 3L * Hp * Wp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long rem = i % ((long)Hp * Wp);
    const int Y = (int)(rem / Wp), X = (int)(rem - (long)Y * Wp);
    if (!(Y >= it && Y < ib && X >= il && X < ir)) img[i] = noise[i % nmod] * sigma;
  }
}

__global__ void canvas_finish_kernel(const float* __restrict__ img, int planes, int Hp, int Wp, int left, int top, int H,
                                     int W, float* __restrict__ out) {
  const long n = (long)planes * H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i / ((long)H * W));
    const long rem = i - (long)c * H * W;
    const int y = (int)(rem / W), x = (int)(rem - (long)y * W);
    float v = img[((long)c * Hp + top + y) * Wp + left + x];
    v = fminf(fmaxf(v, -1.0f), 1.0f);
    out[i] = (v + 1.0f) * 0.5f;
  }
}

// tile <-> packed buffer copy, one float4 per thread
__global__ __launch_bounds__(256) void canvas_exchange_tiles_kernel(float* __restrict__ canvas, float* __restrict__ packed,
                                                                    TileBatch tb, int to_canvas) {
  const int q4 = tb.tile / 4;
  const long n = (long)tb.ntiles * 3 * tb.tile * q4;
  const long plane = (long)tb.Hp * tb.Wp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int xq = (int)(i % q4);
    long r = i / q4;
    const int y = (int)(r % tb.tile);
    r /= tb.tile;
    const int c = (int)(r % 3), t = (int)(r / 3);
    const int* tyx = tb.tile_yx + 3 * (tb.first + t);
    float* cp = canvas + ((long)tyx[2] * 3 + c) * plane + (long)(tyx[0] + y) * tb.Wp + tyx[1] + xq * 4;
    float* pp = packed + i * 4;
    if (to_canvas) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(pp);
      cp[0] = v[0]; cp[1] = v[1]; cp[2] = v[2]; cp[3] = v[3];
    } else {
      *reinterpret_cast<f32x4*>(pp) = f32x4{cp[0], cp[1], cp[2], cp[3]};
    }
  }
}

// unpack of an all-gathered buffer in ONE launch: row j of `gathered` ([world * pw][3][tile][tile]) is the i = j % pw -th tile of rank
// r = j / pw 's part, i.e. tile r * w + off + i of the grid (ranks own slices of w tiles; a part is tiles [off, off + pw) of a slice);
// rows past the end of the grid (short and empty slices) are skipped
__global__ __launch_bounds__(256) void canvas_unpack_gathered_kernel(float* __restrict__ canvas, const float* __restrict__ gathered,
                                                                     TileBatch tb, int w, int off, int pw, int n_grid) {
  const int q4 = tb.tile / 4;
  const long n = (long)tb.ntiles * 3 * tb.tile * q4;         // tb.ntiles = world * pw rows
  const long plane = (long)tb.Hp * tb.Wp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int xq = (int)(i % q4);
    long r = i / q4;
    const int y = (int)(r % tb.tile);
    r /= tb.tile;
    const int c = (int)(r % 3), j = (int)(r / 3);
    const int t = (j / pw) * w + off + (j % pw);
    if (t >= n_grid) continue;
    const int* tyx = tb.tile_yx + 3 * t;
    float* cp = canvas + ((long)tyx[2] * 3 + c) * plane + (long)(tyx[0] + y) * tb.Wp + tyx[1] + xq * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(gathered + i * 4);
    cp[0] = v[0]; cp[1] = v[1]; cp[2] = v[2]; cp[3] = v[3];
  }
}

// ------------------------------------------------------------------ Philox4x32-10 + Box-Muller
__device__ __forceinline__ void philox_round(uint32_t c[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (uint32_t)p1;
  c[3] = (uint32_t)p0;
  c[0] = n0;
  c[2] = n2;
}

__global__ void philox_normal_kernel(float* __restrict__ dst, size_t n, uint64_t seed, uint64_t stream_id,
                                     const int* __restrict__ step_ptr) {
  const uint32_t step = step_ptr ? (uint32_t)*step_ptr : 0u;
  const size_t nq = (n + 3) / 4;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (size_t)gridDim.x * 256) {
    uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32) ^ (step << 8)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float u1 = ((float)c[2 * h] + 1.0f) * 2.3283064365386963e-10f;      // (0, 1]
      const float u2 = (float)c[2 * h + 1] * 2.3283064365386963e-10f;
      const float rad = sqrtf(-2.0f * __logf(u1));
      float sn, cs;
      __sincosf(6.283185307179586f * u2, &sn, &cs);
      z[2 * h] = rad * cs;
      z[2 * h + 1] = rad * sn;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (q * 4 + j < n) dst[q * 4 + j] = z[j];
  }
}

int grid_for(long n) { return (int)std::min<long>((n + 255) / 256, 256L * 16); }

}  // namespace

int init_gather_from_canvas(const float* img, const float* cond, const TileBatch& tb, int passes, int use_cond_mask,
                            void* padded, bool is_bf16, hipStream_t st) {
  return init_gather_from_canvas_edm(img, nullptr, cond, tb, passes, use_cond_mask, nullptr, nullptr, 0, padded, is_bf16, st);
}

int init_gather_from_canvas_edm(const float* img, const float* z, const float* cond, const TileBatch& tb, int passes,
                                int use_cond_mask, const EdmScalars* sc, const int* step_ptr, int edm_pass, void* padded,
                                bool is_bf16, hipStream_t st) {
  if (sc && edm_pass == 0 && !z) SRGD_FAIL("init_gather: EDM pass 0 needs the noise canvas");
  InitSrc s;
  s.x = img; s.cond = cond; s.canvas = 1; s.tile_yx = tb.tile_yx; s.first = tb.first; s.ntiles = tb.ntiles;
  s.use_cond_mask = use_cond_mask; s.H = tb.tile; s.W = tb.tile; s.row_stride = tb.Wp;
  s.plane_stride = (long)tb.Hp * tb.Wp;
  s.edm = sc; s.step_ptr = step_ptr; s.z = z; s.edm_pass = edm_pass;
  const int entries = passes * tb.ntiles;
  const int grid = grid_for((long)entries * (s.H + 6) * (s.W + 8));
  if (is_bf16) hipLaunchKernelGGL((init_gather_kernel<bf16>), dim3(grid), dim3(256), 0, st, s, entries, (bf16*)padded);
  else hipLaunchKernelGGL((init_gather_kernel<float>), dim3(grid), dim3(256), 0, st, s, entries, (float*)padded);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int init_gather_from_nchw(const float* x, const float* cond, int B, int H, int W, void* padded, bool is_bf16,
                          hipStream_t st) {
  InitSrc s;
  s.x = x; s.cond = cond; s.canvas = 0; s.tile_yx = nullptr; s.first = 0; s.ntiles = B; s.use_cond_mask = 1;
  s.H = H; s.W = W; s.row_stride = W; s.plane_stride = (long)H * W;
  s.edm = nullptr; s.step_ptr = nullptr; s.z = nullptr; s.edm_pass = 0;
  const int grid = grid_for((long)B * (H + 6) * (W + 8));
  if (is_bf16) hipLaunchKernelGGL((init_gather_kernel<bf16>), dim3(grid), dim3(256), 0, st, s, B, (bf16*)padded);
  else hipLaunchKernelGGL((init_gather_kernel<float>), dim3(grid), dim3(256), 0, st, s, B, (float*)padded);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int final_conv_to_nchw(const void* act, int B, int H, int W, int C, const float* w, const float* bias, float* out,
                       bool is_bf16, hipStream_t st) {
  const long npix = (long)B * H * W;
  const int grid = (int)((npix + 255) / 256);
  if (C % (is_bf16 ? 8 : 4) != 0) SRGD_FAIL("final_conv: C must be a multiple of the vector width");
  if (is_bf16)
    hipLaunchKernelGGL((final_conv_nchw_kernel<bf16>), dim3(grid), dim3(256), 0, st, (const bf16*)act, npix, H * W, C,
                       w, bias, out);
  else
    hipLaunchKernelGGL((final_conv_nchw_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)act, npix, H * W,
                       C, w, bias, out);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int final_step(const FinalStepArgs& a, const TileBatch& tb, bool is_bf16, hipStream_t st) {
  const long n = (long)tb.ntiles * tb.tile * tb.tile;
  const int grid = (int)((n + 255) / 256);
  if (a.C % (is_bf16 ? 8 : 4) != 0) SRGD_FAIL("final_step: C must be a multiple of the vector width");
  if (((long)tb.tile * tb.tile) % 256 != 0) SRGD_FAIL("final_step: tile area must be a multiple of 256");
  if (is_bf16) hipLaunchKernelGGL((final_step_kernel<bf16>), dim3(grid), dim3(256), 0, st, a, tb);
  else hipLaunchKernelGGL((final_step_kernel<float>), dim3(grid), dim3(256), 0, st, a, tb);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int final_step_edm(const FinalStepArgs& a, const EdmScalars* sc, float* work, size_t canvas_elems, int edm_pass,
                   const TileBatch& tb, bool is_bf16, hipStream_t st) {
  const long n = (long)tb.ntiles * tb.tile * tb.tile;
  const int grid = (int)((n + 255) / 256);
  if (a.C % (is_bf16 ? 8 : 4) != 0) SRGD_FAIL("final_step_edm: C must be a multiple of the vector width");
  if (((long)tb.tile * tb.tile) % 256 != 0) SRGD_FAIL("final_step_edm: tile area must be a multiple of 256");
  if (!sc || !work || (!a.noise && edm_pass != 2)) SRGD_FAIL("final_step_edm: null argument");
  if (is_bf16) hipLaunchKernelGGL((final_step_edm_kernel<bf16>), dim3(grid), dim3(256), 0, st, a, sc, work, canvas_elems, edm_pass, tb);
  else hipLaunchKernelGGL((final_step_edm_kernel<float>), dim3(grid), dim3(256), 0, st, a, sc, work, canvas_elems, edm_pass, tb);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int canvas_prepare_cond(const float* cond01, int planes, int H, int W, int pad_l, int pad_t, int Hp, int Wp, int il,
                        int it, int ir, int ib, float* cond_canvas, hipStream_t st) {
  hipLaunchKernelGGL(canvas_prepare_cond_kernel, dim3(grid_for((long)planes * Hp * Wp)), dim3(256), 0, st, cond01, planes,
                     H, W, pad_l, pad_t, Hp, Wp, il, it, ir, ib, cond_canvas);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int canvas_q_start(const float* cond01, int planes, int H, int W, int pad_l, int pad_t, int Hp, int Wp,
                   const float* noise, float alpha, float sigma, float* img, hipStream_t st) {
  hipLaunchKernelGGL(canvas_q_start_kernel, dim3(grid_for((long)planes * Hp * Wp)), dim3(256), 0, st, cond01, planes, H, W,
                     pad_l, pad_t, Hp, Wp, noise, alpha, sigma, img);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int canvas_ring_renoise(float* img, int planes, const float* noise, int Hp, int Wp, int il, int it, int ir, int ib,
                        const float* sigma_base, int sigma_stride, const int* step_ptr, hipStream_t st) {
  hipLaunchKernelGGL(canvas_ring_renoise_kernel, dim3(grid_for((long)planes * Hp * Wp)), dim3(256), 0, st, img, planes,
                     noise, Hp, Wp, il, it, ir, ib, sigma_base, sigma_stride, step_ptr);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int canvas_finish(const float* img, int planes, int Hp, int Wp, int left, int top, int H, int W, float* out01,
                  hipStream_t st) {
  hipLaunchKernelGGL(canvas_finish_kernel, dim3(grid_for((long)planes * H * W)), dim3(256), 0, st, img, planes, Hp, Wp,
                     left, top, H, W, out01);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int canvas_unpack_gathered(float* canvas, const float* gathered, const TileBatch& tb, int w, int off, int pw, int n_grid, hipStream_t st) {
  if (tb.tile % 4 != 0) SRGD_FAIL("canvas_unpack_gathered: tile edge must be a multiple of 4");
  const long n = (long)tb.ntiles * 3 * tb.tile * (tb.tile / 4);
  hipLaunchKernelGGL(canvas_unpack_gathered_kernel, dim3(grid_for(n)), dim3(256), 0, st, canvas, gathered, tb, w, off, pw, n_grid);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int canvas_exchange_tiles(float* canvas, float* tiles, const TileBatch& tb, bool to_canvas, hipStream_t st) {
  if (tb.tile % 4 != 0) SRGD_FAIL("canvas_exchange_tiles: tile edge must be a multiple of 4");
  const long n = (long)tb.ntiles * 3 * tb.tile * (tb.tile / 4);
  hipLaunchKernelGGL(canvas_exchange_tiles_kernel, dim3(grid_for(n)), dim3(256), 0, st, canvas, tiles, tb, to_canvas ? 1 : 0);
  SRGD_HIP(hipGetLastError());
  return 0;
}

int philox_normal(float* dst, size_t n, uint64_t seed, uint64_t stream_id, const int* step_ptr, hipStream_t st) {
  hipLaunchKernelGGL(philox_normal_kernel, dim3(grid_for((long)((n + 3) / 4))), dim3(256), 0, st, dst, n, seed,
                     stream_id, step_ptr);
  SRGD_HIP(hipGetLastError());
  return 0;
}

}  // namespace srgd
